#!/bin/bash
# usage (on the GPU box): bash tools/pstats.sh <tag> [N]  -- rocprofv3 kernel stats of N Poseidon k=16 proofs through the C ABI (keygen every call)
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
tag=${1:-p}; N=${2:-10}
rm -rf gpurun_out/pprof_$tag
rocprofv3 --kernel-trace --stats -d gpurun_out/pprof_$tag -o run --output-format csv -- python3 tools/proof_profile.py $N > gpurun_out/${tag}_proof.txt 2> gpurun_out/${tag}_perr.txt
f=$(find gpurun_out/pprof_$tag -name '*kernel_stats.csv' | head -1)
python3 - "$f" $N > gpurun_out/${tag}_pstats.txt <<'P'
import csv,sys
N=int(sys.argv[2])+1
tot=0
for r in csv.DictReader(open(sys.argv[1])):
    t=float(r['TotalDurationNs'])/1000/N; tot+=t
    print(r['Name'][:64].ljust(64), ("%.1f"%(int(r['Calls'])/N)).rjust(6), ("%.1f"%(float(r['AverageNs'])/1000)).rjust(9), ("%.1f us/proof"%t).rjust(16))
print("total GPU us per proof (incl. one-time setup kernels / N): %.1f"%tot)
P
cp "$f" gpurun_out/${tag}_proof_kernel_stats.csv
rm -rf gpurun_out/pprof_$tag
head -24 gpurun_out/${tag}_pstats.txt; tail -1 gpurun_out/${tag}_pstats.txt; cat gpurun_out/${tag}_proof.txt
