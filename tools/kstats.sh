#!/bin/bash
# usage (on the GPU box): bash tools/kstats.sh <tag> [bench.py arguments]
#   rocprofv3 kernel stats of a short bench run, table in gpurun_out/<tag>_stats.txt, CSV in gpurun_out/<tag>_kernel_stats.csv
#   default arguments: the Poseidon k=16 step.  Example: bash tools/kstats.sh msm20 --workload msm --k 20
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
tag=${1:-k}
shift
if [ $# -eq 0 ]; then set -- --steps 10 --warmup 2; fi
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o run --output-format csv -- python3 bench.py "$@" --no-cpu-baseline --no-proof --no-extras > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_err.txt
f=$(find gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1)
python3 - "$f" > gpurun_out/${tag}_stats.txt <<'P'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), ("%.1f"%(float(r['AverageNs'])/1000)).rjust(9), r['Percentage'])
P
cp "$f" gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/prof_$tag
cat gpurun_out/${tag}_stats.txt
