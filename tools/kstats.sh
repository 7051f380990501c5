#!/bin/bash
# usage (on the GPU box): bash tools/kstats.sh <tag>  -- rocprofv3 kernel stats of the short bench, table in gpurun_out/<tag>_stats.txt
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
tag=${1:-k}
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o run --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-proof --no-extras > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_err.txt
f=$(find gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1)
python3 - "$f" > gpurun_out/${tag}_stats.txt <<'P'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), ("%.1f"%(float(r['AverageNs'])/1000)).rjust(9), r['Percentage'])
P
cp "$f" gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/prof_$tag
cat gpurun_out/${tag}_stats.txt
