#!/bin/bash
# usage (on the GPU box): bash tools/pmc_sq.sh <tag> <bench args...>  -- SQ counters (one pass) of a bench run, per kernel averages
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
tag=$1; shift
rm -rf gpurun_out/sq_$tag
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d gpurun_out/sq_$tag -o run --output-format csv -- python3 bench.py "$@" --no-cpu-baseline --no-proof --no-extras > /dev/null 2> gpurun_out/sq_${tag}_err.txt || { tail -5 gpurun_out/sq_${tag}_err.txt; exit 1; }
python3 - gpurun_out/sq_$tag/run_counter_collection.csv > gpurun_out/sq_$tag.txt <<'P'
import csv, sys, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()[:48]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, c in acc.items():
    n = len(next(iter(c.values())))
    print(name, "launches", n, " ".join("%s=%.3g" % (k, sum(v) / len(v)) for k, v in sorted(c.items())))
P
cat gpurun_out/sq_$tag.txt
rm -rf gpurun_out/sq_$tag
