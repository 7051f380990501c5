#!/bin/bash
run() { env "$@" python bench.py --workload ntt --k $K --ntt-cols $M --no-cpu-baseline --no-proof --no-extras 2>/dev/null | VV="k=$K m=$M $*" python -c '
import json, os, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(os.environ["VV"], "ms %.3f" % d["ms_per_step"])'; }
for cfg in "19 7" "19 1" "20 4" "16 7" "16 64" "18 64"; do
  set -- $cfg; K=$1; M=$2
  run H2_NOP=1
  run H2_TUNE_NTT32=1
done
