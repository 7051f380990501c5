#!/bin/bash
run() { env "$@" python bench.py --workload ntt --k $K --ntt-cols $M --no-cpu-baseline --no-proof --no-extras 2>/dev/null | VV="k=$K m=$M $*" python -c '
import json, os, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(os.environ["VV"], "ms %.3f" % d["ms_per_step"])'; }
for cfg in "19 7" "20 4" "16 7"; do
  set -- $cfg; K=$1; M=$2
  run H2_NOP=1
  run H2_TUNE_NTT_TWG=0
  run H2_TUNE_NTT_TWG=1
  run H2_TUNE_NTT_MAXR=9
  run H2_TUNE_NTT_LC9=0
  run H2_TUNE_NTT_LC9=2
done
