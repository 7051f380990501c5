#!/bin/bash
# usage (on the GPU box): bash tools/sweep_cwide.sh -- tuning build; MSMs of 2^22 .. 2^24 terms for window widths 16 .. 19 (H2_TUNE_C)
cd "$GRAFT_REPO_ROOT"
H2_BUILD_TUNING=1 python3 -m halo2_prover_amd.build --force > /dev/null 2>&1 || exit 1
for kc in "21 1" "22 1" "22 8" "23 1" "24 8"; do
  set -- $kc
  for c in 16 17 18 19; do
    echo "k=$1 cols=$2 H2_TUNE_C=$c"
    H2_TUNE_C=$c timeout -k 10 300 python3 bench.py --workload msm --k $1 --msm-cols $2 --steps 3 --warmup 1 --no-cpu-baseline --no-proof --no-extras 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read()); print("  ms %.3f  chunk %.3f  windows %d" % (d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["config"]["msm_windows"]))' || exit 1
  done
done
python3 -m halo2_prover_amd.build --force > /dev/null 2>&1
