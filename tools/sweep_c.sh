#!/bin/bash
# usage (on the GPU box): bash tools/sweep_c.sh -- tuning build; the step and the 2^20 MSM for several window widths (H2_TUNE_C)
cd "$GRAFT_REPO_ROOT"
H2_BUILD_TUNING=1 python3 -m halo2_prover_amd.build --force > /dev/null 2>&1 || exit 1
for c in 11 12 13 14; do
  echo "H2_TUNE_C=$c (k=16 step)"
  H2_TUNE_C=$c python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-proof --no-extras 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read()); print("  step ms %.4f  msm phase %.4f  chunk %.4f  windows %d" % (d["ms_per_step"], d["phases_ms"]["msm"], d["roofline"]["avg_kernel_ms"], d["config"]["msm_windows"]))'
done
for c in 15 16; do
  echo "H2_TUNE_C=$c (2^20 MSM)"
  H2_TUNE_C=$c python3 bench.py --workload msm --k 20 --steps 10 --warmup 2 --no-cpu-baseline --no-proof --no-extras 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read()); print("  ms %.4f  chunk %.4f  windows %d" % (d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["config"]["msm_windows"]))'
done
