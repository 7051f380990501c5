#!/bin/bash
# usage (on the GPU box): bash tools/sweep_sort2.sh -- tuning build, then the k=16 step with the one-level (0) and the two-level (1) sort
cd "$GRAFT_REPO_ROOT"
H2_BUILD_TUNING=1 python3 -m halo2_prover_amd.build --force > /dev/null 2>&1 || exit 1
for v in 0 1; do
  echo "H2_TUNE_SORT2=$v"
  H2_TUNE_SORT2=$v python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-proof --no-extras 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read()); print("  step ms %.4f  chunk kernel %.4f  msm phase %.4f ntt %.4f" % (d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["phases_ms"]["msm"], d["phases_ms"]["ntt"]))'
done
H2_TUNE_SORT2=1 bash tools/kstats.sh sort2_k16 > /dev/null
cat gpurun_out/sort2_k16_stats.txt
