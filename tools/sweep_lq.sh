#!/bin/bash
# usage (on the GPU box): bash tools/sweep_lq.sh -- quads per key in the fix-up (H2_TUNE_LQ) with a prebuilt tuning library variants/lib_tune.so
cd "$GRAFT_REPO_ROOT"
cp halo2_prover_amd/libh2hip.so /tmp/libh2hip_keep.so
cp variants/lib_tune.so halo2_prover_amd/libh2hip.so
for round in 1 2; do
for lq in 1 2 3; do
  H2_TUNE_LQ=$lq python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-proof --no-extras 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read()); print("LQ='$lq' step: ms %.4f median %.4f chunk %.4f msm %.4f" % (d["ms_per_step"], d["ms_per_step_median"], d["roofline"]["avg_kernel_ms"], d["phases_ms"]["msm"]))'
done
done
cp /tmp/libh2hip_keep.so halo2_prover_amd/libh2hip.so
