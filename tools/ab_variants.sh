#!/bin/bash
# usage (on the GPU box): bash tools/ab_variants.sh -- A/B of prebuilt library variants (variants/lib_*.so): 2^20 MSM, 2^22 MSM and the k=16 step
cd "$GRAFT_REPO_ROOT"
cp halo2_prover_amd/libh2hip.so /tmp/libh2hip_keep.so
for round in 1 2; do
for v in variants/lib_*.so; do
  cp $v halo2_prover_amd/libh2hip.so
  touch halo2_prover_amd/libh2hip.so
  for kk in 20 22; do
  H2_NO_BUILD=1 python3 bench.py --workload msm --k $kk --steps 10 --warmup 2 --no-cpu-baseline --no-proof --no-extras 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read()); print("'$v' msm 2^'$kk': ms %.4f  chunk %.4f" % (d["ms_per_step"], d["roofline"]["avg_kernel_ms"]))'
  done
  H2_NO_BUILD=1 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-proof --no-extras 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read()); print("'$v' step: ms %.4f median %.4f chunk %.4f msm %.4f" % (d["ms_per_step"], d["ms_per_step_median"], d["roofline"]["avg_kernel_ms"], d["phases_ms"]["msm"]))'
done
done
cp /tmp/libh2hip_keep.so halo2_prover_amd/libh2hip.so
