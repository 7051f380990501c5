#!/bin/bash
# usage (on the GPU box): bash tools/sweep_tmax.sh  -- tuning build, then the 2^20 MSM and the k=16 step for several caps of
# the entries per accumulate thread (H2_TUNE_TMAX); leaves the tuning build in place (rebuild the product library after)
cd "$GRAFT_REPO_ROOT"
H2_BUILD_TUNING=1 python3 -m halo2_prover_amd.build --force > /dev/null 2>&1 || exit 1
for t in 64 86 96 128; do
  echo "TMAX=$t"
  H2_TUNE_TMAX=$t python3 bench.py --workload msm --k 20 --steps 10 --warmup 2 --no-cpu-baseline --no-proof --no-extras | python3 -c '
import json,sys
d=json.loads(sys.stdin.read()); print("  msm 2^20 ms/step %.4f  chunk kernel %.4f" % (d["ms_per_step"], d["roofline"]["avg_kernel_ms"]))'
done
for t in 64 96; do
  echo "TMAX=$t k16 step"
  H2_TUNE_TMAX=$t python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-proof --no-extras | python3 -c '
import json,sys
d=json.loads(sys.stdin.read()); print("  step ms %.4f  chunk kernel %.4f  msm phase %.4f ntt %.4f" % (d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["phases_ms"]["msm"], d["phases_ms"]["ntt"]))'
done
