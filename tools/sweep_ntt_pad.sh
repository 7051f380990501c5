#!/bin/bash
# usage (on the GPU box): bash tools/sweep_ntt_pad.sh -- tuning build; the step with the NTT's LDS request padded (fewer NTT blocks per CU beside the MSM tails)
cd "$GRAFT_REPO_ROOT"
H2_BUILD_TUNING=1 python3 -m halo2_prover_amd.build --force > /dev/null 2>&1 || exit 1
for pad in 0 10000 30000 50000 80000; do
  echo "H2_TUNE_NTT_LDS_PAD=$pad"
  H2_TUNE_NTT_LDS_PAD=$pad python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-proof --no-extras 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read()); print("  step ms %.4f (median %.4f min %.4f)  msm phase %.4f ntt %.4f" % (d["ms_per_step"], d["ms_per_step_median"], d["ms_per_step_min"], d["phases_ms"]["msm"], d["phases_ms"]["ntt"]))'
done
