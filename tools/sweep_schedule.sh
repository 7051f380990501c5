#!/bin/bash
# usage (on the GPU box): bash tools/sweep_schedule.sh [curve] -- the step under the placements of its challenge-free transforms
cd "$GRAFT_REPO_ROOT"
curve=${1:-pallas}
for s in tail early_tail early serial tail early_tail early; do
  python3 bench.py --curve $curve --schedule $s --steps 30 --warmup 3 --no-cpu-baseline --no-proof --no-extras 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read()); print("'$curve' '$s'", "step ms mean %.4f median %.4f min %.4f  msm %.4f ntt %.4f chunk %.4f" % (d["ms_per_step"], d["ms_per_step_median"], d["ms_per_step_min"], d["phases_ms"]["msm"], d["phases_ms"]["ntt"], d["roofline"]["avg_kernel_ms"]))'
done
