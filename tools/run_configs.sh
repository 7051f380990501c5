#!/bin/bash
# BASELINE.json configs 3-5, one GPU's share each, appended as bench.py JSON lines to $1 (default gpurun_out/cfg.jsonl)
set -e
O=${1:-gpurun_out/cfg.jsonl}
: > $O
run() { python bench.py "$@" --no-cpu-baseline --no-proof >> $O; tail -1 $O | python -c '
import json, sys
d = json.loads(sys.stdin.read())
print(d["config"]["workload"], "ms/step %.3f" % d["ms_per_step"], "field-ops/s %.3e" % d["value"])'; }
run --workload msm --k 20 --msm-cols 1
run --workload msm --k 20 --msm-cols 4
run --workload ntt --k 16 --ntt-cols 3
run --workload ntt --k 16 --ntt-cols 16
run --workload ntt --k 16 --ntt-cols 64
run --workload ntt --k 18 --ntt-cols 3
run --workload ntt --k 18 --ntt-cols 64
run --workload ntt --k 20 --ntt-cols 4
run --workload msm --k 24 --msm-cols 8 --steps 3 --warmup 1
run --workload ntt --k 24 --ntt-cols 8 --steps 3 --warmup 1
