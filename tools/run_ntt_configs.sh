#!/bin/bash
run() { python bench.py "$@" --no-cpu-baseline --no-proof --no-extras 2>/dev/null | python -c '
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d["config"]["workload"], "ms/step %.3f" % d["ms_per_step"], "field-ops/s %.3e" % d["value"])'; }
run --workload ntt --k 16 --ntt-cols 3
run --workload ntt --k 16 --ntt-cols 16
run --workload ntt --k 16 --ntt-cols 64
run --workload ntt --k 18 --ntt-cols 3
run --workload ntt --k 18 --ntt-cols 64
run --workload ntt --k 19 --ntt-cols 7
run --workload ntt --k 20 --ntt-cols 4
run --workload ntt --k 24 --ntt-cols 8 --steps 3 --warmup 1
