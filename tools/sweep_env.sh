#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ... -- runs the default bench step with VAR=v and prints ms/step and the phase times
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v python bench.py --steps 10 --no-proof --no-cpu-baseline --no-extras 2>/dev/null | VV="$VAR=$v" python -c '
import json, os, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(os.environ["VV"], round(d["ms_per_step"], 3), round(d["ms_per_step_min"], 3), d["phases_ms"])'
done
