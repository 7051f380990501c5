#!/bin/bash
# usage (on the GPU box): bash tools/pmc_run.sh  -- the two PMC passes of the short bench + the stamped summary in gpurun_out/pmc_traffic.json
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
ARGS="--steps 3 --warmup 0 --no-cpu-baseline --no-proof --no-extras"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_f -o run --output-format csv -- python3 bench.py $ARGS > /dev/null 2> gpurun_out/pmc_f_err.txt || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_w -o run --output-format csv -- python3 bench.py $ARGS > /dev/null 2> gpurun_out/pmc_w_err.txt || exit 1
python3 tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_traffic.json 6 && python3 - <<'P'
import json
d = json.load(open("gpurun_out/pmc_traffic.json"))
for k, v in d.get("kernels", {}).items():
    print(k[:40].ljust(40), {a: (round(b) if isinstance(b, float) else b) for a, b in v.items()})
P
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
