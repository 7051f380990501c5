#!/bin/bash
# NTT tile sweep (needs a tuning build: H2_BUILD_TUNING=1 python -m halo2_prover_amd.build --force): prints phases_ms.ntt of the bench step for several tile configurations
run() { env "$@" python bench.py --steps 5 --no-proof --no-cpu-baseline --no-extras 2>/dev/null | VV="$*" python -c '
import json, os, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(os.environ["VV"], "ntt", d["phases_ms"]["ntt"], "step", round(d["ms_per_step"], 3))'; }
run H2_NOP=1
run H2_TUNE_NTT_LC9=0
run H2_TUNE_NTT_LC9=2
run H2_TUNE_NTT_MAXR=9
run H2_TUNE_NTT_MAXR=9 H2_TUNE_NTT_LC9=2
run H2_TUNE_NTT_MAXR=8
run H2_TUNE_NTT_MAXR=8 H2_TUNE_NTT_LC=3
run H2_TUNE_NTT_MAXR=7 H2_TUNE_NTT_LC=3
run H2_TUNE_NTT_LC=3
