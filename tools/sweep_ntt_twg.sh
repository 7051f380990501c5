#!/bin/bash
# usage (on the GPU box): bash tools/sweep_ntt_twg.sh -- tuning build; radix twiddles from LDS (0) or global memory (1) for every tile shape
cd "$GRAFT_REPO_ROOT"
H2_BUILD_TUNING=1 python3 -m halo2_prover_amd.build --force > /dev/null 2>&1 || exit 1
for v in -1 0 1; do
  echo "H2_TUNE_NTT_TWG=$v"
  for a in "19 7" "16 7" "16 64" "18 64" "20 4"; do H2_TUNE_NTT_TWG=$v python3 tools/time_ntt.py $a 2>/dev/null; done
done
