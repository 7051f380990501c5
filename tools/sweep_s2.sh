#!/bin/bash
# usage (on the GPU box): bash tools/sweep_s2.sh -- tuning build; the two-level sort's knobs on the 2^20 MSM (kernel stats per setting)
cd "$GRAFT_REPO_ROOT"
H2_BUILD_TUNING=1 python3 -m halo2_prover_amd.build --force > /dev/null 2>&1 || exit 1
run() {
  echo "== $*"
  env "$@" bash tools/kstats.sh sw --workload msm --k 20 --steps 6 --warmup 1 | grep -E "msm2_|chunk"
}
run H2_TUNE_S2_GROUP=4
run H2_TUNE_S2_GROUP=2
run H2_TUNE_S2_GROUP=1
run H2_TUNE_S2_GROUP=2 H2_TUNE_S2_STAGE=11264
run H2_TUNE_S2_GROUP=2 H2_TUNE_S2_STAGE=8192
