#!/bin/bash
# usage (on the GPU box): bash tools/collect_round.sh <part>  -- the round's profile artefacts into gpurun_out/ (copy them to profiles/ afterwards)
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
set -e
case "$1" in
a)
  python3 bench.py > gpurun_out/bench_pallas.json 2> gpurun_out/bench_pallas.err
  echo "bench done"
  bash tools/kstats.sh final > /dev/null
  echo "kstats done"
  timeout -k 10 120 ./halo2_prover_amd/build/microbench_tail > gpurun_out/microbench_tail.txt
  python3 bench.py --curve bn254 --no-cpu-baseline > gpurun_out/bench_bn254.json 2> gpurun_out/bench_bn254.err
  echo "bn254 done"
  ;;
b)
  bash tools/run_configs.sh > gpurun_out/configs.jsonl 2> gpurun_out/configs.err
  echo "configs done"
  H2_TRACE=1 python3 tools/proof_bench.py > gpurun_out/proof_gen.json 2> gpurun_out/proof_trace.txt
  echo "proof done"
  H2_PROFILE_KEY_CACHE=1 bash tools/pstats.sh kc 10 > /dev/null
  bash tools/pstats.sh nokc 10 > /dev/null
  python3 bench.py --gpus 2 --no-proof > gpurun_out/bench_gpus2.json 2> gpurun_out/bench_gpus2.err
  echo "gpus2 done"
  ;;
esac
