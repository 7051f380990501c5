#!/bin/bash
# usage (on the GPU box): bash tools/collect_round.sh <part>  -- the round's profile artefacts into gpurun_out/ (copy them to profiles/ afterwards: tools/copy_profiles.sh)
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
set -e
case "$1" in
a)
  python3 bench.py > gpurun_out/bench_pallas.json 2> gpurun_out/bench_pallas.err
  echo "bench done"
  bash tools/kstats.sh final > /dev/null
  echo "kstats done"
  bash tools/kstats.sh msm20 --workload msm --k 20 --steps 10 --warmup 2 > /dev/null
  echo "msm20 kstats done"
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
  python3 tools/proof_bench.py 16 0,0 > gpurun_out/proof_gen_two_contexts.json 2> /dev/null
  python3 tools/all_circuits_bench.py > gpurun_out/all_circuits_k16.json 2> /dev/null
  timeout -k 10 400 python3 bench.py --gpus 2 --no-cpu-baseline > gpurun_out/bench_gpus2.json 2> gpurun_out/bench_gpus2.err
  echo "gpus2 done"
  ;;
c)
  bash tools/pmc_run.sh
  for a in "19 7" "16 7" "19 1" "16 64" "18 64" "20 4"; do python3 tools/time_ntt.py $a 2>/dev/null; done > gpurun_out/ntt_times.txt
  cat gpurun_out/ntt_times.txt
  ;;
esac
