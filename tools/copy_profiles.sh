#!/bin/bash
# copy what tools/collect_round.sh a / b left in gpurun_out/ to profiles/ under the round's names: bash tools/copy_profiles.sh r02
r=${1:-r02}
cp gpurun_out/bench_pallas.json profiles/${r}_bench_poseidon_k16_pallas.json
cp gpurun_out/final_kernel_stats.csv profiles/${r}_bench_poseidon_k16_pallas_kernel_stats.csv
cp gpurun_out/final_bench.json profiles/${r}_bench_poseidon_k16_pallas_under_rocprof.json
cp gpurun_out/microbench_tail.txt profiles/${r}_microbench_tail.txt
cp gpurun_out/bench_bn254.json profiles/${r}_bench_poseidon_k16_bn254.json
cp gpurun_out/cfg.jsonl profiles/${r}_configs_3_4_5.jsonl
cp gpurun_out/proof_gen.json profiles/${r}_proof_gen_k16.json
cp gpurun_out/proof_trace.txt profiles/${r}_proof_trace_k16.txt
cp gpurun_out/bench_gpus2.json profiles/${r}_bench_gpus2_shared_gpu_rehearsal.json
cp gpurun_out/kc_proof_kernel_stats.csv profiles/${r}_proof_poseidon_k16_bn254_kernel_stats_key_cached.csv
cp gpurun_out/nokc_proof_kernel_stats.csv profiles/${r}_proof_poseidon_k16_bn254_kernel_stats.csv
cp gpurun_out/kc_pstats.txt profiles/${r}_proof_poseidon_k16_bn254_per_proof_key_cached.txt
