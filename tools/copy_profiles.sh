#!/bin/bash
# copy what tools/collect_round.sh a / b / c left in gpurun_out/ to profiles/ under the round's names: bash tools/copy_profiles.sh r03
r=${1:-r03}
c() { [ -s "$1" ] && cp "$1" "$2"; }
c gpurun_out/bench_pallas.json profiles/${r}_bench_poseidon_k16_pallas.json
c gpurun_out/final_kernel_stats.csv profiles/${r}_bench_poseidon_k16_pallas_kernel_stats.csv
c gpurun_out/final_bench.json profiles/${r}_bench_poseidon_k16_pallas_under_rocprof.json
c gpurun_out/msm20_kernel_stats.csv profiles/${r}_msm_2e20_pallas_kernel_stats.csv
c gpurun_out/msm20_bench.json profiles/${r}_msm_2e20_pallas_under_rocprof.json
c gpurun_out/microbench_tail.txt profiles/${r}_microbench_tail.txt
c gpurun_out/bench_bn254.json profiles/${r}_bench_poseidon_k16_bn254.json
c gpurun_out/cfg.jsonl profiles/${r}_configs_3_4_5.jsonl
c gpurun_out/proof_gen.json profiles/${r}_proof_gen_k16.json
c gpurun_out/proof_trace.txt profiles/${r}_proof_trace_k16.txt
c gpurun_out/proof_gen_two_contexts.json profiles/${r}_proof_gen_k16_two_contexts_one_gpu.json
c gpurun_out/all_circuits_k16.json profiles/${r}_proof_all_circuits_k16.json
c gpurun_out/bench_gpus2.json profiles/${r}_bench_gpus2_shared_gpu_rehearsal.json
c gpurun_out/kc_proof_kernel_stats.csv profiles/${r}_proof_poseidon_k16_bn254_kernel_stats_key_cached.csv
c gpurun_out/nokc_proof_kernel_stats.csv profiles/${r}_proof_poseidon_k16_bn254_kernel_stats.csv
c gpurun_out/kc_pstats.txt profiles/${r}_proof_poseidon_k16_bn254_per_proof_key_cached.txt
c gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
c gpurun_out/ntt_times.txt profiles/${r}_ntt_times.txt
c gpurun_out/sweep_tmax.txt profiles/${r}_sweep_entries_per_thread.txt
c gpurun_out/sweep_s2.txt profiles/${r}_sweep_two_level_sort.txt
c gpurun_out/sweep_sort2.txt profiles/${r}_sweep_two_level_sort_at_k16.txt
c gpurun_out/sq_ntt19.txt profiles/${r}_sq_counters_ntt_7x2e19.txt
true
