// microbench_valu.hip -- measured integer / FP64 VALU ceilings on gfx950 for the 256-bit modmul.
// SURVEY.md section 8(d) asks for a measured v_mad_u64_u32 ceiling before quoting "% of peak".
// Build: hipcc --offload-arch=gfx950 -O3 -o microbench_valu tools/microbench_valu.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int UNROLL = 8;

__global__ void k_mad(uint64_t* out, uint32_t a0, uint32_t b0, int iters) {
  uint32_t a = a0 + threadIdx.x, b = b0 + blockIdx.x;
  uint64_t acc[UNROLL];
  for (int i = 0; i < UNROLL; i++) acc[i] = i;
  for (int k = 0; k < iters; k++) {
#pragma unroll
    for (int i = 0; i < UNROLL; i++) acc[i] = (uint64_t)a * (uint32_t)(b + i) + acc[i];
    a += (uint32_t)acc[0];
  }
  uint64_t s = 0;
  for (int i = 0; i < UNROLL; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_add32(uint64_t* out, uint32_t a0, int iters) {
  uint32_t acc[UNROLL];
  for (int i = 0; i < UNROLL; i++) acc[i] = a0 + i + threadIdx.x;
  for (int k = 0; k < iters; k++) {
#pragma unroll
    for (int i = 0; i < UNROLL; i++) acc[i] = acc[i] * 1u + (acc[(i + 1) % UNROLL] ^ k);  // xor + add: 2 ops
  }
  uint32_t s = 0;
  for (int i = 0; i < UNROLL; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_add64(uint64_t* out, uint64_t a0, int iters) {
  uint64_t acc[UNROLL];
  for (int i = 0; i < UNROLL; i++) acc[i] = a0 + i + threadIdx.x;
  for (int k = 0; k < iters; k++) {
#pragma unroll
    for (int i = 0; i < UNROLL; i++) acc[i] = acc[i] + acc[(i + 1) % UNROLL];
  }
  uint64_t s = 0;
  for (int i = 0; i < UNROLL; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mullo(uint64_t* out, uint32_t a0, int iters) {
  uint32_t acc[UNROLL];
  for (int i = 0; i < UNROLL; i++) acc[i] = a0 + i + threadIdx.x;
  for (int k = 0; k < iters; k++) {
#pragma unroll
    for (int i = 0; i < UNROLL; i++) acc[i] = acc[i] * (acc[(i + 1) % UNROLL] | 1u);
  }
  uint32_t s = 0;
  for (int i = 0; i < UNROLL; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_dfma(uint64_t* out, double a0, int iters) {
  double acc[UNROLL];
  double m = a0 + threadIdx.x * 1e-9;
  for (int i = 0; i < UNROLL; i++) acc[i] = a0 + i;
  for (int k = 0; k < iters; k++) {
#pragma unroll
    for (int i = 0; i < UNROLL; i++) acc[i] = __fma_rn(acc[i], m, 1.0);
  }
  double s = 0;
  for (int i = 0; i < UNROLL; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}
__global__ void k_ffma(uint64_t* out, float a0, int iters) {
  float acc[UNROLL];
  float m = a0 + threadIdx.x * 1e-6f;
  for (int i = 0; i < UNROLL; i++) acc[i] = a0 + i;
  for (int k = 0; k < iters; k++) {
#pragma unroll
    for (int i = 0; i < UNROLL; i++) acc[i] = __fmaf_rn(acc[i], m, 1.0f);
  }
  float s = 0;
  for (int i = 0; i < UNROLL; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}

// the library's own multiplier, dependent chain
#include "../halo2_prover_amd/csrc/h2_field.hpp"
template <class FP>
__global__ void k_modmul(uint64_t* out, int iters) {
  h2::Fe<FP> a = h2::Fe<FP>::one(), b = h2::Fe<FP>::one();
  a.v[0] += threadIdx.x; b.v[1] += blockIdx.x + 3;
  for (int k = 0; k < iters; k++) a = h2::fe_mul(a, b);
  out[blockIdx.x * blockDim.x + threadIdx.x] = a.v[0] | ((uint64_t)a.v[7] << 32);
}
template <class FP>
__global__ void k_modmul2(uint64_t* out, int iters) {   // two independent chains per thread (ILP)
  h2::Fe<FP> a = h2::Fe<FP>::one(), b = h2::Fe<FP>::one(), c = h2::Fe<FP>::one();
  a.v[0] += threadIdx.x; b.v[1] += blockIdx.x + 3; c.v[2] += threadIdx.x * 7;
  for (int k = 0; k < iters; k++) { a = h2::fe_mul(a, b); c = h2::fe_mul(c, b); }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a.v[0] ^ c.v[3];
}

#include "comba_prototype.hpp"
template <class FP>
__global__ void k_modmul_comba(uint64_t* out, int iters) {
  h2::Fe<FP> a = h2::Fe<FP>::one(), b = h2::Fe<FP>::one();
  a.v[0] += threadIdx.x; b.v[1] += blockIdx.x + 3;
  for (int k = 0; k < iters; k++) a = h2x::fe_mul_comba(a, b);
  out[blockIdx.x * blockDim.x + threadIdx.x] = a.v[0] | ((uint64_t)a.v[7] << 32);
}
// correctness: compare comba with the CIOS multiplier on pseudo-random inputs
template <class FP>
__global__ void k_check(uint32_t* bad) {
  h2::Fe<FP> a = h2::Fe<FP>::one(), b = h2::Fe<FP>::one();
  a.v[0] += threadIdx.x * 2654435761u; a.v[3] ^= blockIdx.x * 40503u; b.v[1] += blockIdx.x + 3; b.v[5] ^= threadIdx.x;
  for (int k = 0; k < 64; k++) {
    h2::Fe<FP> x = h2::fe_mul(a, b), y = h2x::fe_mul_comba(a, b);
    if (x != y) atomicAdd(bad, 1u);
    b = a; a = x;
    a.v[2] ^= k;
    if (k == 7) { for (int i = 0; i < 8; i++) a.v[i] = FP::P(i); a.v[0] -= 1; }      // p - 1
    if (k == 9) { for (int i = 0; i < 8; i++) b.v[i] = FP::P(i); b.v[0] -= 1; }
    if (k == 11) { for (int i = 0; i < 8; i++) a.v[i] = 0xffffffffu; a.v[7] = FP::P(7) - 1; }  // large limbs
  }
}

template <class F>
double time_kernel(F launch, int reps = 3) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double clk = prop.clockRate * 1e3;  // Hz
  printf("device %s, %d CUs, clock %.0f MHz\n", prop.name, cus, clk / 1e6);
  uint64_t* out;
  CHECK(hipMalloc(&out, (size_t)cus * 64 * 1024 * 8));
  const int iters = 4096;
  struct Cfg { const char* name; int blocks; int threads; };
  Cfg cfgs[] = {{"1 wave total", 1, 64}, {"1 wave/SIMD", cus, 256}, {"2 waves/SIMD", cus * 2, 256},
                {"4 waves/SIMD", cus * 4, 256}, {"8 waves/SIMD", cus * 8, 256}};
  for (auto& c : cfgs) {
    const double waves = (double)c.blocks * c.threads / 64.0;
    const double simds = c.blocks == 1 ? 1 : cus * 4.0;
    auto report = [&](const char* what, double ms, double ops_per_thread_iter) {
      const double wave_instr = waves * iters * ops_per_thread_iter;
      const double cyc_per_instr_per_simd = ms * 1e-3 * clk * simds / wave_instr;
      printf("  %-28s %-14s %8.3f ms  %6.2f cycles/wave-instr/SIMD  %8.2f G lane-ops/s\n", what, c.name, ms,
             cyc_per_instr_per_simd, wave_instr * 64 / (ms * 1e-3) / 1e9);
    };
    report("v_mad_u64_u32", time_kernel([&] { hipLaunchKernelGGL(k_mad, dim3(c.blocks), dim3(c.threads), 0, 0, out, 3u, 5u, iters); }), UNROLL);
    report("v_mul_lo_u32", time_kernel([&] { hipLaunchKernelGGL(k_mullo, dim3(c.blocks), dim3(c.threads), 0, 0, out, 3u, iters); }), UNROLL * 2);
    report("v_xor+v_add_u32", time_kernel([&] { hipLaunchKernelGGL(k_add32, dim3(c.blocks), dim3(c.threads), 0, 0, out, 3u, iters); }), UNROLL * 2);
    report("v_lshl_add_u64", time_kernel([&] { hipLaunchKernelGGL(k_add64, dim3(c.blocks), dim3(c.threads), 0, 0, out, 3ull, iters); }), UNROLL);
    report("v_fma_f64", time_kernel([&] { hipLaunchKernelGGL(k_dfma, dim3(c.blocks), dim3(c.threads), 0, 0, out, 1.0000001, iters); }), UNROLL);
    report("v_fma_f32", time_kernel([&] { hipLaunchKernelGGL(k_ffma, dim3(c.blocks), dim3(c.threads), 0, 0, out, 1.0000001f, iters); }), UNROLL);
    const int mi = 512;
    auto report_mul = [&](const char* what, double ms, double muls) {
      const double total = waves * 64 * mi * muls;
      printf("  %-28s %-14s %8.3f ms  %8.1f cycles/wave-modmul/SIMD  %8.2f G modmul/s\n", what, c.name, ms,
             ms * 1e-3 * clk * simds / (waves * mi * muls), total / (ms * 1e-3) / 1e9);
    };
    report_mul("fe_mul<BN254_FQ> chain", time_kernel([&] { hipLaunchKernelGGL(k_modmul<h2::BN254_FQ>, dim3(c.blocks), dim3(c.threads), 0, 0, out, mi); }), 1);
    report_mul("fe_mul<PASTA_FP> chain", time_kernel([&] { hipLaunchKernelGGL(k_modmul<h2::PASTA_FP>, dim3(c.blocks), dim3(c.threads), 0, 0, out, mi); }), 1);
    report_mul("comba<BN254_FQ> chain", time_kernel([&] { hipLaunchKernelGGL(k_modmul_comba<h2::BN254_FQ>, dim3(c.blocks), dim3(c.threads), 0, 0, out, mi); }), 1);
    report_mul("comba<PASTA_FP> chain", time_kernel([&] { hipLaunchKernelGGL(k_modmul_comba<h2::PASTA_FP>, dim3(c.blocks), dim3(c.threads), 0, 0, out, mi); }), 1);
    report_mul("fe_mul<PASTA_FP> 2 chains", time_kernel([&] { hipLaunchKernelGGL(k_modmul2<h2::PASTA_FP>, dim3(c.blocks), dim3(c.threads), 0, 0, out, mi); }), 2);
  }
  uint32_t* bad;
  CHECK(hipMalloc(&bad, 4));
  CHECK(hipMemset(bad, 0, 4));
  hipLaunchKernelGGL(k_check<h2::BN254_FQ>, dim3(64), dim3(256), 0, 0, bad);
  hipLaunchKernelGGL(k_check<h2::BN254_FR>, dim3(64), dim3(256), 0, 0, bad);
  hipLaunchKernelGGL(k_check<h2::PASTA_FP>, dim3(64), dim3(256), 0, 0, bad);
  hipLaunchKernelGGL(k_check<h2::PASTA_FQ>, dim3(64), dim3(256), 0, 0, bad);
  uint32_t hbad = 1;
  CHECK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
  printf("comba vs CIOS mismatches: %u\n", hbad);
  hipFree(out);
  return 0;
}
