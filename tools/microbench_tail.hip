// microbench_tail.hip -- what does one point addition cost in the MSM's tail?
//
// Dependent chains of XYZZ additions on the working form, in registers (no memory in the loop):
//   quad  = 4 lanes per point (h2_curve_quad.hpp), lane = one lane per point (h2_curve29.hpp)
// at 1 / 2 / 4 waves per SIMD.  Prints microseconds per addition (latency of the chain) and chip-wide point
// additions per second.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o microbench_tail tools/microbench_tail.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define H2_TAIL_STAMPS 1
#include "../halo2_prover_amd/csrc/h2_msm.hpp"

using namespace h2;
using CV = PALLAS_CURVE;

__device__ __forceinline__ Xyzz29<CV> seed_point(uint32_t salt) {
  // not a curve point -- the formulas do not care; distinct x so that no exceptional case is hit
  Xyzz29<CV> p;
  for (int i = 0; i < 9; i++) {
    p.x.v[i] = (int32_t)((0x1234567u * (i + 1) + salt * 2654435761u) & L29_MASK);
    p.y.v[i] = (int32_t)((0x7654321u * (i + 3) + salt * 40503u) & L29_MASK);
    p.zz.v[i] = (int32_t)((0x3141592u * (i + 5) + salt * 7919u) & L29_MASK);
    p.zzz.v[i] = (int32_t)((0x2718281u * (i + 7) + salt * 104729u) & L29_MASK);
  }
  p.x.v[8] &= 0xFFFFF; p.y.v[8] &= 0xFFFFF; p.zz.v[8] &= 0xFFFFF; p.zzz.v[8] &= 0xFFFFF;
  return p;
}

__global__ void __launch_bounds__(256) k_quad(uint32_t* out, int iters) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  Xyzz29<CV> a = seed_point(t >> 2), b = seed_point((t >> 2) + 77777u);
  for (int k = 0; k < iters; k++) a = xyzz29_add_quad(a, b);
  if ((t & 3) == 0) xyzz29_store<CV>(out + XYZZ29_WORDS * (size_t)(t >> 2), a);
}
__global__ void __launch_bounds__(256) k_quad_dbl(uint32_t* out, int iters) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  Xyzz29<CV> a = seed_point(t >> 2);
  for (int k = 0; k < iters; k++) a = xyzz29_double_quad(a);
  if ((t & 3) == 0) xyzz29_store<CV>(out + XYZZ29_WORDS * (size_t)(t >> 2), a);
}
__global__ void __launch_bounds__(256) k_lane(uint32_t* out, int iters) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  Xyzz29<CV> a = seed_point(t), b = seed_point(t + 77777u);
  for (int k = 0; k < iters; k++) a = xyzz29_add(a, b);
  xyzz29_store<CV>(out + XYZZ29_WORDS * (size_t)t, a);
}
__global__ void __launch_bounds__(256) k_lane_mixed(uint32_t* out, int iters) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  Xyzz29<CV> a = seed_point(t);
  const Xyzz29<CV> b = seed_point(t + 77777u);
  const Affine29<CV> q{b.x, b.y};
  for (int k = 0; k < iters; k++) a = xyzz29_add_affine(a, q);
  xyzz29_store<CV>(out + XYZZ29_WORDS * (size_t)t, a);
}

__global__ void k_fill(uint32_t* pts, uint32_t count) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < count) xyzz29_store<CV>(pts + XYZZ29_WORDS * (size_t)t, seed_point(t));
}

// the row / column tail of the MSM on synthetic bucket sums, stage by stage (100 MHz stamps from the kernels)
static void tail_stages() {
  const uint32_t m = 4, log_b = 12, lb = 6, B = 1u << log_b, rcn = (1u << (log_b - lb)) + (1u << lb);
  uint32_t *xsum, *rc, *part, *done, *out;
  U128* jac;
  unsigned long long* stamps;
  hipMalloc(&xsum, (size_t)m * B * XYZZ29_WORDS * 4);
  hipMalloc(&rc, (size_t)m * rcn * XYZZ29_WORDS * 4);
  hipMalloc(&part, (size_t)m * MSM_FINAL_MAX_BLOCKS * XYZZ29_WORDS * 4);
  const uint32_t nb = msm_final_blocks(log_b, lb);
  hipMalloc(&done, m * 4);
  hipMalloc(&out, m * XYZZ29_WORDS * 4);
  hipMalloc(&jac, m * 96);
  hipMalloc(&stamps, 16 * 8 * 1024);
  hipMemset(stamps, 0, 16 * 8 * 1024);
  hipMemcpyToSymbol(HIP_SYMBOL(h2_stamps), &stamps, sizeof(stamps));
  hipLaunchKernelGGL(k_fill, dim3((m * B + 255) / 256), dim3(256), 0, 0, xsum, m * B);
  std::vector<unsigned long long> h(16 * nb * m);
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(msm_rowcol_kernel<CV>, dim3(rcn, m), dim3(64 * msm_rowcol_waves(log_b, lb, m)), 0, 0, xsum, rc, done, log_b, lb);
    hipDeviceSynchronize();
    hipMemset(stamps, 0, 16 * 8 * 1024);
    hipLaunchKernelGGL(msm_final_kernel<CV>, dim3(nb, m), dim3(64), 0, 0, rc, part, done, out, jac, log_b, lb);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull;
    for (size_t b = 0; b < nb * m; b++) if (h[b * 16] && h[b * 16] < t0) t0 = h[b * 16];
    printf("  msm_final_kernel stages, run %d (us since the first wave started; row-family blocks first, column 0):\n", rep);
    for (uint32_t b = 0; b < nb; b++) {
      printf("    block %u:", b);
      for (int i = 0; i < 9; i++) {
        if (h[b * 16 + i]) printf(" %7.2f", (double)(h[b * 16 + i] - t0) / 100.0);
        else printf("       -");
      }
      printf("\n");
    }
  }
  printf("  stamps: 0 start, 2 weights and wave tree done, 3 counted, 4 partials loaded, 7 partials added, 8 stored\n");
}

template <class F>
double time_kernel(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

#include <vector>
int main() {
  tail_stages();
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs\n", prop.name, cus);
  void* buf;
  hipMalloc(&buf, (size_t)cus * 8 * 256 * XYZZ29_WORDS * 4);
  const int iters = 64;
  for (int w : {1, 2, 4}) {
    const int blocks = cus * w;                   // 256 threads = 4 waves = one wave per SIMD per unit of w
    const double waves = blocks * 4.0;
    auto rep = [&](const char* what, double ms, double points_per_wave) {
      printf("  %-28s %d wave(s)/SIMD  %7.3f ms  %6.2f us/op (chain)  %8.2f G point-ops/s\n", what, w, ms,
             ms * 1e3 / iters, waves * points_per_wave * iters / (ms * 1e-3) / 1e9);
    };
    rep("quad add (4 lanes/point)", time_kernel([&] { hipLaunchKernelGGL(k_quad, dim3(blocks), dim3(256), 0, 0, (uint32_t*)buf, iters); }), 16);
    rep("quad double", time_kernel([&] { hipLaunchKernelGGL(k_quad_dbl, dim3(blocks), dim3(256), 0, 0, (uint32_t*)buf, iters); }), 16);
    rep("lane add (full XYZZ)", time_kernel([&] { hipLaunchKernelGGL(k_lane, dim3(blocks), dim3(256), 0, 0, (uint32_t*)buf, iters); }), 64);
    rep("lane mixed add (affine)", time_kernel([&] { hipLaunchKernelGGL(k_lane_mixed, dim3(blocks), dim3(256), 0, 0, (uint32_t*)buf, iters); }), 64);
  }
  hipFree(buf);
  return 0;
}
