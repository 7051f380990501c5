// microbench_limb29.hip -- is an UNSATURATED-limb Montgomery product faster on gfx950 than the saturated one?
//
// The library's multiplier (h2_field.hpp, fe_mul_comba) works on 8 x 32-bit limbs: every v_mad_u64_u32 needs a
// v_addc_co_u32 to count the carry out of its 64-bit column accumulator, and the column hand-over costs three moves.
// With 9 limbs of 29 bits a column of a*b + m*p is at most 18 products below 2^60: it cannot overflow 64 bits, so
// there is no carry counter, the hand-over is one 64-bit shift, and with R' = 2^261 the result of a product of two
// values below 8p is below 2p WITHOUT a conditional subtraction.  Price: 81 + (non-trivial modulus limbs) * 9
// multiply-adds instead of 64 + 24 (Pasta) / 64 + 64 (BN254).
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o microbench_limb29 tools/microbench_limb29.hip ; run on the GPU
// box.  Prints G modmul/s at 1 .. 8 waves per SIMD for both forms, and sample (a, b, r) triples that
// tools/check_limb29.py verifies against big integers (r = a b 2^-261 mod p, r < 2p).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../halo2_prover_amd/csrc/h2_field.hpp"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

namespace l29 {
constexpr uint32_t MASK = (1u << 29) - 1;

// limb j (29 bits) of the modulus, from its 32-bit limbs
template <class FP>
constexpr uint32_t P29(int j) {
  const int bit = 29 * j;
  const int w = bit / 32, s = bit % 32;
  uint64_t lo = w < 8 ? FP::P(w) : 0, hi = w + 1 < 8 ? FP::P(w + 1) : 0;
  return (uint32_t)(((lo | (hi << 32)) >> s) & MASK);
}
template <class FP>
constexpr uint32_t INV29() { return FP::INV & MASK; }   // -p^-1 mod 2^29

template <class FP, int J>
__device__ __forceinline__ void mac_p(uint64_t& acc, uint32_t m) {
  constexpr uint32_t pj = P29<FP>(J);
  if constexpr (pj == 0) {
  } else if constexpr (pj == 1) {
    acc += m;
  } else if constexpr ((pj & (pj - 1)) == 0) {
    acc += (uint64_t)m << __builtin_ctz(pj);
  } else {
    acc += (uint64_t)m * pj;
  }
}
template <class FP, int K, int I, int IEND>
__device__ __forceinline__ void col_ab(uint64_t& acc, const uint32_t* a, const uint32_t* b) {
  if constexpr (I <= IEND) {
    acc += (uint64_t)a[I] * b[K - I];
    col_ab<FP, K, I + 1, IEND>(acc, a, b);
  }
}
template <class FP, int K, int I, int IEND>
__device__ __forceinline__ void col_mp(uint64_t& acc, const uint32_t* m) {
  if constexpr (I <= IEND) {
    mac_p<FP, K - I>(acc, m[I]);
    col_mp<FP, K, I + 1, IEND>(acc, m);
  }
}
template <class FP, int K>
__device__ __forceinline__ void columns(uint64_t& acc, const uint32_t* a, const uint32_t* b, uint32_t* m, uint32_t* t) {
  if constexpr (K < 17) {
    col_ab<FP, K, (K < 9 ? 0 : K - 8), (K < 9 ? K : 8)>(acc, a, b);
    if constexpr (K < 9) {
      if constexpr (K > 0) col_mp<FP, K, 0, K - 1>(acc, m);
      m[K] = ((uint32_t)acc * INV29<FP>()) & MASK;
      mac_p<FP, 0>(acc, m[K]);
    } else {
      col_mp<FP, K, K - 8, 8>(acc, m);
      t[K - 9] = (uint32_t)acc & MASK;
    }
    acc >>= 29;
    columns<FP, K + 1>(acc, a, b, m, t);
  }
}
// r = a b 2^-261 mod p (not fully reduced: r < 2p for a, b < 8p); limbs of r below 2^29 except the top one
template <class FP>
__device__ __forceinline__ void mul(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  uint64_t acc = 0;
  uint32_t m[9], t[9];
  columns<FP, 0>(acc, a, b, m, t);
  t[8] = (uint32_t)acc;
#pragma unroll
  for (int i = 0; i < 9; i++) r[i] = t[i];
}
}  // namespace l29

template <class FP>
__global__ void k_mul29(uint32_t* out, int iters) {
  uint32_t a[9], b[9];
#pragma unroll
  for (int i = 0; i < 9; i++) { a[i] = (0x1234567u * (i + 1) + threadIdx.x * 2654435761u) & l29::MASK; b[i] = (0x7654321u * (i + 3) + blockIdx.x * 40503u) & l29::MASK; }
  a[8] &= 0xFFFFF; b[8] &= 0xFFFFF;      // values below 2^252
  for (int k = 0; k < iters; k++) l29::mul<FP>(a, a, b);
  uint32_t* o = out + 9 * (size_t)(blockIdx.x * blockDim.x + threadIdx.x);
#pragma unroll
  for (int i = 0; i < 9; i++) o[i] = a[i];
}
// one product, inputs and output written for the host check
template <class FP>
__global__ void k_sample29(uint32_t* out) {
  uint32_t a[9], b[9], r[9];
  for (int i = 0; i < 9; i++) { a[i] = (0x1234567u * (i + 1) + threadIdx.x * 2654435761u) & l29::MASK; b[i] = (0x7654321u * (i + 3) + threadIdx.x * 40503u) & l29::MASK; }
  a[8] &= 0x1FFFFFF; b[8] &= 0x1FFFFFF;  // up to 2^257 (about 8p): the laziest inputs the bound allows
  l29::mul<FP>(r, a, b);
  uint32_t* o = out + 27 * threadIdx.x;
  for (int i = 0; i < 9; i++) { o[i] = a[i]; o[9 + i] = b[i]; o[18 + i] = r[i]; }
}
template <class FP>
__global__ void k_mul32(uint64_t* out, int iters) {
  h2::Fe<FP> a = h2::Fe<FP>::one(), b = h2::Fe<FP>::one();
  a.v[0] += threadIdx.x; b.v[1] += blockIdx.x + 3;
  for (int k = 0; k < iters; k++) a = h2::fe_mul(a, b);
  out[blockIdx.x * blockDim.x + threadIdx.x] = a.v[0] | ((uint64_t)a.v[7] << 32);
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

template <class FP>
int bench(const char* name, int cus, double clk, void* buf) {
  const int mi = 512;
  struct Cfg { const char* name; int blocks; };
  Cfg cfgs[] = {{"1 wave/SIMD", cus}, {"2 waves/SIMD", cus * 2}, {"4 waves/SIMD", cus * 4}, {"8 waves/SIMD", cus * 8}};
  for (auto& c : cfgs) {
    const double waves = (double)c.blocks * 4;
    auto rep = [&](const char* what, double ms) {
      printf("  %-10s %-22s %-13s %8.3f ms %8.1f cycles/wave-modmul/SIMD %8.2f G modmul/s\n", name, what, c.name, ms,
             ms * 1e-3 * clk * cus * 4.0 / (waves * mi), waves * 64 * mi / (ms * 1e-3) / 1e9);
    };
    rep("8 x 32-bit (library)", time_kernel([&] { hipLaunchKernelGGL(k_mul32<FP>, dim3(c.blocks), dim3(256), 0, 0, (uint64_t*)buf, mi); }));
    rep("9 x 29-bit, R'=2^261", time_kernel([&] { hipLaunchKernelGGL(k_mul29<FP>, dim3(c.blocks), dim3(256), 0, 0, (uint32_t*)buf, mi); }));
  }
  return 0;
}

template <class FP>
void samples(const char* name, uint32_t* dbuf) {
  static uint32_t h[27 * 64];
  hipLaunchKernelGGL(k_sample29<FP>, dim3(1), dim3(64), 0, 0, dbuf);
  hipMemcpy(h, dbuf, sizeof h, hipMemcpyDeviceToHost);
  for (int t = 0; t < 64; t += 9) {
    printf("SAMPLE %s", name);
    for (int i = 0; i < 27; i++) printf(" %x", h[27 * t + i]);
    printf("\n");
  }
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double clk = prop.clockRate * 1e3;
  printf("device %s, %d CUs, clock %.0f MHz\n", prop.name, cus, clk / 1e6);
  void* buf;
  CHECK(hipMalloc(&buf, (size_t)cus * 8 * 256 * 9 * 4 + 4096));
  bench<h2::PASTA_FP>("pasta_fp", cus, clk, buf);
  bench<h2::BN254_FQ>("bn254_fq", cus, clk, buf);
  samples<h2::PASTA_FP>("pasta_fp", (uint32_t*)buf);
  samples<h2::PASTA_FQ>("pasta_fq", (uint32_t*)buf);
  samples<h2::BN254_FQ>("bn254_fq", (uint32_t*)buf);
  samples<h2::BN254_FR>("bn254_fr", (uint32_t*)buf);
  hipFree(buf);
  return 0;
}
