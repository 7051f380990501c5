// h2_poly.hpp -- pointwise polynomial kernels around the NTT: the device pieces of
// halo2_proofs::poly::EvaluationDomain (halo2_proofs @6b43b6b, src/poly/domain.rs -- un-vendored; behaviour
// restated in SURVEY.md App. A.3; reached from /root/reference/circuits/src/utils.rs:83-91,105-120 through
// create_proof): ifft's n^-1 scaling, distribute_powers_zeta (the coset shift a[i] *= g^i),
// divide_by_vanishing_poly (a[i] *= t[i mod period]) and the pointwise add / sub / mul of evaluate_h.
// All are one read + one write of the column: HBM-bound elementwise kernels (64 B per element), kept on
// device so a column never leaves HBM between its NTTs and its MSM.
#pragma once
#include "h2_field.hpp"

namespace h2 {

constexpr int POLY_RUN = 8;  // consecutive elements per thread in the powers kernel

// a[i] *= c   (m columns, stride n)
template <class FP>
__global__ void __launch_bounds__(256) poly_scale_kernel(U128* __restrict__ a, size_t total, Fe<FP> c) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    Fe<FP> x = fe_load<FP>(a + 2 * i);
    fe_store<FP>(a + 2 * i, fe_mul(x, c));
  }
}

// a[col][i] *= g^i : each thread owns POLY_RUN consecutive i, starts from g^(first i) by square-and-multiply
template <class FP>
__global__ void __launch_bounds__(256)
poly_powers_kernel(U128* __restrict__ a, size_t n, Fe<FP> g) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t first = t * POLY_RUN;
  if (first >= n) return;
  U128* col = a + 2 * n * blockIdx.y;
  Fe<FP> cur = fe_pow_u64(g, (uint64_t)first);
  for (int k = 0; k < POLY_RUN && first + k < n; k++) {
    Fe<FP> x = fe_load<FP>(col + 2 * (first + k));
    fe_store<FP>(col + 2 * (first + k), fe_mul(x, cur));
    cur = fe_mul(cur, g);
  }
}

// a[col][i] *= t[i mod period]  (period a power of two)
template <class FP>
__global__ void __launch_bounds__(256)
poly_mul_periodic_kernel(U128* __restrict__ a, size_t total, const U128* __restrict__ t, size_t period_mask) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    Fe<FP> x = fe_load<FP>(a + 2 * i);
    Fe<FP> y = fe_load<FP>(t + 2 * (i & period_mask));
    fe_store<FP>(a + 2 * i, fe_mul(x, y));
  }
}

// a[i] = a[i] (op) b[i],  op: 0 add, 1 sub, 2 mul
template <class FP>
__global__ void __launch_bounds__(256)
poly_pointwise_kernel(U128* __restrict__ a, const U128* __restrict__ b, size_t total, int op) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    Fe<FP> x = fe_load<FP>(a + 2 * i), y = fe_load<FP>(b + 2 * i), r;
    if (op == 0) r = fe_add(x, y);
    else if (op == 1) r = fe_sub(x, y);
    else r = fe_mul(x, y);
    fe_store<FP>(a + 2 * i, r);
  }
}

// a[i] = a[i]^-1 (0 stays 0): the denominators of the permutation grand product
template <class FP>
__global__ void __launch_bounds__(256) poly_inverse_kernel(U128* __restrict__ a, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    Fe<FP> x = fe_load<FP>(a + 2 * i);
    if (!x.is_zero()) fe_store<FP>(a + 2 * i, fe_inv(x));
  }
}

// ---- kate_division: q = (a - a(z)) / (X - z) --------------------------------------------------------------
// halo2_proofs @6b43b6b src/arithmetic.rs `kate_division` (called by the GWC / SHPLONK provers on every opened
// polynomial): the serial recurrence q[i-1] = a[i] + z q[i] from the top coefficient down, q[n-1] = 0.  Written
// Q_i = sum_{j >= i} a[j] z^(j-i) it is a suffix sum with weights, done in three launches over C <= 1024 chunks of
// L coefficients: each chunk's own Horner value, a log-step suffix scan of those values with multiplier
// w = z^L (one block), then the recurrence inside every chunk started from the scanned value.  d_q != d_a.
constexpr uint32_t DIV_MAX_CHUNKS = 1024;

template <class FP>
__global__ void __launch_bounds__(64)
poly_divide_chunk_kernel(const U128* __restrict__ a, size_t n, uint32_t L, uint32_t C, Fe<FP> z, U128* __restrict__ H) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const size_t lo = (size_t)c * L, hi = min(n, lo + L);
  Fe<FP> acc = Fe<FP>::zero();
  for (size_t i = hi; i-- > lo;) acc = fe_add(fe_mul(acc, z), fe_load<FP>(a + 2 * i));
  fe_store<FP>(H + 2 * c, acc);
}

// G[c] = sum_{d > c} H[d] w^(d-c-1)
template <class FP>
__global__ void __launch_bounds__(1024)
poly_divide_scan_kernel(const U128* __restrict__ H, uint32_t C, Fe<FP> w, U128* __restrict__ G) {
  __shared__ U128 lds[2 * DIV_MAX_CHUNKS];
  const uint32_t c = threadIdx.x;
  Fe<FP> y = c < C ? fe_load<FP>(H + 2 * c) : Fe<FP>::zero();
  Fe<FP> wp = w;
  for (uint32_t s = 1; s < C; s <<= 1) {
    fe_store<FP>(lds + 2 * c, y);
    __syncthreads();
    if (c + s < C) y = fe_add(y, fe_mul(wp, fe_load<FP>(lds + 2 * (c + s))));
    __syncthreads();
    wp = fe_mul(wp, wp);
  }
  fe_store<FP>(lds + 2 * c, y);
  __syncthreads();
  if (c < C) fe_store<FP>(G + 2 * c, c + 1 < C ? fe_load<FP>(lds + 2 * (c + 1)) : Fe<FP>::zero());
}

template <class FP>
__global__ void __launch_bounds__(64)
poly_divide_apply_kernel(const U128* __restrict__ a, size_t n, uint32_t L, uint32_t C, Fe<FP> z,
                         const U128* __restrict__ G, U128* __restrict__ q) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const size_t lo = (size_t)c * L, hi = min(n, lo + L);
  Fe<FP> cur = fe_load<FP>(G + 2 * c);
  for (size_t i = hi; i-- > lo;) {
    cur = fe_add(fe_mul(cur, z), fe_load<FP>(a + 2 * i));
    if (i >= 1) fe_store<FP>(q + 2 * (i - 1), cur);
  }
  if (c == C - 1) fe_store<FP>(q + 2 * (n - 1), Fe<FP>::zero());
}

// enqueue; d_ws holds 2 * DIV_MAX_CHUNKS elements
template <class FP>
inline hipError_t poly_divide_linear_launch(const U128* a, size_t n, const Fe<FP>& z, U128* q, U128* d_ws,
                                            hipStream_t stream) {
  uint32_t L = (uint32_t)((n + DIV_MAX_CHUNKS - 1) / DIV_MAX_CHUNKS);
  if (L < 16) L = 16;
  const uint32_t C = (uint32_t)((n + L - 1) / L);
  U128* H = d_ws;
  U128* G = d_ws + 2 * DIV_MAX_CHUNKS;
  const Fe<FP> w = fe_pow_u64(z, (uint64_t)L);
  hipLaunchKernelGGL(poly_divide_chunk_kernel<FP>, dim3((C + 63) / 64), dim3(64), 0, stream, a, n, L, C, z, H);
  hipLaunchKernelGGL(poly_divide_scan_kernel<FP>, dim3(1), dim3(DIV_MAX_CHUNKS), 0, stream, H, C, w, G);
  hipLaunchKernelGGL(poly_divide_apply_kernel<FP>, dim3((C + 63) / 64), dim3(64), 0, stream, a, n, L, C, z, G, q);
  return hipGetLastError();
}

// ---- exclusive prefix product: out[i] = prod_{j < i} a[j], out[0] = 1 ------------------------------------------
// The permutation argument's grand product (halo2_proofs @6b43b6b src/plonk/permutation/prover.rs `Argument::commit`:
// z[0] = last_z, z[i+1] = z[i] * numerator[i] / denominator[i]) is this scan of the per-row ratios, times last_z.
// Same three launches as the division: chunk products, log-step scan of the chunk products in one block, then the
// recurrence inside every chunk.  May run in place (a thread reads a[i] before it writes out[i]).
template <class FP>
__global__ void __launch_bounds__(64)
poly_prefix_chunk_kernel(const U128* __restrict__ a, size_t n, uint32_t L, uint32_t C, U128* __restrict__ H) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const size_t lo = (size_t)c * L, hi = min(n, lo + L);
  Fe<FP> acc = Fe<FP>::one();
  for (size_t i = lo; i < hi; i++) acc = fe_mul(acc, fe_load<FP>(a + 2 * i));
  fe_store<FP>(H + 2 * c, acc);
}

// G[c] = prod_{d < c} H[d]
template <class FP>
__global__ void __launch_bounds__(1024)
poly_prefix_scan_kernel(const U128* __restrict__ H, uint32_t C, U128* __restrict__ G) {
  __shared__ U128 lds[2 * DIV_MAX_CHUNKS];
  const uint32_t c = threadIdx.x;
  Fe<FP> y = c < C ? fe_load<FP>(H + 2 * c) : Fe<FP>::one();
  for (uint32_t s = 1; s < C; s <<= 1) {
    fe_store<FP>(lds + 2 * c, y);
    __syncthreads();
    if (c >= s) y = fe_mul(y, fe_load<FP>(lds + 2 * (c - s)));
    __syncthreads();
  }
  fe_store<FP>(lds + 2 * c, y);
  __syncthreads();
  if (c < C) fe_store<FP>(G + 2 * c, c > 0 ? fe_load<FP>(lds + 2 * (c - 1)) : Fe<FP>::one());
}

template <class FP>
__global__ void __launch_bounds__(64)
poly_prefix_apply_kernel(const U128* a, size_t n, uint32_t L, uint32_t C, const U128* __restrict__ G, U128* out) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const size_t lo = (size_t)c * L, hi = min(n, lo + L);
  Fe<FP> cur = fe_load<FP>(G + 2 * c);
  for (size_t i = lo; i < hi; i++) {
    const Fe<FP> x = fe_load<FP>(a + 2 * i);
    fe_store<FP>(out + 2 * i, cur);
    cur = fe_mul(cur, x);
  }
}

template <class FP>
inline hipError_t poly_prefix_product_launch(const U128* a, size_t n, U128* out, U128* d_ws, hipStream_t stream) {
  uint32_t L = (uint32_t)((n + DIV_MAX_CHUNKS - 1) / DIV_MAX_CHUNKS);
  if (L < 16) L = 16;
  const uint32_t C = (uint32_t)((n + L - 1) / L);
  U128* H = d_ws;
  U128* G = d_ws + 2 * DIV_MAX_CHUNKS;
  hipLaunchKernelGGL(poly_prefix_chunk_kernel<FP>, dim3((C + 63) / 64), dim3(64), 0, stream, a, n, L, C, H);
  hipLaunchKernelGGL(poly_prefix_scan_kernel<FP>, dim3(1), dim3(DIV_MAX_CHUNKS), 0, stream, H, C, G);
  hipLaunchKernelGGL(poly_prefix_apply_kernel<FP>, dim3((C + 63) / 64), dim3(64), 0, stream, a, n, L, C, G, out);
  return hipGetLastError();
}

// ---- the blinding polynomial's coefficients ------------------------------------------------------------------
// out[i] = Scalar::random(ChaCha20Rng::from_seed(seed)) number first + i (halo2_proofs @6b43b6b
// src/plonk/vanishing/prover.rs `Argument::commit`: random_poly from a ChaCha20Rng seeded off the prover's rng;
// SURVEY.md App. A.4).  rand_chacha 0.3.1: 20 rounds, 64-bit block counter in words 12-13, stream id 0; ff's
// `random` reads one 64-byte block as a 512-bit little-endian integer and reduces it: lo R + hi 2^256 R, formed
// as two Montgomery products with R^2 and R^3.
struct ChaChaKey { uint32_t w[8]; };

__device__ __forceinline__ uint32_t chacha_rotl(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
#define H2_CHACHA_QR(a, b, c, d)            \
  a += b; d = chacha_rotl(d ^ a, 16);       \
  c += d; b = chacha_rotl(b ^ c, 12);       \
  a += b; d = chacha_rotl(d ^ a, 8);        \
  c += d; b = chacha_rotl(b ^ c, 7)

template <class FP>
__global__ void __launch_bounds__(256)
chacha20_scalars_kernel(U128* __restrict__ out, size_t n, uint64_t first_block, ChaChaKey key) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t ctr = first_block + i;
  uint32_t in[16] = {0x61707865u, 0x3320646Eu, 0x79622D32u, 0x6B206574u, key.w[0], key.w[1], key.w[2], key.w[3],
                     key.w[4], key.w[5], key.w[6], key.w[7], (uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
  uint32_t x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3], x4 = in[4], x5 = in[5], x6 = in[6], x7 = in[7],
           x8 = in[8], x9 = in[9], x10 = in[10], x11 = in[11], x12 = in[12], x13 = in[13], x14 = in[14], x15 = in[15];
  for (int r = 0; r < 10; r++) {
    H2_CHACHA_QR(x0, x4, x8, x12); H2_CHACHA_QR(x1, x5, x9, x13); H2_CHACHA_QR(x2, x6, x10, x14); H2_CHACHA_QR(x3, x7, x11, x15);
    H2_CHACHA_QR(x0, x5, x10, x15); H2_CHACHA_QR(x1, x6, x11, x12); H2_CHACHA_QR(x2, x7, x8, x13); H2_CHACHA_QR(x3, x4, x9, x14);
  }
  Fe<FP> lo, hi, r2;
  lo.v[0] = x0 + in[0]; lo.v[1] = x1 + in[1]; lo.v[2] = x2 + in[2]; lo.v[3] = x3 + in[3];
  lo.v[4] = x4 + in[4]; lo.v[5] = x5 + in[5]; lo.v[6] = x6 + in[6]; lo.v[7] = x7 + in[7];
  hi.v[0] = x8 + in[8]; hi.v[1] = x9 + in[9]; hi.v[2] = x10 + in[10]; hi.v[3] = x11 + in[11];
  hi.v[4] = x12 + in[12]; hi.v[5] = x13 + in[13]; hi.v[6] = x14 + in[14]; hi.v[7] = x15 + in[15];
#pragma unroll
  for (int k = 0; k < 8; k++) r2.v[k] = FP::R2(k);
  const Fe<FP> r3 = fe_mul(r2, r2);
  // lo, hi < 2^256 need not be reduced: a Montgomery product with one factor < p is < 2p before its final subtraction
  fe_store<FP>(out + 2 * i, fe_add(fe_mul(lo, r2), fe_mul(hi, r3)));
}
#undef H2_CHACHA_QR

inline unsigned poly_grid(size_t total) {
  size_t b = (total + 255) / 256;
  if (b > 256 * 8) b = 256 * 8;  // 8 blocks per CU, grid-stride the rest
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace h2
