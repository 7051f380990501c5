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

inline unsigned poly_grid(size_t total) {
  size_t b = (total + 255) / 256;
  if (b > 256 * 8) b = 256 * 8;  // 8 blocks per CU, grid-stride the rest
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace h2
