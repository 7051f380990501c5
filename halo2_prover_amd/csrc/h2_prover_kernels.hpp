// h2_prover_kernels.hpp -- device kernels of the C++ prover (h2_prover.hip), all over bn256::Fr.
//
// These are the pieces of halo2_proofs::plonk::create_proof that sit BETWEEN the MSM / NTT calls (SURVEY.md App. A.4,
// A.7; reached from /root/reference/circuits/src/utils.rs:83-91,105-120): witness columns, the permutation grand
// product, the quotient numerator, evaluations at the challenge point, the opening combinations.  prover.py runs
// them as ~600 generic pointwise launches; here each is ONE launch:
//   * expr_kernel            the whole quotient numerator -- every gate, the permutation argument, the y-fold and the
//                            division by the vanishing polynomial -- as a straight-line program interpreted per row
//                            of the extended coset (operands: extended columns with rotations, constants, LDS slots)
//   * perm_ratio_kernel      prod (v + beta delta^j w^i + gamma) / prod (v + beta sigma_j + gamma) with one inversion
//                            per 4 rows (Montgomery's trick)
//   * poly_eval_kernels      all evaluations of a proof (different polynomials at different points) in two launches
//   * lincomb_kernel         sum_j c_j a_j, up to 24 columns per launch
//   * coset_extend / coset_shrink   the zero-extension and the zeta^i scaling around the extended-domain NTTs
// All HBM-bound elementwise work except expr_kernel (a few hundred field products per row).
#pragma once
#include "h2_field.hpp"
#include "h2_field29.hpp"

namespace h2 {
namespace pk {

using FR = BN254_FR;
using F = Fe<FR>;

// ---- the 9 x 29-bit lazy form (h2_field29.hpp) for bn256::Fr in these kernels ---------------------------------------------
// A product is ~240 instructions there against ~600 on 8 x 32-bit limbs with carries, and a lone wave runs a chain of
// them about twice as fast -- most kernels below are chains (a scan's recurrence, an inversion, a program's
// instructions).  HBM keeps the API's form x 2^256; on the way in a value is shifted left by five bits while it is
// unpacked, which is the integer x 2^261 + (a multiple of p) < 32 p: the working form of x; minus 16 p it lies in
// (-16 p, 16 p) with limbs of magnitude < 2^29 (a valid operand of fe29_mul on either side).  On the way out one
// product with 2^256 (fe29_to_api) gives the canonical API bytes back.
using W = Fe29<FR>;
template <class FP>
__device__ __forceinline__ Fe29<FP> expr_column_operand(const Fe<FP>& a) {
  Fe29<FP> r;
#pragma unroll
  for (int j = 0; j < 9; j++) {
    // limb j of (a << 5): bits [29 j - 5, 29 j + 24) of a
    const int bit = 29 * j - 5;
    uint32_t limb;
    if (bit < 0) {
      limb = (a.v[0] << 5) & L29_MASK;
    } else {
      const int w = bit >> 5, sh = bit & 31;
      const uint64_t lo = a.v[w], hi = w + 1 < 8 ? a.v[w + 1] : 0;
      limb = (uint32_t)((lo | (hi << 32)) >> sh) & L29_MASK;
    }
    // minus 16 p = (p << 4): limb j of it is bits [29 j - 4, 29 j + 25) of p
    const int pb = 29 * j - 4;
    uint32_t pl;
    if (pb < 0) {
      pl = (FP::P(0) << 4) & L29_MASK;
    } else {
      const int w = pb >> 5, sh = pb & 31;
      const uint64_t lo = w < 8 ? FP::P(w) : 0, hi = w + 1 < 8 ? FP::P(w + 1) : 0;
      pl = (uint32_t)((lo | (hi << 32)) >> sh) & L29_MASK;
    }
    r.v[j] = (int32_t)limb - (int32_t)pl;
  }
  return r;
}

__device__ __forceinline__ W w_load(const U128* p) { return expr_column_operand(fe_load<FR>(p)); }
__device__ __forceinline__ W w_from(const F& a) { return expr_column_operand(a); }
__device__ __forceinline__ void w_store(U128* p, const W& x) { fe_store<FR>(p, fe29_to_api(x)); }
// t in (-p, 3p), any limbs within fe29_norm's reach -> canonical, packed (the API's bytes when t is x 2^256 + j p)
__device__ __forceinline__ F w_canonical_pack(const W& t0) {
  W pl;
#pragma unroll
  for (int i = 0; i < 9; i++) pl.v[i] = (int32_t)fe29_p<FR>(i);
  W t = fe29_norm(t0);
  if (t.v[8] < 0) t = fe29_norm(fe29_add(t, pl));
  W s = fe29_norm(fe29_sub(t, pl));
  if (s.v[8] >= 0) t = s;
  s = fe29_norm(fe29_sub(t, pl));
  if (s.v[8] >= 0) t = s;
  return fe29_pack(t);
}
// a^(p-2), two exponent bits at a time (254 squarings + ~96 products; the exponent is a constant, the branches uniform)
__device__ __forceinline__ W w_inv(const W& a) {
  const W a2 = fe29_mul(a, a), a3 = fe29_mul(a2, a);
  uint32_t e[8];
#pragma unroll
  for (int i = 0; i < 8; i++) e[i] = FR::P(i);
  e[0] -= 2;                                  // bn256::Fr: p[0] = 0xf0000001, no borrow
  W r = fe29_from_api(F::one());
  for (int i = 254; i >= 0; i -= 2) {
    r = fe29_mul(r, r);
    r = fe29_mul(r, r);
    const uint32_t d = (e[i >> 5] >> (i & 31)) & 3u;
    if (d == 1) r = fe29_mul(r, a);
    else if (d == 2) r = fe29_mul(r, a2);
    else if (d == 3) r = fe29_mul(r, a3);
  }
  return r;
}

// column[cell.row] = cell.value for `count` cells (values in Montgomery form); the column was zero-filled before
struct CellRef {
  uint32_t col, row;
};
static __global__ void __launch_bounds__(256)
scatter_cells_kernel(U128* __restrict__ base, size_t col_stride /* elements */, const CellRef* __restrict__ refs,
                     const U128* __restrict__ vals, uint32_t count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  U128* dst = base + 2 * ((size_t)refs[i].col * col_stride + refs[i].row);
  dst[0] = vals[2 * i];
  dst[1] = vals[2 * i + 1];
}

// out[c][i] = i < n ? in[c][i] * zeta^i : 0 for i < en   (zeta^3 = 1: the factor is one of three constants)
static __global__ void __launch_bounds__(256)
coset_extend_kernel(const U128* __restrict__ in, size_t in_stride, U128* __restrict__ out, uint32_t n, uint32_t en, F z1,
                    F z2) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= en) return;
  const U128* src = in + 2 * in_stride * blockIdx.y;
  U128* dst = out + 2 * (size_t)en * blockIdx.y;
  F v = F::zero();
  if (i < n) {
    v = fe_load<FR>(src + 2 * (size_t)i);
    const uint32_t r = i % 3;
    if (r == 1) v = fe_mul(v, z1);
    else if (r == 2) v = fe_mul(v, z2);
  }
  fe_store<FR>(dst + 2 * (size_t)i, v);
}
// a[i] *= zinv^i in place for i < count (after the inverse extended NTT; only the n (d-1) kept coefficients)
static __global__ void __launch_bounds__(256) coset_shrink_kernel(U128* __restrict__ a, uint32_t count, F zi1, F zi2) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint32_t r = i % 3;
  if (r == 0) return;
  F v = fe_load<FR>(a + 2 * (size_t)i);
  fe_store<FR>(a + 2 * (size_t)i, fe_mul(v, r == 1 ? zi1 : zi2));
}

// ---- permutation grand product: ratio[i] = prod_j (v_j + beta delta^j w^i + gamma) / prod_j (v_j + beta sigma_j + gamma)
constexpr int PERM_MAX_COLS = 8;
constexpr int PERM_RUN = 4;      // rows per thread: one inversion per run (Montgomery's trick)
struct PermArgs {
  const U128* value[PERM_MAX_COLS];
  const U128* sigma[PERM_MAX_COLS];
  F beta_delta[PERM_MAX_COLS];   // beta * delta^j
  F beta, gamma;
  int ncols;
};
constexpr int PERM_MAX_SETS = 4;
struct PermBatch {
  PermArgs set[PERM_MAX_SETS];
};
// grid.y = permutation set (the sets only share beta and gamma: all of them in one launch); ratio: one column per set
static __global__ void __launch_bounds__(256)
perm_ratio_kernel(PermBatch B, const U128* __restrict__ omega_col, U128* __restrict__ ratio_base, uint32_t n) {
  const PermArgs& A = B.set[blockIdx.y];
  U128* ratio = ratio_base + 2 * (size_t)n * blockIdx.y;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lo = t * PERM_RUN;
  if (lo >= n) return;
  const W gamma = w_from(A.gamma), beta = w_from(A.beta);
  W num[PERM_RUN], den[PERM_RUN], pre[PERM_RUN];
  const W one = fe29_from_api(F::one());
  W acc = one;
#pragma unroll
  for (int k = 0; k < PERM_RUN; k++) {
    const uint32_t i = lo + k;
    W nu = one, de = one;
    if (i < n) {
      const W w = w_load(omega_col + 2 * (size_t)i);
      for (int j = 0; j < A.ncols; j++) {
        // v + gamma: three terms at most 16 p + p: normalised before it enters a product as the second operand
        const W vg = fe29_add(w_load(A.value[j] + 2 * (size_t)i), gamma);
        nu = fe29_mul(nu, fe29_norm(fe29_add(fe29_mul(w, w_from(A.beta_delta[j])), vg)));
        de = fe29_mul(de, fe29_norm(fe29_add(fe29_mul(w_load(A.sigma[j] + 2 * (size_t)i), beta), vg)));
      }
    }
    num[k] = nu;
    den[k] = de;
    pre[k] = acc;
    acc = fe29_mul(acc, de);
  }
  W inv = w_inv(acc);         // a zero denominator (probability 2^-250 per row) would zero the run, as 1/0 := 0 does
#pragma unroll
  for (int k = PERM_RUN - 1; k >= 0; k--) {
    const uint32_t i = lo + k;
    if (i < n) w_store(ratio + 2 * (size_t)i, fe29_mul(num[k], fe29_mul(pre[k], inv)));
    inv = fe29_mul(inv, den[k]);
  }
}

// a[i] = a[i] * c for rows [lo, hi)
static __global__ void __launch_bounds__(256) scale_range_kernel(U128* __restrict__ a, uint32_t lo, uint32_t hi, F c) {
  const uint32_t i = lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hi) return;
  fe_store<FR>(a + 2 * (size_t)i, fe_mul(fe_load<FR>(a + 2 * (size_t)i), c));
}
// a[i] -= v[i] for i < count
static __global__ void sub_prefix_kernel(U128* __restrict__ a, const U128* __restrict__ v, uint32_t count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  fe_store<FR>(a + 2 * (size_t)i, fe_sub(fe_load<FR>(a + 2 * (size_t)i), fe_load<FR>(v + 2 * (size_t)i)));
}

// ---- out[i] = sum_j c_j a_j[i] ---------------------------------------------------------------------------------
constexpr int LINCOMB_MAX = 24;
struct LincombArgs {
  const U128* a[LINCOMB_MAX];
  F c[LINCOMB_MAX];
  int count;
  int unit_first;   // c[0] == 1: skip its product
};
static __global__ void __launch_bounds__(256)
lincomb_kernel(LincombArgs A, U128* __restrict__ out, uint32_t n, int accumulate) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  F acc = accumulate ? fe_load<FR>(out + 2 * (size_t)i) : F::zero();
  for (int j = 0; j < A.count; j++) {
    const F v = fe_load<FR>(A.a[j] + 2 * (size_t)i);
    acc = fe_add(acc, (j == 0 && A.unit_first) ? v : fe_mul(v, A.c[j]));
  }
  fe_store<FR>(out + 2 * (size_t)i, acc);
}

// ---- evaluations: job q = (polynomial pointer, point); partial[q][block] then out[q] ------------------------------
constexpr int EVAL_RUN = 8;        // coefficients per thread
constexpr int EVAL_BLOCK = 256;
struct EvalJob {
  const U128* poly;
  F point;
};
static __global__ void __launch_bounds__(EVAL_BLOCK)
poly_eval_partial_kernel(const EvalJob* __restrict__ jobs, uint32_t n, U128* __restrict__ partial, uint32_t blocks_per_job) {
  __shared__ U128 red[2 * EVAL_BLOCK];
  const EvalJob job = jobs[blockIdx.y];
  const uint32_t t = blockIdx.x * EVAL_BLOCK + threadIdx.x;
  const uint32_t lo = t * EVAL_RUN;
  F acc = F::zero();
  if (lo < n) {
    const uint32_t hi = min(n, lo + EVAL_RUN);
    for (uint32_t i = hi; i-- > lo;) acc = fe_add(fe_mul(acc, job.point), fe_load<FR>(job.poly + 2 * (size_t)i));
    acc = fe_mul(acc, fe_pow_u64(job.point, lo));
  }
  fe_store<FR>(red + 2 * threadIdx.x, acc);
  __syncthreads();
  for (uint32_t s = EVAL_BLOCK / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s)
      fe_store<FR>(red + 2 * threadIdx.x, fe_add(fe_load<FR>(red + 2 * threadIdx.x), fe_load<FR>(red + 2 * (threadIdx.x + s))));
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    U128* dst = partial + 2 * ((size_t)blockIdx.y * blocks_per_job + blockIdx.x);
    dst[0] = red[0];
    dst[1] = red[1];
  }
}
static __global__ void __launch_bounds__(64)
poly_eval_final_kernel(const U128* __restrict__ partial, uint32_t blocks_per_job, U128* __restrict__ out, uint32_t njobs) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= njobs) return;
  F acc = F::zero();
  for (uint32_t b = 0; b < blocks_per_job; b++) acc = fe_add(acc, fe_load<FR>(partial + 2 * ((size_t)q * blocks_per_job + b)));
  fe_store<FR>(out + 2 * (size_t)q, acc);
}

// ---- several synthetic divisions / prefix products in ONE launch sequence (grid.y = job) -------------------------------
// The same three-kernel scans as h2_poly.hpp (chunk values, a scan of the chunk values in one block,
// recurrence inside every chunk), for independent jobs: the opening witnesses of the GWC points, the permutation
// sets' grand products.  Each scan is a latency chain on 16 waves; side by side they cost what one costs.
constexpr int SCAN_MAX_JOBS = 8;
constexpr uint32_t SCAN_CHUNKS = 4096;
struct ScanBatch {
  const U128* a[SCAN_MAX_JOBS];
  U128* out[SCAN_MAX_JOBS];
  F z[SCAN_MAX_JOBS];        // division only: the point
  F w[SCAN_MAX_JOBS];        // division only: z^L
};
// mode 0: q = (a - a(z)) / (X - z) (suffix sums with weights); mode 1: out[i] = prod_{j < i} a[j]
//
// Both on the 29-bit form (a chain of products on one wave: twice as fast there).  Mode 1 works on true values: shifted
// loads, one extra product (fe29_to_api) per stored element, off the chain.  Mode 0 is LINEAR in the data, so the data
// stay in the API's form read as the working form of x / 32 (plain unpack, no conversion either way): with z in the
// true working form, acc z / R' + a keeps that scaling, and a result only needs to be made canonical to be stored.
constexpr int SCAN_AHEAD = 4;     // elements loaded ahead of the recurrence (the chain itself cannot hide a load)
static __global__ void __launch_bounds__(64)
scan_chunk_kernel(ScanBatch B, int mode, uint32_t n, uint32_t L, uint32_t C, U128* __restrict__ H) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x, job = blockIdx.y;
  if (c >= C) return;
  const U128* a = B.a[job];
  const uint32_t lo = c * L, hi = min(n, lo + L);
  U128* dst = H + 2 * ((size_t)job * SCAN_CHUNKS + c);
  if (mode == 0) {
    W acc = W::zero();
    const W z = w_from(B.z[job]);
    for (uint32_t top = hi; top > lo;) {
      const uint32_t cnt = min((uint32_t)SCAN_AHEAD, top - lo);
      F v[SCAN_AHEAD];
#pragma unroll
      for (int k = 0; k < SCAN_AHEAD; k++)
        if ((uint32_t)k < cnt) v[k] = fe_load<FR>(a + 2 * (size_t)(top - 1 - k));
#pragma unroll
      for (int k = 0; k < SCAN_AHEAD; k++)
        if ((uint32_t)k < cnt) acc = fe29_add(fe29_mul(acc, z), fe29_unpack(v[k]));
      top -= cnt;
    }
    fe_store<FR>(dst, w_canonical_pack(acc));
  } else {
    W acc = fe29_from_api(F::one());
    for (uint32_t at = lo; at < hi;) {
      const uint32_t cnt = min((uint32_t)SCAN_AHEAD, hi - at);
      F v[SCAN_AHEAD];
#pragma unroll
      for (int k = 0; k < SCAN_AHEAD; k++)
        if ((uint32_t)k < cnt) v[k] = fe_load<FR>(a + 2 * (size_t)(at + k));
#pragma unroll
      for (int k = 0; k < SCAN_AHEAD; k++)
        if ((uint32_t)k < cnt) acc = fe29_mul(acc, w_from(v[k]));
      at += cnt;
    }
    w_store(dst, acc);
  }
}
// Scan of the C <= SCAN_CHUNKS chunk values in one block of SCAN_BLOCK_THREADS threads: every thread takes
// SCAN_PER_THREAD consecutive values (a short recurrence in registers), the group totals go through a log-step scan
// in LDS (unpacked, 36 bytes each), and the thread finishes its own values.  The block is a chain of products on ONE
// CU: measured 69 us with 1024 threads x 4 values (16 waves queue for four SIMDs), 74 us with 256 x 16 (long local
// recurrences), 61 us with 512 x 8.  With 4096 chunks of 16 elements the three kernels take 17 + 61 + 28 us for the
// four opening quotients of a proof; with 1024 chunks of 64 they took 58 + 40 + 65.
constexpr int SCAN_BLOCK_THREADS = 512;
constexpr int SCAN_PER_THREAD = SCAN_CHUNKS / SCAN_BLOCK_THREADS;
static __global__ void __launch_bounds__(SCAN_BLOCK_THREADS)
scan_block_kernel(ScanBatch B, int mode, uint32_t C, const U128* __restrict__ H, U128* __restrict__ G) {
  __shared__ int32_t lds[9 * SCAN_BLOCK_THREADS];   // [limb][thread]
  constexpr int K = SCAN_PER_THREAD;
  const uint32_t t = threadIdx.x, job = blockIdx.x;
  H += 2 * (size_t)job * SCAN_CHUNKS;
  G += 2 * (size_t)job * SCAN_CHUNKS;
  auto put = [&](uint32_t i, const W& x) {
#pragma unroll
    for (int l = 0; l < 9; l++) lds[l * SCAN_BLOCK_THREADS + i] = x.v[l];
  };
  auto get = [&](uint32_t i) {
    W x;
#pragma unroll
    for (int l = 0; l < 9; l++) x.v[l] = lds[l * SCAN_BLOCK_THREADS + i];
    return x;
  };
  const W one = fe29_from_api(F::one());
  const uint32_t T = (C + K - 1) / K;               // threads that hold values
  if (mode == 0) {
    // data scaled like the API's bytes (see above); y_c = sum_{j >= c} w^(j-c) v_j, G[c] = y_(c+1)
    W wk[K + 1];                                    // w^0 .. w^K, true working form, normalised
    wk[0] = one;
    wk[1] = fe29_mul(w_from(B.w[job]), one);
#pragma unroll
    for (int k = 2; k <= K; k++) wk[k] = fe29_mul(wk[k - 1], wk[1]);
    W s[K];
#pragma unroll
    for (int k = 0; k < K; k++) s[k] = t * K + k < C ? fe29_unpack(fe_load<FR>(H + 2 * (t * K + k))) : W::zero();
#pragma unroll
    for (int k = K - 2; k >= 0; k--) s[k] = fe29_norm(fe29_add(s[k], fe29_mul(wk[1], s[k + 1])));
    W y = s[0];                                     // the group's total, weights counted from its first chunk
    W wp = wk[K];
    for (uint32_t st = 1; st < T; st <<= 1) {
      put(t, y);
      __syncthreads();
      if (t + st < T) y = fe29_norm(fe29_add(y, fe29_mul(wp, get(t + st))));
      __syncthreads();
      wp = fe29_mul(wp, wp);
    }
    put(t, y);
    __syncthreads();
    const W next = t + 1 < T ? get(t + 1) : W::zero();    // y of the following group's first chunk
    // y_(tK+k) = s_k + w^(K-k) next; G[tK+k] = y_(tK+k+1), and G of the group's last chunk is `next`
#pragma unroll
    for (int k = 0; k < K; k++) {
      const uint32_t c = t * K + k;
      if (c >= C) break;
      const W v = k + 1 < K ? fe29_add(s[k + 1], fe29_mul(wk[K - k - 1], next)) : fe29_mul(next, one);
      fe_store<FR>(G + 2 * c, w_canonical_pack(k + 1 < K ? fe29_mul(fe29_norm(v), one) : v));
    }
  } else {
    // G[c] = prod_{j < c} v_j
    W p[K];
#pragma unroll
    for (int k = 0; k < K; k++) p[k] = t * K + k < C ? w_load(H + 2 * (t * K + k)) : one;
    p[0] = fe29_mul(p[0], one);
#pragma unroll
    for (int k = 1; k < K; k++) p[k] = fe29_mul(p[k - 1], p[k]);
    W y = p[K - 1];
    for (uint32_t st = 1; st < T; st <<= 1) {
      put(t, y);
      __syncthreads();
      if (t >= st) y = fe29_mul(y, get(t - st));
      __syncthreads();
    }
    put(t, y);
    __syncthreads();
    const W before = t > 0 ? get(t - 1) : one;      // product of every group below this one
#pragma unroll
    for (int k = 0; k < K; k++) {
      const uint32_t c = t * K + k;
      if (c >= C) break;
      w_store(G + 2 * c, k == 0 ? before : fe29_mul(before, p[k - 1]));
    }
  }
}
static __global__ void __launch_bounds__(64)
scan_apply_kernel(ScanBatch B, int mode, uint32_t n, uint32_t L, uint32_t C, const U128* __restrict__ G) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x, job = blockIdx.y;
  if (c >= C) return;
  const U128* a = B.a[job];
  U128* out = B.out[job];
  const uint32_t lo = c * L, hi = min(n, lo + L);
  const U128* g = G + 2 * ((size_t)job * SCAN_CHUNKS + c);
  if (mode == 0) {
    W cur = fe29_unpack(fe_load<FR>(g));
    const W z = w_from(B.z[job]);
    for (uint32_t top = hi; top > lo;) {
      const uint32_t cnt = min((uint32_t)SCAN_AHEAD, top - lo);
      F v[SCAN_AHEAD];
#pragma unroll
      for (int k = 0; k < SCAN_AHEAD; k++)
        if ((uint32_t)k < cnt) v[k] = fe_load<FR>(a + 2 * (size_t)(top - 1 - k));
#pragma unroll
      for (int k = 0; k < SCAN_AHEAD; k++)
        if ((uint32_t)k < cnt) {
          const uint32_t i = top - 1 - k;
          cur = fe29_add(fe29_mul(cur, z), fe29_unpack(v[k]));
          if (i >= 1) fe_store<FR>(out + 2 * (size_t)(i - 1), w_canonical_pack(cur));
        }
      top -= cnt;
    }
    if (c == C - 1) fe_store<FR>(out + 2 * (size_t)(n - 1), F::zero());
  } else {
    W cur = fe29_mul(w_load(g), fe29_from_api(F::one()));
    for (uint32_t at = lo; at < hi;) {
      const uint32_t cnt = min((uint32_t)SCAN_AHEAD, hi - at);
      F v[SCAN_AHEAD];
#pragma unroll
      for (int k = 0; k < SCAN_AHEAD; k++)
        if ((uint32_t)k < cnt) v[k] = fe_load<FR>(a + 2 * (size_t)(at + k));
#pragma unroll
      for (int k = 0; k < SCAN_AHEAD; k++)
        if ((uint32_t)k < cnt) {
          w_store(out + 2 * (size_t)(at + k), cur);
          cur = fe29_mul(cur, w_from(v[k]));
        }
      at += cnt;
    }
  }
}

// ---- the quotient numerator as a straight-line program -----------------------------------------------------------------
// operand word: bits 31..30 = kind (0 slot, 1 constant, 2 column, 3 the previous instruction's result); slot / constant:
// index in bits 29..0; column: index in bits 29..8, rotation + 128 in bits 7..0.
// op_dst: op in bits 31..24 (0 add, 1 sub, 2 mul), slot in 23..0 (X_NO_STORE: only the next instruction reads it).
struct XInstr {
  uint32_t op_dst, a, b;
};
constexpr uint32_t X_SLOT = 0u << 30, X_CONST = 1u << 30, X_COL = 2u << 30, X_PREV = 3u << 30;
constexpr uint32_t X_NO_STORE = 0xFFFFFFu;     // destination field of a result that only the next instruction reads
constexpr int EXPR_BLOCK = 64;
constexpr int EXPR_REG_SLOTS = 4;       // slots kept in registers (expr_kernel); the rest is LDS
// magnitudes, in units of p, that the host's compiler (ExprProgram::compile) assumes: a column operand after
// expr_column_operand, and the largest value an instruction may produce before it is reduced by a product with one
constexpr int EXPR_COLUMN_BOUND = 16, EXPR_VALUE_BOUND = 32;

// The arithmetic runs on the 9 x 29-bit lazy form (h2_field29.hpp: for bn256::Fr a product is ~240 instructions
// against ~600 on 8 x 32-bit limbs with carries):
//   * a column holds x 2^256 (canonical); shifted left by five bits while it is unpacked that is the integer
//     x 2^261 + (a multiple of p) < 32 p, the working form of x -- minus 16 p it lies in (-16 p, 16 p);
//   * the constant table is uploaded in the working form (c 2^261 mod p, canonical) and only unpacked;
//   * sums and differences are carry-normalised, products need nothing; the compiler keeps every value below
//     EXPR_VALUE_BOUND p (it multiplies by one where a sum would exceed it), so every product has operands far below
//     the 64 p that fe29_mul and fe29_to_api allow;
//   * the last result goes back to the API form (one product) on its way out.
// Operands that do not depend on the program's own results -- columns and constants -- are fetched TWO instructions
// ahead (a global load is 0.5-2 us, an instruction 0.1-0.5).  LDS: 36 bytes per slot beyond the register slots and row.
static __global__ void __launch_bounds__(EXPR_BLOCK)
expr_kernel(const XInstr* __restrict__ prog, uint32_t ninstr, const U128* const* __restrict__ cols,
            const uint32_t* __restrict__ col_mask, const U128* __restrict__ consts, U128* __restrict__ out, uint32_t step,
            uint32_t en) {
  using W = Fe29<FR>;
  extern __shared__ int32_t slots[];      // [slot][limb][thread]
  const uint32_t tid = threadIdx.x;
  const uint32_t i = blockIdx.x * EXPR_BLOCK + tid;
  // slots 0 .. EXPR_REG_SLOTS-1 live in registers: the slot number comes from the instruction word, the same for the
  // whole wave, so the choice is a scalar branch around nine moves -- nothing next to a 240-instruction product, and
  // the LDS that is left (Poseidon: 3 slots instead of 7) no longer caps the waves per SIMD
  W reg0 = W::zero(), reg1 = W::zero(), reg2 = W::zero(), reg3 = W::zero();
  auto slot_load = [&](uint32_t s) {
    if (s == 0) return reg0;
    if (s == 1) return reg1;
    if (s == 2) return reg2;
    if (s == 3) return reg3;
    W r;
#pragma unroll
    for (int l = 0; l < 9; l++) r.v[l] = slots[((s - EXPR_REG_SLOTS) * 9 + l) * EXPR_BLOCK + tid];
    return r;
  };
  // a column or constant operand as loaded (a slot operand is read when its instruction runs)
  auto fetch = [&](uint32_t code) -> F {
    const uint32_t kind = code & (3u << 30);
    if (kind == X_SLOT || kind == X_PREV) return F::zero();
    if (kind == X_CONST) return fe_load<FR>(consts + 2 * (size_t)(code & 0x3FFFFFFFu));
    const uint32_t c = (code >> 8) & 0x3FFFFFu;
    const int rot = (int)(code & 0xFFu) - 128;
    const uint32_t idx = (i + (uint32_t)(rot * (int)step)) & col_mask[c];
    return fe_load<FR>(cols[c] + 2 * (size_t)idx);
  };
  const XInstr nop{0u, X_SLOT, X_SLOT};
  auto instr_at = [&](uint32_t k) { return k < ninstr ? prog[k] : nop; };
  W r = W::zero();
  auto operand = [&](uint32_t code, const F& pre) -> W {
    const uint32_t kind = code & (3u << 30);
    if (kind == X_SLOT) return slot_load(code & 0x3FFFFFFFu);
    if (kind == X_PREV) return r;
    if (kind == X_CONST) return fe29_unpack(pre);
    return expr_column_operand(pre);
  };
  // one instruction: operands from the prefetched pair, the register or LDS; the pair is refilled for instruction k + 2
  // (the instruction words travel the same way: `ins` was read two instructions ago and is replaced by the one two ahead)
  auto run = [&](uint32_t k, XInstr& slot_ins, F& pa, F& pb) {
    const XInstr ins = slot_ins;
    const W a = operand(ins.a, pa), b = operand(ins.b, pb);
    const XInstr ahead = instr_at(k + 2);
    slot_ins = ahead;
    pa = fetch(ahead.a);
    pb = fetch(ahead.b);
    const uint32_t op = ins.op_dst >> 24;
    if (op == 2) r = fe29_mul(a, b);
    else r = fe29_norm(op == 0 ? fe29_add(a, b) : fe29_sub(a, b));
    const uint32_t s = ins.op_dst & 0xFFFFFFu;
    if (s == 0) reg0 = r;
    else if (s == 1) reg1 = r;
    else if (s == 2) reg2 = r;
    else if (s == 3) reg3 = r;
    else if (s != X_NO_STORE) {
#pragma unroll
      for (int l = 0; l < 9; l++) slots[((s - EXPR_REG_SLOTS) * 9 + l) * EXPR_BLOCK + tid] = r.v[l];
    }
  };
  XInstr i0 = instr_at(0), i1 = instr_at(1);
  F pa0 = fetch(i0.a), pb0 = fetch(i0.b), pa1 = fetch(i1.a), pb1 = fetch(i1.b);
  for (uint32_t k = 0; k < ninstr; k += 2) {
    run(k, i0, pa0, pb0);
    if (k + 1 < ninstr) run(k + 1, i1, pa1, pb1);
  }
  if (i < en) fe_store<FR>(out + 2 * (size_t)i, fe29_to_api(r));     // a domain smaller than one block: the spare lanes computed on wrapped rows
}

}  // namespace pk
}  // namespace h2
