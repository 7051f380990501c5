// h2_ntt.hpp -- NTT over 256-bit Montgomery fields for gfx950 (radix-2 stages fused in pairs).
//
// Device replacement for halo2_proofs::arithmetic::best_fft / recursive_butterfly_arithmetic
// (halo2_proofs @6b43b6b, src/arithmetic.rs -- un-vendored; algorithm restated in SURVEY.md
// App. A.2; reached from /root/reference/circuits/src/utils.rs:83-91,105-120 through
// EvaluationDomain::{lagrange_to_coeff, coeff_to_extended, extended_to_coeff}).
// Contract kept: in place, natural order in and out, A[k] = sum_j a[j] w^(jk), unscaled.
//
// Structure (MI355X-first, not the reference's recursion): n = R1 * R2 * R3 with every radix
// <= 2^10.  Each pass stages a tile of R x C elements in LDS (C consecutive "columns", so every HBM access is a
// run of C*32 contiguous bytes; C = 4, or 2 for R >= 512 so that the tile stays <= 72 KB and two blocks share a CU), runs the log2(R) butterfly stages out
// of LDS with the R/2 twiddles of that radix also held in LDS, applies the inter-pass
// twiddle w^(outer*i*k) and writes the tile back.  The last pass writes the digit-reversed
// position directly, so no bit-reversal sweep over HBM is needed.  Per pass the algorithmic
// HBM traffic is one read and one write of the column (64 B per element).
#pragma once
#include "h2_field.hpp"
#include "h2_tune.hpp"

namespace h2 {

struct NttPass {
  uint32_t log_n;      // size of the whole transform
  uint32_t log_outer;  // log2 of the product of the radices of earlier passes
  uint32_t log_r;      // radix of this pass
  uint32_t log_inner;  // log_n - log_outer - log_r
  uint32_t log_c;      // tile columns
  uint32_t log_r1;     // radix of pass 0 (0 when this is the only pass)
  uint32_t log_r2;     // radix of pass 1 when there are three passes, else 0
  uint32_t is_final;
  uint32_t has_scale;  // final pass multiplies every output by `scale` (EvaluationDomain::ifft's n^-1)
  uint32_t tw_global;  // 29-bit kernel: radix twiddles read from the table in global memory (L1/L2) instead of LDS
};

__device__ __forceinline__ uint32_t h2_bitrev(uint32_t x, uint32_t bits) {
  return bits ? (__brev(x) >> (32 - bits)) : 0;
}

// tw[i] = omega^i for i < half_n.  One thread fills TW_RUN consecutive entries.
constexpr int TW_RUN = 16;
template <class FP>
__global__ void __launch_bounds__(256) ntt_twiddle_kernel(U128* tw, Fe<FP> omega, uint32_t half_n) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t start = (uint64_t)t * TW_RUN;
  if (start >= half_n) return;
  Fe<FP> cur = fe_pow_u64(omega, start);
  for (int k = 0; k < TW_RUN && start + k < half_n; k++) {
    fe_store<FP>(tw + 2 * (start + k), cur);
    cur = fe_mul(cur, omega);
  }
}

// One pass.  grid.x = tiles per column, grid.y = batch column.  Dynamic LDS:
// (R*C + R/2) elements in two 16-byte planes (plane h holds limbs 4h..4h+3 of every element),
// so a wave's ds_read_b128 / ds_write_b128 of consecutive elements is conflict-free.
template <class FP>
__global__ void __launch_bounds__(1024)
ntt_pass_kernel(const U128* __restrict__ in, U128* __restrict__ out, const U128* __restrict__ tw, NttPass P,
                size_t col_stride /* elements */, Fe<FP> scale /* used when P.has_scale (final pass) */) {
  extern __shared__ U128 lds[];
  const uint32_t R = 1u << P.log_r, C = 1u << P.log_c;
  const uint32_t RC = R * C;
  const uint32_t n_half_log = P.log_n - 1;
  U128* tile0 = lds;             // plane 0 of the tile
  U128* tile1 = lds + RC;        // plane 1
  U128* twl0 = lds + 2 * RC;     // plane 0 of the radix twiddles
  U128* twl1 = twl0 + (R >> 1);  // plane 1

  const U128* src = in + 2 * col_stride * blockIdx.y;
  U128* dst = out + 2 * col_stride * blockIdx.y;
  const uint32_t tid = threadIdx.x, nthr = blockDim.x;
  const uint32_t tile = blockIdx.x;

  // tile coordinates
  uint64_t in_base;        // element address of (j = 0, cc = 0)
  uint64_t in_j_stride;    // element stride between consecutive j
  uint64_t in_c_stride;    // element stride between consecutive cc
  uint32_t i_first = 0;    // inner index of cc = 0 (non-final passes)
  uint64_t out_base, out_k_stride, out_c_stride;
  if (!P.is_final) {
    const uint32_t chunks = 1u << (P.log_inner - P.log_c);
    const uint32_t ic = tile & (chunks - 1), o = tile >> (P.log_inner - P.log_c);
    i_first = ic << P.log_c;
    in_base = ((uint64_t)o << (P.log_r + P.log_inner)) + i_first;
    in_j_stride = (uint64_t)1 << P.log_inner;
    in_c_stride = 1;
    out_base = in_base; out_k_stride = in_j_stride; out_c_stride = 1;
  } else {
    // C consecutive k1 (same k2); each row is R contiguous elements
    const uint32_t groups_log = P.log_r1 - P.log_c;  // log2(R1 / C)
    const uint32_t k1c = tile & ((1u << groups_log) - 1), k2 = tile >> groups_log;
    const uint32_t k1 = k1c << P.log_c;
    in_base = (((uint64_t)k1 << P.log_r2) + k2) << P.log_r;
    in_j_stride = 1;
    in_c_stride = (uint64_t)1 << (P.log_r2 + P.log_r);
    out_base = (uint64_t)k1 + ((uint64_t)k2 << P.log_r1);
    out_k_stride = (uint64_t)1 << (P.log_r1 + P.log_r2);
    out_c_stride = 1;
  }

  // radix twiddles w_R^i = w^(i * n/R), i < R/2
  for (uint32_t i = tid; i < (R >> 1); i += nthr) {
    const U128* t = tw + 2 * ((uint64_t)i << (P.log_n - P.log_r));
    twl0[i] = t[0];
    twl1[i] = t[1];
  }
  // load the tile, bit-reversing j on the way in
  if (!P.is_final) {
    for (uint32_t e = tid; e < RC; e += nthr) {
      const uint32_t cc = e & (C - 1), j = e >> P.log_c;
      const U128* g = src + 2 * (in_base + (uint64_t)j * in_j_stride + cc);
      const uint32_t l = (h2_bitrev(j, P.log_r) << P.log_c) + cc;
      tile0[l] = g[0];
      tile1[l] = g[1];
    }
  } else {
    for (uint32_t e = tid; e < RC; e += nthr) {
      const uint32_t j = e & (R - 1), cc = e >> P.log_r;
      const U128* g = src + 2 * (in_base + (uint64_t)cc * in_c_stride + j);
      const uint32_t l = (h2_bitrev(j, P.log_r) << P.log_c) + cc;
      tile0[l] = g[0];
      tile1[l] = g[1];
    }
  }
  __syncthreads();

  // butterfly stages.  Two radix-2 stages (s, s+1) are done per trip through LDS: a thread holds the four elements
  // base + {0, 1, 2, 3} * 2^s in registers, so the tile crosses LDS (and the block synchronises) log2(R)/2 times
  // instead of log2(R).  An odd log2(R) starts with the lone stage 0, whose twiddles are all 1.
  auto lds_get = [&](uint32_t i) {
    const U128 a0 = tile0[i], a1 = tile1[i];
    Fe<FP> x;
    x.v[0] = a0.x; x.v[1] = a0.y; x.v[2] = a0.z; x.v[3] = a0.w;
    x.v[4] = a1.x; x.v[5] = a1.y; x.v[6] = a1.z; x.v[7] = a1.w;
    return x;
  };
  auto lds_put = [&](uint32_t i, const Fe<FP>& u) {
    tile0[i] = U128{u.v[0], u.v[1], u.v[2], u.v[3]};
    tile1[i] = U128{u.v[4], u.v[5], u.v[6], u.v[7]};
  };
  auto tw_get = [&](uint32_t i) {
    const U128 t0 = twl0[i], t1 = twl1[i];
    Fe<FP> t;
    t.v[0] = t0.x; t.v[1] = t0.y; t.v[2] = t0.z; t.v[3] = t0.w;
    t.v[4] = t1.x; t.v[5] = t1.y; t.v[6] = t1.z; t.v[7] = t1.w;
    return t;
  };
  uint32_t s = 0;
  if (P.log_r & 1) {
    for (uint32_t w = tid; w < (RC >> 1); w += nthr) {
      const uint32_t cc = w & (C - 1), b = w >> P.log_c;
      const uint32_t i0 = (b << (P.log_c + 1)) + cc, i1 = i0 + C;
      const Fe<FP> x = lds_get(i0), y = lds_get(i1);
      lds_put(i0, fe_add(x, y));
      lds_put(i1, fe_sub(x, y));
    }
    __syncthreads();
    s = 1;
  }
  for (; s < P.log_r; s += 2) {
    const uint32_t h = 1u << s;
    for (uint32_t w = tid; w < (RC >> 2); w += nthr) {
      const uint32_t cc = w & (C - 1), b = w >> P.log_c;
      const uint32_t pos = b & (h - 1), grp = b >> s;
      const uint32_t i0 = (((grp << (s + 2)) + pos) << P.log_c) + cc;
      const uint32_t step = h << P.log_c;
      Fe<FP> e0 = lds_get(i0), e1 = lds_get(i0 + step), e2 = lds_get(i0 + 2 * step), e3 = lds_get(i0 + 3 * step);
      const uint32_t tb = pos << (P.log_r - 2 - s);
      if (s != 0) {                      // s == 0: pos == 0, the twiddles of stage s and of the pair (e0, e2) are 1
        const Fe<FP> ta = tw_get(pos << (P.log_r - 1 - s));
        e1 = fe_mul(e1, ta);
        e3 = fe_mul(e3, ta);
      }
      const Fe<FP> a0 = fe_add(e0, e1), a1 = fe_sub(e0, e1);
      Fe<FP> a2 = fe_add(e2, e3), a3 = fe_sub(e2, e3);
      if (s != 0) a2 = fe_mul(a2, tw_get(tb));
      a3 = fe_mul(a3, tw_get(tb + (R >> 2)));
      lds_put(i0, fe_add(a0, a2));
      lds_put(i0 + step, fe_add(a1, a3));
      lds_put(i0 + 2 * step, fe_sub(a0, a2));
      lds_put(i0 + 3 * step, fe_sub(a1, a3));
    }
    __syncthreads();
  }

  // write back (with the inter-pass twiddle w^(outer * i * k) on non-final passes)
  for (uint32_t e = tid; e < RC; e += nthr) {
    const uint32_t cc = e & (C - 1), k = e >> P.log_c;
    U128 a0 = tile0[e], a1 = tile1[e];
    if (!P.is_final) {
      const uint64_t ex = ((uint64_t)(i_first + cc) * k) << P.log_outer;  // < n
      if (ex != 0) {
        Fe<FP> x;
        x.v[0] = a0.x; x.v[1] = a0.y; x.v[2] = a0.z; x.v[3] = a0.w;
        x.v[4] = a1.x; x.v[5] = a1.y; x.v[6] = a1.z; x.v[7] = a1.w;
        const uint64_t half_n = (uint64_t)1 << n_half_log;
        const bool negate = ex >= half_n;
        const uint64_t ti = negate ? ex - half_n : ex;
        Fe<FP> t = fe_load<FP>(tw + 2 * ti);
        x = fe_mul(x, t);
        if (negate) x = fe_neg(x);
        a0 = U128{x.v[0], x.v[1], x.v[2], x.v[3]};
        a1 = U128{x.v[4], x.v[5], x.v[6], x.v[7]};
      }
    } else if (P.has_scale) {
      Fe<FP> x;
      x.v[0] = a0.x; x.v[1] = a0.y; x.v[2] = a0.z; x.v[3] = a0.w;
      x.v[4] = a1.x; x.v[5] = a1.y; x.v[6] = a1.z; x.v[7] = a1.w;
      x = fe_mul(x, scale);
      a0 = U128{x.v[0], x.v[1], x.v[2], x.v[3]};
      a1 = U128{x.v[4], x.v[5], x.v[6], x.v[7]};
    }
    U128* g = dst + 2 * (out_base + (uint64_t)k * out_k_stride + (uint64_t)cc * out_c_stride);
    g[0] = a0;
    g[1] = a1;
  }
}

// ---- host side -------------------------------------------------------------------------------
struct NttPlan {
  int npass;
  NttPass pass[3];
  uint32_t threads[3];
  size_t lds_bytes[3];
  uint32_t tiles[3];
};

constexpr uint32_t NTT_MAX_LOG_R = 10;  // R <= 1024: tile R*2 elements = 64 KiB + 16 KiB twiddles

inline NttPlan ntt_make_plan(uint32_t log_n, uint32_t max_log_r = NTT_MAX_LOG_R) {
  NttPlan pl{};
  static const int tune_r = tune_int("H2_TUNE_NTT_MAXR", 0);     // tuning builds only (h2_tune.hpp)
  static const int tune_c9 = tune_int("H2_TUNE_NTT_LC9", -1);
  static const int tune_c = tune_int("H2_TUNE_NTT_LC", -1);
  if (tune_r > 0) max_log_r = (uint32_t)tune_r;
  uint32_t np = log_n == 0 ? 1 : (log_n + max_log_r - 1) / max_log_r;
  if (np > 3) np = 3;  // callers reject log_n > 30
  uint32_t radix[3] = {0, 0, 0};
  uint32_t left = log_n;
  for (uint32_t p = 0; p < np; p++) {
    radix[p] = (left + (np - p) - 1) / (np - p);
    left -= radix[p];
  }
  pl.npass = (int)np;
  uint32_t outer = 0;
  for (uint32_t p = 0; p < np; p++) {
    NttPass& P = pl.pass[p];
    P.log_n = log_n;
    P.log_outer = outer;
    P.log_r = radix[p];
    P.log_inner = log_n - outer - radix[p];
    P.is_final = (p == np - 1);
    P.log_r1 = np >= 2 ? radix[0] : 0;
    P.log_r2 = np == 3 ? radix[1] : 0;
    // tile columns: keep the tile at <= 72 KB of LDS so that at least two blocks share a CU (one computes while the
    // other waits at a barrier): 2 columns for R >= 512, 4 below (measured: R = 1024 x 4 columns, one block per CU,
    // is 13 % slower on the 2^19 transforms; R = 256 x 2 columns is 5-30 % slower than x 4 on the 2^16 ones)
    uint32_t lc = radix[p] >= 9 ? 1 : 2;
    if (radix[p] >= 9 && tune_c9 >= 0) lc = (uint32_t)tune_c9;
    if (radix[p] < 9 && tune_c >= 0) lc = (uint32_t)tune_c;
    if (P.is_final) {
      if (lc > P.log_r1) lc = P.log_r1;
    } else {
      if (lc > P.log_inner) lc = P.log_inner;
    }
    P.log_c = lc;
    const uint32_t rc = 1u << (P.log_r + lc);
    pl.lds_bytes[p] = ((size_t)rc + ((size_t)1 << P.log_r) / 2) * 32;
    if (pl.lds_bytes[p] < 64) pl.lds_bytes[p] = 64;
    uint32_t thr = rc / 4;                      // one radix-4 butterfly per thread and double stage
    if (thr < 64) thr = 64;
    if (thr > 1024) thr = 1024;
    pl.threads[p] = thr;
    pl.tiles[p] = 1u << (log_n - P.log_r - lc);
    outer += radix[p];
  }
  return pl;
}

// Enqueue the transform of m columns (column stride = n elements) on `stream`.
// data: in place; scratch: m*n elements when the plan has more than one pass.
template <class FP>
inline hipError_t ntt_launch(U128* data, U128* scratch, const U128* tw, uint32_t log_n, size_t m,
                             hipStream_t stream, const Fe<FP>* scale = nullptr) {
  if (log_n == 0 || m == 0) return hipSuccess;
  NttPlan pl = ntt_make_plan(log_n);
  const Fe<FP> sc = scale ? *scale : Fe<FP>::zero();
  const size_t n = (size_t)1 << log_n;
  for (int p = 0; p < pl.npass; p++) {
    const U128* src;
    U128* dst;
    if (pl.npass == 1) { src = data; dst = data; }
    else if (p == 0) { src = data; dst = scratch; }
    else if (p == pl.npass - 1) { src = scratch; dst = data; }
    else { src = scratch; dst = scratch; }
    dim3 grid(pl.tiles[p], (unsigned)m);
    NttPass P = pl.pass[p];
    P.has_scale = (scale && P.is_final) ? 1u : 0u;
    hipLaunchKernelGGL(ntt_pass_kernel<FP>, grid, dim3(pl.threads[p]), pl.lds_bytes[p], stream, src, dst, tw, P, n,
                       sc);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
// once per device (h2_init): the tile kernel may use the whole 160 KiB of LDS
template <class FP>
inline hipError_t ntt_kernel_setup() {
  return hipFuncSetAttribute((const void*)ntt_pass_kernel<FP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

template <class FP>
inline hipError_t ntt_build_twiddles(U128* tw, const Fe<FP>& omega, uint32_t log_n, hipStream_t stream) {
  if (log_n == 0) return hipSuccess;
  const uint32_t half_n = 1u << (log_n - 1);
  const uint32_t threads = (half_n + TW_RUN - 1) / TW_RUN;
  const uint32_t blocks = (threads + 255) / 256;
  hipLaunchKernelGGL(ntt_twiddle_kernel<FP>, dim3(blocks), dim3(256), 0, stream, tw, omega, half_n);
  return hipGetLastError();
}

}  // namespace h2
