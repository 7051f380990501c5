// h2_ntt29.hpp -- the NTT pass of h2_ntt.hpp with its butterflies on the 9 x 29-bit lazy form (h2_field29.hpp).
//
// Same tiling, same pass structure, same HBM format (4 x u64 Montgomery limbs, R = 2^256) as ntt_pass_kernel; what
// changes is the arithmetic between the tile load and the tile store:
//   * x 2^256 mod p (the API's bytes) is the R' = 2^261 form of x / 32, and the transform is linear: the DATA need no
//     conversion at all, only the twiddle tables are built in R' form (ntt_twiddle29_kernel);
//   * a product costs 81 + 9k multiply-adds without carry counters or a conditional subtraction (1.37x faster than
//     the 32-bit product scanning form on MI355X, profiles/r01_microbench_limb29.txt), additions and subtractions are
//     nine limb operations without carries; every value written back to LDS is carry-normalised (limbs in [0, 2^29),
//     the top limb takes the sign);
//   * magnitudes: inputs are canonical (< p); a radix-4 double stage adds at most two products (each in (-p/2, 3p/2))
//     to an element, the very first one (all twiddles 1) at most quadruples it: <= 4p + 3p * 4 = 16p after the five
//     double stages of a 1024-row tile, well inside fe29_mul's |a| |b| <= 64 p^2 with a canonical twiddle as b;
//   * leaving the tile every element goes through one product anyway -- the inter-pass twiddle, the iNTT's 1/n, or
//     (first column of a non-final pass, unscaled final pass) a product with 1 -- which brings it back to
//     (-p/2, 3p/2); two conditional additions make it canonical and it is packed to 256 bits.
// LDS: 36 bytes per element in three planes (two of 16 bytes, one of 4: ds_read_b128 x 2 + ds_read_b32), radix
// twiddles packed (32 bytes) and unpacked on use.
#pragma once
#include "h2_field29.hpp"
#include "h2_ntt.hpp"

namespace h2 {

// tw[i] = omega^i in R' form (canonical, packed to 8 x u32) for i < half_n
template <class FP>
__global__ void __launch_bounds__(256) ntt_twiddle29_kernel(U128* tw, Fe<FP> omega, uint32_t half_n) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t start = (uint64_t)t * TW_RUN;
  if (start >= half_n) return;
  Fe<FP> cur = fe_pow_u64(omega, start);
  for (int k = 0; k < TW_RUN && start + k < half_n; k++) {
    Fe<FP> r = cur;
#pragma unroll
    for (int d = 0; d < 5; d++) r = fe_dbl(r);            // x 2^256 -> x 2^261
    fe_store<FP>(tw + 2 * (start + k), r);
    cur = fe_mul(cur, omega);
  }
}

template <class FP>
__global__ void __launch_bounds__(1024)
ntt29_pass_kernel(const U128* __restrict__ in, U128* __restrict__ out, const U128* __restrict__ tw, NttPass P,
                  size_t col_stride /* elements */, Fe<FP> scale29 /* R' form, canonical; used when P.has_scale */) {
  extern __shared__ U128 lds[];
  const uint32_t R = 1u << P.log_r, C = 1u << P.log_c;
  const uint32_t RC = R * C;
  const uint32_t n_half_log = P.log_n - 1;
  U128* tile0 = lds;                                   // limbs 0..3
  U128* tile1 = lds + RC;                              // limbs 4..7
  int32_t* tile2 = reinterpret_cast<int32_t*>(lds + 2 * RC);     // limb 8
  U128* twl0 = lds + 2 * RC + ((RC + 3) >> 2);         // radix twiddles, packed, two planes (absent when P.tw_global)
  U128* twl1 = twl0 + (R >> 1);

  const U128* src = in + 2 * col_stride * blockIdx.y;
  U128* dst = out + 2 * col_stride * blockIdx.y;
  const uint32_t tid = threadIdx.x, nthr = blockDim.x;
  const uint32_t tile = blockIdx.x;

  uint64_t in_base, in_j_stride, in_c_stride;
  uint32_t i_first = 0;
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
    const uint32_t groups_log = P.log_r1 - P.log_c;
    const uint32_t k1c = tile & ((1u << groups_log) - 1), k2 = tile >> groups_log;
    const uint32_t k1 = k1c << P.log_c;
    in_base = (((uint64_t)k1 << P.log_r2) + k2) << P.log_r;
    in_j_stride = 1;
    in_c_stride = (uint64_t)1 << (P.log_r2 + P.log_r);
    out_base = (uint64_t)k1 + ((uint64_t)k2 << P.log_r1);
    out_k_stride = (uint64_t)1 << (P.log_r1 + P.log_r2);
    out_c_stride = 1;
  }

  auto lds_put = [&](uint32_t i, const Fe29<FP>& u) {
    tile0[i] = U128{(uint32_t)u.v[0], (uint32_t)u.v[1], (uint32_t)u.v[2], (uint32_t)u.v[3]};
    tile1[i] = U128{(uint32_t)u.v[4], (uint32_t)u.v[5], (uint32_t)u.v[6], (uint32_t)u.v[7]};
    tile2[i] = u.v[8];
  };
  auto lds_get = [&](uint32_t i) {
    const U128 a0 = tile0[i], a1 = tile1[i];
    Fe29<FP> x;
    x.v[0] = (int32_t)a0.x; x.v[1] = (int32_t)a0.y; x.v[2] = (int32_t)a0.z; x.v[3] = (int32_t)a0.w;
    x.v[4] = (int32_t)a1.x; x.v[5] = (int32_t)a1.y; x.v[6] = (int32_t)a1.z; x.v[7] = (int32_t)a1.w;
    x.v[8] = tile2[i];
    return x;
  };
  const uint32_t tw_shift = P.log_n - P.log_r;
  auto tw_get = [&](uint32_t i) {
    if (P.tw_global) return fe29_unpack(fe_load<FP>(tw + 2 * ((uint64_t)i << tw_shift)));
    const U128 t0 = twl0[i], t1 = twl1[i];
    Fe<FP> t;
    t.v[0] = t0.x; t.v[1] = t0.y; t.v[2] = t0.z; t.v[3] = t0.w;
    t.v[4] = t1.x; t.v[5] = t1.y; t.v[6] = t1.z; t.v[7] = t1.w;
    return fe29_unpack(t);
  };

  // radix twiddles w_R^i = w^(i * n/R), i < R/2 (R' form, packed)
  if (!P.tw_global)
    for (uint32_t i = tid; i < (R >> 1); i += nthr) {
      const U128* t = tw + 2 * ((uint64_t)i << (P.log_n - P.log_r));
      twl0[i] = t[0];
      twl1[i] = t[1];
    }
  // load the tile, bit-reversing j on the way in; the API's bytes are read as they are (x 2^256 = R' form of x / 32)
  // (four elements per thread: unrolled so that the four loads are in flight together)
  if (!P.is_final) {
#pragma unroll 4
    for (uint32_t e = tid; e < RC; e += nthr) {
      const uint32_t cc = e & (C - 1), j = e >> P.log_c;
      const U128* g = src + 2 * (in_base + (uint64_t)j * in_j_stride + cc);
      lds_put((h2_bitrev(j, P.log_r) << P.log_c) + cc, fe29_unpack(fe_load<FP>(g)));
    }
  } else {
#pragma unroll 4
    for (uint32_t e = tid; e < RC; e += nthr) {
      const uint32_t j = e & (R - 1), cc = e >> P.log_r;
      const U128* g = src + 2 * (in_base + (uint64_t)cc * in_c_stride + j);
      lds_put((h2_bitrev(j, P.log_r) << P.log_c) + cc, fe29_unpack(fe_load<FP>(g)));
    }
  }
  __syncthreads();

  uint32_t s = 0;
  if (P.log_r & 1) {
    for (uint32_t w = tid; w < (RC >> 1); w += nthr) {
      const uint32_t cc = w & (C - 1), b = w >> P.log_c;
      const uint32_t i0 = (b << (P.log_c + 1)) + cc, i1 = i0 + C;
      const Fe29<FP> x = lds_get(i0), y = lds_get(i1);
      lds_put(i0, fe29_norm(fe29_add(x, y)));
      lds_put(i1, fe29_norm(fe29_sub(x, y)));
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
      Fe29<FP> e0 = lds_get(i0), e1 = lds_get(i0 + step), e2 = lds_get(i0 + 2 * step), e3 = lds_get(i0 + 3 * step);
      const uint32_t tb = pos << (P.log_r - 2 - s);
      if (s != 0) {                      // s == 0: pos == 0, the twiddles of stage s and of the pair (e0, e2) are 1
        const Fe29<FP> ta = tw_get(pos << (P.log_r - 1 - s));
        e1 = fe29_mul(e1, ta);
        e3 = fe29_mul(e3, ta);
      }
      const Fe29<FP> a0 = fe29_add(e0, e1), a1 = fe29_sub(e0, e1);
      Fe29<FP> a2 = fe29_add(e2, e3), a3 = fe29_sub(e2, e3);
      if (s != 0) a2 = fe29_mul(a2, tw_get(tb));
      a3 = fe29_mul(a3, tw_get(tb + (R >> 2)));     // e2 - e3 of two stored values: limbs within +-2^29
      lds_put(i0, fe29_norm(fe29_add(a0, a2)));
      lds_put(i0 + step, fe29_norm(fe29_add(a1, a3)));
      lds_put(i0 + 2 * step, fe29_norm(fe29_sub(a0, a2)));
      lds_put(i0 + 3 * step, fe29_norm(fe29_sub(a1, a3)));
    }
    __syncthreads();
  }

  // write back: one product per element brings it to (-p/2, 3p/2) -- the inter-pass twiddle w^(outer * i * k), the
  // scale of a scaled transform, or 1 -- then canonical, packed
  Fe29<FP> one29;
  {
    Fe<FP> o = Fe<FP>::one();
#pragma unroll
    for (int d = 0; d < 5; d++) o = fe_dbl(o);
    one29 = fe29_unpack(o);
  }
  const Fe29<FP> sc29 = fe29_unpack(scale29);
  Fe29<FP> pl;
#pragma unroll
  for (int i = 0; i < 9; i++) pl.v[i] = (int32_t)fe29_p<FP>(i);
  for (uint32_t e = tid; e < RC; e += nthr) {
    const uint32_t cc = e & (C - 1), k = e >> P.log_c;
    Fe29<FP> x = lds_get(e);
    if (!P.is_final) {
      const uint64_t ex = ((uint64_t)(i_first + cc) * k) << P.log_outer;  // < n
      if (ex != 0) {
        const uint64_t half_n = (uint64_t)1 << n_half_log;
        const bool negate = ex >= half_n;
        const uint64_t ti = negate ? ex - half_n : ex;
        x = fe29_mul(x, fe29_unpack(fe_load<FP>(tw + 2 * ti)));
        if (negate) x = fe29_norm(fe29_sub(pl, x));          // p - x, still in (-p/2, 3p/2)
      } else {
        x = fe29_mul(x, one29);
      }
    } else {
      x = fe29_mul(x, P.has_scale ? sc29 : one29);
    }
    const Fe<FP> r = fe29_canonical_pack(x);
    U128* g = dst + 2 * (out_base + (uint64_t)k * out_k_stride + (uint64_t)cc * out_c_stride);
    g[0] = U128{r.v[0], r.v[1], r.v[2], r.v[3]};
    g[1] = U128{r.v[4], r.v[5], r.v[6], r.v[7]};
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------
inline size_t ntt29_lds_bytes(const NttPass& P) {
  const size_t rc = (size_t)1 << (P.log_r + P.log_c), r = (size_t)1 << P.log_r;
  size_t b = rc * 32 + ((rc * 4 + 15) & ~(size_t)15) + (P.tw_global ? 0 : (r / 2) * 32);
  return b < 64 ? 64 : b;
}
// a tile that leaves room for a second block on the CU only without its radix twiddles reads them from global memory
// (the 16 KB table of a 1024-row pass stays in the vector L1 / L2)
inline bool ntt29_tw_global(const NttPass& P) {
  static const int tune = tune_int("H2_TUNE_NTT_TWG", -1);     // tuning builds only (h2_tune.hpp)
  if (tune >= 0) return tune != 0;
  const size_t rc = (size_t)1 << (P.log_r + P.log_c), r = (size_t)1 << P.log_r;
  const size_t with = rc * 36 + (r / 2) * 32, without = rc * 36;
  return with > 80 * 1024 && without <= 80 * 1024;
}

// Enqueue the transform of m columns (column stride = n elements) on `stream`; data in place, scratch m*n elements
// when the plan has more than one pass; tw from ntt29_build_twiddles; scale (optional) in the API's Montgomery form.
template <class FP>
inline hipError_t ntt29_launch(U128* data, U128* scratch, const U128* tw, uint32_t log_n, size_t m,
                               hipStream_t stream, const Fe<FP>* scale = nullptr) {
  if (log_n == 0 || m == 0) return hipSuccess;
  NttPlan pl = ntt_make_plan(log_n);
  Fe<FP> sc = scale ? *scale : Fe<FP>::zero();
  for (int d = 0; d < 5; d++) sc = fe_dbl(sc);          // x 2^256 -> x 2^261: the scale as a working-form constant
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
    P.tw_global = ntt29_tw_global(P) ? 1u : 0u;
    hipLaunchKernelGGL(ntt29_pass_kernel<FP>, grid, dim3(pl.threads[p]), ntt29_lds_bytes(P), stream, src, dst, tw, P, n, sc);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
template <class FP>
inline hipError_t ntt29_kernel_setup() {
  return hipFuncSetAttribute((const void*)ntt29_pass_kernel<FP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
template <class FP>
inline hipError_t ntt29_build_twiddles(U128* tw, const Fe<FP>& omega, uint32_t log_n, hipStream_t stream) {
  if (log_n == 0) return hipSuccess;
  const uint32_t half_n = 1u << (log_n - 1);
  const uint32_t threads = (half_n + TW_RUN - 1) / TW_RUN;
  hipLaunchKernelGGL(ntt_twiddle29_kernel<FP>, dim3((threads + 255) / 256), dim3(256), 0, stream, tw, omega, half_n);
  return hipGetLastError();
}

}  // namespace h2
