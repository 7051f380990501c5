// h2_ntt29.hpp -- the NTT pass kernel: butterflies on the 9 x 29-bit lazy form (h2_field29.hpp).
//
// Device replacement for halo2_proofs::arithmetic::best_fft / recursive_butterfly_arithmetic (plan and contract in
// h2_ntt.hpp).  HBM keeps the API's format (4 x u64 Montgomery limbs, R = 2^256); between the tile load and the tile
// store everything is on the working form:
//   * x 2^256 mod p (the API's bytes) is the R' = 2^261 form of x / 32, and the transform is linear: the DATA need no
//     conversion at all, only the twiddle tables are built in R' form;
//   * a product costs 81 + 9k multiply-adds without carry counters or a conditional subtraction, additions and
//     subtractions are nine limb operations without carries; every value written back to LDS is carry-normalised
//     (limbs in [0, 2^29), the top limb takes the sign);
//   * magnitudes: a tile's inputs are below 3p/2 (canonical API data, or what a non-final pass stored: see below); a
//     radix-4 double stage adds at most two products (each in (-p/2, 3p/2)) to an element, the very first one (all
//     twiddles 1) at most quadruples it: <= 4 * 1.5p + 3p * 4 = 18p after the five double stages of a 1024-row tile,
//     well inside fe29_mul's |a| |b| <= 64 p^2 with a canonical twiddle as b.
// Round 3: what the pass did besides products:
//   * leaving a NON-FINAL pass an element x is multiplied by the inter-pass twiddle anyway.  It is first moved to
//     32p + x (or 32p - x where the exponent asks for -w: the negation costs nothing) -- positive, limbs below 2^30 --
//     so the product lies in [0, 1.5p) and its limbs pack to 256 bits as they are: no carry pass, no conditional
//     additions.  The stored value is a representative below 2^256, not the canonical one; only the next pass reads it;
//   * leaving the FINAL pass nothing is multiplied (round 2: a 240-instruction product with 1 on every element of every
//     forward transform): t = floor(x / 2^252) is read off the top limb and x - T[t] with T[t] = floor(t 2^252 / p) p
//     from a 300-entry table lies in [0, p + 2^252): one conditional subtraction makes it canonical;
//   * a transform scaled by a constant (EvaluationDomain::ifft's 1/n) in two passes uses inter-pass twiddles that carry
//     the constant (table built once per (omega, log n, constant)): no product in its final pass either;
//   * the radix twiddles are kept UNPACKED (36 bytes in three planes, in LDS or -- 1024-row tiles -- global memory):
//     three loads per use instead of two loads and a 27-instruction unpack, three times per radix-4 group.
// LDS: 36 bytes per element in three planes (two of 16 bytes, one of 4: ds_read_b128 x 2 + ds_read_b32).
#pragma once
#include "h2_field29.hpp"
#include "h2_ntt.hpp"
#include <algorithm>

namespace h2 {

// ---- tables ----------------------------------------------------------------------------------------------------------
// One allocation per (field, omega, log n, constant):
//   main   [n/2] x 32 B   first * omega^i in R' form, canonical, packed (the inter-pass twiddles; first = 1 or the constant)
//   radix  per pass: [R/2] unpacked twiddles omega^(i n / R) in three planes (16 B, 16 B, 4 B)
//   canon  [NTT_CANON_N] multiples of p, unpacked, one 36-byte row each: T[t + NTT_CANON_OFF] = floor(t 2^252 / p) p
constexpr int NTT_CANON_OFF = 192, NTT_CANON_N = 384;   // |x| <= 18p < 18 * 2^255: |t| = |x| / 2^252 <= 144
struct NttTables {
  size_t off_radix[3], off_canon, total;                // bytes from the start of the allocation
};
inline NttTables ntt29_tables(uint32_t log_n) {
  NttTables t{};
  const NttPlan pl = ntt_make_plan(log_n);
  size_t o = (((size_t)1 << log_n) / 2) * 32;
  if (o < 64) o = 64;
  for (int p = 0; p < 3; p++) {
    t.off_radix[p] = o;
    if (p < pl.npass) o += ((((size_t)1 << pl.pass[p].log_r) / 2) * 36 + 255) & ~(size_t)255;
  }
  t.off_canon = o;
  o += (size_t)NTT_CANON_N * 36;
  t.total = (o + 255) & ~(size_t)255;
  return t;
}

// main[i] = first * omega^i (R' form, canonical, packed) for i < half_n
template <class FP>
__global__ void __launch_bounds__(256) ntt_twiddle29_kernel(U128* tw, Fe<FP> omega, Fe<FP> first, uint32_t half_n) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t start = (uint64_t)t * TW_RUN;
  if (start >= half_n) return;
  Fe<FP> cur = fe_mul(first, fe_pow_u64(omega, start));
  for (int k = 0; k < TW_RUN && start + k < half_n; k++) {
    Fe<FP> r = cur;
#pragma unroll
    for (int d = 0; d < 5; d++) r = fe_dbl(r);            // x 2^256 -> x 2^261
    fe_store<FP>(tw + 2 * (start + k), r);
    cur = fe_mul(cur, omega);
  }
}
// radix[i] = omega^(i << shift), unpacked, planes of `half_r` entries: limbs 0..3 | limbs 4..7 | limb 8
template <class FP>
__global__ void __launch_bounds__(256) ntt_radix29_kernel(uint32_t* radix, Fe<FP> omega, uint32_t half_r, uint32_t shift) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= half_r) return;
  Fe<FP> r = fe_pow_u64(omega, (uint64_t)i << shift);
#pragma unroll
  for (int d = 0; d < 5; d++) r = fe_dbl(r);
  const Fe29<FP> u = fe29_unpack(r);
  uint32_t* p0 = radix + 4 * (size_t)i;
  uint32_t* p1 = radix + 4 * (size_t)half_r + 4 * (size_t)i;
  uint32_t* p2 = radix + 8 * (size_t)half_r + i;
#pragma unroll
  for (int l = 0; l < 4; l++) {
    p0[l] = (uint32_t)u.v[l];
    p1[l] = (uint32_t)u.v[4 + l];
  }
  p2[0] = (uint32_t)u.v[8];
}
// canon[t + OFF] = floor(t 2^252 / p) * p as nine normalised limbs (the top one signed): thread t finds the quotient q
// with 0 <= t 2^252 - q p < p by exact limb arithmetic starting from a floating-point estimate
template <class FP>
__global__ void __launch_bounds__(64) ntt_canon29_kernel(int32_t* canon) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= NTT_CANON_N) return;
  const int t = idx - NTT_CANON_OFF;
  double pd = 0;                                          // p / 2^232
  for (int j = 0; j < 9; j++) pd += (double)fe29_p<FP>(j) * exp2((double)(29 * j - 232));
  int q = (int)floor((double)t * 1048576.0 / pd);        // t 2^252 / p = t 2^20 / (p / 2^232)
  int64_t d[9];
  for (int tries = 0; tries < 8; tries++) {
    // d = t 2^252 - q p, carry-normalised (top limb signed)
    int64_t carry = 0;
    for (int j = 0; j < 9; j++) {
      int64_t v = -(int64_t)q * (int64_t)fe29_p<FP>(j) + carry + (j == 8 ? (int64_t)t * 1048576 : 0);
      if (j < 8) {
        d[j] = v & L29_MASK;
        carry = v >> 29;
      } else {
        d[j] = v;
      }
    }
    if (d[8] < 0) { q--; continue; }
    // d >= p ?
    int64_t borrow = 0;
    for (int j = 0; j < 9; j++) {
      const int64_t v = d[j] - (int64_t)fe29_p<FP>(j) + borrow;
      borrow = j < 8 ? (v >> 29) : v;                    // after the last limb: the sign of d - p
    }
    if (borrow >= 0) { q++; continue; }
    break;
  }
  // the row: q p, normalised
  int64_t carry = 0;
  for (int j = 0; j < 9; j++) {
    const int64_t v = (int64_t)q * (int64_t)fe29_p<FP>(j) + carry;
    if (j < 8) {
      canon[9 * idx + j] = (int32_t)(v & L29_MASK);
      carry = v >> 29;
    } else {
      canon[9 * idx + j] = (int32_t)v;
    }
  }
}

template <class FP>
__global__ void __launch_bounds__(1024)
ntt29_pass_kernel(const U128* __restrict__ in, U128* __restrict__ out, const U128* __restrict__ tw,
                  const uint32_t* __restrict__ radix /* this pass's unpacked radix twiddles */,
                  const int32_t* __restrict__ canon, NttPass P, size_t col_stride /* elements */,
                  Fe<FP> scale29 /* R' form, canonical; used when P.has_scale */) {
  extern __shared__ U128 lds[];
  const uint32_t R = 1u << P.log_r, C = 1u << P.log_c;
  const uint32_t RC = R * C;
  const uint32_t n_half_log = P.log_n - 1;
  U128* tile0 = lds;                                   // limbs 0..3
  U128* tile1 = lds + RC;                              // limbs 4..7
  int32_t* tile2 = reinterpret_cast<int32_t*>(lds + 2 * RC);     // limb 8
  U128* twl0 = lds + 2 * RC + ((RC + 3) >> 2);         // radix twiddles, unpacked, three planes (absent when P.tw_global)
  U128* twl1 = twl0 + (R >> 1);
  uint32_t* twl2 = reinterpret_cast<uint32_t*>(twl1 + (R >> 1));

  const U128* src = in + 2 * col_stride * blockIdx.y;
  U128* dst = out + 2 * col_stride * blockIdx.y;
  const uint32_t tid = threadIdx.x, nthr = blockDim.x;
  const uint32_t tile = blockIdx.x;

  // every stride is a power of two: addresses are shifts and adds (as 64-bit products they were 55 quarter-rate
  // multiplies per element in the write-back loop alone)
  uint64_t in_base, out_base;
  uint32_t in_j_shift, in_c_shift, out_k_shift;
  uint32_t i_first = 0;
  if (!P.is_final) {
    const uint32_t chunks = 1u << (P.log_inner - P.log_c);
    const uint32_t ic = tile & (chunks - 1), o = tile >> (P.log_inner - P.log_c);
    i_first = ic << P.log_c;
    in_base = ((uint64_t)o << (P.log_r + P.log_inner)) + i_first;
    in_j_shift = P.log_inner;
    in_c_shift = 0;
    out_base = in_base; out_k_shift = in_j_shift;
  } else {
    const uint32_t groups_log = P.log_r1 - P.log_c;
    const uint32_t k1c = tile & ((1u << groups_log) - 1), k2 = tile >> groups_log;
    const uint32_t k1 = k1c << P.log_c;
    in_base = (((uint64_t)k1 << P.log_r2) + k2) << P.log_r;
    in_j_shift = 0;
    in_c_shift = P.log_r2 + P.log_r;
    out_base = (uint64_t)k1 + ((uint64_t)k2 << P.log_r1);
    out_k_shift = P.log_r1 + P.log_r2;
  }
  src += 2 * in_base;
  dst += 2 * out_base;

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
  const U128* rg0 = reinterpret_cast<const U128*>(radix);
  const U128* rg1 = rg0 + (R >> 1);
  const uint32_t* rg2 = radix + 8 * (size_t)(R >> 1);
  auto tw_get = [&](uint32_t i, bool from_table = false) {
    U128 t0, t1;
    Fe29<FP> t;
    if (P.tw_global || from_table) {
      t0 = rg0[i]; t1 = rg1[i];
      t.v[8] = (int32_t)rg2[i];
    } else {
      t0 = twl0[i]; t1 = twl1[i];
      t.v[8] = (int32_t)twl2[i];
    }
    t.v[0] = (int32_t)t0.x; t.v[1] = (int32_t)t0.y; t.v[2] = (int32_t)t0.z; t.v[3] = (int32_t)t0.w;
    t.v[4] = (int32_t)t1.x; t.v[5] = (int32_t)t1.y; t.v[6] = (int32_t)t1.z; t.v[7] = (int32_t)t1.w;
    return t;
  };

  // radix twiddles w_R^i = w^(i * n/R), i < R/2 (R' form, unpacked)
  if (!P.tw_global)
    for (uint32_t i = tid; i < (R >> 1); i += nthr) {
      twl0[i] = rg0[i];
      twl1[i] = rg1[i];
      twl2[i] = rg2[i];
    }
  // ---- leaving the tile: element k of column cc of the tile, carry-normalised ------------------------------------------
  Fe29<FP> pl, p32;                                     // p and 32 p as limbs (32 p: the limbs of p shifted five bits up)
#pragma unroll
  for (int i = 0; i < 9; i++) pl.v[i] = (int32_t)fe29_p<FP>(i);
  {
    int64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int64_t v = ((int64_t)fe29_p<FP>(i) << 5) + carry;
      p32.v[i] = i < 8 ? (int32_t)(v & L29_MASK) : (int32_t)v;
      carry = v >> 29;
    }
  }
  auto emit = [&](uint32_t k, uint32_t cc, Fe29<FP> x) {
    Fe<FP> r;
    if (!P.is_final || P.has_scale) {
      // one product: the inter-pass twiddle w^(outer * i * k) (the table's entry 0 is 1, or the constant of a scaled
      // transform) or the final pass's scale.  32p +- x is positive (|x| <= 18p) with limbs below 2^30, so the product
      // is in [0, 50 p / 128 + p) and its limbs are those of a non-negative integer below 2^256: packed as they are
      Fe29<FP> t;
      bool negate = false;
      if (!P.is_final) {
        const uint32_t ex = ((i_first + cc) * k) << P.log_outer;            // < n <= 2^30
        const uint32_t half_n = 1u << n_half_log;
        negate = ex >= half_n;
        t = fe29_unpack(fe_load<FP>(tw + 2 * (size_t)(negate ? ex - half_n : ex)));
      } else {
        t = fe29_unpack(scale29);
      }
      x = negate ? fe29_sub(p32, x) : fe29_add(p32, x);
      x = fe29_mul(x, t);
      if (P.is_final) r = fe29_canonical_pack(x);       // a scaled final pass (one or three passes): canonical bytes
      else r = fe29_pack(x);
    } else {
      // no product: t = floor(x / 2^252) from the top limb (x is carry-normalised), x - T[t] in [0, p + 2^252)
      int32_t t = (x.v[8] >> 20) + NTT_CANON_OFF;
      t = t < 0 ? 0 : (t >= NTT_CANON_N ? NTT_CANON_N - 1 : t);       // (never: |x| <= 18p)
      const int32_t* row = canon + 9 * t;
      Fe29<FP> y;
#pragma unroll
      for (int i = 0; i < 9; i++) y.v[i] = x.v[i] - row[i];
      y = fe29_norm(y);
      const Fe29<FP> z = fe29_norm(fe29_sub(y, pl));
      r = fe29_pack(z.v[8] >= 0 ? z : y);
    }
    U128* g = dst + 2 * (((size_t)k << out_k_shift) + cc);
    g[0] = U128{r.v[0], r.v[1], r.v[2], r.v[3]};
    g[1] = U128{r.v[4], r.v[5], r.v[6], r.v[7]};
  };
  // element j of column cc of the tile, as the API's bytes (x 2^256 = the R' form of x / 32): one of the two shifts is 0
  auto load_el = [&](uint32_t j, uint32_t cc) {
    return fe29_unpack(fe_load<FP>(src + 2 * (((size_t)j << in_j_shift) + ((size_t)cc << in_c_shift))));
  };
  // one radix-4 group (two fused radix-2 stages s, s + 1) of elements e0..e3 at LDS distance `step`
  auto radix4 = [&](Fe29<FP>& e0, Fe29<FP>& e1, Fe29<FP>& e2, Fe29<FP>& e3, uint32_t s, uint32_t pos) {
    if (s != 0) {                        // s == 0: pos == 0, the twiddles of stage s and of the pair (e0, e2) are 1
      const Fe29<FP> ta = tw_get(pos << (P.log_r - 1 - s));
      e1 = fe29_mul(e1, ta);
      e3 = fe29_mul(e3, ta);
    }
    const uint32_t tb = pos << (P.log_r - 2 - s);
    const Fe29<FP> a0 = fe29_add(e0, e1), a1 = fe29_sub(e0, e1);
    Fe29<FP> a2 = fe29_add(e2, e3), a3 = fe29_sub(e2, e3);
    if (s != 0) a2 = fe29_mul(a2, tw_get(tb));
    // e2 - e3 of two stored values: limbs within +-2^29.  (s == 0 is the stage fused with the tile load: it runs before
    // the block's first barrier, when the LDS copy of the twiddles is still being written -- its one twiddle, the
    // fourth root of unity, comes from the table itself)
    a3 = fe29_mul(a3, tw_get(tb + (R >> 2), s == 0));
    e0 = fe29_norm(fe29_add(a0, a2));
    e1 = fe29_norm(fe29_add(a1, a3));
    e2 = fe29_norm(fe29_sub(a0, a2));
    e3 = fe29_norm(fe29_sub(a1, a3));
  };

  // Round 3 (late): the FIRST stage works on the elements as they arrive from HBM and the LAST one hands its results to
  // the exit code in registers -- two of a 1024-row tile's six LDS round trips and two of its six barriers are gone.
  // The thread that owns rows t, t + R/4, t + R/2, t + 3R/4 of the tile (its bit-reversed positions are the four
  // neighbours 4 brev(t) + 0..3) loads them -- consecutive threads still read consecutive addresses, as before.
  // Tiles of fewer than 8 rows (transforms of 2 or 4 elements) keep the plain sequence: load, stages, write back.
  const bool fused = P.log_r >= 3;
  uint32_t s = 0;
  if (fused) {
    if (P.log_r & 1) {
      const uint32_t lq = P.log_r - 1;                   // pairs (t, t + R/2): one radix-2 stage, twiddle 1
      for (uint32_t w = tid; w < (RC >> 1); w += nthr) {
        const uint32_t cc = P.is_final ? w >> lq : w & (C - 1), t = P.is_final ? w & ((1u << lq) - 1) : w >> P.log_c;
        const Fe29<FP> x = load_el(t, cc), y = load_el(t + (R >> 1), cc);
        const uint32_t i0 = (h2_bitrev(t, lq) << (P.log_c + 1)) + cc;
        lds_put(i0, fe29_norm(fe29_add(x, y)));
        lds_put(i0 + C, fe29_norm(fe29_sub(x, y)));
      }
      s = 1;
    } else {
      const uint32_t lq = P.log_r - 2;
      for (uint32_t w = tid; w < (RC >> 2); w += nthr) {
        const uint32_t cc = P.is_final ? w >> lq : w & (C - 1), t = P.is_final ? w & ((1u << lq) - 1) : w >> P.log_c;
        // LDS neighbours q = 0..3 of 4 brev(t) hold rows t + brev2(q) R/4
        Fe29<FP> e0 = load_el(t, cc), e1 = load_el(t + (R >> 1), cc), e2 = load_el(t + (R >> 2), cc),
                 e3 = load_el(t + 3 * (R >> 2), cc);
        radix4(e0, e1, e2, e3, 0, 0);
        const uint32_t i0 = (h2_bitrev(t, lq) << (P.log_c + 2)) + cc;
        lds_put(i0, e0);
        lds_put(i0 + C, e1);
        lds_put(i0 + 2 * C, e2);
        lds_put(i0 + 3 * C, e3);
      }
      s = 2;
    }
    __syncthreads();
  } else {
    // load the tile, bit-reversing j on the way in
    for (uint32_t e = tid; e < RC; e += nthr) {
      const uint32_t j = P.is_final ? e & (R - 1) : e >> P.log_c, cc = P.is_final ? e >> P.log_r : e & (C - 1);
      lds_put((h2_bitrev(j, P.log_r) << P.log_c) + cc, load_el(j, cc));
    }
    __syncthreads();
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
  }
  const uint32_t s_end = fused ? P.log_r - 2 : P.log_r;
  for (; s < s_end; s += 2) {
    const uint32_t h = 1u << s;
    for (uint32_t w = tid; w < (RC >> 2); w += nthr) {
      const uint32_t cc = w & (C - 1), b = w >> P.log_c;
      const uint32_t pos = b & (h - 1), grp = b >> s;
      const uint32_t i0 = (((grp << (s + 2)) + pos) << P.log_c) + cc;
      const uint32_t step = h << P.log_c;
      Fe29<FP> e0 = lds_get(i0), e1 = lds_get(i0 + step), e2 = lds_get(i0 + 2 * step), e3 = lds_get(i0 + 3 * step);
      radix4(e0, e1, e2, e3, s, pos);
      lds_put(i0, e0);
      lds_put(i0 + step, e1);
      lds_put(i0 + 2 * step, e2);
      lds_put(i0 + 3 * step, e3);
    }
    __syncthreads();
  }
  if (fused) {
    // the last double stage (s = log r - 2: one group, pos = the row): rows pos + q R/4 leave from registers
    const uint32_t step = (R >> 2) << P.log_c;
    for (uint32_t w = tid; w < (RC >> 2); w += nthr) {
      const uint32_t cc = w & (C - 1), pos = w >> P.log_c;
      const uint32_t i0 = (pos << P.log_c) + cc;
      Fe29<FP> e0 = lds_get(i0), e1 = lds_get(i0 + step), e2 = lds_get(i0 + 2 * step), e3 = lds_get(i0 + 3 * step);
      radix4(e0, e1, e2, e3, P.log_r - 2, pos);
      emit(pos, cc, e0);
      emit(pos + (R >> 2), cc, e1);
      emit(pos + 2 * (R >> 2), cc, e2);
      emit(pos + 3 * (R >> 2), cc, e3);
    }
  } else {
    for (uint32_t e = tid; e < RC; e += nthr) emit(e >> P.log_c, e & (C - 1), lds_get(e));
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------
inline size_t ntt29_lds_bytes(const NttPass& P) {
  const size_t rc = (size_t)1 << (P.log_r + P.log_c), r = (size_t)1 << P.log_r;
  size_t b = rc * 32 + ((rc * 4 + 15) & ~(size_t)15) + (P.tw_global ? 0 : (r / 2) * 36 + 16);
  return b < 64 ? 64 : b;
}
// a tile that leaves room for a second block on the CU only without its radix twiddles reads them from global memory
// (the 18 KB table of a 1024-row pass stays in the vector L1 / L2)
inline bool ntt29_tw_global(const NttPass& P) {
  static const int tune = tune_int("H2_TUNE_NTT_TWG", -1);     // tuning builds only (h2_tune.hpp)
  if (tune >= 0) return tune != 0;
  const size_t rc = (size_t)1 << (P.log_r + P.log_c), r = (size_t)1 << P.log_r;
  const size_t with = rc * 36 + (r / 2) * 36, without = rc * 36;
  return with > 80 * 1024 && without <= 80 * 1024;
}
// a scaled transform whose constant rides in the inter-pass twiddles (two passes): the table must have been built
// with that constant (ntt29_build_tables(..., scale))
inline bool ntt29_scale_in_table(uint32_t log_n) { return ntt_make_plan(log_n).npass == 2; }

// Enqueue the transform of m columns (column stride = n elements) on `stream`; data in place, scratch m*n elements
// when the plan has more than one pass; `tables` from ntt29_build_tables; scale (optional) in the API's Montgomery
// form -- with ntt29_scale_in_table(log_n) the tables must carry it.
template <class FP>
inline hipError_t ntt29_launch(U128* data, U128* scratch, const void* tables, uint32_t log_n, size_t m,
                               hipStream_t stream, const Fe<FP>* scale = nullptr) {
  if (log_n == 0 || m == 0) return hipSuccess;
  NttPlan pl = ntt_make_plan(log_n);
  const NttTables tb = ntt29_tables(log_n);
  const U128* tw = (const U128*)tables;
  const int32_t* canon = (const int32_t*)((const char*)tables + tb.off_canon);
  Fe<FP> sc = scale ? *scale : Fe<FP>::zero();
  for (int d = 0; d < 5; d++) sc = fe_dbl(sc);          // x 2^256 -> x 2^261: the scale as a working-form constant
  const bool in_table = scale && ntt29_scale_in_table(log_n);
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
    P.has_scale = (scale && P.is_final && !in_table) ? 1u : 0u;
    P.tw_global = ntt29_tw_global(P) ? 1u : 0u;
    const uint32_t* radix = (const uint32_t*)((const char*)tables + tb.off_radix[p]);
    static const size_t lds_pad = (size_t)tune_int("H2_TUNE_NTT_LDS_PAD", 0);      // tuning builds: fewer blocks per CU
    hipLaunchKernelGGL(ntt29_pass_kernel<FP>, grid, dim3(pl.threads[p]), std::min<size_t>(ntt29_lds_bytes(P) + lds_pad, 160 * 1024), stream, src, dst, tw, radix,
                       canon, P, n, sc);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
template <class FP>
inline hipError_t ntt29_kernel_setup() {
  return hipFuncSetAttribute((const void*)ntt29_pass_kernel<FP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
// `tables`: ntt29_tables(log_n).total bytes.  scale (API Montgomery form) or null.
template <class FP>
inline hipError_t ntt29_build_tables(void* tables, const Fe<FP>& omega, uint32_t log_n, hipStream_t stream,
                                     const Fe<FP>* scale = nullptr) {
  if (log_n == 0) return hipSuccess;
  const NttPlan pl = ntt_make_plan(log_n);
  const NttTables tb = ntt29_tables(log_n);
  const uint32_t half_n = 1u << (log_n - 1);
  const uint32_t threads = (half_n + TW_RUN - 1) / TW_RUN;
  const Fe<FP> first = (scale && ntt29_scale_in_table(log_n)) ? *scale : Fe<FP>::one();
  hipLaunchKernelGGL(ntt_twiddle29_kernel<FP>, dim3((threads + 255) / 256), dim3(256), 0, stream, (U128*)tables, omega, first, half_n);
  for (int p = 0; p < pl.npass; p++) {
    const uint32_t half_r = (1u << pl.pass[p].log_r) / 2;
    if (half_r == 0) continue;
    hipLaunchKernelGGL(ntt_radix29_kernel<FP>, dim3((half_r + 255) / 256), dim3(256), 0, stream,
                       (uint32_t*)((char*)tables + tb.off_radix[p]), omega, half_r, log_n - pl.pass[p].log_r);
  }
  hipLaunchKernelGGL(ntt_canon29_kernel<FP>, dim3((NTT_CANON_N + 63) / 64), dim3(64), 0, stream,
                     (int32_t*)((char*)tables + tb.off_canon));
  return hipGetLastError();
}

}  // namespace h2
