// h2_field.hpp -- 256-bit Montgomery prime fields for gfx950 (and the host side of the library).
//
// Replaces, on the device, the field types the reference's hot path computes in:
// halo2curves 0.3.2 bn256::{Fr,Fq} and pasta_curves 0.5.1 {Fp,Fq} (un-vendored dependencies
// pinned at /root/reference/circuits/Cargo.lock:854-856,1126-1128; imported by the reference at
// circuits/src/utils.rs:5 and circuits/src/wasm.rs:20).  Memory layout is exactly theirs:
// 4 x u64 little-endian limbs, Montgomery form with R = 2^256 -- seen here as 8 x u32 limbs,
// because CDNA4's integer multiplier is 32 x 32 -> 64 (v_mad_u64_u32).
//
// Every function is __host__ __device__: the host uses the same code for the few scalar
// operations outside the kernels (roots of unity, final window combination).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define H2_HD __host__ __device__ __forceinline__

namespace h2 {

#include "h2_constants.inc"

template <class FP>
struct Fe {
  uint32_t v[8];

  static H2_HD Fe zero() {
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
  }
  static H2_HD Fe one() {
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = FP::ONE(i);
    return r;
  }
  H2_HD bool is_zero() const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= v[i];
    return o == 0;
  }
  H2_HD bool operator==(const Fe& b) const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= v[i] ^ b.v[i];
    return o == 0;
  }
  H2_HD bool operator!=(const Fe& b) const { return !(*this == b); }
};

// 16-byte vector view used for all global / LDS traffic of field elements
struct alignas(16) U128 {
  uint32_t x, y, z, w;
};

template <class FP>
H2_HD Fe<FP> fe_load(const void* p) {
  const U128* q = reinterpret_cast<const U128*>(p);
  U128 lo = q[0], hi = q[1];
  Fe<FP> r;
  r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
  r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
  return r;
}
template <class FP>
H2_HD void fe_store(void* p, const Fe<FP>& a) {
  U128* q = reinterpret_cast<U128*>(p);
  q[0] = U128{a.v[0], a.v[1], a.v[2], a.v[3]};
  q[1] = U128{a.v[4], a.v[5], a.v[6], a.v[7]};
}

// r = a - p if a >= p else a   (a < 2p, optional carry word `top`)
template <class FP>
H2_HD void fe_reduce_once(uint32_t* r, const uint32_t* t, uint32_t top) {
  uint32_t s[8];
  uint64_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t d = (uint64_t)t[i] - FP::P(i) - br;
    s[i] = (uint32_t)d;
    br = (d >> 63) & 1;
  }
  bool ge = (top != 0) || (br == 0);
#pragma unroll
  for (int i = 0; i < 8; i++) r[i] = ge ? s[i] : t[i];
}

template <class FP>
H2_HD Fe<FP> fe_add(const Fe<FP>& a, const Fe<FP>& b) {
  uint32_t t[8];
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (uint64_t)a.v[i] + b.v[i];
    t[i] = (uint32_t)c;
    c >>= 32;
  }
  Fe<FP> r;
  fe_reduce_once<FP>(r.v, t, (uint32_t)c);  // p < 2^255: c is always 0, kept for generality
  return r;
}

template <class FP>
H2_HD Fe<FP> fe_sub(const Fe<FP>& a, const Fe<FP>& b) {
  uint32_t t[8];
  uint64_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t d = (uint64_t)a.v[i] - b.v[i] - br;
    t[i] = (uint32_t)d;
    br = (d >> 63) & 1;
  }
  // add p back when the subtraction borrowed
  uint32_t mask = (uint32_t)0 - (uint32_t)br;
  uint64_t c = 0;
  Fe<FP> r;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (uint64_t)t[i] + (FP::P(i) & mask);
    r.v[i] = (uint32_t)c;
    c >>= 32;
  }
  return r;
}

template <class FP>
H2_HD Fe<FP> fe_neg(const Fe<FP>& a) {
  return fe_sub(Fe<FP>::zero(), a);
}
template <class FP>
H2_HD Fe<FP> fe_dbl(const Fe<FP>& a) {
  return fe_add(a, a);
}

// Montgomery product a*b*R^-1 mod p, CIOS over 32-bit limbs: the portable form (host side of the
// library; also the statement the device form below is tested against).
template <class FP>
H2_HD Fe<FP> fe_mul_cios(const Fe<FP>& a, const Fe<FP>& b) {
  uint32_t t[10];
#pragma unroll
  for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      uint64_t x = (uint64_t)a.v[j] * b.v[i] + t[j] + c;
      t[j] = (uint32_t)x;
      c = x >> 32;
    }
    uint64_t x = (uint64_t)t[8] + c;
    t[8] = (uint32_t)x;
    t[9] = (uint32_t)(x >> 32);
    uint32_t m = t[0] * FP::INV;
    c = ((uint64_t)m * FP::P(0) + t[0]) >> 32;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      uint64_t y = (uint64_t)m * FP::P(j) + t[j] + c;
      t[j - 1] = (uint32_t)y;
      c = y >> 32;
    }
    x = (uint64_t)t[8] + c;
    t[7] = (uint32_t)x;
    t[8] = t[9] + (uint32_t)(x >> 32);
  }
  Fe<FP> r;
  fe_reduce_once<FP>(r.v, t, t[8]);
  return r;
}

#if defined(__HIP_DEVICE_COMPILE__)
// gfx950 form: Montgomery product scanning (Comba / FIPS).  Measured on MI355X (tools/microbench_valu.hip):
// v_mad_u64_u32 issues every ~5.3 cycles per wave and SIMD, only ~2.3x a 32-bit add, so the cost of a
// 256-bit product is set by the adds and moves AROUND the 128 multiplies.  Each column keeps a 64-bit
// accumulator that v_mad_u64_u32 updates in place (no register-pair shuffling) plus a 32-bit count of
// carry-outs fed by one v_addc_co_u32 per product: 2 instructions per limb product instead of ~4.5 for
// the CIOS form above (1.3x-1.4x faster end to end).  The modulus limbs are compile-time constants:
// for the Pasta primes (p = 2^254 + t, p = 1 mod 2^32) the reduction columns collapse to three
// multiplies, an add and a shift.
namespace detail {
// acc += a * b (64-bit, in place); cnt += carry out
__device__ __forceinline__ void mac_vv(uint64_t& acc, uint32_t& cnt, uint32_t a, uint32_t b) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(acc), "+v"(cnt) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ void mac_vs(uint64_t& acc, uint32_t& cnt, uint32_t a, uint32_t b_const) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(acc), "+v"(cnt) : "v"(a), "s"(b_const) : "vcc");
}
__device__ __forceinline__ void add64(uint64_t& acc, uint32_t& cnt, uint64_t x) {
  const uint64_t n = acc + x;
  cnt += (n < x) ? 1u : 0u;
  acc = n;
}
// acc += m * P[J] with the constant limb folded: 0 -> nothing, 1 -> add, 2^s -> shifted add
template <class FP, int J>
__device__ __forceinline__ void mac_modulus(uint64_t& acc, uint32_t& cnt, uint32_t m) {
  constexpr uint32_t pj = FP::P(J);
  if constexpr (pj == 0) {
  } else if constexpr (pj == 1) {
    add64(acc, cnt, (uint64_t)m);
  } else if constexpr ((pj & (pj - 1)) == 0) {
    add64(acc, cnt, (uint64_t)m << __builtin_ctz(pj));
  } else {
    mac_vs(acc, cnt, m, pj);
  }
}
template <class FP, int K, int I>
__device__ __forceinline__ void col_ab(uint64_t& acc, uint32_t& cnt, const uint32_t* a, const uint32_t* b) {
  if constexpr (I <= (K < 8 ? K : 7)) {
    mac_vv(acc, cnt, a[I], b[K - I]);
    col_ab<FP, K, I + 1>(acc, cnt, a, b);
  }
}
template <class FP, int K, int I, int IEND>
__device__ __forceinline__ void col_mp(uint64_t& acc, uint32_t& cnt, const uint32_t* m) {
  if constexpr (I <= IEND) {
    mac_modulus<FP, K - I>(acc, cnt, m[I]);
    col_mp<FP, K, I + 1, IEND>(acc, cnt, m);
  }
}
// column K of  a*b + m*p : low half fixes m[K] so that the column's low word vanishes
template <class FP, int K>
__device__ __forceinline__ void columns(uint64_t& acc, uint32_t& cnt, const uint32_t* a, const uint32_t* b,
                                        uint32_t* m, uint32_t* t) {
  if constexpr (K < 16) {
    col_ab<FP, K, (K < 8 ? 0 : K - 7)>(acc, cnt, a, b);
    if constexpr (K < 8) {
      col_mp<FP, K, 0, K - 1>(acc, cnt, m);
      m[K] = (uint32_t)acc * FP::INV;
      mac_modulus<FP, 0>(acc, cnt, m[K]);
    } else {
      if constexpr (K < 15) col_mp<FP, K, K - 7, 7>(acc, cnt, m);
      t[K - 8] = (uint32_t)acc;
    }
    acc = (acc >> 32) | ((uint64_t)cnt << 32);
    cnt = 0;
    columns<FP, K + 1>(acc, cnt, a, b, m, t);
  }
}
}  // namespace detail

template <class FP>
__device__ __forceinline__ Fe<FP> fe_mul_comba(const Fe<FP>& a, const Fe<FP>& b) {
  uint64_t acc = 0;
  uint32_t cnt = 0;
  uint32_t m[8], t[9];
  detail::columns<FP, 0>(acc, cnt, a.v, b.v, m, t);
  t[8] = (uint32_t)acc;
  Fe<FP> r;
  fe_reduce_once<FP>(r.v, t, t[8]);
  return r;
}

// Latency form of the same product, for kernels that run ONE wave per SIMD (the MSM tails on quads,
// h2_curve_quad.hpp).  fe_mul_comba threads all ~90 multiply-adds through one accumulator, so a lone wave waits
// out the full latency of every v_mad_u64_u32 (measured 1600-2350 cycles per product against 1116 when four waves
// share the SIMD).  Here every column K of a*b + m*p has its OWN accumulator: the 64 limb products are issued row
// by row, so consecutive multiply-adds hit different columns and are independent; only the reduction walks the
// columns in order (carry in, m[K], carry out), scattering m[K] * p[J] into the columns above it as soon as m[K]
// exists.  ~6 % more VALU instructions but a fifth of the s_nops, ~45 more live registers -- which a lone wave has to
// spare.  Measured in the throughput-bound kernels instead of fe_mul_comba it is SLOWER (accumulate kernel 0.437 ->
// 0.460 ms, NTT passes 0.63 -> 0.70 ms per step): there other waves fill the s_nop slots and only the VALU count matters.
namespace detail {
// Blocks of independent multiply-adds: the carry of each v_mad_u64_u32 goes to its OWN scalar register pair and is
// folded into the column's carry count only after the whole block has been issued, so the in-order pipeline never
// waits on the instruction just before (one mad + its addc back to back would).
__device__ __forceinline__ void mul4_first(uint64_t* acc, uint32_t a, const uint32_t* b) {
  asm("v_mad_u64_u32 %0, vcc, %4, %5, 0\n\t"
      "v_mad_u64_u32 %1, vcc, %4, %6, 0\n\t"
      "v_mad_u64_u32 %2, vcc, %4, %7, 0\n\t"
      "v_mad_u64_u32 %3, vcc, %4, %8, 0"
      : "=&v"(acc[0]), "=&v"(acc[1]), "=&v"(acc[2]), "=&v"(acc[3])
      : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]) : "vcc");
}
__device__ __forceinline__ void mac4_vv(uint64_t* acc, uint32_t* cnt, uint32_t a, const uint32_t* b) {
  uint64_t c0, c1, c2, c3;
  asm("v_mad_u64_u32 %0, %8, %12, %13, %0\n\t"
      "v_mad_u64_u32 %1, %9, %12, %14, %1\n\t"
      "v_mad_u64_u32 %2, %10, %12, %15, %2\n\t"
      "v_mad_u64_u32 %3, %11, %12, %16, %3\n\t"
      "v_addc_co_u32_e64 %4, %8, 0, %4, %8\n\t"
      "v_addc_co_u32_e64 %5, %9, 0, %5, %9\n\t"
      "v_addc_co_u32_e64 %6, %10, 0, %6, %10\n\t"
      "v_addc_co_u32_e64 %7, %11, 0, %7, %11"
      : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(cnt[0]), "+v"(cnt[1]), "+v"(cnt[2]), "+v"(cnt[3]),
        "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3)
      : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}
// three accumulating columns and a fresh fourth (the last product of rows 1..7 opens column I + 7)
__device__ __forceinline__ void mac3_vv_first(uint64_t* acc, uint32_t* cnt, uint32_t a, const uint32_t* b) {
  uint64_t c0, c1, c2;
  asm("v_mad_u64_u32 %0, %7, %10, %11, %0\n\t"
      "v_mad_u64_u32 %1, %8, %10, %12, %1\n\t"
      "v_mad_u64_u32 %2, %9, %10, %13, %2\n\t"
      "v_mad_u64_u32 %3, vcc, %10, %14, 0\n\t"
      "v_addc_co_u32_e64 %4, %7, 0, %4, %7\n\t"
      "v_addc_co_u32_e64 %5, %8, 0, %5, %8\n\t"
      "v_addc_co_u32_e64 %6, %9, 0, %6, %9"
      : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "=&v"(acc[3]), "+v"(cnt[0]), "+v"(cnt[1]), "+v"(cnt[2]),
        "=&s"(c0), "=&s"(c1), "=&s"(c2)
      : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]) : "vcc");
}
// m * (four / three consecutive modulus limbs), limbs as scalar constants
__device__ __forceinline__ void mac4_vs(uint64_t* acc, uint32_t* cnt, uint32_t m, uint32_t p0, uint32_t p1, uint32_t p2,
                                        uint32_t p3) {
  uint64_t c0, c1, c2, c3;
  asm("v_mad_u64_u32 %0, %8, %12, %13, %0\n\t"
      "v_mad_u64_u32 %1, %9, %12, %14, %1\n\t"
      "v_mad_u64_u32 %2, %10, %12, %15, %2\n\t"
      "v_mad_u64_u32 %3, %11, %12, %16, %3\n\t"
      "v_addc_co_u32_e64 %4, %8, 0, %4, %8\n\t"
      "v_addc_co_u32_e64 %5, %9, 0, %5, %9\n\t"
      "v_addc_co_u32_e64 %6, %10, 0, %6, %10\n\t"
      "v_addc_co_u32_e64 %7, %11, 0, %7, %11"
      : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(cnt[0]), "+v"(cnt[1]), "+v"(cnt[2]), "+v"(cnt[3]),
        "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3)
      : "v"(m), "s"(p0), "s"(p1), "s"(p2), "s"(p3));
}
__device__ __forceinline__ void mac3_vs(uint64_t* acc, uint32_t* cnt, uint32_t m, uint32_t p0, uint32_t p1, uint32_t p2) {
  uint64_t c0, c1, c2;
  asm("v_mad_u64_u32 %0, %6, %9, %10, %0\n\t"
      "v_mad_u64_u32 %1, %7, %9, %11, %1\n\t"
      "v_mad_u64_u32 %2, %8, %9, %12, %2\n\t"
      "v_addc_co_u32_e64 %3, %6, 0, %3, %6\n\t"
      "v_addc_co_u32_e64 %4, %7, 0, %4, %7\n\t"
      "v_addc_co_u32_e64 %5, %8, 0, %5, %8"
      : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(cnt[0]), "+v"(cnt[1]), "+v"(cnt[2]),
        "=&s"(c0), "=&s"(c1), "=&s"(c2)
      : "v"(m), "s"(p0), "s"(p1), "s"(p2));
}
template <class FP, int I>
__device__ __forceinline__ void rows_ab(uint64_t* acc, uint32_t* cnt, const uint32_t* a, const uint32_t* b) {
  if constexpr (I == 0) {
    mul4_first(acc, a[0], b);
    mul4_first(acc + 4, a[0], b + 4);
    rows_ab<FP, 1>(acc, cnt, a, b);
  } else if constexpr (I < 8) {
    mac4_vv(acc + I, cnt + I, a[I], b);
    mac3_vv_first(acc + I + 4, cnt + I + 4, a[I], b + 4);
    rows_ab<FP, I + 1>(acc, cnt, a, b);
  }
}
// a limb that needs a real multiply (0, 1 and powers of two are folded by mac_modulus)
template <class FP, int J>
constexpr bool limb_is_mul() {
  if constexpr (J > 7) {
    return false;
  } else {
    constexpr uint32_t pj = FP::P(J);
    return pj != 0 && (pj & (pj - 1)) != 0;
  }
}
// columns K + J0 .. K + J0 + N - 1 += m * p[J0 ..]
template <class FP, int K, int J0, int N>
__device__ __forceinline__ void scatter_mp(uint64_t* acc, uint32_t* cnt, uint32_t m) {
  if constexpr (N == 4 && limb_is_mul<FP, J0>() && limb_is_mul<FP, J0 + 1>() && limb_is_mul<FP, J0 + 2>() &&
                limb_is_mul<FP, J0 + 3>()) {
    mac4_vs(acc + K + J0, cnt + K + J0, m, FP::P(J0), FP::P(J0 + 1), FP::P(J0 + 2), FP::P(J0 + 3));
  } else if constexpr (N == 4 && !limb_is_mul<FP, J0>() && limb_is_mul<FP, J0 + 1>() && limb_is_mul<FP, J0 + 2>() &&
                       limb_is_mul<FP, J0 + 3>()) {
    mac_modulus<FP, J0>(acc[K + J0], cnt[K + J0], m);      // the Pasta shape: p[0] = 1, p[1..3] full limbs
    mac3_vs(acc + K + J0 + 1, cnt + K + J0 + 1, m, FP::P(J0 + 1), FP::P(J0 + 2), FP::P(J0 + 3));
  } else {
    mac_modulus<FP, J0>(acc[K + J0], cnt[K + J0], m);
    if constexpr (N > 1) scatter_mp<FP, K, J0 + 1, N - 1>(acc, cnt, m);
  }
}
template <class FP, int K>
__device__ __forceinline__ void reduce_cols(uint64_t* acc, uint32_t* cnt, uint64_t& carry, uint32_t* t) {
  if constexpr (K < 16) {
    add64(acc[K], cnt[K], carry);
    if constexpr (K < 8) {
      const uint32_t m = (uint32_t)acc[K] * FP::INV;
      scatter_mp<FP, K, 0, 4>(acc, cnt, m);       // limb 0 clears the low word of column K
      scatter_mp<FP, K, 4, 4>(acc, cnt, m);
    } else {
      t[K - 8] = (uint32_t)acc[K];
    }
    carry = (acc[K] >> 32) | ((uint64_t)cnt[K] << 32);
    reduce_cols<FP, K + 1>(acc, cnt, carry, t);
  }
}
}  // namespace detail

template <class FP>
__device__ __forceinline__ Fe<FP> fe_mul_rows(const Fe<FP>& a, const Fe<FP>& b) {
  uint64_t acc[16];
  uint32_t cnt[16], t[9];
#pragma unroll
  for (int k = 0; k < 16; k++) cnt[k] = 0;
  acc[15] = 0;                                     // column 15 only receives carries
  detail::rows_ab<FP, 0>(acc, cnt, a.v, b.v);
  uint64_t carry = 0;
  detail::reduce_cols<FP, 0>(acc, cnt, carry, t);
  t[8] = (uint32_t)carry;
  Fe<FP> r;
  fe_reduce_once<FP>(r.v, t, t[8]);
  return r;
}
#endif  // __HIP_DEVICE_COMPILE__

template <class FP>
H2_HD Fe<FP> fe_mul(const Fe<FP>& a, const Fe<FP>& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return fe_mul_comba(a, b);
#else
  return fe_mul_cios(a, b);
#endif
}
// the product for lone-wave (latency-bound) kernels
template <class FP>
H2_HD Fe<FP> fe_mul_lat(const Fe<FP>& a, const Fe<FP>& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return fe_mul_rows(a, b);
#else
  return fe_mul_cios(a, b);
#endif
}
template <class FP>
H2_HD Fe<FP> fe_sqr(const Fe<FP>& a) {
  return fe_mul(a, a);
}

// canonical (non-Montgomery) representation: a * R^-1
template <class FP>
H2_HD Fe<FP> fe_from_mont(const Fe<FP>& a) {
  Fe<FP> one = Fe<FP>::zero();
  one.v[0] = 1;
  return fe_mul(a, one);
}
template <class FP>
H2_HD Fe<FP> fe_to_mont(const Fe<FP>& a) {
  Fe<FP> r2;
#pragma unroll
  for (int i = 0; i < 8; i++) r2.v[i] = FP::R2(i);
  return fe_mul(a, r2);
}

// a^e for a 64-bit exponent (twiddle bases, powers of omega)
template <class FP>
H2_HD Fe<FP> fe_pow_u64(const Fe<FP>& a, uint64_t e) {
  Fe<FP> acc = Fe<FP>::one(), base = a;
  while (e) {
    if (e & 1) acc = fe_mul(acc, base);
    base = fe_sqr(base);
    e >>= 1;
  }
  return acc;
}

// a^(p-2); not constant time, not on any hot path
template <class FP>
H2_HD Fe<FP> fe_inv(const Fe<FP>& a) {
  uint32_t e[8];
#pragma unroll
  for (int i = 0; i < 8; i++) e[i] = FP::P(i);
  e[0] -= 2;  // p odd, p[0] >= 3 for all four primes? p[0]=1 for Pasta: handle borrow below
  if (FP::P(0) < 2) {
    // borrow through the limbs (Pasta: p = ...00000001)
    e[0] = FP::P(0) - 2;  // wraps
    int i = 1;
    while (i < 8) {
      uint32_t old = e[i];
      e[i] = old - 1;
      if (old != 0) break;
      i++;
    }
  }
  Fe<FP> acc = Fe<FP>::one();
  for (int i = 255; i >= 0; i--) {
    acc = fe_sqr(acc);
    if ((e[i >> 5] >> (i & 31)) & 1) acc = fe_mul(acc, a);
  }
  return acc;
}

}  // namespace h2
