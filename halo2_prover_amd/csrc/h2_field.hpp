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

#endif  // __HIP_DEVICE_COMPILE__

template <class FP>
H2_HD Fe<FP> fe_mul(const Fe<FP>& a, const Fe<FP>& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return fe_mul_comba(a, b);
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
