// h2_field29.hpp -- the MSM's working representation of base-field elements: 9 signed limbs of 29 bits,
// Montgomery form with R' = 2^261, values kept only loosely reduced.
//
// Why (measured, tools/microbench_limb29.hip -> profiles/r01_microbench_limb29.txt): on 8 x 32-bit limbs every
// v_mad_u64_u32 of the Montgomery product needs a v_addc_co_u32 to count the carry out of its 64-bit column and the
// column hand-over costs three moves (h2_field.hpp, fe_mul_comba).  A column of 29-bit limb products -- at most nine
// a_i b_j and nine m_i p_j -- cannot overflow 64 bits, so there is no carry counter, the hand-over is one shift, and
// because R' is 2^7 times the modulus there is no conditional subtraction either.  Price: 81 + 9 k multiply-adds
// (k = non-trivial modulus limbs: 4 for the Pasta primes, 9 for BN254) instead of 64 + 8 k'.  Result on MI355X:
// 167 against 122 G modmul/s with four waves per SIMD, 144 against 88 for a lone wave (Pasta Fp).
//
// Invariants (the callers in h2_curve29.hpp are written against these):
//   value   x R' mod p, as the integer  sum_i v[i] 2^(29 i)  with SIGNED limbs -- it may be negative or exceed p.
//   fe29_mul(a, b): needs |a_i| < 2^30 and |b_j| < 2^29 (or the other way round: the sums of 18 limb products must
//           stay below 2^63), and |a| |b| <= 64 p^2.  Returns limbs in [0, 2^29) (top limb signed, small) and a
//           value in (-p/2, 3p/2).
//   fe29_add / fe29_sub / fe29_neg: limb-wise, no carries: limb magnitudes add up.  A difference of two normalised
//           values has limbs of magnitude < 2^29 and can go straight into a product; anything built from three or
//           more terms is passed through fe29_norm (carry propagation: limbs back in [0, 2^29), value unchanged).
//   exact zero (all limbs 0) only ever arises from the identity's coordinates, so `is_zero_exact` is the identity
//   test; `fe29_is_zero_mod_p` is the real test for the exceptional cases of the addition formulas.
//
// Conversions need no new constants: the API form x 2^256 is the R' form of x / 32, so going in is five doublings
// in the 32-bit-limb field, and going out is one product with FP::ONE (= 2^256 mod p) read as a plain integer.
#pragma once
#include "h2_field.hpp"

namespace h2 {

constexpr uint32_t L29_MASK = (1u << 29) - 1;

template <class FP>
struct Fe29 {
  int32_t v[9];
  static H2_HD Fe29 zero() {
    Fe29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = 0;
    return r;
  }
  H2_HD bool is_zero_exact() const {
    int32_t o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o |= v[i];
    return o == 0;
  }
};

// limb j (29 bits) of the modulus, from its 32-bit limbs
template <class FP>
H2_HD constexpr uint32_t fe29_p(int j) {
  const int bit = 29 * j;
  const int w = bit / 32, s = bit % 32;
  const uint64_t lo = w < 8 ? FP::P(w) : 0, hi = w + 1 < 8 ? FP::P(w + 1) : 0;
  return (uint32_t)(((lo | (hi << 32)) >> s) & L29_MASK);
}

// 8 x 32 bits (a non-negative integer below 2^256) -> 9 x 29 bits
template <class FP>
H2_HD Fe29<FP> fe29_unpack(const Fe<FP>& a) {
  Fe29<FP> r;
#pragma unroll
  for (int j = 0; j < 9; j++) {
    const int bit = 29 * j, w = bit >> 5, s = bit & 31;
    const uint64_t lo = a.v[w], hi = w + 1 < 8 ? a.v[w + 1] : 0;
    r.v[j] = (int32_t)((uint32_t)((lo | (hi << 32)) >> s) & L29_MASK);
  }
  return r;
}
// 9 normalised limbs of a non-negative integer below 2^256 -> 8 x 32 bits
template <class FP>
H2_HD Fe<FP> fe29_pack(const Fe29<FP>& a) {
  Fe<FP> r;
#pragma unroll
  for (int w = 0; w < 8; w++) {
    const int bit = 32 * w, j = bit / 29, off = bit % 29;
    uint64_t x = (uint64_t)(uint32_t)a.v[j] >> off;
    x |= (uint64_t)(uint32_t)a.v[j + 1] << (29 - off);
    if (j + 2 < 9) x |= (uint64_t)(uint32_t)a.v[j + 2] << (58 - off);
    r.v[w] = (uint32_t)x;
  }
  return r;
}

template <class FP>
H2_HD Fe29<FP> fe29_add(const Fe29<FP>& a, const Fe29<FP>& b) {
  Fe29<FP> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
  return r;
}
template <class FP>
H2_HD Fe29<FP> fe29_sub(const Fe29<FP>& a, const Fe29<FP>& b) {
  Fe29<FP> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] - b.v[i];
  return r;
}
template <class FP>
H2_HD Fe29<FP> fe29_neg(const Fe29<FP>& a) {
  Fe29<FP> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = -a.v[i];
  return r;
}
// carry propagation: same value, limbs 0..7 in [0, 2^29), the top limb takes the sign
template <class FP>
H2_HD Fe29<FP> fe29_norm(const Fe29<FP>& a) {
  Fe29<FP> r;
  int32_t carry = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int32_t x = a.v[i] + carry;
    r.v[i] = (int32_t)((uint32_t)x & L29_MASK);
    carry = x >> 29;                      // arithmetic
  }
  r.v[8] = a.v[8] + carry;
  return r;
}

namespace detail29 {
// acc += m * p[J], the constant limb folded: 0 -> nothing, 1 -> add, 2^s -> shifted add
template <class FP, int J>
H2_HD void mac_p(int64_t& acc, int32_t m) {
  constexpr uint32_t pj = fe29_p<FP>(J);
  if constexpr (pj == 0) {
  } else if constexpr (pj == 1) {
    acc += m;
  } else if constexpr ((pj & (pj - 1)) == 0) {
    acc += (int64_t)m << __builtin_ctz(pj);
  } else {
    acc += (int64_t)m * (int32_t)pj;
  }
}
// m with (acc + m p) = 0 mod 2^29: m = acc * (-p^-1) mod 2^29.  For the Pasta primes p = 1 mod 2^29, so that is -acc:
// a subtraction instead of a quarter-rate v_mul_lo_u32, nine times per product
template <class FP>
H2_HD int32_t mont_m(uint32_t acc_lo) {
  constexpr uint32_t ninv = FP::INV & L29_MASK;
  if constexpr (ninv == L29_MASK) return (int32_t)((0u - acc_lo) & L29_MASK);
  else return (int32_t)((acc_lo * ninv) & L29_MASK);
}
template <class FP, int K, int I, int IEND>
H2_HD void col_ab(int64_t& acc, const int32_t* a, const int32_t* b) {
  if constexpr (I <= IEND) {
    acc += (int64_t)a[I] * b[K - I];
    col_ab<FP, K, I + 1, IEND>(acc, a, b);
  }
}
template <class FP, int K, int I, int IEND>
H2_HD void col_mp(int64_t& acc, const int32_t* m) {
  if constexpr (I <= IEND) {
    mac_p<FP, K - I>(acc, m[I]);
    col_mp<FP, K, I + 1, IEND>(acc, m);
  }
}
// column K of a*b + m*p; the low half fixes m[K] so that the column's low 29 bits vanish
template <class FP, int K>
H2_HD void columns(int64_t& acc, const int32_t* a, const int32_t* b, int32_t* m, int32_t* t) {
  if constexpr (K < 17) {
    col_ab<FP, K, (K < 9 ? 0 : K - 8), (K < 9 ? K : 8)>(acc, a, b);
    if constexpr (K < 9) {
      if constexpr (K > 0) col_mp<FP, K, 0, K - 1>(acc, m);
      m[K] = mont_m<FP>((uint32_t)acc);
      mac_p<FP, 0>(acc, m[K]);
    } else {
      col_mp<FP, K, K - 8, 8>(acc, m);
      t[K - 9] = (int32_t)((uint32_t)acc & L29_MASK);
    }
    acc >>= 29;                                                               // arithmetic: exact below K = 9
    columns<FP, K + 1>(acc, a, b, m, t);
  }
}
// column K of a^2: cross terms with the doubled operand, the square of the middle limb once
template <class FP, int K, int I, int IEND>
H2_HD void col_sq(int64_t& acc, const int32_t* a, const int32_t* a2) {
  if constexpr (I <= IEND) {
    if constexpr (I < K - I) acc += (int64_t)a[I] * a2[K - I];
    else if constexpr (I == K - I) acc += (int64_t)a[I] * a[I];
    col_sq<FP, K, I + 1, IEND>(acc, a, a2);
  }
}
template <class FP, int K>
H2_HD void columns_sq(int64_t& acc, const int32_t* a, const int32_t* a2, int32_t* m, int32_t* t) {
  if constexpr (K < 17) {
    col_sq<FP, K, (K < 9 ? 0 : K - 8), K / 2>(acc, a, a2);
    if constexpr (K < 9) {
      if constexpr (K > 0) col_mp<FP, K, 0, K - 1>(acc, m);
      m[K] = mont_m<FP>((uint32_t)acc);
      mac_p<FP, 0>(acc, m[K]);
    } else {
      col_mp<FP, K, K - 8, 8>(acc, m);
      t[K - 9] = (int32_t)((uint32_t)acc & L29_MASK);
    }
    acc >>= 29;
    columns_sq<FP, K + 1>(acc, a, a2, m, t);
  }
}
// column K of a b + c d
template <class FP, int K>
H2_HD void columns2(int64_t& acc, const int32_t* a, const int32_t* b, const int32_t* c, const int32_t* d, int32_t* m, int32_t* t) {
  if constexpr (K < 17) {
    col_ab<FP, K, (K < 9 ? 0 : K - 8), (K < 9 ? K : 8)>(acc, a, b);
    col_ab<FP, K, (K < 9 ? 0 : K - 8), (K < 9 ? K : 8)>(acc, c, d);
    if constexpr (K < 9) {
      if constexpr (K > 0) col_mp<FP, K, 0, K - 1>(acc, m);
      m[K] = mont_m<FP>((uint32_t)acc);
      mac_p<FP, 0>(acc, m[K]);
    } else {
      col_mp<FP, K, K - 8, 8>(acc, m);
      t[K - 9] = (int32_t)((uint32_t)acc & L29_MASK);
    }
    acc >>= 29;
    columns2<FP, K + 1>(acc, a, b, c, d, m, t);
  }
}
}  // namespace detail29

// The limb products are signed 32 x 32 -> 64 multiply-adds: ONE v_mad_i64_i32 each.  But where the optimiser can PROVE a
// limb non-negative (a value just unpacked or masked) it rewrites that operand's sign extension as a zero extension,
// and the back end then has neither its signed nor its unsigned pattern for sext(a) * zext(b): it multiplies 32 x 64
// bits -- two v_mad_u64_u32, two moves, and for the signed top limbs two v_mul_lo_u32 and a v_add3_u32 more (seen in
// msm_chunk_kernel's ISA: 1211 multiply-adds + 112 v_mul_lo_u32 per point addition against 1062 limb products).  An
// empty asm statement per operand limb hides the range from the optimiser and costs no instruction.
H2_HD int32_t fe29_opaque(int32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(x));
#endif
  return x;
}
template <class FP>
H2_HD void fe29_opaque_limbs(int32_t* o, const Fe29<FP>& a) {
#pragma unroll
  for (int i = 0; i < 9; i++) o[i] = fe29_opaque(a.v[i]);
}

// a b / R' (see the invariants at the top)
template <class FP>
H2_HD Fe29<FP> fe29_mul(const Fe29<FP>& a, const Fe29<FP>& b) {
  int64_t acc = 0;
  int32_t m[9], av[9], bv[9];
  fe29_opaque_limbs(av, a);
  fe29_opaque_limbs(bv, b);
  Fe29<FP> r;
  detail29::columns<FP, 0>(acc, av, bv, m, r.v);
  r.v[8] = (int32_t)acc;
  return r;
}
// the same product with the limbs' ranges left visible to the optimiser (the ceiling microbenchmark times both forms
// and reports the faster: on two operands it knows to be non-negative the compiler picks v_mad_u64_u32 throughout)
template <class FP>
H2_HD Fe29<FP> fe29_mul_plain(const Fe29<FP>& a, const Fe29<FP>& b) {
  int64_t acc = 0;
  int32_t m[9];
  Fe29<FP> r;
  detail29::columns<FP, 0>(acc, a.v, b.v, m, r.v);
  r.v[8] = (int32_t)acc;
  return r;
}
// a^2 / R': the cross terms a_i a_j (i < j) once, against the doubled limbs 2 a_j -- 45 limb products instead of 81.
// Needs |a_i| < 2^29 (what fe29_mul(a, a) needs): a column is at most four cross terms below 2^59, one square below
// 2^58 and nine m p terms below 2^58, under 2^63.
template <class FP>
H2_HD Fe29<FP> fe29_sqr(const Fe29<FP>& a) {
  int64_t acc = 0;
  int32_t m[9], av[9], a2[9];
  fe29_opaque_limbs(av, a);
#pragma unroll
  for (int i = 0; i < 9; i++) a2[i] = fe29_opaque(av[i] * 2);
  Fe29<FP> r;
  detail29::columns_sq<FP, 0>(acc, av, a2, m, r.v);
  r.v[8] = (int32_t)acc;
  return r;
}
// (a b - c d) / R' with ONE reduction: both products are accumulated column by column before m is chosen (saves the 9 k
// multiply-adds and the carry chain of a second reduction).  Needs every limb below 2^29 in magnitude: a column is at
// most 18 limb products and nine m p terms below 2^58 each, under 2^63; |a b - c d| <= 64 p^2 as for a single product.
template <class FP>
H2_HD Fe29<FP> fe29_mul_sub(const Fe29<FP>& a, const Fe29<FP>& b, const Fe29<FP>& c, const Fe29<FP>& d) {
  int64_t acc = 0;
  int32_t m[9], av[9], bv[9], nc[9], dv[9];
  fe29_opaque_limbs(av, a);
  fe29_opaque_limbs(bv, b);
  fe29_opaque_limbs(dv, d);
#pragma unroll
  for (int i = 0; i < 9; i++) nc[i] = fe29_opaque(-c.v[i]);
  Fe29<FP> r;
  detail29::columns2<FP, 0>(acc, av, bv, nc, dv, m, r.v);
  r.v[8] = (int32_t)acc;
  return r;
}

// API form (x 2^256 mod p, canonical) -> working form (x 2^261 mod p, canonical, normalised limbs)
template <class FP>
H2_HD Fe29<FP> fe29_from_api(const Fe<FP>& a) {
  Fe<FP> t = a;
#pragma unroll
  for (int i = 0; i < 5; i++) t = fe_dbl(t);
  return fe29_unpack(t);
}
// working form (any value of magnitude < 64 p, limbs of magnitude < 2^30) -> API form, canonical
template <class FP>
H2_HD Fe<FP> fe29_to_api(const Fe29<FP>& a) {
  Fe<FP> one;
#pragma unroll
  for (int i = 0; i < 8; i++) one.v[i] = FP::ONE(i);                 // 2^256 mod p as a plain integer
  Fe29<FP> t = fe29_mul(a, fe29_unpack(one));                        // in (-p/2, 3p/2), limbs normalised
  Fe29<FP> pl;
#pragma unroll
  for (int i = 0; i < 9; i++) pl.v[i] = (int32_t)fe29_p<FP>(i);
  if (t.v[8] < 0) t = fe29_norm(fe29_add(t, pl));
  const Fe29<FP> s = fe29_norm(fe29_sub(t, pl));
  if (s.v[8] >= 0) t = s;
  return fe29_pack(t);
}

// t in (-p/2, 3p/2) with normalised limbs -> canonical, packed
template <class FP>
H2_HD Fe<FP> fe29_canonical_pack(Fe29<FP> t) {
  Fe29<FP> pl;
#pragma unroll
  for (int i = 0; i < 9; i++) pl.v[i] = (int32_t)fe29_p<FP>(i);
  if (t.v[8] < 0) t = fe29_norm(fe29_add(t, pl));
  const Fe29<FP> s = fe29_norm(fe29_sub(t, pl));
  if (s.v[8] >= 0) t = s;
  return fe29_pack(t);
}

// a^(p-2) on the working form, two exponent bits at a time (254 squarings + ~96 products); a normalised, of small
// magnitude (a product's result); the exponent is a constant, so the branches are uniform
template <class FP>
H2_HD Fe29<FP> fe29_inv(const Fe29<FP>& a) {
  const Fe29<FP> a2 = fe29_mul(a, a), a3 = fe29_mul(a2, a);
  uint32_t e[8];
  uint32_t borrow = 2;
#pragma unroll
  for (int i = 0; i < 8; i++) {               // p - 2, with the borrow of the Pasta primes (p[0] = 1)
    const uint32_t w = FP::P(i);
    e[i] = w - borrow;
    borrow = w < borrow ? 1u : 0u;
  }
  Fe29<FP> r = fe29_from_api(Fe<FP>::one());
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

// x == 0 mod p for a loosely reduced x (|x| < 16 p): if x = j p then j = x[0] p^-1 mod 2^29 is tiny -- anything else
// is rejected by that one limb; the rare survivors are reduced completely
template <class FP>
H2_HD bool fe29_is_zero_mod_p(const Fe29<FP>& a) {
  constexpr uint32_t pinv = (0u - (FP::INV & L29_MASK)) & L29_MASK;  // p^-1 mod 2^29
  const uint32_t j = ((uint32_t)a.v[0] * pinv) & L29_MASK;
  if (j > 16u && j < (1u << 29) - 16u) return false;
  return fe29_to_api(a).is_zero();
}

}  // namespace h2
