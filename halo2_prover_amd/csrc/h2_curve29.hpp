// h2_curve29.hpp -- the XYZZ group law of h2_curve.hpp on the MSM's working field representation
// (h2_field29.hpp: 9 x 29-bit signed limbs, R' = 2^261, lazy reduction).
//
// Same formulas, same explicit handling of the exceptional cases (P + P, P - P, identity operands); what changes is
// the bookkeeping the lazy representation asks for.  Every coordinate that is STORED (accumulators, partial sums) has
// normalised limbs; with that, each product below has one operand with limbs < 2^29 in magnitude and the other
// < 2^30, and every pair of operand values satisfies |a| |b| <= 64 p^2 (bounds in units of p are noted on the right).
//
// Memory: a table point is 64 bytes -- (x, y) as 2 x 256-bit integers in R' form, canonical, identity (0, 0) --
// unpacked on load; an XYZZ partial sum is 36 words (4 coordinates x 9 limbs, 144 bytes).
#pragma once
#include "h2_curve.hpp"
#include "h2_field29.hpp"

namespace h2 {

constexpr int XYZZ29_WORDS = 36;   // 32-bit words per stored XYZZ point

template <class CV>
struct Affine29 {
  using F = Fe29<typename CV::Base>;
  F x, y;
  H2_HD bool is_identity() const { return x.is_zero_exact() && y.is_zero_exact(); }
};

template <class CV>
struct Xyzz29 {
  using F = Fe29<typename CV::Base>;
  F x, y, zz, zzz;                                       // |x| < 5p, |y| < 2p, zz, zzz in (-p/2, 3p/2)
  H2_HD bool is_identity() const { return zz.is_zero_exact(); }
  static H2_HD Xyzz29 identity() { return Xyzz29{F::zero(), F::zero(), F::zero(), F::zero()}; }
};

// table entry -> working form; `negate`: the signed digit asks for -P
template <class CV>
H2_HD Affine29<CV> affine29_load(const void* p, bool negate) {
  using B = typename CV::Base;
  const char* c = reinterpret_cast<const char*>(p);
  Affine29<CV> a{fe29_unpack(fe_load<B>(c)), fe29_unpack(fe_load<B>(c + 32))};
  if (negate) a.y = fe29_neg(a.y);                       // limbs in (-2^29, 0]: still a valid product operand
  return a;
}
// an affine point in the API form (Montgomery R = 2^256) -> table entry bytes
template <class CV>
H2_HD void affine29_store_table(void* p, const Affine<CV>& a) {
  using B = typename CV::Base;
  char* c = reinterpret_cast<char*>(p);
  if (a.is_identity()) {
    fe_store<B>(c, Fe<B>::zero());
    fe_store<B>(c + 32, Fe<B>::zero());
    return;
  }
  fe_store<B>(c, fe29_pack(fe29_from_api(a.x)));
  fe_store<B>(c + 32, fe29_pack(fe29_from_api(a.y)));
}

template <class CV>
H2_HD Xyzz29<CV> xyzz29_load(const uint32_t* p) {
  Xyzz29<CV> r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    r.x.v[i] = (int32_t)p[i];
    r.y.v[i] = (int32_t)p[9 + i];
    r.zz.v[i] = (int32_t)p[18 + i];
    r.zzz.v[i] = (int32_t)p[27 + i];
  }
  return r;
}
template <class CV>
H2_HD void xyzz29_store(uint32_t* p, const Xyzz29<CV>& a) {
#pragma unroll
  for (int i = 0; i < 9; i++) {
    p[i] = (uint32_t)a.x.v[i];
    p[9 + i] = (uint32_t)a.y.v[i];
    p[18 + i] = (uint32_t)a.zz.v[i];
    p[27 + i] = (uint32_t)a.zzz.v[i];
  }
}

// 1 in the working form: 2^261 mod p = 32 * (2^256 mod p)
template <class CV>
H2_HD Fe29<typename CV::Base> fe29_one() {
  using B = typename CV::Base;
  return fe29_from_api(Fe<B>::one());
}

template <class CV>
H2_HD Xyzz29<CV> xyzz29_from_affine(const Affine29<CV>& a) {
  if (a.is_identity()) return Xyzz29<CV>::identity();
  const Fe29<typename CV::Base> one = fe29_one<CV>();
  return Xyzz29<CV>{a.x, fe29_norm(a.y), one, one};      // y may carry the digit's sign: store it normalised
}

// 2 * (affine point), "mdbl-2008-s-1"
template <class CV>
H2_HD Xyzz29<CV> xyzz29_double_affine(const Affine29<CV>& a) {
  using F = Fe29<typename CV::Base>;
  if (a.is_identity() || fe29_is_zero_mod_p(a.y)) return Xyzz29<CV>::identity();
  const F u = fe29_norm(fe29_add(a.y, a.y));             // |u| < 2
  const F v = fe29_sqr(u);
  const F w = fe29_mul(u, v);
  const F s = fe29_mul(a.x, v);
  const F xx = fe29_sqr(a.x);
  const F m = fe29_norm(fe29_add(fe29_add(xx, xx), xx)); // < 4.5
  const F x3 = fe29_norm(fe29_sub(fe29_sub(fe29_sqr(m), s), s));
  const F y3 = fe29_mul_sub(m, fe29_sub(s, x3), w, a.y);                                 // one reduction for both products
  return Xyzz29<CV>{x3, y3, v, w};
}

// 2 * P, "dbl-2008-s-1" with a = 0
template <class CV>
H2_HD Xyzz29<CV> xyzz29_double(const Xyzz29<CV>& p) {
  using F = Fe29<typename CV::Base>;
  if (p.is_identity() || fe29_is_zero_mod_p(p.y)) return Xyzz29<CV>::identity();
  const F u = fe29_norm(fe29_add(p.y, p.y));             // |u| < 4
  const F v = fe29_sqr(u);                               // 16
  const F w = fe29_mul(u, v);                            // 4 * 1.5
  const F s = fe29_mul(p.x, v);                          // 5 * 1.5
  const F xx = fe29_sqr(p.x);                            // 25
  const F m = fe29_norm(fe29_add(fe29_add(xx, xx), xx)); // < 4.5
  const F x3 = fe29_norm(fe29_sub(fe29_sub(fe29_sqr(m), s), s));                        // (-3.5, 2.5)
  const F y3 = fe29_mul_sub(m, fe29_sub(s, x3), w, p.y);                                // 4.5 * 5 + 1.5 * 2, one reduction
  return Xyzz29<CV>{x3, y3, fe29_mul(v, p.zz), fe29_mul(w, p.zzz)};
}

// acc + (affine q), "madd-2008-s"
template <class CV>
H2_HD Xyzz29<CV> xyzz29_add_affine(const Xyzz29<CV>& acc, const Affine29<CV>& q) {
  using F = Fe29<typename CV::Base>;
  if (q.is_identity()) return acc;
  if (acc.is_identity()) return xyzz29_from_affine(q);
  const F u2 = fe29_mul(q.x, acc.zz);
  const F s2 = fe29_mul(q.y, acc.zzz);
  const F p = fe29_sub(u2, acc.x);                       // (-3.5, 6.5); limbs of magnitude < 2^29
  const F r = fe29_sub(s2, acc.y);                       // (-2.5, 3.5)
  if (fe29_is_zero_mod_p(p)) {
    if (fe29_is_zero_mod_p(r)) return xyzz29_double_affine(q);
    return Xyzz29<CV>::identity();
  }
  const F pp = fe29_sqr(p);                              // 42
  const F ppp = fe29_mul(p, pp);                         // 6.5 * 1.5
  const F qq = fe29_mul(acc.x, pp);                      // 5 * 1.5
  const F x3 = fe29_norm(fe29_sub(fe29_sub(fe29_sub(fe29_sqr(r), ppp), qq), qq));       // (-5, 3)
  const F y3 = fe29_mul_sub(r, fe29_sub(qq, x3), acc.y, ppp);                            // 3.5 * 6.5 + 2 * 1.5, one reduction
  return Xyzz29<CV>{x3, y3, fe29_mul(acc.zz, pp), fe29_mul(acc.zzz, ppp)};
}

// a + b, "add-2008-s"
template <class CV>
H2_HD Xyzz29<CV> xyzz29_add(const Xyzz29<CV>& a, const Xyzz29<CV>& b) {
  using F = Fe29<typename CV::Base>;
  if (a.is_identity()) return b;
  if (b.is_identity()) return a;
  const F u1 = fe29_mul(a.x, b.zz);                      // 5 * 1.5
  const F u2 = fe29_mul(b.x, a.zz);
  const F s1 = fe29_mul(a.y, b.zzz);                     // 2 * 1.5
  const F s2 = fe29_mul(b.y, a.zzz);
  const F p = fe29_sub(u2, u1);                          // (-2, 2)
  const F r = fe29_sub(s2, s1);
  if (fe29_is_zero_mod_p(p)) {
    if (fe29_is_zero_mod_p(r)) return xyzz29_double(a);
    return Xyzz29<CV>::identity();
  }
  const F pp = fe29_sqr(p);
  const F ppp = fe29_mul(p, pp);
  const F qq = fe29_mul(u1, pp);
  const F x3 = fe29_norm(fe29_sub(fe29_sub(fe29_sub(fe29_sqr(r), ppp), qq), qq));
  const F y3 = fe29_mul_sub(r, fe29_sub(qq, x3), s1, ppp);
  return Xyzz29<CV>{x3, y3, fe29_mul(fe29_mul(a.zz, b.zz), pp), fe29_mul(fe29_mul(a.zzz, b.zzz), ppp)};
}

// working form -> the 32-bit-limb XYZZ point in the API's Montgomery form (for the final conversions)
template <class CV>
H2_HD Xyzz<CV> xyzz29_to_api(const Xyzz29<CV>& p) {
  if (p.is_identity()) return Xyzz<CV>::identity();
  return Xyzz<CV>{fe29_to_api(p.x), fe29_to_api(p.y), fe29_to_api(p.zz), fe29_to_api(p.zzz)};
}

}  // namespace h2
