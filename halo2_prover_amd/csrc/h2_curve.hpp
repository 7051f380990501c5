// h2_curve.hpp -- a = 0 short-Weierstrass group law (BN254 G1, Pallas, Vesta) for gfx950.
//
// Stands in for halo2curves 0.3.2 bn256::{G1Affine,G1} and pasta_curves 0.5.1 {Ep,Eq}
// (/root/reference/circuits/Cargo.lock:854-856,1126-1128; used through
// /root/reference/circuits/src/utils.rs:5,83-91,105-120).  Memory layouts are the reference's:
// affine = (x, y) Montgomery limbs with identity (0, 0); the value returned across the C ABI
// is Jacobian (x, y, z) with identity z = 0.
//
// Bucket accumulators use extended Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ,
// ZZ^3 = ZZZ^2): the mixed addition costs 8M + 2S and needs no inversion.  All the
// exceptional cases of the incomplete formulas (P + P, P - P, identity operands) are
// resolved explicitly -- structured witness columns do produce them (SURVEY.md section 7,
// hard part 2).
#pragma once
#include "h2_field.hpp"

namespace h2 {

template <class CV>
struct Affine {
  using F = Fe<typename CV::Base>;
  F x, y;
  H2_HD bool is_identity() const { return x.is_zero() && y.is_zero(); }
  static H2_HD Affine identity() { return Affine{F::zero(), F::zero()}; }
};

template <class CV>
struct Xyzz {
  using F = Fe<typename CV::Base>;
  F x, y, zz, zzz;
  H2_HD bool is_identity() const { return zz.is_zero(); }
  static H2_HD Xyzz identity() { return Xyzz{F::zero(), F::zero(), F::zero(), F::zero()}; }
};

template <class CV>
H2_HD Affine<CV> affine_load(const void* p) {
  using B = typename CV::Base;
  const char* c = reinterpret_cast<const char*>(p);
  return Affine<CV>{fe_load<B>(c), fe_load<B>(c + 32)};
}
template <class CV>
H2_HD Affine<CV> affine_neg(const Affine<CV>& a) {
  return Affine<CV>{a.x, fe_neg(a.y)};
}
template <class CV>
H2_HD Xyzz<CV> xyzz_load(const void* p) {
  using B = typename CV::Base;
  const char* c = reinterpret_cast<const char*>(p);
  return Xyzz<CV>{fe_load<B>(c), fe_load<B>(c + 32), fe_load<B>(c + 64), fe_load<B>(c + 96)};
}
template <class CV>
H2_HD void xyzz_store(void* p, const Xyzz<CV>& a) {
  using B = typename CV::Base;
  char* c = reinterpret_cast<char*>(p);
  fe_store<B>(c, a.x);
  fe_store<B>(c + 32, a.y);
  fe_store<B>(c + 64, a.zz);
  fe_store<B>(c + 96, a.zzz);
}

template <class CV>
H2_HD Xyzz<CV> xyzz_from_affine(const Affine<CV>& a) {
  using F = Fe<typename CV::Base>;
  if (a.is_identity()) return Xyzz<CV>::identity();
  return Xyzz<CV>{a.x, a.y, F::one(), F::one()};
}

// 2 * (affine point), "mdbl-2008-s-1"
template <class CV>
H2_HD Xyzz<CV> xyzz_double_affine(const Affine<CV>& a) {
  using F = Fe<typename CV::Base>;
  if (a.is_identity() || a.y.is_zero()) return Xyzz<CV>::identity();
  F u = fe_dbl(a.y);
  F v = fe_sqr(u);
  F w = fe_mul(u, v);
  F s = fe_mul(a.x, v);
  F xx = fe_sqr(a.x);
  F m = fe_add(fe_dbl(xx), xx);
  F x3 = fe_sub(fe_sub(fe_sqr(m), s), s);
  F y3 = fe_sub(fe_mul(m, fe_sub(s, x3)), fe_mul(w, a.y));
  return Xyzz<CV>{x3, y3, v, w};
}

// 2 * P, "dbl-2008-s-1" with a = 0
template <class CV>
H2_HD Xyzz<CV> xyzz_double(const Xyzz<CV>& p) {
  using F = Fe<typename CV::Base>;
  if (p.is_identity() || p.y.is_zero()) return Xyzz<CV>::identity();
  F u = fe_dbl(p.y);
  F v = fe_sqr(u);
  F w = fe_mul(u, v);
  F s = fe_mul(p.x, v);
  F xx = fe_sqr(p.x);
  F m = fe_add(fe_dbl(xx), xx);
  F x3 = fe_sub(fe_sub(fe_sqr(m), s), s);
  F y3 = fe_sub(fe_mul(m, fe_sub(s, x3)), fe_mul(w, p.y));
  return Xyzz<CV>{x3, y3, fe_mul(v, p.zz), fe_mul(w, p.zzz)};
}

// acc + (affine q), "madd-2008-s"
template <class CV>
H2_HD Xyzz<CV> xyzz_add_affine(const Xyzz<CV>& acc, const Affine<CV>& q) {
  using F = Fe<typename CV::Base>;
  if (q.is_identity()) return acc;
  if (acc.is_identity()) return xyzz_from_affine(q);
  F u2 = fe_mul(q.x, acc.zz);
  F s2 = fe_mul(q.y, acc.zzz);
  F p = fe_sub(u2, acc.x);
  F r = fe_sub(s2, acc.y);
  if (p.is_zero()) {
    if (r.is_zero()) return xyzz_double_affine(q);
    return Xyzz<CV>::identity();
  }
  F pp = fe_sqr(p);
  F ppp = fe_mul(p, pp);
  F qq = fe_mul(acc.x, pp);
  F x3 = fe_sub(fe_sub(fe_sub(fe_sqr(r), ppp), qq), qq);
  F y3 = fe_sub(fe_mul(r, fe_sub(qq, x3)), fe_mul(acc.y, ppp));
  return Xyzz<CV>{x3, y3, fe_mul(acc.zz, pp), fe_mul(acc.zzz, ppp)};
}

// a + b, "add-2008-s"
template <class CV>
H2_HD Xyzz<CV> xyzz_add(const Xyzz<CV>& a, const Xyzz<CV>& b) {
  using F = Fe<typename CV::Base>;
  if (a.is_identity()) return b;
  if (b.is_identity()) return a;
  F u1 = fe_mul(a.x, b.zz);
  F u2 = fe_mul(b.x, a.zz);
  F s1 = fe_mul(a.y, b.zzz);
  F s2 = fe_mul(b.y, a.zzz);
  F p = fe_sub(u2, u1);
  F r = fe_sub(s2, s1);
  if (p.is_zero()) {
    if (r.is_zero()) return xyzz_double(a);
    return Xyzz<CV>::identity();
  }
  F pp = fe_sqr(p);
  F ppp = fe_mul(p, pp);
  F qq = fe_mul(u1, pp);
  F x3 = fe_sub(fe_sub(fe_sub(fe_sqr(r), ppp), qq), qq);
  F y3 = fe_sub(fe_mul(r, fe_sub(qq, x3)), fe_mul(s1, ppp));
  return Xyzz<CV>{x3, y3, fe_mul(fe_mul(a.zz, b.zz), pp), fe_mul(fe_mul(a.zzz, b.zzz), ppp)};
}

// XYZZ -> Jacobian representative (X*ZZ, Y*ZZZ, ZZ): x = X/ZZ = X*ZZ/ZZ^2, y = Y/ZZZ = Y*ZZZ/ZZ^3
template <class CV>
H2_HD void xyzz_to_jacobian(const Xyzz<CV>& p, Fe<typename CV::Base>& jx, Fe<typename CV::Base>& jy,
                            Fe<typename CV::Base>& jz) {
  using F = Fe<typename CV::Base>;
  if (p.is_identity()) {
    jx = F::zero(); jy = F::zero(); jz = F::zero();
    return;
  }
  jx = fe_mul(p.x, p.zz);
  jy = fe_mul(p.y, p.zzz);
  jz = p.zz;
}

// XYZZ -> affine (one inversion); identity -> (0, 0)
template <class CV>
H2_HD Affine<CV> xyzz_to_affine(const Xyzz<CV>& p) {
  using F = Fe<typename CV::Base>;
  if (p.is_identity()) return Affine<CV>::identity();
  F zi = fe_inv(p.zzz);        // 1/ZZZ = 1/Z^3
  F t = fe_mul(zi, p.zz);      // ZZ/ZZZ = 1/Z
  F zz_inv = fe_sqr(t);        // 1/Z^2 = 1/ZZ
  return Affine<CV>{fe_mul(p.x, zz_inv), fe_mul(p.y, zi)};
}

}  // namespace h2
