// h2_pairing.hpp -- BN254 optimal-ate pairing check on the host: the last step of KZG verification,
//   e(left, [s]G2) * e(right, -G2) == 1
// (halo2_proofs @6b43b6b src/poly/kzg/msm.rs `DualMSM::check`, reached from
// /root/reference/circuits/src/utils.rs:125-158 through verify_proof).  A transcription of
// halo2_prover_amd/pairing.py (same tower, same line functions; the final exponentiation uses the usual addition chain); not a
// hot path: two Miller loops and one final exponentiation per proof.  Nothing in /root/reference pins the pairing on
// its own ("parity unpinned"); it is checked by bilinearity and by accepting the recorded proofs.
#pragma once
#include "h2_host.hpp"

namespace h2 {
namespace bn {

// ---- Fq2 = Fq[u] / (u^2 + 1) ------------------------------------------------------------------------------------
struct F2 {
  Fq a, b;
  bool operator==(const F2& o) const { return a == o.a && b == o.b; }
  bool is_zero() const { return a.is_zero() && b.is_zero(); }
};
inline F2 operator+(const F2& x, const F2& y) { return {x.a + y.a, x.b + y.b}; }
inline F2 operator-(const F2& x, const F2& y) { return {x.a - y.a, x.b - y.b}; }
inline F2 operator-(const F2& x) { return {-x.a, -x.b}; }
inline F2 operator*(const F2& x, const F2& y) { return {x.a * y.a - x.b * y.b, x.a * y.b + x.b * y.a}; }
inline F2 sqr(const F2& x) { return {(x.a + x.b) * (x.a - x.b), (x.a * x.b) + (x.a * x.b)}; }
inline F2 conj(const F2& x) { return {x.a, -x.b}; }
inline F2 scale(const F2& x, const Fq& k) { return {x.a * k, x.b * k}; }
inline F2 mul_xi(const F2& x) {  // times xi = 9 + u
  const Fq nine = Fq::from_u64(9);
  return {nine * x.a - x.b, x.a + nine * x.b};
}
inline F2 inv(const F2& x) {
  const Fq t = (x.a * x.a + x.b * x.b).inv();
  return {x.a * t, -(x.b * t)};
}
inline F2 f2_zero() { return {Fq::zero(), Fq::zero()}; }
inline F2 f2_one() { return {Fq::one(), Fq::zero()}; }
// x^e for a multi-limb exponent given as little-endian 64-bit words
inline F2 pow(const F2& x, const std::vector<uint64_t>& e) {
  F2 r = f2_one();
  for (size_t i = e.size() * 64; i-- > 0;) {
    r = sqr(r);
    if ((e[i >> 6] >> (i & 63)) & 1) r = r * x;
  }
  return r;
}

// ---- Fq6 = Fq2[v] / (v^3 - xi) ------------------------------------------------------------------------------------
struct F6 {
  F2 c0, c1, c2;
  bool operator==(const F6& o) const { return c0 == o.c0 && c1 == o.c1 && c2 == o.c2; }
};
inline F6 operator+(const F6& x, const F6& y) { return {x.c0 + y.c0, x.c1 + y.c1, x.c2 + y.c2}; }
inline F6 operator-(const F6& x, const F6& y) { return {x.c0 - y.c0, x.c1 - y.c1, x.c2 - y.c2}; }
inline F6 operator-(const F6& x) { return {-x.c0, -x.c1, -x.c2}; }
inline F6 operator*(const F6& a, const F6& b) {
  const F2 t0 = a.c0 * b.c0, t1 = a.c1 * b.c1, t2 = a.c2 * b.c2;
  return {t0 + mul_xi((a.c1 + a.c2) * (b.c1 + b.c2) - (t1 + t2)),
          ((a.c0 + a.c1) * (b.c0 + b.c1) - (t0 + t1)) + mul_xi(t2),
          ((a.c0 + a.c2) * (b.c0 + b.c2) - (t0 + t2)) + t1};
}
inline F6 mul_v(const F6& a) { return {mul_xi(a.c2), a.c0, a.c1}; }
inline F6 inv(const F6& a) {
  const F2 c0 = sqr(a.c0) - mul_xi(a.c1 * a.c2);
  const F2 c1 = mul_xi(sqr(a.c2)) - a.c0 * a.c1;
  const F2 c2 = sqr(a.c1) - a.c0 * a.c2;
  const F2 t = inv(a.c0 * c0 + mul_xi(a.c2 * c1 + a.c1 * c2));
  return {c0 * t, c1 * t, c2 * t};
}
inline F6 f6_zero() { return {f2_zero(), f2_zero(), f2_zero()}; }
inline F6 f6_one() { return {f2_one(), f2_zero(), f2_zero()}; }

// ---- Fq12 = Fq6[w] / (w^2 - v) ------------------------------------------------------------------------------------
struct F12 {
  F6 c0, c1;
  bool operator==(const F12& o) const { return c0 == o.c0 && c1 == o.c1; }
};
inline F12 operator*(const F12& a, const F12& b) {
  const F6 t0 = a.c0 * b.c0, t1 = a.c1 * b.c1;
  return {t0 + mul_v(t1), (a.c0 + a.c1) * (b.c0 + b.c1) - (t0 + t1)};
}
inline F12 conj(const F12& a) { return {a.c0, -a.c1}; }   // the p^6 Frobenius
inline F12 inv(const F12& a) {
  const F6 t = inv(a.c0 * a.c0 - mul_v(a.c1 * a.c1));
  return {a.c0 * t, -(a.c1 * t)};
}
inline F12 f12_one() { return {f6_one(), f6_zero()}; }
inline F12 pow(const F12& x, const std::vector<uint64_t>& e) {
  F12 r = f12_one();
  bool started = false;
  for (size_t i = e.size() * 64; i-- > 0;) {
    if (started) r = r * r;
    if ((e[i >> 6] >> (i & 63)) & 1) {
      r = started ? r * x : x;
      started = true;
    }
  }
  return r;
}

// ---- little multi-precision helpers for the exponents (p - 1) / 3, (p^2 - 1) / 2, (p^6 + 1) / r ... -----------------
using Big = std::vector<uint64_t>;   // little-endian words
inline Big big_mul(const Big& a, const Big& b) {
  Big r(a.size() + b.size(), 0);
  for (size_t i = 0; i < a.size(); i++) {
    unsigned __int128 carry = 0;
    for (size_t j = 0; j < b.size(); j++) {
      carry += (unsigned __int128)a[i] * b[j] + r[i + j];
      r[i + j] = (uint64_t)carry;
      carry >>= 64;
    }
    r[i + b.size()] += (uint64_t)carry;
  }
  return r;
}
inline Big big_add_small(Big a, uint64_t k) {
  for (size_t i = 0; i < a.size() && k; i++) {
    const uint64_t s = a[i] + k;
    k = s < a[i] ? 1 : 0;
    a[i] = s;
  }
  if (k) a.push_back(k);
  return a;
}
inline Big big_sub_small(Big a, uint64_t k) {
  for (size_t i = 0; i < a.size() && k; i++) {
    const uint64_t s = a[i] - k;
    k = a[i] < k ? 1 : 0;
    a[i] = s;
  }
  return a;
}
inline int big_cmp(const Big& a, const Big& b) {
  const size_t n = a.size() > b.size() ? a.size() : b.size();
  for (size_t i = n; i-- > 0;) {
    const uint64_t x = i < a.size() ? a[i] : 0, y = i < b.size() ? b[i] : 0;
    if (x != y) return x < y ? -1 : 1;
  }
  return 0;
}
// floor(a / d) by binary long division (a few thousand bit steps, once per process)
inline Big big_div(const Big& a, const Big& d) {
  Big q(a.size(), 0), rem(d.size() + 1, 0);
  for (size_t i = a.size() * 64; i-- > 0;) {
    // rem = rem * 2 + bit
    uint64_t carry = (a[i >> 6] >> (i & 63)) & 1;
    for (size_t j = 0; j < rem.size(); j++) {
      const uint64_t nc = rem[j] >> 63;
      rem[j] = (rem[j] << 1) | carry;
      carry = nc;
    }
    if (big_cmp(rem, d) >= 0) {
      unsigned __int128 borrow = 0;
      for (size_t j = 0; j < rem.size(); j++) {
        const uint64_t dj = j < d.size() ? d[j] : 0;
        const unsigned __int128 s = (unsigned __int128)rem[j] - dj - borrow;   // wraps when it borrows
        rem[j] = (uint64_t)s;
        borrow = (s >> 64) & 1;
      }
      q[i >> 6] |= 1ull << (i & 63);
    }
  }
  return q;
}
inline Big big_of_modulus_q() {
  Big p(4);
  for (int i = 0; i < 4; i++) p[i] = (uint64_t)BN254_FQ::P(2 * i) | ((uint64_t)BN254_FQ::P(2 * i + 1) << 32);
  return p;
}
inline Big big_of_modulus_r() {
  Big p(4);
  for (int i = 0; i < 4; i++) p[i] = (uint64_t)BN254_FR::P(2 * i) | ((uint64_t)BN254_FR::P(2 * i + 1) << 32);
  return p;
}

struct G2 {
  F2 x, y;
  bool inf = true;
};

struct Consts {
  F2 g12, g13, g22, g23, twist_b;
  F2 frob1[6], frob2[6];   // xi^(i (p - 1) / 6), xi^(i (p^2 - 1) / 6): w^i -> its image under the p- and p^2-power maps
};
inline const Consts& consts() {
  static const Consts c = [] {
    Consts k;
    const Big p = big_of_modulus_q(), p2 = big_mul(p, p);
    const F2 xi{Fq::from_u64(9), Fq::one()};
    k.g12 = pow(xi, big_div(big_sub_small(p, 1), Big{3}));
    k.g13 = pow(xi, big_div(big_sub_small(p, 1), Big{2}));
    k.g22 = pow(xi, big_div(big_sub_small(p2, 1), Big{3}));
    k.g23 = pow(xi, big_div(big_sub_small(p2, 1), Big{2}));
    k.twist_b = F2{Fq::from_u64(3), Fq::zero()} * inv(xi);
    const F2 f1 = pow(xi, big_div(big_sub_small(p, 1), Big{6})), f2 = pow(xi, big_div(big_sub_small(p2, 1), Big{6}));
    k.frob1[0] = k.frob2[0] = f2_one();
    for (int i = 1; i < 6; i++) {
      k.frob1[i] = k.frob1[i - 1] * f1;
      k.frob2[i] = k.frob2[i - 1] * f2;
    }
    return k;
  }();
  return c;
}

inline bool g2_on_curve(const G2& q) {
  if (q.inf) return true;
  return sqr(q.y) == sqr(q.x) * q.x + consts().twist_b;
}

// the line through the untwisted points t, q (t == q: the tangent) evaluated at P = (xp, yp), and t + q on the twist:
// psi(x', y') = (x' w^2, y' w^3), l(P) = yP - lambda xP w + (lambda x1 - y1) w^3
inline F12 line(G2& t, const G2& q, bool tangent, const Fq& xp, const Fq& yp) {
  F2 lam;
  if (tangent) lam = scale(sqr(t.x), Fq::from_u64(3)) * inv(scale(t.y, Fq::from_u64(2)));
  else lam = (q.y - t.y) * inv(q.x - t.x);
  const F2 x3 = sqr(lam) - t.x - q.x;
  const F2 y3 = lam * (t.x - x3) - t.y;
  F12 l;
  l.c0 = {F2{yp, Fq::zero()}, f2_zero(), f2_zero()};
  l.c1 = {-scale(lam, xp), lam * t.x - t.y, f2_zero()};
  t.x = x3;
  t.y = y3;
  return l;
}

inline F12 miller_loop(const G1& p, const G2& q) {
  if (p.inf || q.inf) return f12_one();
  const Consts& k = consts();
  const unsigned __int128 loop = (unsigned __int128)6 * 4965661367192848881ull + 2;   // 6x + 2
  int top = 127;
  while (!((loop >> top) & 1)) top--;
  F12 f = f12_one();
  G2 t = q;
  for (int i = top - 1; i >= 0; i--) {
    const F12 l = line(t, t, true, p.x, p.y);
    f = (f * f) * l;
    if ((loop >> i) & 1) f = f * line(t, q, false, p.x, p.y);
  }
  G2 q1{conj(q.x) * k.g12, conj(q.y) * k.g13, false};
  G2 q2{q.x * k.g22, -(q.y * k.g23), false};        // -pi^2(Q)
  f = f * line(t, q1, false, p.x, p.y);
  f = f * line(t, q2, false, p.x, p.y);
  return f;
}

// x -> x^(p^power), power = 1, 2, 3.  In the basis 1, v, v^2, w, v w, v^2 w the element is sum_i c_i w^i with
// i = 0, 2, 4, 1, 3, 5; c w^i maps to c^(p^k) (w^(p^k))^i and w^(p^k) = w xi^((p^k - 1) / 6).
inline F12 frobenius(const F12& a, int power) {
  const Consts& k = consts();
  if (power == 3) return frobenius(frobenius(a, 2), 1);
  if (power == 2)
    return {{a.c0.c0, a.c0.c1 * k.frob2[2], a.c0.c2 * k.frob2[4]},
            {a.c1.c0 * k.frob2[1], a.c1.c1 * k.frob2[3], a.c1.c2 * k.frob2[5]}};
  return {{conj(a.c0.c0), conj(a.c0.c1) * k.frob1[2], conj(a.c0.c2) * k.frob1[4]},
          {conj(a.c1.c0) * k.frob1[1], conj(a.c1.c1) * k.frob1[3], conj(a.c1.c2) * k.frob1[5]}};
}

// f^(-x) for an f of the cyclotomic subgroup (where the conjugate is the inverse), x the curve parameter
inline F12 exp_by_neg_x(const F12& f) {
  return conj(pow(f, Big{4965661367192848881ull}));
}

// f^((p^12 - 1) / r * 2x(6x^2 + 3x + 1)): the easy part (p^6 - 1)(p^2 + 1) with one inversion and a Frobenius map, the
// hard part by the addition chain of Fuentes-Castaneda, Knapp, Rodriguez-Henriquez ("Faster hashing to G2", as used by
// the common BN implementations): three exponentiations by the 63-bit x instead of one by a 1270-bit number.  The
// extra factor 2x(6x^2 + 3x + 1) < r is coprime to r, so the result is 1 exactly when the pairing product is.
// (The Python mirror, pairing.py, keeps the plain exponentiation by (p^6 + 1) / r: the two agree on every check.)
inline F12 final_exponentiation(const F12& f) {
  const F12 t = conj(f) * inv(f);                  // f^(p^6 - 1)
  const F12 r = frobenius(t, 2) * t;               // ^(p^2 + 1)
  const F12 y0 = exp_by_neg_x(r);
  const F12 y1 = y0 * y0;
  const F12 y2 = y1 * y1;
  const F12 y3 = y2 * y1;
  const F12 y4 = exp_by_neg_x(y3);
  const F12 y5 = y4 * y4;
  const F12 y6 = conj(exp_by_neg_x(y5));
  const F12 y7 = y6 * y4;
  const F12 y8 = y7 * conj(y3);
  const F12 y9 = y8 * y1;
  const F12 y10 = y8 * y4;
  const F12 y11 = y10 * r;
  const F12 y13 = frobenius(y9, 1) * y11;
  const F12 y14 = frobenius(y8, 2) * y13;
  const F12 y15 = frobenius(conj(r) * y9, 3);
  return y15 * y14;
}

// prod_i e(P_i, Q_i) == 1
inline bool pairing_check(const std::vector<std::pair<G1, G2>>& pairs) {
  F12 f = f12_one();
  for (const auto& pq : pairs) f = f * miller_loop(pq.first, pq.second);
  return final_exponentiation(f) == f12_one();
}

}  // namespace bn
}  // namespace h2
