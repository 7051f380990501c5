// h2_curve_quad.hpp -- XYZZ point arithmetic with FOUR LANES PER POINT (gfx950 DPP quad_perm), on the MSM's working
// field representation (h2_field29.hpp / h2_curve29.hpp).
//
// The tail of the MSM (summing a bucket's pieces, the bucket weights, the final tree) works on few points with long
// dependency chains: about one wave per SIMD, every addition waiting for the previous one.  A full XYZZ addition is
// 14 field multiplications executed one after the other by a lane, but its dependency DEPTH is 4:
//     level 1   u1 = x1 zz2      u2 = x2 zz1      s1 = y1 zzz2     s2 = y2 zzz1
//     level 2   pp = p^2         rr = r^2         zz12 = zz1 zz2   zzz12 = zzz1 zzz2        (p = u2-u1, r = s2-s1)
//     level 3   ppp = p pp       q = u1 pp        zz3 = zz12 pp
//     level 4   r (q - x3)       s1 ppp           zzz3 = zzz12 ppp                          (x3 = rr - ppp - 2q)
// Here the four lanes of a quad hold the same two points; at each level every lane selects the operands of ITS
// product (mask arithmetic on the lane's role), all four execute the same multiplication, and the results are passed
// around with v_mov_b32 quad_perm broadcasts (one VALU instruction per word, no LDS).  A lone wave is issue-bound
// (~4.5 cycles per VALU instruction whatever the dependencies), so what counts is instructions per operation: about a
// third of the one-lane form.  History of the measurements (Poseidon k = 16 bench, per MSM phase, 32-bit limbs):
// weights 192 -> 131 us and tree 190 -> 131 us with the first version, 86 / 96 us once the operand select was
// written as mask arithmetic (as `q == 0 ? a0 : ...` it compiled to exec-mask branches and the points lived in
// scratch), fix-up 188 -> 117 us; on the 29-bit working form 64 / 65 / 74 us.
//
// Contract: a and b are REPLICATED across the quad (all four lanes hold identical values) and so is the result.
// All four lanes of a quad must be active and take the same branches (they do: the data is replicated).
#pragma once
#include "h2_curve.hpp"
#include "h2_curve29.hpp"

namespace h2 {

template <int R>
__device__ __forceinline__ uint32_t quad_bcast_u32(uint32_t v) {
  // dpp_ctrl quad_perm:[R,R,R,R]; all rows / banks enabled; bound_ctrl irrelevant (source lane is in the quad)
  uint32_t r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, R * 0x55, 0xf, 0xf, true);
  asm volatile("" : "+v"(r));   // keep it a plain v_mov_b32_dpp (see the note below)
  return r;
}
// the operand of lane role q among four candidates.  Written as mask arithmetic ((m & a) | (~m & b) is one
// v_bfi_b32): as `q == 0 ? a0 : ...` the compiler turned every word into exec-mask branches.
__device__ __forceinline__ uint32_t quad_bfi(uint32_t m, uint32_t a, uint32_t b) { return (m & a) | (~m & b); }

// The broadcast words feed ordinary VALU code, and with ROCm 7.2 the compiler's folding of v_mov_b32_dpp into the
// consuming instruction produced wrong results for every input (tests/test_gpu_parity.py::
// test_device_group_law_on_the_working_form failed on ops 0 and 1 while the one-lane forms passed): quad_bcast_u32
// pins its result with an empty asm statement.
template <int R, class FP>
__device__ __forceinline__ Fe29<FP> quad_bcast(const Fe29<FP>& a) {
  Fe29<FP> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = (int32_t)quad_bcast_u32<R>((uint32_t)a.v[i]);
  return r;
}
template <class FP>
__device__ __forceinline__ Fe29<FP> quad_select(uint32_t q, const Fe29<FP>& a0, const Fe29<FP>& a1, const Fe29<FP>& a2,
                                                const Fe29<FP>& a3) {
  Fe29<FP> r;
  const uint32_t m1 = 0u - (q & 1u), m2 = 0u - ((q >> 1) & 1u);
#pragma unroll
  for (int i = 0; i < 9; i++)
    r.v[i] = (int32_t)quad_bfi(m2, quad_bfi(m1, (uint32_t)a3.v[i], (uint32_t)a2.v[i]),
                               quad_bfi(m1, (uint32_t)a1.v[i], (uint32_t)a0.v[i]));
  return r;
}

template <class CV>
__device__ __forceinline__ Xyzz29<CV> xyzz29_double_quad(const Xyzz29<CV>& p) {
  using F = Fe29<typename CV::Base>;
  if (p.is_identity() || fe29_is_zero_mod_p(p.y)) return Xyzz29<CV>::identity();
  const uint32_t q = threadIdx.x & 3;
  const F u = fe29_norm(fe29_add(p.y, p.y));
  // level 1: v = u^2 (lane 0), xx = x^2 (lanes 1..3)
  const F o1 = quad_select(q, u, p.x, p.x, p.x);
  const F m1 = fe29_mul(o1, o1);
  const F v = quad_bcast<0>(m1), xx = quad_bcast<1>(m1);
  const F m = fe29_norm(fe29_add(fe29_add(xx, xx), xx));
  // level 2: w = u v, s = x v, mm = m^2, zz3 = v zz
  const F m2 = fe29_mul(quad_select(q, u, p.x, m, v), quad_select(q, v, v, m, p.zz));
  const F w = quad_bcast<0>(m2), s = quad_bcast<1>(m2), mm = quad_bcast<2>(m2), zz3 = quad_bcast<3>(m2);
  const F x3 = fe29_norm(fe29_sub(fe29_sub(mm, s), s));
  // level 3: m (s - x3), w y, zzz3 = w zzz
  const F m3 = fe29_mul(quad_select(q, m, w, w, w), quad_select(q, fe29_sub(s, x3), p.y, p.zzz, p.zzz));
  const F y3 = fe29_norm(fe29_sub(quad_bcast<0>(m3), quad_bcast<1>(m3)));
  return Xyzz29<CV>{x3, y3, zz3, quad_bcast<2>(m3)};
}

template <class CV>
__device__ __forceinline__ Xyzz29<CV> xyzz29_add_quad(const Xyzz29<CV>& a, const Xyzz29<CV>& b) {
  using F = Fe29<typename CV::Base>;
  if (a.is_identity()) return b;
  if (b.is_identity()) return a;
  const uint32_t q = threadIdx.x & 3;
  // level 1: u1, u2, s1, s2
  const F m1 = fe29_mul(quad_select(q, a.x, b.x, a.y, b.y), quad_select(q, b.zz, a.zz, b.zzz, a.zzz));
  const F u1 = quad_bcast<0>(m1), u2 = quad_bcast<1>(m1), s1 = quad_bcast<2>(m1), s2 = quad_bcast<3>(m1);
  const F p = fe29_sub(u2, u1), r = fe29_sub(s2, s1);
  if (fe29_is_zero_mod_p(p)) {
    if (fe29_is_zero_mod_p(r)) return xyzz29_double_quad(a);
    return Xyzz29<CV>::identity();
  }
  // level 2: pp, rr, zz12, zzz12
  const F m2 = fe29_mul(quad_select(q, p, r, a.zz, a.zzz), quad_select(q, p, r, b.zz, b.zzz));
  const F pp = quad_bcast<0>(m2), rr = quad_bcast<1>(m2);
  // level 3: ppp = p pp, qq = u1 pp, zz3 = zz12 pp (lane 2's own m2); lane 3 repeats lane 2's shape and keeps m2
  const F m3 = fe29_mul(quad_select(q, p, u1, m2, m2), pp);
  const F ppp = quad_bcast<0>(m3), qq = quad_bcast<1>(m3), zz3 = quad_bcast<2>(m3);
  const F x3 = fe29_norm(fe29_sub(fe29_sub(fe29_sub(rr, ppp), qq), qq));
  // level 4: r (qq - x3), s1 ppp, -, zzz3 = zzz12 ppp (lane 3's m2 is zzz12)
  const F m4 = fe29_mul(quad_select(q, r, s1, s1, m2), quad_select(q, fe29_sub(qq, x3), ppp, ppp, ppp));
  const F y3 = fe29_norm(fe29_sub(quad_bcast<0>(m4), quad_bcast<1>(m4)));
  return Xyzz29<CV>{x3, y3, zz3, quad_bcast<3>(m4)};
}

}  // namespace h2
