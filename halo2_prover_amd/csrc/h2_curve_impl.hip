// h2_curve_impl.hip -- kernels and launchers of ONE curve (selected by -DH2_CURVE_ID).
#include "h2_curve_ops.hpp"
#include "h2_msm.hpp"
#include "h2_ntt.hpp"
#include "h2_ntt29.hpp"
#include "h2_tune.hpp"
#include "h2_poly.hpp"
#include "h2_group_fft.hpp"

#include <cstring>

#ifndef H2_CURVE_ID
#error "compile with -DH2_CURVE_ID=0 (bn254), 1 (pallas) or 2 (vesta)"
#endif

namespace h2 {
namespace {

#if H2_CURVE_ID == 0
using CV = BN254_CURVE;
#elif H2_CURVE_ID == 1
using CV = PALLAS_CURVE;
#else
using CV = VESTA_CURVE;
#endif
using FS = typename CV::Scalar;
using FB = typename CV::Base;

hipError_t kernel_setup() {
  hipError_t e = msm_kernel_setup<CV>();
  if (e != hipSuccess) return e;
  return ntt29_kernel_setup<FS>();
}
hipError_t table_build(const void* d_bases, void* d_table, void* d_scratch, uint32_t n, const MsmGeom& g, uint32_t* d_bad, hipStream_t s) {
  hipLaunchKernelGGL(msm_table_kernel<CV>, dim3((n + 255) / 256), dim3(256), 0, s, (const U128*)d_bases,
                     (U128*)d_table, (uint32_t*)d_scratch, n, g, d_bad);
  return hipGetLastError();
}
hipError_t msm_launch_(const void* d_table, const void* const* per_column, uint32_t n_bases, const void* d_scalars, size_t n, size_t col_stride, size_t m,
                       const MsmGeom& g, char* ws_base, const MsmWorkspace& ws, hipStream_t s, hipEvent_t ev_start,
                       hipEvent_t ev_stop, hipEvent_t ev_tail, void* d_out_jac, bool zeroed) {
  return msm_launch<CV>((const U128*)d_table, (const U128* const*)per_column, n_bases, (const U128*)d_scalars, n, col_stride, m, g, ws_base, ws, s, ev_start,
                        ev_stop, ev_tail, (U128*)d_out_jac, zeroed);
}
hipError_t srs_powers(void* d_out_affine, const uint64_t s_mont[4], uint32_t n, hipStream_t s) {
  Fe<FS> sv;
  memcpy(sv.v, s_mont, 32);
  hipLaunchKernelGGL(srs_powers_kernel<CV>, dim3((n + 255) / 256), dim3(256), 0, s, (U128*)d_out_affine, sv, n);
  return hipGetLastError();
}
hipError_t fixed_base_mul(void* d_out_affine, const void* d_scalars, uint32_t n, hipStream_t s) {
  hipLaunchKernelGGL(fixed_base_mul_kernel<CV>, dim3((n + 255) / 256), dim3(256), 0, s, (U128*)d_out_affine,
                     (const U128*)d_scalars, n);
  return hipGetLastError();
}
hipError_t to_affine(const void* d_xyzz, void* d_out, uint32_t m, hipStream_t s) {
  hipLaunchKernelGGL(msm_to_affine_kernel<CV>, dim3((m + 63) / 64), dim3(64), 0, s, (const uint32_t*)d_xyzz,
                     (U128*)d_out, m);
  return hipGetLastError();
}
hipError_t msm_small(const void* const* d_points, const void* const* d_scalars, const uint32_t* m, uint32_t count, void* d_work,
                     void* d_out_jac, hipStream_t s) {
  if (count == 0 || count > MSM_SMALL_MAX) return hipErrorInvalidValue;
  MsmSmallBatch J{};
  uint32_t mmax = 0;
  for (uint32_t j = 0; j < count; j++) {
    J.points[j] = (const U128*)d_points[j];
    J.scalars[j] = (const U128*)d_scalars[j];
    J.m[j] = m[j];
    mmax = std::max(mmax, m[j]);
  }
  const uint32_t blocks = (mmax + 15) / 16;
  if (blocks == 0) return hipErrorInvalidValue;
  uint32_t* part = (uint32_t*)d_work;
  uint32_t* counter = part + (size_t)count * blocks * XYZZ29_WORDS;
  hipError_t e = hipMemsetAsync(counter, 0, 4 * count, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(msm_small_kernel<CV>, dim3(blocks, count), dim3(64), 0, s, J, part, counter, (U128*)d_out_jac);
  return hipGetLastError();
}
hipError_t points_sum(const void* d_in_jac, void* d_out_jac, uint32_t groups, uint32_t count, hipStream_t s) {
  hipLaunchKernelGGL(points_sum_kernel<CV>, dim3(count), dim3(64), 0, s, (const U128*)d_in_jac,
                     (U128*)d_out_jac, groups, count);
  return hipGetLastError();
}
size_t ntt_table_bytes(uint32_t log_n) { return ntt29_tables(log_n).total; }
bool ntt_scale_in_table(uint32_t log_n) { return ntt29_scale_in_table(log_n); }
hipError_t ntt_twiddles(void* d_tw, const uint64_t omega[4], uint32_t log_n, hipStream_t s, const uint64_t* scale) {
  Fe<FS> w, sc;
  memcpy(w.v, omega, 32);
  if (scale) memcpy(sc.v, scale, 32);
  return ntt29_build_tables<FS>(d_tw, w, log_n, s, scale ? &sc : nullptr);
}
hipError_t ntt_launch_(void* d_data, void* d_scratch, const void* d_tw, uint32_t log_n, size_t m, hipStream_t s,
                       const uint64_t* scale) {
  Fe<FS> sc;
  if (scale) memcpy(sc.v, scale, 32);
  return ntt29_launch<FS>((U128*)d_data, (U128*)d_scratch, d_tw, log_n, m, s, scale ? &sc : nullptr);
}
size_t group_fft_scratch(uint32_t log_n) { return gfft_scratch_bytes(log_n); }
hipError_t group_fft(const void* d_in_jac, void* d_out_jac, void* d_scratch, const uint64_t omega[4], uint32_t log_n, hipStream_t s) {
  Fe<FS> w;
  memcpy(w.v, omega, 32);
  return gfft_launch<CV>((const U128*)d_in_jac, (U128*)d_out_jac, d_scratch, w, log_n, s);
}
hipError_t poly_scale(void* d_a, size_t total, const uint64_t c[4], hipStream_t s) {
  Fe<FS> cv;
  memcpy(cv.v, c, 32);
  hipLaunchKernelGGL(poly_scale_kernel<FS>, dim3(poly_grid(total)), dim3(256), 0, s, (U128*)d_a, total, cv);
  return hipGetLastError();
}
hipError_t poly_powers(void* d_a, size_t n, size_t m, const uint64_t g[4], hipStream_t s) {
  Fe<FS> gv;
  memcpy(gv.v, g, 32);
  const size_t threads = (n + POLY_RUN - 1) / POLY_RUN;
  hipLaunchKernelGGL(poly_powers_kernel<FS>, dim3((unsigned)((threads + 255) / 256), (unsigned)m), dim3(256), 0, s,
                     (U128*)d_a, n, gv);
  return hipGetLastError();
}
hipError_t poly_mul_periodic(void* d_a, size_t total, const void* d_t, size_t period, hipStream_t s) {
  hipLaunchKernelGGL(poly_mul_periodic_kernel<FS>, dim3(poly_grid(total)), dim3(256), 0, s, (U128*)d_a, total,
                     (const U128*)d_t, period - 1);
  return hipGetLastError();
}
hipError_t poly_inverse(void* d_a, size_t total, hipStream_t s) {
  hipLaunchKernelGGL(poly_inverse_kernel<FS>, dim3(poly_grid(total)), dim3(256), 0, s, (U128*)d_a, total);
  return hipGetLastError();
}
hipError_t poly_divide_linear(const void* d_a, size_t n, const uint64_t z[4], void* d_q, void* d_ws, hipStream_t s) {
  Fe<FS> zv;
  memcpy(zv.v, z, 32);
  return poly_divide_linear_launch<FS>((const U128*)d_a, n, zv, (U128*)d_q, (U128*)d_ws, s);
}
hipError_t poly_prefix_product(const void* d_a, size_t n, void* d_out, void* d_ws, hipStream_t s) {
  return poly_prefix_product_launch<FS>((const U128*)d_a, n, (U128*)d_out, (U128*)d_ws, s);
}
hipError_t chacha20_scalars(void* d_out, size_t n, uint64_t first_block, const uint32_t key[8], hipStream_t s) {
  ChaChaKey k;
  memcpy(k.w, key, 32);
  hipLaunchKernelGGL(chacha20_scalars_kernel<FS>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (U128*)d_out, n,
                     first_block, k);
  return hipGetLastError();
}
hipError_t poly_pointwise(void* d_a, const void* d_b, size_t total, int op, hipStream_t s) {
  hipLaunchKernelGGL(poly_pointwise_kernel<FS>, dim3(poly_grid(total)), dim3(256), 0, s, (U128*)d_a,
                     (const U128*)d_b, total, op);
  return hipGetLastError();
}

// the ceiling the MSM kernels are priced against (bench.py `modmul_ceiling`): dependent products of the working
// form (h2_field29.hpp) over the base field, `blocks` x 256 threads, best of three launches
template <bool PLAIN>
__global__ void __launch_bounds__(256) modmul_rate_kernel(uint32_t* out, int iters) {
  Fe29<FB> a, b;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    a.v[i] = (int32_t)((0x1234567u * (i + 1) + threadIdx.x * 2654435761u) & L29_MASK);
    b.v[i] = (int32_t)((0x7654321u * (i + 3) + blockIdx.x * 40503u) & L29_MASK);
  }
  a.v[8] &= 0xFFFFF;
  b.v[8] &= 0xFFFFF;      // values below 2^252
  for (int k = 0; k < iters; k++) a = PLAIN ? fe29_mul_plain(a, b) : fe29_mul(a, b);
  uint32_t* o = out + 9 * (size_t)(blockIdx.x * blockDim.x + threadIdx.x);
#pragma unroll
  for (int i = 0; i < 9; i++) o[i] = (uint32_t)a.v[i];
}
hipError_t modmul_rate(int blocks, int iters, hipStream_t s, double* modmul_per_s) {
  void* buf = nullptr;
  hipError_t e = hipMalloc(&buf, (size_t)blocks * 256 * 9 * 4);
  if (e != hipSuccess) return e;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < 8 && e == hipSuccess; r++) {          // both forms of the product, the first launch of each a warm-up
    (void)hipEventRecord(e0, s);
    if (r < 4) hipLaunchKernelGGL(modmul_rate_kernel<false>, dim3(blocks), dim3(256), 0, s, (uint32_t*)buf, iters);
    else hipLaunchKernelGGL(modmul_rate_kernel<true>, dim3(blocks), dim3(256), 0, s, (uint32_t*)buf, iters);
    (void)hipEventRecord(e1, s);
    e = hipEventSynchronize(e1);
    float ms = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if ((r & 3) > 0 && ms < best) best = ms;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(buf);
  if (e == hipSuccess) *modmul_per_s = (double)blocks * 256.0 * iters / (best * 1e-3);
  return e;
}

template <class FP>
int selftest_field_t(int op, const uint64_t* a_, const uint64_t* b_, uint64_t* out) {
  Fe<FP> a, b, r;
  memcpy(a.v, a_, 32);
  memcpy(b.v, b_, 32);
  switch (op) {
    case 9: r = fe29_to_api(fe29_mul(fe29_from_api(a), fe29_from_api(b))); break;            // working-form product
    case 10: r = fe29_to_api(fe29_add(fe29_from_api(a), fe29_from_api(b))); break;
    case 11: r = fe29_to_api(fe29_sub(fe29_from_api(a), fe29_from_api(b))); break;
    case 12: {   // a long lazy chain: ((a - b)^2 - a b - 2 b) stays within the working form's bounds
      const Fe29<FP> x = fe29_from_api(a), y = fe29_from_api(b);
      const Fe29<FP> d = fe29_sub(x, y);
      r = fe29_to_api(fe29_norm(fe29_sub(fe29_sub(fe29_sub(fe29_sqr(d), fe29_mul(x, y)), y), y)));
      break;
    }
    case 0: r = fe_add(a, b); break;
    case 1: r = fe_sub(a, b); break;
    case 2: r = fe_mul(a, b); break;
    case 3: r = fe_inv(a); break;
    case 4: r = fe_to_mont(a); break;
    case 5: r = fe_from_mont(a); break;
    case 6: r = fe_neg(a); break;
    default: return -1;
  }
  memcpy(out, r.v, 32);
  return 0;
}
int selftest_field(int which, int op, const uint64_t* a, const uint64_t* b, uint64_t* out) {
  return which == 0 ? selftest_field_t<FB>(op, a, b, out) : selftest_field_t<FS>(op, a, b, out);
}
template <class FP>
__global__ void selftest_field_kernel(int op, const U128* a_, const U128* b_, U128* out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fe<FP> a = fe_load<FP>(a_ + 2 * (size_t)i), b = fe_load<FP>(b_ + 2 * (size_t)i), r;
  switch (op) {
    case 0: r = fe_add(a, b); break;
    case 1: r = fe_sub(a, b); break;
    case 2: r = fe_mul(a, b); break;
    case 3: r = fe_inv(a); break;
    case 4: r = fe_to_mont(a); break;
    case 5: r = fe_from_mont(a); break;
    case 7: r = fe_mul_cios(a, b); break;
    case 9: r = fe29_to_api(fe29_mul(fe29_from_api(a), fe29_from_api(b))); break;
    default: r = fe_neg(a); break;
  }
  fe_store<FP>(out + 2 * (size_t)i, r);
}
hipError_t selftest_field_device(int which, int op, const void* d_a, const void* d_b, void* d_out, uint32_t n,
                                 hipStream_t s) {
  if (which == 0)
    hipLaunchKernelGGL(selftest_field_kernel<FB>, dim3((n + 63) / 64), dim3(64), 0, s, op, (const U128*)d_a,
                       (const U128*)d_b, (U128*)d_out, n);
  else
    hipLaunchKernelGGL(selftest_field_kernel<FS>, dim3((n + 63) / 64), dim3(64), 0, s, op, (const U128*)d_a,
                       (const U128*)d_b, (U128*)d_out, n);
  return hipGetLastError();
}
// n pairs (p, q) of affine points in the API form, four lanes per pair; out: affine, API form
__global__ void __launch_bounds__(64)
selftest_curve_kernel(int op, const U128* p_, const U128* q_, U128* out, uint32_t n) {
  const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  if (i >= n) return;
  const Affine<CV> pa = affine_load<CV>(p_ + 4 * (size_t)i), qa = affine_load<CV>(q_ + 4 * (size_t)i);
  const Affine29<CV> p9{fe29_from_api(pa.x), fe29_from_api(pa.y)}, q9{fe29_from_api(qa.x), fe29_from_api(qa.y)};
  const Xyzz29<CV> P = xyzz29_from_affine(p9), Q = xyzz29_from_affine(q9);
  Xyzz29<CV> r;
  switch (op) {
    case 0: r = xyzz29_add_quad(P, Q); break;
    case 1: r = xyzz29_double_quad(P); break;
    case 2: r = xyzz29_add(P, Q); break;
    case 3: r = xyzz29_double(P); break;
    case 4: r = xyzz29_add_quad(xyzz29_double_quad(P), Q); break;
    default: {
      const uint32_t k = qa.x.v[0];
      r = Xyzz29<CV>::identity();
      if (k) {
        const int top = 31 - __clz(k);
        r = P;
        for (int bit = top - 1; bit >= 0; bit--) {
          r = xyzz29_double_quad(r);
          if ((k >> bit) & 1) r = xyzz29_add_quad(r, P);
        }
      }
    }
  }
  const Affine<CV> a = xyzz_to_affine(xyzz29_to_api(r));
  if ((threadIdx.x & 3) == 0) {
    fe_store<FB>(out + 4 * (size_t)i, a.x);
    fe_store<FB>(out + 4 * (size_t)i + 2, a.y);
  }
}
hipError_t selftest_curve_device(int op, const void* d_p, const void* d_q, void* d_out, uint32_t n, hipStream_t s) {
  hipLaunchKernelGGL(selftest_curve_kernel, dim3((4 * n + 63) / 64), dim3(64), 0, s, op, (const U128*)d_p,
                     (const U128*)d_q, (U128*)d_out, n);
  return hipGetLastError();
}
int selftest_curve(int op, const uint64_t* p_, const uint64_t* q_, uint64_t* out) {
  Affine<CV> p, q;
  memcpy(p.x.v, p_, 32); memcpy(p.y.v, p_ + 4, 32);
  memcpy(q.x.v, q_, 32); memcpy(q.y.v, q_ + 4, 32);
  Xyzz<CV> r;
  switch (op) {
    case 0: r = xyzz_add_affine(xyzz_from_affine(p), q); break;
    case 1: r = xyzz_double_affine(p); break;
    case 2: r = xyzz_add(xyzz_add_affine(xyzz_from_affine(p), q), xyzz_from_affine(q)); break;
    case 3: {
      uint32_t k = (uint32_t)q_[0];
      r = Xyzz<CV>::identity();
      Xyzz<CV> base = xyzz_from_affine(p);
      for (int bit = 31; bit >= 0; bit--) {
        r = xyzz_double(r);
        if ((k >> bit) & 1) r = xyzz_add(r, base);
      }
      break;
    }
    case 10: case 11: case 12: case 13: {          // the same four operations on the working representation
      const Affine29<CV> p9{fe29_from_api(p.x), fe29_from_api(p.y)}, q9{fe29_from_api(q.x), fe29_from_api(q.y)};
      Xyzz29<CV> r9;
      if (op == 10) r9 = xyzz29_add_affine(xyzz29_from_affine(p9), q9);
      else if (op == 11) r9 = xyzz29_double_affine(p9);
      else if (op == 12) r9 = xyzz29_add(xyzz29_add_affine(xyzz29_from_affine(p9), q9), xyzz29_from_affine(q9));
      else {
        const uint32_t k = (uint32_t)q_[0];
        r9 = Xyzz29<CV>::identity();
        const Xyzz29<CV> base = xyzz29_from_affine(p9);
        for (int bit = 31; bit >= 0; bit--) {
          r9 = xyzz29_double(r9);
          if ((k >> bit) & 1) r9 = xyzz29_add(r9, base);
        }
      }
      r = xyzz29_to_api(r9);
      break;
    }
    default: return -1;
  }
  Affine<CV> a = xyzz_to_affine(r);
  memcpy(out, a.x.v, 32);
  memcpy(out + 4, a.y.v, 32);
  return 0;
}
// out[0..3] = c, W, B, nbits; out[4 + w] = encoded digit of window w (host run of msm_digit_step);
// out[4 + W + w] = first bit of window w; out[4 + 2W + w] = its width
int selftest_digits(const uint64_t* scalar_mont, size_t n_for_geometry, uint32_t* out, uint32_t cap) {
  MsmGeom g = msm_geometry(n_for_geometry, FS::NUM_BITS);
  if (cap < 4 + 3 * g.W) return -1;
  for (uint32_t w = 0; w < g.W; w++) {
    out[4 + g.W + w] = g.off[w];
    out[4 + 2 * g.W + w] = g.width[w];
  }
  Fe<FS> s;
  memcpy(s.v, scalar_mont, 32);
  s = fe_from_mont(s);
  out[0] = g.c; out[1] = g.W; out[2] = g.B; out[3] = g.nbits;
  uint32_t carry = 0;
  MsmDigits dg(s.v);        // the kernels' form (windows taken in order from a shifted scalar) must agree with the indexed one
  for (uint32_t w = 0; w < g.W; w++) {
    out[4 + w] = msm_digit_step(s.v, g, w, carry);
    if (dg.next(g, w) != out[4 + w]) return -3;
  }
  return carry ? -2 : 0;  // a carry out of the top window would lose value
}

const CurveOps OPS = {CV::ID,      FS::ID,      FS::NUM_BITS, kernel_setup, table_build, msm_launch_,    srs_powers, fixed_base_mul, msm_small,
                      to_affine,   points_sum, ntt_table_bytes, ntt_scale_in_table, ntt_twiddles, ntt_launch_, group_fft_scratch, group_fft, poly_scale, poly_powers, poly_mul_periodic,
                      poly_pointwise, poly_inverse, poly_divide_linear, poly_prefix_product, chacha20_scalars, selftest_field, selftest_curve,
                      selftest_field_device, selftest_curve_device, selftest_digits, modmul_rate};

}  // namespace

#if H2_CURVE_ID == 0
const CurveOps* curve_ops_bn254() { return &OPS; }
#elif H2_CURVE_ID == 1
const CurveOps* curve_ops_pallas() { return &OPS; }
#else
const CurveOps* curve_ops_vesta() { return &OPS; }
#endif

}  // namespace h2
