// h2_group_fft.hpp -- best_fft over GROUP elements (halo2_proofs::arithmetic::FftGroup for C::Curve).
//
// The reference's only use is ParamsKZG::new -> g_to_lagrange (reached from /root/reference/circuits/src/utils.rs:59-61):
// g_lagrange = n^-1 * best_fft(g as projective points, omega^-1, k).  Contract kept: in place, natural order in and
// out, A[k] = sum_j [omega^(jk)] a[j], unscaled; butterflies t = [w] b, b = a - t, a = a + t with the first twiddle of a
// block skipped (App. A.2).  Setup-time work (the product surface builds g_lagrange as [L_i(s)] G instead, h2_setup);
// this entry point exists so that a patched halo2_proofs can hand best_fft its G1 slices too.
//
// One thread per butterfly and stage, points kept between the stages as XYZZ on the 29-bit working form (144 bytes),
// the twiddle's scalar multiplication a plain double-and-add over the scalar field's bits (one-lane arithmetic of
// h2_curve29.hpp with all exceptional cases): n/2 * log n multiplications of ~380 point operations each -- about 20 ms
// for 2^16 points on MI355X against minutes on a CPU core; nothing here is tuned beyond that.
#pragma once
#include "h2_curve29.hpp"

namespace h2 {

// tw[i] = omega^i as a CANONICAL integer (8 x u32), i < half_n
template <class FS>
__global__ void __launch_bounds__(256) gfft_twiddle_kernel(U128* tw, Fe<FS> omega, uint32_t half_n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= half_n) return;
  fe_store<FS>(tw + 2 * (size_t)i, fe_from_mont(fe_pow_u64(omega, (uint64_t)i)));
}

// W[bitrev(i)] = in[i] (Jacobian, API form) as XYZZ on the working form
template <class CV>
__global__ void __launch_bounds__(256) gfft_load_kernel(const U128* __restrict__ in_jac, uint32_t* __restrict__ W, uint32_t n,
                                                       uint32_t log_n) {
  using B = typename CV::Base;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const U128* p = in_jac + 6 * (size_t)i;
  const Fe<B> x = fe_load<B>(p), y = fe_load<B>(p + 2), z = fe_load<B>(p + 4);
  Xyzz29<CV> r = Xyzz29<CV>::identity();
  if (!z.is_zero()) {
    const Fe29<B> z9 = fe29_from_api(z), zz = fe29_mul(z9, z9);
    r = Xyzz29<CV>{fe29_from_api(x), fe29_from_api(y), zz, fe29_mul(zz, z9)};
  }
  const uint32_t dst = log_n ? (__brev(i) >> (32 - log_n)) : 0u;
  xyzz29_store<CV>(W + XYZZ29_WORDS * (size_t)dst, r);
}

// stage s: butterflies (i0, i1 = i0 + 2^s) with twiddle omega^(pos * n / 2^(s+1))
template <class CV>
__global__ void __launch_bounds__(256) gfft_stage_kernel(uint32_t* __restrict__ W, const U128* __restrict__ tw, uint32_t n,
                                                        uint32_t log_n, uint32_t s) {
  using S = typename CV::Scalar;
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= (n >> 1)) return;
  const uint32_t half = 1u << s, pos = b & (half - 1), grp = b >> s;
  const uint32_t i0 = (grp << (s + 1)) + pos, i1 = i0 + half;
  const Xyzz29<CV> a = xyzz29_load<CV>(W + XYZZ29_WORDS * (size_t)i0), q = xyzz29_load<CV>(W + XYZZ29_WORDS * (size_t)i1);
  Xyzz29<CV> t = q;
  if (pos != 0) {                                             // the first twiddle of a block is 1
    const Fe<S> k = fe_load<S>(tw + 2 * ((size_t)pos << (log_n - 1 - s)));
    t = Xyzz29<CV>::identity();
    int top = (int)S::NUM_BITS - 1;
    while (top > 0 && !((k.v[top >> 5] >> (top & 31)) & 1u)) top--;
    for (int bit = top; bit >= 0; bit--) {
      t = xyzz29_double(t);
      if ((k.v[bit >> 5] >> (bit & 31)) & 1u) t = xyzz29_add(t, q);
    }
  }
  Xyzz29<CV> nt = t;
  nt.y = fe29_norm(fe29_neg(t.y));
  xyzz29_store<CV>(W + XYZZ29_WORDS * (size_t)i0, xyzz29_add(a, t));
  xyzz29_store<CV>(W + XYZZ29_WORDS * (size_t)i1, xyzz29_add(a, nt));
}

// out[i] = W[i] as a Jacobian point in the API form
template <class CV>
__global__ void __launch_bounds__(256) gfft_store_kernel(const uint32_t* __restrict__ W, U128* __restrict__ out_jac, uint32_t n) {
  using B = typename CV::Base;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fe<B> jx, jy, jz;
  xyzz_to_jacobian(xyzz29_to_api(xyzz29_load<CV>(W + XYZZ29_WORDS * (size_t)i)), jx, jy, jz);
  U128* o = out_jac + 6 * (size_t)i;
  fe_store<B>(o, jx);
  fe_store<B>(o + 2, jy);
  fe_store<B>(o + 4, jz);
}

// scratch: n * 144 bytes of points + (n / 2) * 32 bytes of twiddles (at least 64)
inline size_t gfft_scratch_bytes(uint32_t log_n) {
  const size_t n = (size_t)1 << log_n;
  return n * (XYZZ29_WORDS * 4) + std::max<size_t>(64, (n / 2) * 32);
}
template <class CV>
inline hipError_t gfft_launch(const U128* d_in_jac, U128* d_out_jac, void* d_scratch, const Fe<typename CV::Scalar>& omega,
                              uint32_t log_n, hipStream_t stream) {
  const uint32_t n = 1u << log_n;
  uint32_t* W = (uint32_t*)d_scratch;
  U128* tw = (U128*)((char*)d_scratch + (size_t)n * (XYZZ29_WORDS * 4));
  if (n > 1)
    hipLaunchKernelGGL(gfft_twiddle_kernel<typename CV::Scalar>, dim3((n / 2 + 255) / 256), dim3(256), 0, stream, tw, omega, n / 2);
  hipLaunchKernelGGL(gfft_load_kernel<CV>, dim3((n + 255) / 256), dim3(256), 0, stream, d_in_jac, W, n, log_n);
  for (uint32_t s = 0; s < log_n; s++)
    hipLaunchKernelGGL(gfft_stage_kernel<CV>, dim3((n / 2 + 255) / 256), dim3(256), 0, stream, W, tw, n, log_n, s);
  hipLaunchKernelGGL(gfft_store_kernel<CV>, dim3((n + 255) / 256), dim3(256), 0, stream, W, d_out_jac, n);
  return hipGetLastError();
}

}  // namespace h2
