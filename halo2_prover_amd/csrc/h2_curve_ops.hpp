// h2_curve_ops.hpp -- per-curve launch table.  Each curve's kernels are compiled in their own
// translation unit (h2_curve_impl.hip with -DH2_CURVE_ID=0/1/2) so the three build in parallel;
// h2_capi.hip holds only host logic and calls through this table.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace h2 {

struct U128;
struct MsmGeom;
struct MsmWorkspace;

struct CurveOps {
  int curve_id;
  int scalar_field_id;
  uint32_t scalar_bits;
  // raise the dynamic-LDS limits of this curve's kernels on the current device (once per device, from h2_init)
  hipError_t (*kernel_setup)();
  // MSM
  // d_bad: device counter (zeroed by the caller) of points that are not on the curve
  // d_scratch: MSM_TABLE_SCRATCH (80) bytes per window and point
  hipError_t (*table_build)(const void* d_bases, void* d_table, void* d_scratch, uint32_t n, const MsmGeom& g, uint32_t* d_bad, hipStream_t s);
  hipError_t (*msm_launch)(const void* d_table, const void* const* per_column_tables, uint32_t n_bases, const void* d_scalars, size_t n, size_t col_stride, size_t m,
                           const MsmGeom& g, char* ws_base, const MsmWorkspace& ws, hipStream_t s,
                           hipEvent_t ev_start, hipEvent_t ev_stop, hipEvent_t ev_tail,
                           void* d_out_jac /* optional: m Jacobian results, written by the MSM's last kernel */,
                           bool zeroed /* the workspace's zeroed region is zero already (left so by the previous launch sequence) */);
  hipError_t (*srs_powers)(void* d_out_affine, const uint64_t s_mont[4], uint32_t n, hipStream_t s);
  hipError_t (*fixed_base_mul)(void* d_out_affine, const void* d_scalars, uint32_t n, hipStream_t s);
  // `count` <= MSM_SMALL_MAX independent MSMs of a few dozen arbitrary points each (no table), side by side:
  // d_work = count * (ceil(max m / 16) * 144 bytes) and count 4-byte counters behind them (zeroed here);
  // count Jacobian points out
  hipError_t (*msm_small)(const void* const* d_points_affine, const void* const* d_scalars, const uint32_t* m, uint32_t count,
                          void* d_work, void* d_out_jac, hipStream_t s);
  hipError_t (*to_affine)(const void* d_xyzz, void* d_out, uint32_t m, hipStream_t s);
  // d_out[j] = sum_g d_in[g * count + j], Jacobian points in the API form
  hipError_t (*points_sum)(const void* d_in_jac, void* d_out_jac, uint32_t groups, uint32_t count, hipStream_t s);
  // NTT over the scalar field
  // the tables of one (omega, log n [, constant]): inter-pass twiddles, unpacked radix twiddles per pass, the
  // canonicalisation table (h2_ntt29.hpp).  ntt_scale_in_table: a transform scaled by a constant takes that constant
  // from tables built with it (two-pass plans); otherwise the final pass multiplies
  size_t (*ntt_table_bytes)(uint32_t log_n);
  bool (*ntt_scale_in_table)(uint32_t log_n);
  hipError_t (*ntt_twiddles)(void* d_tables, const uint64_t omega[4], uint32_t log_n, hipStream_t s, const uint64_t* scale /* or null */);
  hipError_t (*ntt_launch)(void* d_data, void* d_scratch, const void* d_tw, uint32_t log_n, size_t m,
                           hipStream_t s, const uint64_t* scale /* 4 limbs or null */);
  // best_fft over group elements (FftGroup for the curve: g_to_lagrange): n = 2^log_n Jacobian points (API form) in,
  // the transform out (may alias), natural order, unscaled; d_scratch: group_fft_scratch(log_n) bytes
  size_t (*group_fft_scratch)(uint32_t log_n);
  hipError_t (*group_fft)(const void* d_in_jac, void* d_out_jac, void* d_scratch, const uint64_t omega[4], uint32_t log_n,
                          hipStream_t s);
  // pointwise polynomial kernels over the scalar field (EvaluationDomain pieces)
  hipError_t (*poly_scale)(void* d_a, size_t total, const uint64_t c[4], hipStream_t s);
  hipError_t (*poly_powers)(void* d_a, size_t n, size_t m, const uint64_t g[4], hipStream_t s);
  hipError_t (*poly_mul_periodic)(void* d_a, size_t total, const void* d_t, size_t period, hipStream_t s);
  hipError_t (*poly_pointwise)(void* d_a, const void* d_b, size_t total, int op, hipStream_t s);
  hipError_t (*poly_inverse)(void* d_a, size_t total, hipStream_t s);
  // d_q = (d_a - d_a(z)) / (X - z); d_ws: 2 * 1024 elements of scratch
  hipError_t (*poly_divide_linear)(const void* d_a, size_t n, const uint64_t z[4], void* d_q, void* d_ws, hipStream_t s);
  // d_out[i] = prod_{j < i} d_a[j] (d_out[0] = 1; may alias d_a); d_ws as above
  hipError_t (*poly_prefix_product)(const void* d_a, size_t n, void* d_out, void* d_ws, hipStream_t s);
  // d_out[i] = Scalar::random of ChaCha20 block first_block + i (key = the 32 seed bytes as 8 little-endian words)
  hipError_t (*chacha20_scalars)(void* d_out, size_t n, uint64_t first_block, const uint32_t key[8], hipStream_t s);
  // host self-test hooks (host instantiation of the same templates)
  int (*selftest_field)(int which /* 0 = base field, 1 = scalar field */, int op, const uint64_t* a,
                        const uint64_t* b, uint64_t* out);
  int (*selftest_curve)(int op, const uint64_t* p, const uint64_t* q, uint64_t* out);
  // the same field ops run by a device kernel on n element pairs (device pointers)
  hipError_t (*selftest_field_device)(int which, int op, const void* d_a, const void* d_b, void* d_out, uint32_t n,
                                      hipStream_t s);
  // the working-form group law on the device, four lanes per pair (op 0/1: quad add / double, 2/3: lane add /
  // double, 5: [k]p by the weight kernel's double-and-add with a different k per quad)
  hipError_t (*selftest_curve_device)(int op, const void* d_p, const void* d_q, void* d_out, uint32_t n, hipStream_t s);
  int (*selftest_digits)(const uint64_t* scalar_mont, size_t n_for_geometry, uint32_t* out, uint32_t cap);
  // dependent working-form products of the base field on `blocks` x 256 threads: measured modmul/s (best of 3)
  hipError_t (*modmul_rate)(int blocks, int iters, hipStream_t s, double* modmul_per_s);
};

const CurveOps* curve_ops_bn254();
const CurveOps* curve_ops_pallas();
const CurveOps* curve_ops_vesta();

}  // namespace h2
