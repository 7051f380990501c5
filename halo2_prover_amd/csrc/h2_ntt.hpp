// h2_ntt.hpp -- NTT over 256-bit Montgomery fields for gfx950 (radix-2 stages fused in pairs): the plan.  The pass kernel
// is in h2_ntt29.hpp.
//
// Device replacement for halo2_proofs::arithmetic::best_fft / recursive_butterfly_arithmetic
// (halo2_proofs @6b43b6b, src/arithmetic.rs -- un-vendored; algorithm restated in SURVEY.md
// App. A.2; reached from /root/reference/circuits/src/utils.rs:83-91,105-120 through
// EvaluationDomain::{lagrange_to_coeff, coeff_to_extended, extended_to_coeff}).
// Contract kept: in place, natural order in and out, A[k] = sum_j a[j] w^(jk), unscaled.
//
// Structure (MI355X-first, not the reference's recursion): n = R1 * R2 * R3 with every radix
// <= 2^10.  Each pass stages a tile of R x C elements in LDS (C consecutive "columns", so every HBM access is a
// run of C*32 contiguous bytes; C = 4, or 2 for R >= 512 so that the tile stays <= 72 KB and two blocks share a CU), runs the log2(R) butterfly stages out
// of LDS with the R/2 twiddles of that radix also held in LDS, applies the inter-pass
// twiddle w^(outer*i*k) and writes the tile back.  The last pass writes the digit-reversed
// position directly, so no bit-reversal sweep over HBM is needed.  Per pass the algorithmic
// HBM traffic is one read and one write of the column (64 B per element).
#pragma once
#include "h2_field.hpp"
#include "h2_tune.hpp"

namespace h2 {

struct NttPass {
  uint32_t log_n;      // size of the whole transform
  uint32_t log_outer;  // log2 of the product of the radices of earlier passes
  uint32_t log_r;      // radix of this pass
  uint32_t log_inner;  // log_n - log_outer - log_r
  uint32_t log_c;      // tile columns
  uint32_t log_r1;     // radix of pass 0 (0 when this is the only pass)
  uint32_t log_r2;     // radix of pass 1 when there are three passes, else 0
  uint32_t is_final;
  uint32_t has_scale;  // final pass multiplies every output by `scale` (EvaluationDomain::ifft's n^-1)
  uint32_t tw_global;  // 29-bit kernel: radix twiddles read from the table in global memory (L1/L2) instead of LDS
};

__device__ __forceinline__ uint32_t h2_bitrev(uint32_t x, uint32_t bits) {
  return bits ? (__brev(x) >> (32 - bits)) : 0;
}

// twiddle tables are filled TW_RUN consecutive entries per thread
constexpr int TW_RUN = 16;

// (Round 1's pass kernel on 8 x 32-bit limbs lived here; the 29-bit kernel of h2_ntt29.hpp replaced it in round 2 --
// 8-14 % faster per transform, DESIGN.md section 4.2 -- and round 3 removed it from the build.)

// ---- host side -------------------------------------------------------------------------------
struct NttPlan {
  int npass;
  NttPass pass[3];
  uint32_t threads[3];
  size_t lds_bytes[3];
  uint32_t tiles[3];
};

constexpr uint32_t NTT_MAX_LOG_R = 10;  // R <= 1024: tile R*2 elements = 64 KiB + 16 KiB twiddles

inline NttPlan ntt_make_plan(uint32_t log_n, uint32_t max_log_r = NTT_MAX_LOG_R) {
  NttPlan pl{};
  static const int tune_r = tune_int("H2_TUNE_NTT_MAXR", 0);     // tuning builds only (h2_tune.hpp)
  static const int tune_c9 = tune_int("H2_TUNE_NTT_LC9", -1);
  static const int tune_c = tune_int("H2_TUNE_NTT_LC", -1);
  if (tune_r > 0) max_log_r = (uint32_t)tune_r;
  uint32_t np = log_n == 0 ? 1 : (log_n + max_log_r - 1) / max_log_r;
  if (np > 3) np = 3;  // callers reject log_n > 30
  uint32_t radix[3] = {0, 0, 0};
  uint32_t left = log_n;
  for (uint32_t p = 0; p < np; p++) {
    radix[p] = (left + (np - p) - 1) / (np - p);
    left -= radix[p];
  }
  pl.npass = (int)np;
  uint32_t outer = 0;
  for (uint32_t p = 0; p < np; p++) {
    NttPass& P = pl.pass[p];
    P.log_n = log_n;
    P.log_outer = outer;
    P.log_r = radix[p];
    P.log_inner = log_n - outer - radix[p];
    P.is_final = (p == np - 1);
    P.log_r1 = np >= 2 ? radix[0] : 0;
    P.log_r2 = np == 3 ? radix[1] : 0;
    // tile columns: keep the tile at <= 72 KB of LDS so that at least two blocks share a CU (one computes while the
    // other waits at a barrier): 2 columns for R >= 512, 4 below (measured: R = 1024 x 4 columns, one block per CU,
    // is 13 % slower on the 2^19 transforms; R = 256 x 2 columns is 5-30 % slower than x 4 on the 2^16 ones)
    uint32_t lc = radix[p] >= 9 ? 1 : 2;
    if (radix[p] >= 9 && tune_c9 >= 0) lc = (uint32_t)tune_c9;
    if (radix[p] < 9 && tune_c >= 0) lc = (uint32_t)tune_c;
    if (P.is_final) {
      if (lc > P.log_r1) lc = P.log_r1;
    } else {
      if (lc > P.log_inner) lc = P.log_inner;
    }
    P.log_c = lc;
    const uint32_t rc = 1u << (P.log_r + lc);
    pl.lds_bytes[p] = ((size_t)rc + ((size_t)1 << P.log_r) / 2) * 32;
    if (pl.lds_bytes[p] < 64) pl.lds_bytes[p] = 64;
    uint32_t thr = rc / 4;                      // one radix-4 butterfly per thread and double stage
    if (thr < 64) thr = 64;
    if (thr > 1024) thr = 1024;
    pl.threads[p] = thr;
    pl.tiles[p] = 1u << (log_n - P.log_r - lc);
    outer += radix[p];
  }
  return pl;
}

}  // namespace h2
