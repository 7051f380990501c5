// h2_msm.hpp -- Pippenger bucket MSM for gfx950 with HBM-resident precomputed bases.
//
// Device replacement for halo2_proofs::arithmetic::best_multiexp / multiexp_serial
// (halo2_proofs @6b43b6b, src/arithmetic.rs -- un-vendored; algorithm restated in SURVEY.md
// App. A.1; reached from /root/reference/circuits/src/utils.rs:83-91,105-120 through
// ParamsKZG::commit / commit_lagrange).  Contract kept: result = sum_i coeffs[i] * bases[i]
// as a group element; zero coefficients contribute nothing.
//
// MI355X-first structure (not the reference's chunk-per-thread serial Pippenger):
//  * The bases are an SRS that every commitment of a proof reuses, and the GPU has 288 GB of
//    HBM: at registration the table T[w][i] = 2^(off_w) * P_i (affine) is built once.  Every
//    window of a scalar then selects a point of the same weight, so ALL windows share ONE set
//    of 2^(c-1) buckets per column: no per-window running sums, and the 255 serial doublings
//    of the window combination disappear.
//  * nbits+1 scalar bits are split into W windows of c or c-1 bits (balanced), so no window is
//    degenerate; signed digits (|d| <= 2^(width-1)) halve the bucket count, the sign rides in
//    bit 31 of the sorted entry and negates y on the fly.
//  * sort by bucket: count pass (LDS histogram per tile; its one returning global atomic per tile and bucket also
//    hands the tile its base inside the bucket's list) -> exclusive scan -> scatter (LDS cursors), giving per
//    bucket the contiguous list of table indices to add (4 bytes per entry; the entry's bucket is implied by
//    `offsets`; both passes decompose the scalar themselves, no digits array is stored).
//  * accumulate = segmented reduction with perfect load balance: every thread adds exactly T
//    consecutive sorted entries (mixed XYZZ additions, 8M + 2S) whatever bucket they belong
//    to, writes complete runs straight to the bucket and its cut-off head / tail runs to
//    partial slots; a fix-up kernel sums each bucket's pieces (buckets cut into hundreds of pieces --
//    degenerate columns -- first go through a wave-per-128-pieces reduction), then the bucket weights (b+1) by a
//    row / column split of the bucket index (msm_rowcol_kernel, msm_final_kernel: two plain sums per bucket and a
//    few small multiplications per column).  The whole tail runs with FOUR LANES PER POINT (h2_curve_quad.hpp).
//  * the sort's tiles are handed out XCD by XCD (msm_tile_id): contiguous (column, tile) ranges and a counter set
//    per XCD.
//  * all base-field arithmetic of these kernels is done on 9 x 29-bit signed limbs with lazy reduction
//    (h2_field29.hpp, h2_curve29.hpp); the table holds the points in that Montgomery form (R' = 2^261) and the m
//    results are converted back to the API's form by the final kernel.
#pragma once
#include "h2_curve29.hpp"
#include "h2_curve_quad.hpp"
#include "h2_tune.hpp"
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdlib>

namespace h2 {

// tools/microbench_tail.hip times the stages of the tail kernels with the 100 MHz counter; nothing in the library build
#ifdef H2_TAIL_STAMPS
__device__ unsigned long long* h2_stamps;
#define H2_STAMP(slot)                                                                                              \
  do {                                                                                                              \
    if (threadIdx.x == 0) h2_stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (slot)] = wall_clock64(); \
  } while (0)
#else
#define H2_STAMP(slot)
#endif

constexpr uint32_t MSM_SIGN = 0x80000000u;
constexpr uint32_t MSM_MAX_WINDOWS = 48;
constexpr uint32_t MSM_HOT_SPAN = 256;    // keys cut into more pieces than this take the hierarchical path
constexpr uint32_t MSM_HOT_SEG = 128;     // pieces summed by one wave of msm_hot_reduce_kernel
constexpr uint32_t MSM_NOT_HOT = 0xFFFFFFFFu;
constexpr uint32_t MSM_SORT_THREADS = 1024;   // block size of the digits / scatter kernels
constexpr uint32_t MSM_CHUNK_WAVES = 3;   // resident waves per SIMD of the accumulate kernel (159 VGPRs)
constexpr uint32_t MSM_MAX_C_ONE_LEVEL = 16;   // one-level sort: B*4 bytes of LDS histogram must fit one CU: 2^15 * 4 = 128 KiB
constexpr uint32_t MSM_MAX_C = 19;        // two-level sort: 1024 coarse bins x 256 fine buckets per column

struct MsmGeom {
  uint32_t c;        // widest window, bits
  uint32_t W;        // windows
  uint32_t B;        // buckets per column = 2^(c-1)
  uint32_t nbits;    // scalar field bits
  uint32_t wbase, wextra;          // window w has wbase + (w < wextra) bits and starts where window w - 1 ends
  uint8_t off[MSM_MAX_WINDOWS];    // first bit of window w
  uint8_t width[MSM_MAX_WINDOWS];  // bits of window w (c or c-1)
};

inline MsmGeom msm_geometry(size_t n, uint32_t nbits) {
  uint32_t lg = 0;
  while (((size_t)1 << (lg + 1)) <= n) lg++;
  // measured on the Poseidon proof shape (bench.py --k 12 / 14 / 16 / 18) and on single 2^20 columns: wider windows
  // save additions in the accumulate kernel (W = ceil(256 / c)) but every bucket costs ~16 point operations in the
  // tail, so the best width is log2 n - 3 up to 2^15 and log2 n - 4 from 2^16 (round 3: with the accumulate kernel 17 %
  // faster the fixed costs weigh more -- at 2^16 c = 11 / 12 / 13 / 14 give 2.93 / 2.69 / 2.73 / 2.85 ms per step,
  // tools/sweep_c.sh)
  int c = (int)lg - (lg <= 15 ? 3 : 4);
  // from 2^21 terms the widest window the two-level sort reaches (19 bits: 14 windows instead of 16, 2^18 buckets per
  // column) pays for its tail: measured -3% at 2^21, -8% at 2^22, -13% at 2^23, -9% for 8 columns of 2^24
  // (profiles/r03_sweep_cwide.log; 17 and 18 bits keep 16 / 15 windows and do not); at 2^20 the tail costs more than the
  // eighth of the additions it saves
  c = lg <= 20 ? std::min(c, 16) : 19;
  c = tune_int("H2_TUNE_C", c);             // tuning builds only (h2_tune.hpp)
  if (c < 6) c = 6;
  if (c > (int)MSM_MAX_C) c = (int)MSM_MAX_C;
  MsmGeom g{};
  g.nbits = nbits;
  const uint32_t total = nbits + 1;  // one spare bit: the top window never carries out
  g.W = (total + c - 1) / c;
  const uint32_t base = total / g.W, extra = total % g.W;
  g.wbase = base;
  g.wextra = extra;
  uint32_t o = 0;
  for (uint32_t w = 0; w < g.W; w++) {
    g.off[w] = (uint8_t)o;
    g.width[w] = (uint8_t)(base + (w < extra ? 1 : 0));
    o += g.width[w];
  }
  g.c = base + (extra ? 1 : 0);
  g.B = 1u << (g.c - 1);
  return g;
}

// ---- table build: T[w][i] = 2^(off_w) * P_i, affine -----------------------------------------
// `bad` counts the points that are neither the identity (0, 0) nor on the curve y^2 = x^3 + b (or whose
// coordinates are not canonical): the reference reads its params with curve checks (SerdeFormat::RawBytes)
//
// One thread per point walks the 255 doublings once, in Jacobian form with a single Z (for a = 0 the XYZZ doubling
// needs neither ZZ nor ZZZ: Z3 = Z * 2Y is all that has to be tracked), leaves (X, Y) of every window in the table slot
// and (Z, running product of the Z so far) in `scratch`, inverts the LAST running product, and walks back dividing:
// ONE inversion per point instead of one per window and point (20 at 2^16: they were 80 % of the kernel, 7 ms of the
// 8 that registering 2^16 bases took; ParamsKZG::read rebuilds two such tables on every call of the reference's flow).
// The chain runs on the 29-bit working form (the table's own form): 2.3 -> ~1 ms for 2^16 BN254 points.
// Intermediate values are 9 limbs each: X and Y's limbs 0..6 sit in the 64-byte table slot, Y's limbs 7..8, Z and the
// running product in the 80-byte scratch slot (MSM_TABLE_SCRATCH bytes per window and point).
constexpr size_t MSM_TABLE_SCRATCH = 80;
template <class CV>
__global__ void __launch_bounds__(256)
msm_table_kernel(const U128* __restrict__ bases, U128* __restrict__ table, uint32_t* __restrict__ scratch, uint32_t n, MsmGeom g,
                 uint32_t* __restrict__ bad) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  using B = typename CV::Base;
  using W = Fe29<B>;
  Affine<CV> p = affine_load<CV>(bases + 4 * (size_t)i);
  if (!p.is_identity()) {
    Fe<B> b;
#pragma unroll
    for (int k = 0; k < 8; k++) b.v[k] = CV::B(k);
    uint32_t rx[8], ry[8];
    fe_reduce_once<B>(rx, p.x.v, 0);                    // canonical <=> unchanged by the conditional subtraction
    fe_reduce_once<B>(ry, p.y.v, 0);
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 8; k++) ok = ok && rx[k] == p.x.v[k] && ry[k] == p.y.v[k];
    if (!ok || fe_sqr(p.y) != fe_add(fe_mul(fe_sqr(p.x), p.x), b)) {
      atomicAdd(bad, 1u);
      p = Affine<CV>::identity();                       // keep the kernel's arithmetic on valid points
    }
  }
  if (p.is_identity()) {                                // every multiple of the identity is the identity
    for (uint32_t w = 0; w < g.W; w++) affine29_store_table<CV>(table + 4 * ((size_t)w * n + i), p);
    return;
  }
  auto slot_of = [&](uint32_t w) { return reinterpret_cast<uint32_t*>(table + 4 * ((size_t)w * n + i)); };
  auto scr_of = [&](uint32_t w) { return scratch + (MSM_TABLE_SCRATCH / 4) * ((size_t)w * n + i); };
  auto put9 = [](uint32_t* q, const W& v) {
#pragma unroll
    for (int l = 0; l < 9; l++) q[l] = (uint32_t)v.v[l];
  };
  auto get9 = [](const uint32_t* q) {
    W v;
#pragma unroll
    for (int l = 0; l < 9; l++) v.v[l] = (int32_t)q[l];
    return v;
  };
  W X = fe29_from_api(p.x), Y = fe29_from_api(p.y), Z = fe29_from_api(Fe<B>::one()), run = Z;
  for (uint32_t w = 0; w < g.W; w++) {
    uint32_t* slot = slot_of(w);
    uint32_t* scr = scr_of(w);
    put9(slot, X);
#pragma unroll
    for (int l = 0; l < 7; l++) slot[9 + l] = (uint32_t)Y.v[l];
    scr[0] = (uint32_t)Y.v[7];
    scr[1] = (uint32_t)Y.v[8];
    run = fe29_mul(run, Z);
    put9(scr + 2, Z);
    put9(scr + 11, run);
    if (w + 1 < g.W) {
      for (uint32_t k = 0; k < g.width[w]; k++) {
        // dbl-2008-s-1 (XYZZ) for a = 0, with Z instead of ZZ / ZZZ: the order of these curves is odd, Y != 0
        const W U = fe29_norm(fe29_add(Y, Y));
        const W V = fe29_mul(U, U), Wd = fe29_mul(U, V), S = fe29_mul(X, V);
        const W xx = fe29_mul(X, X), M = fe29_norm(fe29_add(fe29_add(xx, xx), xx));
        const W X3 = fe29_norm(fe29_sub(fe29_sub(fe29_mul(M, M), S), S));
        Y = fe29_norm(fe29_sub(fe29_mul(M, fe29_norm(fe29_sub(S, X3))), fe29_mul(Wd, Y)));
        X = X3;
        Z = fe29_mul(Z, U);
      }
    }
  }
  W inv = fe29_inv(run);                                // 1 / (Z_0 Z_1 ... Z_(W-1))
  for (uint32_t w = g.W; w-- > 0;) {
    uint32_t* slot = slot_of(w);
    const uint32_t* scr = scr_of(w);
    const W zw = get9(scr + 2);
    W zinv = inv;                                       // w == 0: Z_0 = 1 and inv is 1 by now
    if (w > 0) zinv = fe29_mul(inv, get9(scr_of(w - 1) + 11));
    inv = fe29_mul(inv, zw);
    const W zi2 = fe29_mul(zinv, zinv);
    const W x9 = get9(slot);
    W y9;
#pragma unroll
    for (int l = 0; l < 7; l++) y9.v[l] = (int32_t)slot[9 + l];
    y9.v[7] = (int32_t)scr[0];
    y9.v[8] = (int32_t)scr[1];
    // the table's entry: canonical, packed working form (what affine29_store_table writes for an API-form point)
    const Fe<B> ax = fe29_canonical_pack(fe29_mul(x9, zi2)), ay = fe29_canonical_pack(fe29_mul(y9, fe29_mul(zi2, zinv)));
    U128* t = table + 4 * ((size_t)w * n + i);
    fe_store<B>(t, ax);
    fe_store<B>(t + 2, ay);
  }
}

// ---- signed window digits -------------------------------------------------------------------
// One step: window w of the canonical scalar v (8 x u32), carry in/out.
// Returns 0 (no contribution) or |d| | sign<<31 with 1 <= |d| <= 2^(width_w - 1) <= B.
// Invariant: sum_w d_w * 2^(off_w) = v; the top window never carries out (spare bit).
H2_HD uint32_t msm_digit_step(const uint32_t v[8], const MsmGeom& g, uint32_t w, uint32_t& carry) {
  const uint32_t width = g.width[w], bit = g.off[w];
  const uint32_t mask = (1u << width) - 1, halfw = 1u << (width - 1);
  const uint32_t limb = bit >> 5, off = bit & 31;
  uint64_t two = v[limb];
  if (limb + 1 < 8) two |= (uint64_t)v[limb + 1] << 32;
  uint32_t raw = ((uint32_t)(two >> off) & mask) + carry;
  if (raw > halfw) {
    carry = 1;
    const uint32_t mag = (1u << width) - raw;  // digit = raw - 2^width <= 0 (0 when raw == 2^width)
    return mag ? (mag | MSM_SIGN) : 0u;
  }
  carry = 0;
  return raw;
}

// The same digits taken IN ORDER from a scalar that is shifted down window by window: the windows are consecutive
// (off_(w+1) = off_w + width_w) and their widths are base + 1 for the first `extra` windows, base for the rest, so no
// table is indexed and no register is addressed dynamically (msm_digit_step's v[bit >> 5] put the scalar in scratch
// memory: two scratch loads per digit, ~3 us per million entries in every sort kernel).
struct MsmDigits {
  uint32_t v[8];
  uint32_t carry;
  H2_HD explicit MsmDigits(const uint32_t s[8]) : carry(0) {
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = s[i];
  }
  // window w's digit (0, or |d| | sign << 31); call with w = 0, 1, ..., W - 1 in turn
  H2_HD uint32_t next(const MsmGeom& g, uint32_t w) {
    const uint32_t width = g.wbase + (w < g.wextra ? 1u : 0u);          // 1 <= width <= 16
    const uint32_t mask = (1u << width) - 1, halfw = 1u << (width - 1);
    const uint32_t raw = (v[0] & mask) + carry;
#pragma unroll
    for (int i = 0; i < 7; i++) v[i] = (v[i] >> width) | (v[i + 1] << (32 - width));
    v[7] >>= width;
    if (raw > halfw) {
      carry = 1;
      const uint32_t mag = (1u << width) - raw;
      return mag ? (mag | MSM_SIGN) : 0u;
    }
    carry = 0;
    return raw;
  }
};

// Workgroups are handed to the 8 XCDs round-robin by linear block id, and every XCD has an L2 of its own.  The sort's
// two kernels therefore give XCD x a CONTIGUOUS range of (column, tile) pairs and a counter set of its own:
//     group x = blockIdx.x % 8,   virtual id v = x * per + blockIdx.x / 8   (per = ceil(total / 8)),
//     (column, tile) = (v / tiles, v % tiles).
// A bucket's list is then laid out group by group (group x starts behind the counts of the groups below it), so the
// stores that land in one 64-byte sector nearly all come from one XCD.  Measured (Poseidon k = 16 shape): the scatter
// kernel 65 -> 52 us.  WRITE_SIZE did NOT fall (162 -> 152 MB for 21 MB of entries): on this chip every store
// instruction's bytes leave the L2 as they are written (MI355X_MICROARCH.md, stores of each flavour), so a scattered
// 4-byte store is one fabric write whatever the L2 holds -- only wider runs per store instruction change that
// (msm_scatter_staged_kernel: 33 MB).
// If the hardware mapped blocks differently only the locality would suffer: both kernels compute the same (group, tile).
constexpr uint32_t MSM_XCDS = 8;
struct MsmTileId {
  uint32_t group, col, tile;
  bool live;
};
// (a function of the block index alone, so that the host can check the mapping: h2_selftest_msm_tiles)
H2_HD MsmTileId msm_tile_id_of(uint32_t block, uint32_t tiles, uint32_t m) {
  const uint32_t total = tiles * m, per = (total + MSM_XCDS - 1) / MSM_XCDS;
  MsmTileId t;
  t.group = block % MSM_XCDS;
  const uint32_t v = t.group * per + block / MSM_XCDS;
  // the grid is rounded up to a multiple of 8 blocks: the surplus blocks of the last groups are dead.  Without the
  // `v < total` test such a block would take (column, tile) = (v / tiles >= m, ...) and read scalars past the last
  // column (the memory-access fault recorded in round 2)
  t.live = block / MSM_XCDS < per && v < total;
  t.col = v / tiles;
  t.tile = v % tiles;
  return t;
}
__device__ __forceinline__ MsmTileId msm_tile_id(uint32_t tiles, uint32_t m) { return msm_tile_id_of(blockIdx.x, tiles, m); }
inline uint32_t msm_tile_grid(uint32_t tiles, uint32_t m) {
  return (tiles * m + MSM_XCDS - 1) / MSM_XCDS * MSM_XCDS;
}

}  // namespace h2
#include "h2_tune.hpp"
#include "h2_msm_sort2.hpp"
namespace h2 {

// ---- the one-level sort (short columns: B <= 4096 buckets per column) ------------------------------------------------
// Count pass: every scalar's signed digits (0 or |d| | sign<<31) go into an LDS histogram of the tile;
// gcounts[group][col*B + |d|-1] += the tile's count with one RETURNING global atomic per non-empty bucket, whose
// result -- where this tile's entries start inside the group's part of the bucket's list -- is kept in tile_base.
// grid = msm_tile_grid(tiles, m), LDS B*4.
template <class CV>
__global__ void __launch_bounds__(1024)
msm_digits_kernel(const U128* __restrict__ scalars, uint32_t* __restrict__ gcounts, uint32_t* __restrict__ tile_base,
                  uint32_t* __restrict__ tile_hist /* null, or the tile's own counts for the staged scatter */,
                  uint32_t n, size_t col_stride /* elements */, uint32_t tile, uint32_t tiles, uint32_t m, MsmGeom g) {
  using S = typename CV::Scalar;
  extern __shared__ uint32_t hist[];
  const MsmTileId id = msm_tile_id(tiles, m);
  if (!id.live) return;
  const uint32_t col = id.col;
  for (uint32_t b = threadIdx.x; b < g.B; b += blockDim.x) hist[b] = 0;
  __syncthreads();
  const uint32_t lo = id.tile * tile, hi = min(lo + tile, n);
  for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    MsmDigits dg(fe_from_mont(fe_load<S>(scalars + 2 * (col_stride * col + i))).v);
    for (uint32_t w = 0; w < g.W; w++) {
      const uint32_t enc = dg.next(g, w);
      if (enc) atomicAdd(&hist[(enc & ~MSM_SIGN) - 1], 1u);
    }
  }
  __syncthreads();
  // the value the atomic returns is where this tile's entries start inside the group's run: kept for the scatter
  uint32_t* tb = tile_base + ((size_t)col * tiles + id.tile) * g.B;
  uint32_t* gc = gcounts + (size_t)id.group * ((size_t)m * g.B) + (size_t)col * g.B;
  uint32_t* th = tile_hist ? tile_hist + ((size_t)col * tiles + id.tile) * g.B : nullptr;
  for (uint32_t b = threadIdx.x; b < g.B; b += blockDim.x) {
    const uint32_t h = hist[b];
    tb[b] = h ? atomicAdd(&gc[b], h) : 0u;
    if (th) th[b] = h;
  }
}

// counts[key] = the key's entries over all groups (the multi-kernel scan's input)
static __global__ void __launch_bounds__(256) msm_group_fold_kernel(const uint32_t* __restrict__ gcounts, uint32_t* __restrict__ counts, size_t K) {
  const size_t key = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (key >= K) return;
  uint32_t run = 0;
  for (uint32_t x = 0; x < MSM_XCDS; x++) run += gcounts[x * K + key];
  counts[key] = run;
}

// ---- exclusive scan over K counts (three small kernels) -------------------------------------
constexpr uint32_t SCAN_BLOCK = 1024;  // elements per block (256 threads x 4)
static __global__ void __launch_bounds__(256) scan_reduce_kernel(const uint32_t* in, uint32_t* block_sums, size_t K) {
  __shared__ uint32_t sh[256];
  const size_t base = (size_t)blockIdx.x * SCAN_BLOCK;
  uint32_t s = 0;
  for (uint32_t k = 0; k < 4; k++) {
    const size_t idx = base + threadIdx.x * 4 + k;
    if (idx < K) s += in[idx];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (uint32_t st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) block_sums[blockIdx.x] = sh[0];
}
// single block: exclusive scan of nb block sums in place; total -> *total_out
static __global__ void __launch_bounds__(1024)
scan_blocksums_kernel(uint32_t* block_sums, uint32_t nb, uint32_t* total_out) {
  __shared__ uint32_t sh[1024];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nb; base += 1024) {
    const uint32_t idx = base + threadIdx.x;
    const uint32_t v = idx < nb ? block_sums[idx] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t st = 1; st < 1024; st <<= 1) {
      uint32_t t = threadIdx.x >= st ? sh[threadIdx.x - st] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    const uint32_t incl = sh[threadIdx.x];
    if (idx < nb) block_sums[idx] = carry + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_out = carry;
}
// The same scan in ONE block for K <= SCAN_LDS_MAX (every proof-sized batch): the counts are staged in LDS with
// coalesced loads, thread t scans `per` consecutive LDS words, the 1024 partial sums are scanned with shuffles, and
// the offsets leave coalesced.  (A first single-block version that read global memory with per-thread strides took
// 21 us against 14 us for the three launches below.)
constexpr uint32_t SCAN_LDS_MAX = 32 * 1024;
static __global__ void __launch_bounds__(1024)
scan_lds_kernel(const uint32_t* __restrict__ gcounts, uint32_t* __restrict__ offsets, uint32_t K, uint32_t per) {
  extern __shared__ uint32_t buf[];            // K words (+1 spare per 32 to spread banks is not needed: per is odd)
  __shared__ uint32_t wave_sum[16];
  // a key's count is the sum over the groups (msm_group_fold_kernel's work, without its launch)
  for (uint32_t i = threadIdx.x; i < K; i += 1024) {
    uint32_t run = 0;
#pragma unroll
    for (uint32_t x = 0; x < MSM_XCDS; x++) run += gcounts[(size_t)x * K + i];
    buf[i] = run;
  }
  __syncthreads();
  const uint32_t lo = threadIdx.x * per, hi = min(K, lo + per);
  uint32_t s = 0;
  for (uint32_t i = lo; i < hi; i++) s += buf[i];
  // inclusive scan of s over the block: within the wave by shuffles, across the 16 waves through LDS
  uint32_t incl = s;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wave_sum[wave] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (uint32_t w = 0; w < wave; w++) base += wave_sum[w];
  uint32_t run = base + incl - s;
  for (uint32_t i = lo; i < hi; i++) {
    const uint32_t v = buf[i];
    buf[i] = run;
    run += v;
  }
  if (threadIdx.x == 1023) offsets[K] = run;   // the last thread's running total is the grand total
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < K; i += 1024) offsets[i] = buf[i];
}
// offsets[i] = exclusive prefix (offsets[K] = total)
static __global__ void __launch_bounds__(256)
scan_apply_kernel(const uint32_t* in, const uint32_t* block_sums, uint32_t* offsets, size_t K) {
  __shared__ uint32_t sh[256];
  const size_t base = (size_t)blockIdx.x * SCAN_BLOCK;
  uint32_t v[4], s = 0;
  for (uint32_t k = 0; k < 4; k++) {
    const size_t idx = base + threadIdx.x * 4 + k;
    v[k] = idx < K ? in[idx] : 0;
    s += v[k];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (uint32_t st = 1; st < 256; st <<= 1) {
    uint32_t t = threadIdx.x >= st ? sh[threadIdx.x - st] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t run = block_sums[blockIdx.x] + sh[threadIdx.x] - s;
  for (uint32_t k = 0; k < 4; k++) {
    const size_t idx = base + threadIdx.x * 4 + k;
    if (idx < K) {
      offsets[idx] = run;
    }
    run += v[k];
    if (idx + 1 == K) offsets[K] = run;
  }
}

// Entries per accumulate thread, decided ON THE DEVICE from the number of entries the sort actually produced
// (E = offsets[K]): the host sizes T for the worst case (every window of every scalar non-zero), but the columns of a
// real proof are mostly zeros (a Poseidon witness is ~60 rows of 65 536, fixed columns likewise): with the host's T a
// sparse launch put 64 dependent additions on each of a few hundred threads of an idle GPU (1.9 ms for keygen's 16
// sparse columns).  Every kernel that cuts the sorted list into chunks calls this with the same arguments.  Never
// above the host's T, so the chunk count stays within the arrays sized for it.
__device__ __forceinline__ uint32_t msm_effective_t(uint32_t E, uint32_t t_host) {
  uint32_t t = (E + MSM_CHUNK_WAVES * 65536u - 1) / (MSM_CHUNK_WAVES * 65536u);
  if (t < 8) t = 8;
  return t < t_host ? t : t_host;
}

// Degenerate columns (a permutation grand product that is 1 on almost every row, an all-ones selector) put tens of
// thousands of entries into one bucket.  The chunk kernel does not care -- every thread still adds T entries -- but
// the bucket then has thousands of pieces.  One thread per key: keys with more than MSM_HOT_SPAN pieces reserve
// ceil(span / MSM_HOT_SEG) slots and emit one task per slot.
// The same pass gives every chunk its first key: chunk_first[j] = key for the chunks that START inside the key's
// list (each chunk start lies in exactly one non-empty list, so every slot below ceil(E / T) gets one writer).
// It needs the scan's `offsets` only -- like the scatter, which does not need IT: every block of the staged scatter kernel
// first runs this pass for its share of the keys (MsmKeysArgs::keys_per_block: a few dozen keys on its first wave
// while the other waves start on the scatter), one launch and its ~9 us of stream time less per MSM.  (As blocks of
// their own in that launch -- in front or behind -- the keys blocks each took a CU's LDS and pushed eight scatter blocks
// into a second round: no gain.)  The other sorts launch msm_keys_kernel.  A block covers keys [key_lo, key_lo +
// key_count); `long_key`: blockDim.x + 1 words of LDS.  One block also stores the per-column table pointers of a
// multi-table launch.
constexpr uint32_t MSM_MAX_MULTI = 16;    // columns of a launch that may each bring their own bases
struct MsmTableList {
  const U128* t[MSM_MAX_MULTI];
};
struct MsmKeysArgs {
  uint32_t* chunk_first;
  uint32_t* hot_slot;
  uint32_t* tasks;          // 2 words each: key, segment
  uint32_t* task_count;
  uint32_t max_tasks, T_host;
  uint32_t keys_per_block;  // staged scatter: every block runs the keys pass for this many keys first (0: it does not)
  uint32_t n_tables;        // per-column tables to store (0: the launch has one table)
  const U128** tables_dst;
  MsmTableList tables;
};
__device__ __forceinline__ void msm_keys_block(const uint32_t* __restrict__ offsets, size_t K, const MsmKeysArgs& A, size_t key_lo,
                                               uint32_t key_count /* <= blockDim.x */, bool store_tables,
                                               uint32_t* long_key /* LDS: blockDim.x + 1 words */) {
  // long lists (degenerate columns: tens of thousands of chunks under one key) are filled by the whole block
  uint32_t* n_long = long_key + blockDim.x;
  const uint32_t T = msm_effective_t(offsets[K], A.T_host);
  if (threadIdx.x == 0) *n_long = 0;
  if (store_tables && threadIdx.x < A.n_tables) A.tables_dst[threadIdx.x] = A.tables.t[threadIdx.x];
  __syncthreads();
  const size_t key = key_lo + threadIdx.x;
  const bool live = threadIdx.x < key_count && key < K;
  const uint32_t s = live ? offsets[key] : 0, e = live ? offsets[key + 1] : 0;
  if (live) {
    const uint32_t j0 = (s + T - 1) / T;
    const uint64_t j1 = ((uint64_t)e + T - 1) / T;             // chunks j0 .. j1-1 start inside [s, e)
    if (j1 > (uint64_t)j0 + 64) long_key[atomicAdd(n_long, 1u)] = (uint32_t)key;
    else
      for (uint32_t j = j0; j < j1; j++) A.chunk_first[j] = (uint32_t)key;
  }
  __syncthreads();
  const uint32_t nl = *n_long;
  for (uint32_t q = 0; q < nl; q++) {
    const uint32_t lk = long_key[q];
    const uint32_t ls = offsets[lk], le = offsets[lk + 1];
    const uint64_t j1 = ((uint64_t)le + T - 1) / T;
    for (uint64_t j = (uint64_t)(ls + T - 1) / T + threadIdx.x; j < j1; j += blockDim.x) A.chunk_first[j] = lk;
  }
  if (!live) return;
  uint32_t slot = MSM_NOT_HOT;
  if (e > s) {
    const uint32_t span = (e - 1) / T - s / T + 1;
    if (span > MSM_HOT_SPAN) {
      const uint32_t nseg = (span + MSM_HOT_SEG - 1) / MSM_HOT_SEG;
      const uint32_t first = atomicAdd(A.task_count, nseg);
      if (first + nseg <= A.max_tasks) {     // cannot fail by construction (see msm_workspace); stay in bounds anyway
        slot = first;
        for (uint32_t q = 0; q < nseg; q++) {
          A.tasks[2 * (first + q)] = (uint32_t)key;
          A.tasks[2 * (first + q) + 1] = q;
        }
      }
    }
  }
  A.hot_slot[key] = slot;
}
constexpr uint32_t MSM_KEYS_THREADS = 256;
static __global__ void __launch_bounds__(MSM_KEYS_THREADS)
msm_keys_kernel(const uint32_t* __restrict__ offsets, size_t K, MsmKeysArgs A) {
  __shared__ uint32_t long_key[MSM_KEYS_THREADS + 1];
  msm_keys_block(offsets, K, A, (size_t)blockIdx.x * blockDim.x, blockDim.x, blockIdx.x == 0, long_key);
}

// ---- scatter: same tiling as the digits kernel, which already reserved the tile's range in every bucket's list:
// LDS cursors = list start + tile base, then one LDS atomic and one store per entry.
// sorted_ref[pos] = (w * n_bases + i) | sign.  Only ONE word per entry is written: every scattered 4-byte store
// costs a 64-byte sector at the memory side (measured: 8x write amplification), so the entry's key is not stored --
// the accumulate kernel recovers it from `offsets`.
template <class CV>
__global__ void __launch_bounds__(1024)
msm_scatter_kernel(const U128* __restrict__ scalars, const uint32_t* __restrict__ offsets,
                   const uint32_t* __restrict__ gcounts, const uint32_t* __restrict__ tile_base,
                   uint32_t* __restrict__ sorted_ref, uint32_t n, size_t col_stride /* elements */, uint32_t n_bases,
                   uint32_t tile, uint32_t tiles, uint32_t m, MsmGeom g) {
  using S = typename CV::Scalar;
  extern __shared__ uint32_t hist[];
  const MsmTileId id = msm_tile_id(tiles, m);
  if (!id.live) return;
  const uint32_t col = id.col;
  // LDS cursors: list start + the group's start inside the list + this tile's base inside the group's run (handed out
  // by the digits kernel's atomics)
  const uint32_t* tb = tile_base + ((size_t)col * tiles + id.tile) * g.B;
  const uint32_t* gc = gcounts + (size_t)col * g.B;
  const size_t K = (size_t)m * g.B;
  for (uint32_t b = threadIdx.x; b < g.B; b += blockDim.x) {
    uint32_t at = offsets[(size_t)col * g.B + b] + tb[b];
    for (uint32_t x = 0; x < id.group; x++) at += gc[x * K + b];     // the groups below this one come first in the list
    hist[b] = at;
  }
  __syncthreads();
  const uint32_t lo = id.tile * tile, hi = min(lo + tile, n);
  // the digits are recomputed rather than stored by the first kernel: 32 bytes of scalar instead of 4 W bytes of digits
  for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    MsmDigits dg(fe_from_mont(fe_load<S>(scalars + 2 * (col_stride * col + i))).v);
    for (uint32_t w = 0; w < g.W; w++) {
      const uint32_t enc = dg.next(g, w);
      if (enc) {
        const uint32_t pos = atomicAdd(&hist[(enc & ~MSM_SIGN) - 1], 1u);
        sorted_ref[pos] = (w * n_bases + i) | (enc & MSM_SIGN);
      }
    }
  }
}

// The same scatter with the tile's entries STAGED in LDS in bucket order and written out in that order: lane j of a
// store instruction holds staged entry j, so the ~4 entries of one (tile, bucket) run -- consecutive in the bucket's
// list -- sit in neighbouring lanes and leave as one request instead of four.  (A scattered 4-byte store is one 32-byte
// fabric write on this chip whatever the L2 holds: 152 MB of WRITE_SIZE for 21 MB of entries with the direct kernel.)
// LDS: cursors and (global - local) offsets per bucket, 4 + 2 bytes per staged entry; the host picks this kernel when
// a tile with a few entries per bucket fits (msm_workspace) and the direct one otherwise (wide windows, long columns).
template <class CV>
__global__ void __launch_bounds__(1024)
msm_scatter_staged_kernel(const U128* __restrict__ scalars, const uint32_t* __restrict__ offsets,
                          const uint32_t* __restrict__ gcounts, const uint32_t* __restrict__ tile_base,
                          const uint32_t* __restrict__ tile_hist, uint32_t* __restrict__ sorted_ref, uint32_t n,
                          size_t col_stride /* elements */, uint32_t n_bases, uint32_t tile, uint32_t tiles, uint32_t m,
                          MsmGeom g, uint32_t stage_cap /* entries the staging area holds = tile * W */, MsmKeysArgs keys) {
  using S = typename CV::Scalar;
  extern __shared__ uint32_t hist[];
  __shared__ uint32_t wave_sum[16];
  __shared__ uint32_t total_s;
  uint32_t* cur = hist;                                   // B cursors into the staging area
  uint32_t* delta = hist + g.B;                           // B: (position in sorted_ref) - (position in the staging area)
  uint32_t* sref = hist + 2 * (size_t)g.B;                // stage_cap entries
  uint16_t* sbkt = reinterpret_cast<uint16_t*>(sref + stage_cap);   // their buckets
  const MsmTileId id = msm_tile_id(tiles, m);
  if (!id.live) return;
  const uint32_t col = id.col;
  if (keys.keys_per_block) {                              // this block's share of the keys pass (msm_keys_block)
    const uint32_t v = col * tiles + id.tile;
    msm_keys_block(offsets, (size_t)m * g.B, keys, (size_t)v * keys.keys_per_block, keys.keys_per_block, v == 0, hist);
    __syncthreads();                                      // the pass is done with the LDS the scatter uses from here on
  }
  const uint32_t* tb = tile_base + ((size_t)col * tiles + id.tile) * g.B;
  const uint32_t* th = tile_hist + ((size_t)col * tiles + id.tile) * g.B;
  const uint32_t* gc = gcounts + (size_t)col * g.B;
  const size_t K = (size_t)m * g.B;
  // exclusive scan of the tile's counts over the buckets: thread t owns `per` consecutive buckets
  const uint32_t per = (g.B + blockDim.x - 1) / blockDim.x;
  const uint32_t b_lo = min(g.B, threadIdx.x * per), b_hi = min(g.B, b_lo + per);
  uint32_t s = 0;
  for (uint32_t b = b_lo; b < b_hi; b++) s += th[b];
  uint32_t incl = s;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wave_sum[wave] = incl;
  __syncthreads();
  uint32_t run = incl - s;
  for (uint32_t w = 0; w < wave; w++) run += wave_sum[w];
  for (uint32_t b = b_lo; b < b_hi; b++) {
    uint32_t at = offsets[(size_t)col * g.B + b] + tb[b];
    for (uint32_t x = 0; x < id.group; x++) at += gc[x * K + b];     // the groups below this one come first in the list
    cur[b] = run;
    delta[b] = at - run;                                              // modulo 2^32
    run += th[b];
  }
  if (threadIdx.x == blockDim.x - 1) total_s = run;
  __syncthreads();
  const uint32_t lo = id.tile * tile, hi = min(lo + tile, n);
  for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    MsmDigits dg(fe_from_mont(fe_load<S>(scalars + 2 * (col_stride * col + i))).v);
    for (uint32_t w = 0; w < g.W; w++) {
      const uint32_t enc = dg.next(g, w);
      if (enc) {
        const uint32_t b = (enc & ~MSM_SIGN) - 1;
        const uint32_t pos = atomicAdd(&cur[b], 1u);
        if (pos < stage_cap) {             // always: the counts come from the same digits (msm_digits_kernel)
          sref[pos] = (w * n_bases + i) | (enc & MSM_SIGN);
          sbkt[pos] = (uint16_t)b;
        }
      }
    }
  }
  __syncthreads();
  const uint32_t total = min(total_s, stage_cap);
  for (uint32_t j = threadIdx.x; j < total; j += blockDim.x) sorted_ref[delta[sbkt[j]] + j] = sref[j];
}

// ---- accumulate: every thread adds T consecutive sorted entries ---------------------------------
template <class CV>
__device__ __forceinline__ Affine29<CV> msm_fetch(const U128* __restrict__ table, uint32_t entry) {
  return affine29_load<CV>(table + 4 * (size_t)(entry & ~MSM_SIGN), (entry & MSM_SIGN) != 0);
}

// Chunk t covers sorted entries [t*T, min((t+1)*T, E)), E = offsets[K] read on the device.  A run (maximal stretch of one key inside the chunk) that holds the key's
// whole list goes to bucket_sum[key]; a cut-off first run goes to head[t], a cut-off last run to tail[t].
// `col_tables` (optional): one table per column -- columns of ONE launch may commit against different bases of the same
// length (a proof's permutation products over g_lagrange and its random polynomial over g share a launch); the
// column of a key is key >> log_b.  Null: every column uses `table`.
template <class CV>
__global__ void __launch_bounds__(256, MSM_CHUNK_WAVES)
msm_chunk_kernel(const U128* __restrict__ table, const U128* const* __restrict__ col_tables, uint32_t log_b,
                 const uint32_t* __restrict__ sorted_ref,
                 const uint32_t* __restrict__ chunk_first, const uint32_t* __restrict__ offsets, size_t K, uint32_t T_host,
                 uint32_t* __restrict__ bucket_sum, uint32_t* __restrict__ head, uint32_t* __restrict__ tail) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t E = offsets[K];
  const uint32_t T = msm_effective_t(E, T_host);
  const uint64_t lo64 = (uint64_t)t * T;
  if (lo64 >= E) return;
  const uint32_t lo = (uint32_t)lo64, hi = (uint32_t)min((uint64_t)E, lo64 + T);
  uint32_t key = chunk_first[t];
  uint32_t next = offsets[key + 1];          // first entry of the following list
  if (col_tables) table = col_tables[key >> log_b];
  bool first = true;
  Xyzz29<CV> a = Xyzz29<CV>::identity();
  U128 quad = U128{0, 0, 0, 0};
  for (uint32_t e = lo; e < hi; e++) {
    // refs are fetched as aligned 16-byte vectors whatever T is (the first one may start below lo, the last one may
    // reach up to 12 bytes past E: the array is allocated with that slack)
    if (e == lo || (e & 3u) == 0) quad = *reinterpret_cast<const U128*>(sorted_ref + (e & ~3u));
    const uint32_t sel = e & 3u;
    const uint32_t ref = sel == 0 ? quad.x : sel == 1 ? quad.y : sel == 2 ? quad.z : quad.w;
    if (e >= next) {
      // the run of `key` ended inside the chunk
      if (first && offsets[key] != lo) xyzz29_store<CV>(head + XYZZ29_WORDS * (size_t)t, a);
      else xyzz29_store<CV>(bucket_sum + XYZZ29_WORDS * (size_t)key, a);
      first = false;
      a = Xyzz29<CV>::identity();
      do {                                  // skip empty lists
        key++;
        next = offsets[key + 1];
      } while (e >= next);
      if (col_tables) table = col_tables[key >> log_b];
    }
    a = xyzz29_add_affine(a, msm_fetch<CV>(table, ref));
  }
  const bool ends_here = next == hi;
  const bool starts_here = !first || offsets[key] == lo;
  if (starts_here && ends_here) xyzz29_store<CV>(bucket_sum + XYZZ29_WORDS * (size_t)key, a);
  else if (first) xyzz29_store<CV>(head + XYZZ29_WORDS * (size_t)t, a);   // one run spanning the whole chunk, or a cut first run
  else xyzz29_store<CV>(tail + XYZZ29_WORDS * (size_t)t, a);
}

// shuffle an XYZZ point down by `delta` lanes
template <class CV>
__device__ __forceinline__ Xyzz29<CV> xyzz_shfl_down(const Xyzz29<CV>& p, uint32_t delta) {
  Xyzz29<CV> r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    r.x.v[i] = __shfl_down(p.x.v[i], delta, 64);
    r.y.v[i] = __shfl_down(p.y.v[i], delta, 64);
    r.zz.v[i] = __shfl_down(p.zz.v[i], delta, 64);
    r.zzz.v[i] = __shfl_down(p.zzz.v[i], delta, 64);
  }
  return r;
}

// Hand-off of a block's result to the block that arrives last at a counter (msm_final_kernel, msm_small_kernel), in the
// form MI355X_MICROARCH.md prescribes for data written by one XCD and read on another: the storing lane waits for its
// stores, releases at agent scope, waits again (ROCm 7.2 can drop the wait behind the write-back when it believes
// nothing is outstanding -- the count could overtake the data) and only then adds to the counter; the block that
// sees the last count acquires before it loads.
__device__ __forceinline__ void h2_publish_release() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void h2_consume_acquire() {
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// One level of a shuffle tree over quads: the lanes below `d` (of a group whose lane index is `lane`) add the point held
// `d` lanes further up.  The other lanes keep their value: their partner lies outside the group (a lane past the end of
// the wave reads ITSELF, and adding a point to itself sends the whole wave through the doubling path as well -- 6.2 us
// per level instead of 3.6, measured with tools/microbench_tail.hip).
template <class CV>
__device__ __forceinline__ Xyzz29<CV> xyzz_fold_down(const Xyzz29<CV>& a, uint32_t d, uint32_t lane) {
  const Xyzz29<CV> o = xyzz_shfl_down(a, d);
  return lane < d ? xyzz29_add_quad(a, o) : a;
}

// piece p of a key whose list starts at entry s and spans chunks j0..: p = 0 is chunk j0's tail (or head when the
// list starts exactly at the chunk), p >= 1 is the head of chunk j0 + p
template <class CV>
__device__ __forceinline__ Xyzz29<CV> msm_piece(const uint32_t* __restrict__ head, const uint32_t* __restrict__ tail,
                                               uint32_t s, uint32_t j0, uint32_t T, uint32_t p) {
  const uint32_t j = j0 + p;
  const uint32_t* src = (p == 0 && s != j0 * T) ? tail : head;
  return xyzz29_load<CV>(src + XYZZ29_WORDS * (size_t)j);
}

// one wave per task = 16 quads (4 lanes per point, h2_curve_quad.hpp): quad g folds pieces g, g+16, ... of the
// task's MSM_HOT_SEG pieces, then a 4-level shuffle tree across the quads
template <class CV>
__global__ void __launch_bounds__(64)
msm_hot_reduce_kernel(const uint32_t* __restrict__ offsets, size_t K, uint32_t T_host, const uint32_t* __restrict__ hot_slot,
                      const uint32_t* __restrict__ tasks, const uint32_t* __restrict__ task_count, uint32_t max_tasks,
                      const uint32_t* __restrict__ head, const uint32_t* __restrict__ tail, uint32_t* __restrict__ hot_part) {
  __builtin_amdgcn_s_setprio(3);   // a dependent chain on a mostly idle SIMD: issue ahead of co-resident throughput kernels
  const uint32_t ntask = min(*task_count, max_tasks);
  const uint32_t T = msm_effective_t(offsets[K], T_host);
  const uint32_t quad = threadIdx.x >> 2;
  for (uint32_t t = blockIdx.x; t < ntask; t += gridDim.x) {
    const uint32_t key = tasks[2 * t], q = tasks[2 * t + 1];
    const uint32_t s = offsets[key], e = offsets[key + 1];
    const uint32_t j0 = s / T, span = (e - 1) / T - j0 + 1;
    const uint32_t lo = q * MSM_HOT_SEG, hi = min(span, lo + MSM_HOT_SEG);
    Xyzz29<CV> a = Xyzz29<CV>::identity();
    for (uint32_t p = lo + quad; p < hi; p += 16) a = xyzz29_add_quad(a, msm_piece<CV>(head, tail, s, j0, T, p));
    for (uint32_t d = 32; d >= 4; d >>= 1) a = xyzz_fold_down(a, d, threadIdx.x);
    if (threadIdx.x == 0) xyzz29_store<CV>(hot_part + XYZZ29_WORDS * (size_t)(hot_slot[key] + q), a);
  }
}

// Fix-up: G lanes = G/4 quads per key (G = 2^log_g, 4 <= G <= 64; 4 lanes per point, h2_curve_quad.hpp).  The
// quads of a key share out its pieces (or its hot partials), then a shuffle tree across the quads; lane 0 writes
// xsum[key], the bucket's point sum.
template <class CV>
__global__ void __launch_bounds__(256)
msm_fixup_kernel(const uint32_t* __restrict__ offsets, size_t K, uint32_t T_host, uint32_t log_g,
                 const uint32_t* __restrict__ bucket_sum, const uint32_t* __restrict__ head,
                 const uint32_t* __restrict__ tail, const uint32_t* __restrict__ hot_slot,
                 const uint32_t* __restrict__ hot_part, uint32_t* __restrict__ xsum) {
  __builtin_amdgcn_s_setprio(3);   // a dependent chain on a mostly idle SIMD: issue ahead of co-resident throughput kernels
  const uint32_t G = 1u << log_g;
  const uint32_t T = msm_effective_t(offsets[K], T_host);
  const size_t gt = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t key = gt >> log_g;
  const uint32_t lane = (uint32_t)gt & (G - 1);
  const uint32_t quad = lane >> 2, nquad = G >> 2;
  // all lanes of a wave stay in the shuffle tree together; out-of-range keys work on identities
  const bool live = key < K;
  Xyzz29<CV> x = Xyzz29<CV>::identity();
  if (live) {
    const uint32_t s = offsets[key], e = offsets[key + 1];
    if (e > s) {
      const uint32_t j0 = s / T, j1 = (e - 1) / T;
      if (j0 == j1) {
        if (quad == 0) x = xyzz29_load<CV>(bucket_sum + XYZZ29_WORDS * key);
      } else if (hot_slot[key] != MSM_NOT_HOT) {
        const uint32_t nseg = (j1 - j0 + 1 + MSM_HOT_SEG - 1) / MSM_HOT_SEG;
        const uint32_t* part = hot_part + XYZZ29_WORDS * (size_t)hot_slot[key];
        for (uint32_t q = quad; q < nseg; q += nquad) x = xyzz29_add_quad(x, xyzz29_load<CV>(part + XYZZ29_WORDS * (size_t)q));
      } else if (quad <= j1 - j0) {
        // the next piece is on its way while the current one is added (a piece is 144 bytes from HBM: ~1.5 us exposed
        // per addition otherwise, the chain has nothing else to do)
        Xyzz29<CV> nxt = msm_piece<CV>(head, tail, s, j0, T, quad);
        for (uint32_t p = quad; p <= j1 - j0; p += nquad) {
          const Xyzz29<CV> cur = nxt;
          if (p + nquad <= j1 - j0) nxt = msm_piece<CV>(head, tail, s, j0, T, p + nquad);
          x = xyzz29_add_quad(x, cur);
        }
      }
    }
  }
  for (uint32_t d = G >> 1; d >= 4; d >>= 1) x = xyzz_fold_down(x, d, lane);
  if (live && lane == 0) xyzz29_store<CV>(xsum + XYZZ29_WORDS * key, x);
}

// ---- bucket weights without multiplying every bucket --------------------------------------------------------------
// A column's result is sum_b (b + 1) * x_b over its B = 2^log_b buckets.  Write b = hi * 2^lb + lo and let
//     R_hi = sum_lo x_b   (a row of 2^lb consecutive buckets),      C_lo = sum_hi x_b   (a column, stride 2^lb);
// then   sum_b (b + 1) x_b  =  2^lb * sum_hi hi * R_hi  +  sum_lo (lo + 1) * C_lo      (sum_lo C_lo = sum_b x_b).
// So the B buckets only take part in two plain sums each (msm_rowcol_kernel, 2 B additions per column), and the
// double-and-add multiplications are left for the 2^hb + 2^lb row / column sums, with multipliers of hb resp. lb bits
// (msm_final_kernel).  Before: one 12-bit double-and-add per bucket (12 doublings + ~6 additions, 75 us per launch at
// the proof shape -- VALU-bound, 16 384 quads on 1024 SIMDs) and a two-level tree behind it (2 x 42 us).
//
// One block of 1, 2 or 4 waves per row or column sum: the quads (4 lanes per point, h2_curve_quad.hpp) fold the points
// quad, quad + nq, ..., a 4-level shuffle tree joins a wave's 16 quads, the waves' sums meet in LDS.  The factor 2^lb of
// the row family is applied HERE, by lb doublings of every row sum (and of the last column's sum, whose multiplier
// cols = 2^lb * 1 is the one that does not fit lb bits): the row waves have fewer points to add than the column waves
// when hb > lb, and a doubling done by 2^hb waves side by side is off the final kernel's one chain.
// The last column's multiplier is cols = 2^lb, one bit more than the column family's digits hold, so it rides in the row
// family (item 0, whose own multiplier would be 0).  If 2^lb fits the ROW family's base-4 digits (hb odd, or hb > lb) it
// is simply that item's multiplier; otherwise msm_rowcol_kernel doubles that sum lb times like a row and the multiplier
// is 1.
H2_HD bool msm_special_predoubled(uint32_t log_b, uint32_t lb) {
  const uint32_t hb = log_b - lb, ndig = ((hb + 1) >> 1) ? ((hb + 1) >> 1) : 1u;
  return lb >= 2 * ndig;
}
template <class CV>
__global__ void __launch_bounds__(256)
msm_rowcol_kernel(const uint32_t* __restrict__ xsum, uint32_t* __restrict__ rc, uint32_t* __restrict__ done,
                  uint32_t log_b, uint32_t lb, U128* zero_ptr = nullptr, uint32_t zero_u128 = 0) {
  __builtin_amdgcn_s_setprio(3);   // a dependent chain on a mostly idle SIMD: issue ahead of co-resident throughput kernels
  __shared__ uint32_t xw[3][XYZZ29_WORDS];
  const uint32_t col = blockIdx.y;
  if (blockIdx.x == 0 && threadIdx.x == 0) done[col] = 0;       // msm_final_kernel's arrival counter
  // the launch sequence's zeroed region (misc + the sort's counters: every kernel that used it has finished) is left
  // ZERO again for the next launch sequence on this workspace, which then needs no memset of its own (a fill kernel
  // and its gaps: ~12 us of stream time in front of every MSM)
  if (zero_ptr) {
    const uint32_t nthr = gridDim.x * gridDim.y * blockDim.x;
    for (uint32_t i = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < zero_u128; i += nthr)
      zero_ptr[i] = U128{0, 0, 0, 0};
  }
  const uint32_t rows = 1u << (log_b - lb), cols = 1u << lb;
  const bool is_row = blockIdx.x < rows;
  const uint32_t first = is_row ? blockIdx.x * cols : blockIdx.x - rows;
  const uint32_t stride = is_row ? 1u : cols, len = is_row ? cols : rows;
  const uint32_t* base = xsum + XYZZ29_WORDS * ((size_t)col << log_b);
  const uint32_t quad = threadIdx.x >> 2, nq = blockDim.x >> 2, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  // the first point is taken as it is: an addition onto the identity costs as much as any other
  Xyzz29<CV> a = quad < len ? xyzz29_load<CV>(base + XYZZ29_WORDS * (size_t)(first + quad * stride)) : Xyzz29<CV>::identity();
  if (quad + nq < len) {
    // the next point is on its way while the current one is added (144 bytes from L2 / HBM: the chain has nothing else to do)
    Xyzz29<CV> nxt = xyzz29_load<CV>(base + XYZZ29_WORDS * (size_t)(first + (quad + nq) * stride));
    for (uint32_t j = quad + nq; j < len; j += nq) {
      const Xyzz29<CV> cur = nxt;
      if (j + nq < len) nxt = xyzz29_load<CV>(base + XYZZ29_WORDS * (size_t)(first + (j + nq) * stride));
      a = xyzz29_add_quad(a, cur);
    }
  }
  for (uint32_t d = 32; d >= 4; d >>= 1) a = xyzz_fold_down(a, d, lane);
  if (blockDim.x > 64) {
    if (wave > 0 && lane == 0) xyzz29_store<CV>(xw[wave - 1], a);
    __syncthreads();
    if (wave > 0) return;
    const uint32_t nw = blockDim.x >> 6;
    if (quad >= 1) a = quad < nw ? xyzz29_load<CV>(xw[quad - 1]) : Xyzz29<CV>::identity();
    for (uint32_t d = 2 * nw; d >= 4; d >>= 1) a = xyzz_fold_down(a, d, lane);
  }
  if (is_row || (blockIdx.x == rows + cols - 1 && msm_special_predoubled(log_b, lb)))
    for (uint32_t k = 0; k < lb; k++) a = xyzz29_double_quad(a);
  if (threadIdx.x == 0) xyzz29_store<CV>(rc + XYZZ29_WORDS * ((size_t)col * (rows + cols) + blockIdx.x), a);
}
// waves per row / column sum: as many as shorten the chains (a wave wants >= 2 points per quad) while the launch stays
// at about one wave per SIMD -- two of these chains on one SIMD run at half speed each
inline uint32_t msm_rowcol_waves(uint32_t log_b, uint32_t lb, size_t m) {
  const uint32_t rows = 1u << (log_b - lb), cols = 1u << lb;
  const uint32_t shortest = rows < cols ? rows : cols;
  uint32_t nw = 1;
  while (nw < 4 && shortest >= 64 * nw && (size_t)(rows + cols) * m * (2 * nw) <= 1024) nw *= 2;
  return nw;
}

// word-wise choice among three points by a per-quad digit (0 -> the identity, which is all zeros)
template <class CV>
__device__ __forceinline__ Xyzz29<CV> xyzz29_pick(uint32_t dig, const Xyzz29<CV>& p1, const Xyzz29<CV>& p2, const Xyzz29<CV>& p3) {
  const int32_t m1 = -(int32_t)(dig == 1), m2 = -(int32_t)(dig == 2), m3 = -(int32_t)(dig == 3);
  Xyzz29<CV> r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    r.x.v[i] = (m1 & p1.x.v[i]) | (m2 & p2.x.v[i]) | (m3 & p3.x.v[i]);
    r.y.v[i] = (m1 & p1.y.v[i]) | (m2 & p2.y.v[i]) | (m3 & p3.y.v[i]);
    r.zz.v[i] = (m1 & p1.zz.v[i]) | (m2 & p2.zz.v[i]) | (m3 & p3.zz.v[i]);
    r.zzz.v[i] = (m1 & p1.zzz.v[i]) | (m2 & p2.zzz.v[i]) | (m3 & p3.zzz.v[i]);
  }
  return r;
}

// One-wave blocks, each on a SIMD of its own (two waves sharing a SIMD run these chains at half speed: the
// 4-lanes-per-point arithmetic keeps a SIMD's issue port busy with ONE wave), ONE item per quad, so the chain is as
// long as one multiplication whatever the bucket count (with a fixed eight blocks a 2^20-term MSM -- 256 + 127 items --
// went round four times: 205 us):
//     row family     blocks [0, ceil(rows / 16)):   item 0 = the last column's sum, whose multiplier cols = 2^lb does not
//                    fit the column family's lb bits (the rows' own item 0 would have multiplier 0): multiplier 2^lb
//                    where the row digits hold it, else 1 on a sum msm_rowcol_kernel has doubled lb times
//                    (msm_special_predoubled),
//                    item hi = 2^lb R_hi with multiplier hi                                    (< 2^hb)
//     column family  the blocks behind them:        item lo = C_lo with multiplier lo + 1, lo < cols - 1   (< 2^lb)
// multiplied two bits at a time (x, 2x, 3x, then per digit two doublings and one addition of the quad's own choice),
// a shuffle tree over the wave's 16 quads, and the block that arrives last (a counter per column, zeroed by
// msm_rowcol_kernel) adds the blocks' partials (one or more per quad, then a tree over the quads that hold one) and writes the column's MSM: XYZZ on the
// working form to out[col] and, when out_jac is given, the Jacobian point in the API's form.
//
// The whole chain is ONE loop around one doubling and one addition (a little program counter decides what each step
// does): written as straight-line code the kernel was 180 KB -- every inlined addition is ~24 KB -- and each copy ran
// exactly once, from a cold instruction cache (4.9 us against 3.6 us for an addition; tools/microbench_tail.hip).
constexpr uint32_t MSM_FINAL_MAX_BLOCKS = 64;      // 2^18 buckets: 512 row items + 511 column items, 16 per block
H2_HD uint32_t msm_final_row_blocks(uint32_t log_b, uint32_t lb) { return ((1u << (log_b - lb)) + 15u) / 16u; }
H2_HD uint32_t msm_final_blocks(uint32_t log_b, uint32_t lb) {
  return msm_final_row_blocks(log_b, lb) + ((1u << lb) - 1u + 15u) / 16u;
}
template <class CV>
__global__ void __launch_bounds__(64)
msm_final_kernel(const uint32_t* __restrict__ rc, uint32_t* part /* m x MSM_FINAL_MAX_BLOCKS points */, uint32_t* done,
                 uint32_t* __restrict__ out, U128* __restrict__ out_jac, uint32_t log_b, uint32_t lb) {
  __builtin_amdgcn_s_setprio(3);
  using P = Xyzz29<CV>;
  const uint32_t col = blockIdx.y;
  const uint32_t hb = log_b - lb, rows = 1u << hb, cols = 1u << lb;
  const uint32_t nb_row = msm_final_row_blocks(log_b, lb), nb = gridDim.x;
  const uint32_t fam = blockIdx.x >= nb_row ? 1u : 0u, lane = threadIdx.x, quad = threadIdx.x >> 2;
  const uint32_t cnt = fam ? cols - 1 : rows;
  const uint32_t ndig = max(1u, ((fam ? lb : hb) + 1) >> 1);     // base-4 digits of the family's multipliers
  const uint32_t* src = rc + XYZZ29_WORDS * ((size_t)col * (rows + cols));
  const uint32_t i = (fam ? blockIdx.x - nb_row : blockIdx.x) * 16 + quad;   // this quad's item
  const uint32_t n_weigh = 3 * ndig - 1;                         // double, add x, (double, double, add)*
  // the last block: quad q takes partials q, q + 16, ... (n_more further ones), then a tree over the quads that hold
  // one -- three levels for the six blocks of a 2048-bucket column, not the full four
  const uint32_t n_more = nb > 16 ? (nb - 1) / 16 : 0;
  uint32_t lv2 = 0;
  while ((1u << lv2) < min(nb, 16u)) lv2++;
  const uint32_t n_tree = n_weigh + 4, n_all = n_tree + n_more + lv2;
  P x = P::identity(), x2 = P::identity(), x3 = P::identity();
  uint32_t k = 0;
  if (i < cnt) {
    const uint32_t idx = fam ? rows + i : (i == 0 ? rows + cols - 1 : i);
    x = xyzz29_load<CV>(src + XYZZ29_WORDS * (size_t)idx);
    k = fam ? i + 1 : (i == 0 ? (msm_special_predoubled(log_b, lb) ? 1u : cols) : i);
  }
  P r = x;
  H2_STAMP(0);
#pragma nounroll
  for (uint32_t pc = 0; pc < n_all; pc++) {
    bool is_double = false, wanted = true;
    P o = P::identity();
    if (pc == 0) {
      is_double = true;                                          // 2x
    } else if (pc == 1) {
      o = x;                                                     // 3x
    } else if (pc < n_weigh) {
      const uint32_t t = pc - 2, d = ndig - 2 - t / 3;
      if (t % 3 < 2) is_double = true;
      else o = xyzz29_pick((k >> (2 * d)) & 3u, x, x2, x3);
    } else if (pc < n_tree) {
      const uint32_t d = 32u >> (pc - n_weigh);
      o = xyzz_shfl_down(r, d);
      wanted = lane < d;
    } else if (pc < n_tree + n_more) {
      // the last block: every quad holds partial `quad` and now adds partials `quad + 16`, `quad + 32`, ...
      const uint32_t j = quad + 16 * (pc - n_tree + 1);
      if (j < nb) o = xyzz29_load<CV>(part + XYZZ29_WORDS * ((size_t)col * MSM_FINAL_MAX_BLOCKS + j));
    } else {
      const uint32_t d = (2u << lv2) >> (pc - n_tree - n_more);       // 4 * 2^(lv2 - 1), ..., 8, 4 lanes
      o = xyzz_shfl_down(r, d);
      wanted = lane < d;
    }
    if (is_double) r = xyzz29_double_quad(r);
    else if (wanted) r = xyzz29_add_quad(r, o);
    if (pc == 0) x2 = r;
    else if (pc == 1) {
      x3 = r;
      r = xyzz29_pick((k >> (2 * (ndig - 1))) & 3u, x, x2, x3);
    }
    if (pc + 1 == n_tree) {
      // this wave's partial is complete: publish it, and only the block that arrives last goes on
      H2_STAMP(2);
      uint32_t arrived = 0;
      if (threadIdx.x == 0) {
        xyzz29_store<CV>(part + XYZZ29_WORDS * ((size_t)col * MSM_FINAL_MAX_BLOCKS + blockIdx.x), r);
        h2_publish_release();                            // the partial is visible device-wide before the count
        arrived = atomicAdd(done + col, 1u);
      }
      arrived = __shfl(arrived, 0, 64);
      H2_STAMP(3);
      if (arrived != nb - 1) return;
      h2_consume_acquire();
      r = P::identity();
      if (quad < nb) r = xyzz29_load<CV>(part + XYZZ29_WORDS * ((size_t)col * MSM_FINAL_MAX_BLOCKS + quad));
      H2_STAMP(4);
    }
  }
  H2_STAMP(7);
  if (threadIdx.x == 0) xyzz29_store<CV>(out + XYZZ29_WORDS * (size_t)col, r);
  if (out_jac && threadIdx.x < 4) {
    // X zz, Y zzz, zz on lanes 0, 1, 2 at once, each lane converting and storing its own coordinate
    using F = Fe29<typename CV::Base>;
    using B = typename CV::Base;
    const uint32_t q = threadIdx.x & 3u;
    const F one = fe29_one<CV>();
    const F prod = fe29_mul(quad_select(q, r.x, r.y, r.zz, r.zz), quad_select(q, r.zz, r.zzz, one, one));
    Fe<B> v = fe29_to_api(prod);
    if (r.is_identity()) v = Fe<B>::zero();
    if (q < 3) fe_store<B>(out_jac + 6 * (size_t)col + 2 * q, v);
  }
  H2_STAMP(8);
}

// ---- finish ---------------------------------------------------------------------------------------------
// (the Jacobian result in the API's form is written by msm_final_kernel itself)
// XYZZ -> affine (m points), for h2_msm_batch's normalised output
template <class CV>
__global__ void msm_to_affine_kernel(const uint32_t* __restrict__ in, U128* __restrict__ out_aff, uint32_t m) {
  const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= m) return;
  using B = typename CV::Base;
  Affine<CV> a = xyzz_to_affine(xyzz29_to_api(xyzz29_load<CV>(in + XYZZ29_WORDS * (size_t)col)));
  fe_store<B>(out_aff + 4 * (size_t)col, a.x);
  fe_store<B>(out_aff + 4 * (size_t)col + 2, a.y);
}

// out[j] = sum_{g < groups} in[g * count + j], Jacobian in and out (API form): the partial sums of an MSM split by
// point range over several GPUs are added here after the all-gather.
// One wave per output point: lane g converts partial g (lanes beyond `groups` take further partials in turn), then a
// shuffle tree -- log2(64) additions deep instead of `groups` (a one-thread loop over 8 partials took ~0.1 ms).
template <class CV>
__global__ void __launch_bounds__(64)
points_sum_kernel(const U128* __restrict__ in_jac, U128* __restrict__ out_jac, uint32_t groups, uint32_t count) {
  const uint32_t j = blockIdx.x;
  if (j >= count) return;
  using B = typename CV::Base;
  Xyzz<CV> acc = Xyzz<CV>::identity();
  for (uint32_t g = threadIdx.x; g < groups; g += 64) {
    const U128* p = in_jac + 6 * ((size_t)g * count + j);
    const Fe<B> x = fe_load<B>(p), y = fe_load<B>(p + 2), z = fe_load<B>(p + 4);
    if (z.is_zero()) continue;
    const Fe<B> zz = fe_sqr(z);
    acc = xyzz_add(acc, Xyzz<CV>{x, y, zz, fe_mul(zz, z)});
  }
  uint32_t span = 1;
  while (span < groups && span < 64) span <<= 1;
  for (uint32_t d = span >> 1; d >= 1; d >>= 1) {
    Xyzz<CV> o;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      o.x.v[i] = __shfl_down(acc.x.v[i], d, 64);
      o.y.v[i] = __shfl_down(acc.y.v[i], d, 64);
      o.zz.v[i] = __shfl_down(acc.zz.v[i], d, 64);
      o.zzz.v[i] = __shfl_down(acc.zzz.v[i], d, 64);
    }
    acc = xyzz_add(acc, o);
  }
  if (threadIdx.x == 0) {
    Fe<B> x, y, z;
    xyzz_to_jacobian(acc, x, y, z);
    fe_store<B>(out_jac + 6 * (size_t)j, x);
    fe_store<B>(out_jac + 6 * (size_t)j + 2, y);
    fe_store<B>(out_jac + 6 * (size_t)j + 4, z);
  }
}

// ---- a few dozen terms against points that are not an SRS (the verifier's two combinations of a proof's commitments):
// no table, no sort -- one QUAD per term runs a 255-step double-and-add on the 4-lanes-per-point arithmetic (1.5 ms,
// whatever m <= a few thousand is), the wave's 16 quads fold with shuffles, and the block that arrives last adds the
// blocks' partials and writes the Jacobian result; grid.y = independent MSMs side by side (they are latency chains on a
// few waves: two cost what one costs).  Registering such a vector as bases cost 15 ms for 30 points.  `counter`: zero on entry.
constexpr uint32_t MSM_SMALL_MAX = 4;     // independent small MSMs per launch (grid.y)
struct MsmSmallBatch {
  const U128* points[MSM_SMALL_MAX];      // m affine points each, API form
  const U128* scalars[MSM_SMALL_MAX];     // m scalars each, Montgomery form
  uint32_t m[MSM_SMALL_MAX];
};
template <class CV>
__global__ void __launch_bounds__(64)
msm_small_kernel(MsmSmallBatch J, uint32_t* part /* gridDim.y x gridDim.x points */, uint32_t* counter /* gridDim.y, zero */,
                 U128* __restrict__ out_jac /* gridDim.y Jacobian points */) {
  using B = typename CV::Base;
  using S = typename CV::Scalar;
  const uint32_t job = blockIdx.y, m = J.m[job];
  const U128* points = J.points[job];
  const U128* scalars = J.scalars[job];
  part += XYZZ29_WORDS * (size_t)job * gridDim.x;
  const uint32_t quad = threadIdx.x >> 2, t = blockIdx.x * 16 + quad;
  Xyzz29<CV> r = Xyzz29<CV>::identity();
  if (t < m) {
    const Fe<B> x = fe_load<B>(points + 4 * (size_t)t), y = fe_load<B>(points + 4 * (size_t)t + 2);
    const Fe<S> k = fe_from_mont(fe_load<S>(scalars + 2 * (size_t)t));
    if (!(x.is_zero() && y.is_zero())) {
      const Xyzz29<CV> p = xyzz29_from_affine(Affine29<CV>{fe29_from_api(x), fe29_from_api(y)});
      for (int bit = 255; bit >= 0; bit--) {
        r = xyzz29_double_quad(r);
        if ((k.v[bit >> 5] >> (bit & 31)) & 1) r = xyzz29_add_quad(r, p);
      }
    }
  }
  for (uint32_t d = 32; d >= 4; d >>= 1) r = xyzz_fold_down(r, d, threadIdx.x);
  uint32_t arrived = 0;
  if (threadIdx.x == 0) {
    xyzz29_store<CV>(part + XYZZ29_WORDS * (size_t)blockIdx.x, r);
    h2_publish_release();
    arrived = atomicAdd(counter + job, 1u);
  }
  arrived = __shfl(arrived, 0, 64);
  if (arrived != gridDim.x - 1) return;
  h2_consume_acquire();
  Xyzz29<CV> acc = Xyzz29<CV>::identity();
  for (uint32_t b = quad; b < gridDim.x; b += 16) acc = xyzz29_add_quad(acc, xyzz29_load<CV>(part + XYZZ29_WORDS * (size_t)b));
  for (uint32_t d = 32; d >= 4; d >>= 1) acc = xyzz_fold_down(acc, d, threadIdx.x);
  if (threadIdx.x == 0) {
    Fe<B> jx, jy, jz;
    xyzz_to_jacobian(xyzz29_to_api(acc), jx, jy, jz);
    fe_store<B>(out_jac + 6 * (size_t)job, jx);
    fe_store<B>(out_jac + 6 * (size_t)job + 2, jy);
    fe_store<B>(out_jac + 6 * (size_t)job + 4, jz);
  }
}

// ---- SRS generation: g[i] = [s^i] G  (ParamsKZG::new's coefficient-basis vector) -----------------
// Device counterpart of the setup loop reached from /root/reference/circuits/src/utils.rs:59-61
// (SURVEY.md section 3.2).  One thread per point: s^i by square-and-multiply, then a 255-step
// double-and-add on the generator and one inversion to affine.
template <class CV>
__global__ void __launch_bounds__(256)
srs_powers_kernel(U128* __restrict__ out, Fe<typename CV::Scalar> s, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  using FS = typename CV::Scalar;
  using FB = typename CV::Base;
  Fe<FS> k = fe_from_mont(fe_pow_u64(s, (uint64_t)i));
  Affine<CV> g;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    g.x.v[j] = CV::GX(j);
    g.y.v[j] = CV::GY(j);
  }
  Xyzz<CV> r = Xyzz<CV>::identity();
  for (int bit = 255; bit >= 0; bit--) {
    r = xyzz_double(r);
    if ((k.v[bit >> 5] >> (bit & 31)) & 1) r = xyzz_add_affine(r, g);
  }
  Affine<CV> a = xyzz_to_affine(r);
  fe_store<FB>(out + 4 * (size_t)i, a.x);
  fe_store<FB>(out + 4 * (size_t)i + 2, a.y);
}

// out[i] = [k_i] G for n given scalars (Montgomery form): the g_lagrange vector of ParamsKZG::new is
// [L_i(s)] G with L_i the Lagrange basis evaluated at the toxic scalar
template <class CV>
__global__ void __launch_bounds__(256)
fixed_base_mul_kernel(U128* __restrict__ out, const U128* __restrict__ scalars, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  using FS = typename CV::Scalar;
  using FB = typename CV::Base;
  Fe<FS> k = fe_from_mont(fe_load<FS>(scalars + 2 * (size_t)i));
  Affine<CV> g;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    g.x.v[j] = CV::GX(j);
    g.y.v[j] = CV::GY(j);
  }
  Xyzz<CV> r = Xyzz<CV>::identity();
  for (int bit = 255; bit >= 0; bit--) {
    r = xyzz_double(r);
    if ((k.v[bit >> 5] >> (bit & 31)) & 1) r = xyzz_add_affine(r, g);
  }
  Affine<CV> a = xyzz_to_affine(r);
  fe_store<FB>(out + 4 * (size_t)i, a.x);
  fe_store<FB>(out + 4 * (size_t)i + 2, a.y);
}

// ---- workspace layout -------------------------------------------------------------------------
// The arena is a sequence of named regions.  msm_check() proves, on the host and before anything is enqueued, that the
// index ranges of every kernel of the launch sequence lie inside their regions; a guard build of the layout
// (msm_workspace(..., guard)) puts a red zone behind every region, which tests fill with a pattern before the launch
// and inspect after it (h2_selftest_msm_guard): an overrun that stays inside the arena corrupts silently otherwise.
constexpr uint32_t MSM_WS_REGIONS = 32;
constexpr uint8_t MSM_GUARD_BYTE = 0xA5;
struct MsmRegion {
  const char* name;
  size_t off, bytes;
};
struct MsmWorkspace {
  size_t n, m;          // the shape the layout was made for
  size_t K;             // keys = m * B
  size_t E;             // max entries = m * W * n
  size_t nblk;          // scan blocks
  uint32_t T;           // sorted entries per accumulate thread
  size_t nchunks;       // ceil(E / T)
  uint32_t log_g;       // lanes per key in the fix-up kernel = 2^log_g
  uint32_t tile;        // scalars per block in the digits / scatter kernels
  uint32_t staged;      // the scatter stages its tile in LDS (msm_scatter_staged_kernel); stage_lds bytes of dynamic LDS
  size_t stage_lds;
  uint32_t sort2;       // the two-level sort (h2_msm_sort2.hpp) instead of digits / scan / scatter
  Sort2Geom s2;
  uint32_t lb;          // low bits of a bucket index in the row / column split of the weights (msm_rowcol_kernel)
  uint32_t rc;          // row + column sums per column = 2^(log_b - lb) + 2^lb
  size_t off_counts, off_gcounts, off_offsets, off_tile_base, off_tile_hist, off_blocksums, off_ref, off_key, off_misc, off_bsum,
      off_head, off_tail, off_xsum, off_rc, off_part, off_done, off_tree2, off_hot_slot, off_hot_tasks, off_hot_part, total;
  size_t off_cstart, off_group_base, off_mid_ref, off_mid_lo;   // two-level sort only
  size_t zero_bytes;    // misc + the counters behind it: cleared by one memset per launch
  uint32_t max_tasks;
  uint32_t guard;       // bytes of red zone behind every region (0 in the product path)
  uint32_t n_regions;
  MsmRegion regions[MSM_WS_REGIONS];
};
inline size_t h2_align256(size_t x) { return (x + 255) & ~(size_t)255; }

// which sort a launch of m columns of n scalars uses
inline bool msm_use_sort2(size_t n, size_t m, const MsmGeom& g) {
  const int forced = tune_int("H2_TUNE_SORT2", -1);                 // tuning builds only (h2_tune.hpp)
  const Sort2Geom s = msm_sort2_geom(n, g);
  const bool fits = g.B >= S2_MAX_F / 8 && s.F <= S2_MAX_F && (size_t)s.Hc * m <= S2_MAX_H;
  if (forced >= 0) return forced != 0 && fits;
  // wide windows: a tile of the one-level sort has less than one entry per bucket, its stores leave one by one
  return fits && g.B > 4096;
}

inline MsmWorkspace msm_workspace(size_t n, size_t m, const MsmGeom& g, uint32_t guard = 0, size_t n_bases = 0,
                                  bool allow_pack = true) {
  if (!n_bases) n_bases = n;
  MsmWorkspace ws{};
  ws.n = n;
  ws.m = m;
  ws.guard = guard;
  ws.K = m * g.B;
  ws.E = m * g.W * n;
  ws.nblk = (ws.K + SCAN_BLOCK - 1) / SCAN_BLOCK;
  // Additions per thread.  The accumulate kernel is one resident round of VALU-bound waves, so it lasts as long as
  // the SIMDs that hold the most waves: pick T so that the launch is a WHOLE number w of waves per SIMD (1024 SIMDs
  // x 64 lanes = 65536 threads per unit of w) and w * T is smallest; fewer, longer chunks also mean fewer pieces for
  // the fix-up.  Big inputs take T = 64 and several rounds, where the rounding no longer matters.
  const uint64_t t_max = (uint64_t)tune_int("H2_TUNE_TMAX", 64);
  uint64_t T = t_max;
  {
    uint64_t best = ~0ull;
    for (uint64_t w = MSM_CHUNK_WAVES; w >= 2; w--) {
      uint64_t t = (ws.E + w * 65536 - 1) / (w * 65536);
      if (t < 8) t = 8;
      if (t > t_max) continue;
      const uint64_t cost = w * t * (w == 2 ? 21 : 20);      // two waves per SIMD hide a little less latency
      if (cost < best) { best = cost; T = t; }
    }
  }
  ws.T = (uint32_t)T;
  ws.nchunks = (ws.E + T - 1) / T;
  {
    // the device may cut finer than T when the sort produced fewer entries than the worst case (msm_effective_t):
    // never more chunks than ceil(E / 8) nor than MSM_CHUNK_WAVES waves per SIMD -- size arrays and grid for that
    const size_t fine = std::min<size_t>((ws.E + 7) / 8, (size_t)MSM_CHUNK_WAVES * 65536 + 1);
    if (ws.nchunks < fine) ws.nchunks = fine;
  }
  // pieces per key ~ list length / T + 1
  const double span = (double)g.W * (double)n / (double)g.B / (double)T + 1.0;
  // quads per key in the fix-up: no more than the pieces need, and few enough that the launch stays around
  // two waves per SIMD (K * Q * 4 lanes <= 160k): wider groups waste most of their quads in the shuffle tree
  uint32_t lq = 0;
  while ((1u << lq) < span && lq < 4) lq++;
  while (lq > 0 && (ws.K << lq) > 40960) lq--;
  lq = (uint32_t)tune_int("H2_TUNE_LQ", (int)lq);   // tuning builds only (h2_tune.hpp)
  uint32_t lg = lq + 2;
  ws.log_g = lg;
  ws.sort2 = msm_use_sort2(n, m, g) ? 1u : 0u;
  ws.s2 = msm_sort2_geom(n, g, allow_pack ? n_bases : 0);
  // digits / scatter tiling: about 1024 blocks over the launch, at least one wave of scalars per block
  size_t tile = (n * m + 1023) / 1024;
  // a tile should carry a few entries per bucket, or zeroing / flushing the LDS histogram dominates
  const size_t dense = (size_t)2 * g.B / g.W;
  if (tile < dense) tile = dense;
  if (tile < 256) tile = 256;
  tile *= (size_t)tune_int("H2_TUNE_TILE_MUL", 2);   // measured with 1024-thread blocks (MSM_SORT_THREADS): longer runs per (tile, bucket), fewer sector writes
  if (tile > n) tile = n;
  // the staged scatter: buckets fit 16 bits, and a tile with >= 2 entries per bucket fits the CU's LDS next to the two
  // per-bucket arrays.  One block per CU then, so the tile is sized for a whole number of rounds of 256 blocks, as
  // few as fit, two at most (m = 4 at 2^16: 256 blocks of 1024 scalars, 152 KB; m = 5: 512 blocks of 640).
  ws.staged = 0;
  ws.stage_lds = 0;
  if (!ws.sort2) {
    const size_t cap = 160 * 1024 - 512;
    auto need = [&](size_t t) { return (size_t)8 * g.B + (size_t)6 * t * g.W + 64; };
    if (g.B <= 65536 && n * m >= 8192 && need(dense) <= cap) {
      // (more than two rounds of one-block-per-CU tiles cost more than the stores save: keygen's 16 sparse columns took
      // 214 us per launch staged in four rounds against 85 us direct)
      for (size_t rounds = 1; rounds <= 2; rounds++) {
        size_t t = (n * m + 256 * rounds - 1) / (256 * rounds);
        if (t < dense) t = dense;
        if (t < 256) t = 256;
        if (t > n) t = n;
        if (need(t) <= cap) {
          tile = t;
          ws.staged = 1;
          ws.stage_lds = need(t);
          break;
        }
      }
    }
  }
  ws.tile = (uint32_t)tile;
  ws.lb = (g.c - 1) / 2;
  ws.rc = (1u << (g.c - 1 - ws.lb)) + (1u << ws.lb);
  size_t o = 0;
  auto region = [&](const char* name, size_t bytes) {
    const size_t at = o;
    if (ws.n_regions < MSM_WS_REGIONS) ws.regions[ws.n_regions++] = MsmRegion{name, at, bytes};
    o = h2_align256(o + bytes) + (guard ? h2_align256(guard) : 0);
    return at;
  };
  const size_t tiles = (n + ws.tile - 1) / ws.tile;
  if (ws.sort2) {
    const size_t H = (size_t)ws.s2.Hc * m, t2 = (size_t)ws.s2.tiles * m;
    // misc[0] = hot task counter, misc + 16: per-column table pointers; then the coarse bins' counters
    ws.zero_bytes = 256 + H * 4;
    ws.off_misc = region("zeroed: misc + coarse counters", ws.zero_bytes);
    ws.off_gcounts = ws.off_misc + 256;
    ws.off_counts = ws.off_gcounts;
    ws.off_cstart = region("coarse bin starts", (H + 1) * 4);
    ws.off_offsets = region("offsets", (ws.K + 1) * 4);
    ws.off_tile_base = region("tile bases", t2 * ws.s2.Hc * 4);
    ws.off_tile_hist = region("tile counts", t2 * ws.s2.Hc * 4);
    ws.off_group_base = region("group bases", (size_t)ws.s2.groups * m * ws.s2.Hc * 4);
    ws.off_mid_ref = region("coarse-sorted entries", ws.E * 4);
    ws.off_mid_lo = region("coarse-sorted low key bits", ws.s2.pack_shift ? 0 : ws.E);
    ws.off_blocksums = ws.off_cstart;     // unused
  } else {
    // misc (256 B), per-key totals (only the multi-kernel scan reads them), the per-XCD-group counters
    ws.zero_bytes = 256 + h2_align256(ws.K * 4) + MSM_XCDS * ws.K * 4;
    ws.off_misc = region("zeroed: misc + counts + group counters", ws.zero_bytes);
    ws.off_counts = ws.off_misc + 256;
    ws.off_gcounts = ws.off_counts + h2_align256(ws.K * 4);
    ws.off_offsets = region("offsets", (ws.K + 1) * 4);
    ws.off_tile_base = region("tile bases", tiles * ws.K * 4);                       // tiles x (m * B) words
    ws.off_tile_hist = region("tile counts", ws.staged ? tiles * ws.K * 4 : 0);
    ws.off_blocksums = region("scan block sums", (ws.nblk + 1) * 4);
  }
  ws.off_ref = region("sorted entries", ws.E * 4 + 16);                              // + slack for the last 16-byte read
  ws.off_key = region("chunk first keys", (ws.nchunks + 1) * 4);
  ws.off_bsum = region("bucket sums", ws.K * (XYZZ29_WORDS * 4));
  ws.off_head = region("chunk heads", ws.nchunks * (XYZZ29_WORDS * 4));
  ws.off_tail = region("chunk tails", ws.nchunks * (XYZZ29_WORDS * 4));
  ws.off_xsum = region("bucket point sums", ws.K * (XYZZ29_WORDS * 4));
  ws.off_rc = region("row / column sums", m * ws.rc * (XYZZ29_WORDS * 4));
  ws.off_part = region("final partials", m * MSM_FINAL_MAX_BLOCKS * (XYZZ29_WORDS * 4));
  ws.off_done = region("final counters", m * 4);
  ws.off_tree2 = region("results", m * (XYZZ29_WORDS * 4));
  // hot keys: a key with span > MSM_HOT_SPAN emits ceil(span / SEG) <= span / SEG + 1 <= span / SEG + span / SPAN
  // tasks, and the spans of all keys add up to at most nchunks + K_hot <= nchunks * (1 + 1 / SPAN)
  ws.max_tasks = (uint32_t)(ws.nchunks / MSM_HOT_SEG + 2 * (ws.nchunks / MSM_HOT_SPAN) + 16);
  ws.off_hot_slot = region("hot slots", ws.K * 4);
  ws.off_hot_tasks = region("hot tasks", (size_t)ws.max_tasks * 8);
  ws.off_hot_part = region("hot partials", (size_t)ws.max_tasks * (XYZZ29_WORDS * 4));
  ws.total = o;
  return ws;
}

// Host-side proof that the launch sequence stays inside the regions above: every kernel's largest index against the
// bytes of the array it indexes, the grid sizes against the hardware's limits, the dynamic LDS against the CU's.
// Returns null, or the first violated condition (msm_launch then enqueues nothing).
#define MSM_REQUIRE(cond) \
  do {                    \
    if (!(cond)) return #cond; \
  } while (0)
inline const char* msm_check(const MsmWorkspace& ws, const MsmGeom& g, size_t n, size_t m, size_t col_stride,
                             uint32_t n_bases, size_t arena_bytes) {
  auto bytes_at = [&](size_t off) -> size_t {
    for (uint32_t r = 0; r < ws.n_regions; r++)
      if (off >= ws.regions[r].off && off < ws.regions[r].off + ws.regions[r].bytes)    // (an empty region holds nothing)
        return ws.regions[r].off + ws.regions[r].bytes - off;
    return 0;
  };
  MSM_REQUIRE(ws.n == n && ws.m == m);                                   // the layout was made for this launch
  MSM_REQUIRE(n >= 1 && m >= 1 && n <= n_bases);
  MSM_REQUIRE(m == 1 || col_stride >= n);
  MSM_REQUIRE(ws.n_regions < MSM_WS_REGIONS);
  MSM_REQUIRE(ws.total <= arena_bytes);
  for (uint32_t r = 0; r + 1 < ws.n_regions; r++)
    MSM_REQUIRE(ws.regions[r].off + ws.regions[r].bytes + ws.guard <= ws.regions[r + 1].off);
  MSM_REQUIRE(ws.regions[ws.n_regions - 1].off + ws.regions[ws.n_regions - 1].bytes + ws.guard <= ws.total);
  MSM_REQUIRE(g.W >= 1 && g.W <= MSM_MAX_WINDOWS && g.B == (1u << (g.c - 1)) && g.c <= MSM_MAX_C);
  MSM_REQUIRE(ws.sort2 || g.c <= MSM_MAX_C_ONE_LEVEL);                    // the one-level sort's LDS histogram
  MSM_REQUIRE((uint64_t)g.W * n_bases < (1ull << 31));                  // an entry is w * n_bases + i below the sign bit
  MSM_REQUIRE(ws.K == m * (size_t)g.B && ws.E == m * (size_t)g.W * n);
  MSM_REQUIRE(ws.E < (1ull << 31) && ws.K < (1ull << 31));
  MSM_REQUIRE(bytes_at(ws.off_misc) >= ws.zero_bytes && ws.zero_bytes >= 256);
  MSM_REQUIRE(ws.zero_bytes % 16 == 0 && ws.off_misc % 16 == 0 && ws.zero_bytes / 16 < (1ull << 32));   // re-zeroed by msm_rowcol_kernel
  MSM_REQUIRE(bytes_at(ws.off_offsets) >= (ws.K + 1) * 4);
  if (ws.sort2) {
    const Sort2Geom& s = ws.s2;
    const size_t H = (size_t)s.Hc * m;
    MSM_REQUIRE(s.F * s.Hc == g.B && s.F == (1u << s.lo_bits) && s.F <= S2_MAX_F && s.Hc <= S2_MAX_HC && H <= S2_MAX_H);
    MSM_REQUIRE(s.tile >= 1 && s.tile <= S2_THREADS && (size_t)s.tile * g.W <= S2_STAGE);
    MSM_REQUIRE((size_t)s.tiles * s.tile >= n && (size_t)(s.tiles - 1) * s.tile < n);
    MSM_REQUIRE(s.tiles <= 0x7FFFFFFFu && m <= 65535);                  // grid (tiles, m)
    MSM_REQUIRE(ws.off_gcounts == ws.off_misc + 256 && ws.zero_bytes >= 256 + H * 4);
    MSM_REQUIRE(bytes_at(ws.off_cstart) >= (H + 1) * 4);
    MSM_REQUIRE(bytes_at(ws.off_tile_base) >= (size_t)s.tiles * m * s.Hc * 4);
    MSM_REQUIRE(bytes_at(ws.off_tile_hist) >= (size_t)s.tiles * m * s.Hc * 4);
    MSM_REQUIRE(bytes_at(ws.off_mid_ref) >= ws.E * 4 && (s.pack_shift || bytes_at(ws.off_mid_lo) >= ws.E));
    MSM_REQUIRE(s.group >= 1 && s.group <= S2_GROUP && (size_t)s.groups * s.group >= s.tiles && bytes_at(ws.off_group_base) >= (size_t)s.groups * m * s.Hc * 4);
    MSM_REQUIRE(!s.pack_shift || (((uint64_t)g.W * n_bases <= (1ull << s.pack_shift)) && s.pack_shift + s.lo_bits <= 31));
    MSM_REQUIRE((size_t)s.Hc * 8 <= 64 * 1024);
    MSM_REQUIRE(msm_sort2_lds_scatter(s, g) <= 160 * 1024 - 512 && msm_sort2_lds_fine() + 4 * S2_MAX_F * 4 <= 160 * 1024 - 512);
  } else {
    const size_t tiles = (n + ws.tile - 1) / ws.tile;
    MSM_REQUIRE(ws.tile >= 1 && tiles * m < (1ull << 31) - MSM_XCDS);
    MSM_REQUIRE(msm_tile_grid((uint32_t)tiles, (uint32_t)m) >= tiles * m);
    MSM_REQUIRE((size_t)g.B * 4 <= 128 * 1024);                          // the LDS histogram of the digits / scatter kernels
    MSM_REQUIRE(ws.off_counts == ws.off_misc + 256 && ws.off_gcounts >= ws.off_counts + ws.K * 4);
    MSM_REQUIRE(ws.off_gcounts + MSM_XCDS * ws.K * 4 <= ws.off_misc + ws.zero_bytes);
    MSM_REQUIRE(bytes_at(ws.off_tile_base) >= tiles * ws.K * 4);
    if (ws.staged) {
      MSM_REQUIRE(bytes_at(ws.off_tile_hist) >= tiles * ws.K * 4);
      MSM_REQUIRE(g.B <= 65536);                                         // staged buckets are 16-bit
      MSM_REQUIRE(ws.stage_lds >= (size_t)8 * g.B + (size_t)6 * ws.tile * g.W && ws.stage_lds <= 160 * 1024 - 512);
    }
    if (ws.K > SCAN_LDS_MAX) {
      MSM_REQUIRE(ws.nblk * SCAN_BLOCK >= ws.K && ws.nblk < (1ull << 31));
      MSM_REQUIRE(bytes_at(ws.off_blocksums) >= (ws.nblk + 1) * 4);
    }
  }
  MSM_REQUIRE(bytes_at(ws.off_ref) >= ((ws.E + 3) & ~(size_t)3) * 4);  // 16-byte reads of the last entries
  // chunks the device may cut (msm_effective_t): t = max(8, ceil(E' / (3 * 65536))) capped at T, for any E' <= E
  MSM_REQUIRE(ws.T >= 8 && ws.nchunks >= (ws.E + ws.T - 1) / ws.T);
  MSM_REQUIRE(ws.nchunks >= std::min<size_t>((ws.E + 7) / 8, (size_t)MSM_CHUNK_WAVES * 65536 + 1));
  MSM_REQUIRE((ws.nchunks + 255) / 256 < (1ull << 31));
  MSM_REQUIRE(bytes_at(ws.off_key) >= ws.nchunks * 4);
  MSM_REQUIRE(bytes_at(ws.off_head) >= ws.nchunks * (XYZZ29_WORDS * 4) && bytes_at(ws.off_tail) >= ws.nchunks * (XYZZ29_WORDS * 4));
  MSM_REQUIRE(bytes_at(ws.off_bsum) >= ws.K * (XYZZ29_WORDS * 4) && bytes_at(ws.off_xsum) >= ws.K * (XYZZ29_WORDS * 4));
  MSM_REQUIRE(ws.log_g >= 2 && ws.log_g <= 6 && ((ws.K << ws.log_g) + 255) / 256 < (1ull << 31));
  MSM_REQUIRE(ws.lb < g.c && ws.rc == (1u << (g.c - 1 - ws.lb)) + (1u << ws.lb));
  MSM_REQUIRE(bytes_at(ws.off_rc) >= m * ws.rc * (XYZZ29_WORDS * 4));
  MSM_REQUIRE(msm_final_blocks(g.c - 1, ws.lb) <= MSM_FINAL_MAX_BLOCKS);
  MSM_REQUIRE(bytes_at(ws.off_part) >= m * MSM_FINAL_MAX_BLOCKS * (XYZZ29_WORDS * 4));
  MSM_REQUIRE(bytes_at(ws.off_done) >= m * 4 && bytes_at(ws.off_tree2) >= m * (XYZZ29_WORDS * 4));
  MSM_REQUIRE(bytes_at(ws.off_hot_slot) >= ws.K * 4);
  MSM_REQUIRE(bytes_at(ws.off_hot_tasks) >= (size_t)ws.max_tasks * 8);
  MSM_REQUIRE(bytes_at(ws.off_hot_part) >= (size_t)ws.max_tasks * (XYZZ29_WORDS * 4));
  MSM_REQUIRE(m <= 65535);                                               // grid.y of the row / column and final kernels
  return nullptr;
}
#undef MSM_REQUIRE

// guard builds of the layout: count the red-zone bytes that no longer hold the pattern (one block per region)
static __global__ void __launch_bounds__(256)
msm_guard_check_kernel(const uint8_t* __restrict__ arena, MsmWorkspace ws, uint32_t* __restrict__ bad /* per region */) {
  const uint32_t r = blockIdx.x;
  const size_t from = ws.regions[r].off + ws.regions[r].bytes;
  const size_t to = r + 1 < ws.n_regions ? ws.regions[r + 1].off : ws.total;
  uint32_t c = 0;
  for (size_t i = from + threadIdx.x; i < to; i += blockDim.x) c += arena[i] != MSM_GUARD_BYTE;
  if (c) atomicAdd(&bad[r], c);
}

// once per device (h2_init): the sort kernels and the one-block scan use more than the default 64 KiB of dynamic LDS
template <class CV>
inline hipError_t msm_kernel_setup() {
  hipError_t e;
  const int lds = (int)((1u << (MSM_MAX_C_ONE_LEVEL - 1)) * 4);
  if ((e = hipFuncSetAttribute((const void*)msm_digits_kernel<CV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute((const void*)msm_scatter_kernel<CV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute((const void*)msm_scatter_staged_kernel<CV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute((const void*)msm2_scatter_kernel<CV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute((const void*)msm2_fine_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)msm_sort2_lds_fine())) != hipSuccess) return e;
  return hipFuncSetAttribute((const void*)scan_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SCAN_LDS_MAX * 4));
}

// Enqueue m MSMs of n terms against `table` (built for n_bases points with geometry g); column j's scalars start
// col_stride elements after column j-1's.
// Result: m XYZZ points at ws_base + off_tree2.  ev_start / ev_stop (optional) bracket the
// accumulate (chunk) kernel for the roofline measurement.
template <class CV>
inline hipError_t msm_launch(const U128* table, const U128* const* per_column /* host array of m tables, or null */,
                             uint32_t n_bases, const U128* d_scalars, size_t n, size_t col_stride,
                             size_t m, const MsmGeom& g, char* ws_base, const MsmWorkspace& ws, hipStream_t stream,
                             hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, hipEvent_t ev_tail = nullptr,
                             U128* d_out_jac = nullptr /* m Jacobian points in the API's form, written by the last kernel */,
                             bool zeroed = false /* [off_misc, + zero_bytes) is zero already: the previous launch sequence on
                                                    this workspace left it so (same off_misc, at least as many bytes) */) {
  uint32_t* counts = (uint32_t*)(ws_base + ws.off_counts);
  uint32_t* gcounts = (uint32_t*)(ws_base + ws.off_gcounts);
  uint32_t* offsets = (uint32_t*)(ws_base + ws.off_offsets);
  uint32_t* tile_base = (uint32_t*)(ws_base + ws.off_tile_base);
  uint32_t* blocksums = (uint32_t*)(ws_base + ws.off_blocksums);
  uint32_t* sref = (uint32_t*)(ws_base + ws.off_ref);
  uint32_t* chunk_first = (uint32_t*)(ws_base + ws.off_key);
  uint32_t* misc = (uint32_t*)(ws_base + ws.off_misc);
  uint32_t* bsum = (uint32_t*)(ws_base + ws.off_bsum);
  uint32_t* head = (uint32_t*)(ws_base + ws.off_head);
  uint32_t* tail = (uint32_t*)(ws_base + ws.off_tail);
  uint32_t* xsum = (uint32_t*)(ws_base + ws.off_xsum);
  uint32_t* rc = (uint32_t*)(ws_base + ws.off_rc);
  uint32_t* tree2 = (uint32_t*)(ws_base + ws.off_tree2);
  uint32_t* hot_slot = (uint32_t*)(ws_base + ws.off_hot_slot);
  uint32_t* hot_tasks = (uint32_t*)(ws_base + ws.off_hot_tasks);
  uint32_t* hot_part = (uint32_t*)(ws_base + ws.off_hot_part);
  hipError_t e;
  if (per_column && m > MSM_MAX_MULTI) return hipErrorInvalidValue;
  // one memset: misc (256 B) and the sort's counters behind it.  Nothing else needs clearing: every slot of bucket_sum /
  // head / tail that a later kernel reads has been written by the accumulate kernel (the fix-up decides from
  // `offsets` which slots exist).
  if (!zeroed && (e = hipMemsetAsync(misc, 0, ws.zero_bytes, stream)) != hipSuccess) return e;
  uint32_t* tile_hist = (ws.staged || ws.sort2) ? (uint32_t*)(ws_base + ws.off_tile_hist) : nullptr;
  // the keys pass (msm_keys_block): after the scan, beside or behind the scatter
  const U128** d_tables = nullptr;
  uint32_t log_b = 0;
  MsmKeysArgs keys_args{};
  keys_args.chunk_first = chunk_first;
  keys_args.hot_slot = hot_slot;
  keys_args.tasks = hot_tasks;
  keys_args.task_count = misc;
  keys_args.max_tasks = ws.max_tasks;
  keys_args.T_host = ws.T;
  if (per_column) {
    d_tables = (const U128**)(misc + 16);                          // 128 bytes of the 256-byte misc block
    keys_args.n_tables = (uint32_t)m;
    keys_args.tables_dst = d_tables;
    for (size_t j = 0; j < m; j++) keys_args.tables.t[j] = per_column[j];
    while ((1u << log_b) < g.B) log_b++;
  }
  bool keys_merged = false;
  if (ws.sort2) {
    const Sort2Geom& s2 = ws.s2;
    const uint32_t H = s2.Hc * (uint32_t)m;
    uint32_t* cstart = (uint32_t*)(ws_base + ws.off_cstart);
    uint32_t* mid_ref = (uint32_t*)(ws_base + ws.off_mid_ref);
    uint8_t* mid_lo = (uint8_t*)(ws_base + ws.off_mid_lo);
    uint32_t* group_base = (uint32_t*)(ws_base + ws.off_group_base);
    hipLaunchKernelGGL(msm2_count_kernel<CV>, dim3(s2.groups, (unsigned)m), dim3(S2_THREADS), (size_t)s2.Hc * 8, stream, d_scalars,
                       gcounts, tile_base, tile_hist, group_base, (uint32_t)n, col_stride, s2, g);
    hipLaunchKernelGGL(msm2_coarse_scan_kernel, dim3(1), dim3(1024), 0, stream, gcounts, cstart, H, offsets + ws.K);
    hipLaunchKernelGGL(msm2_scatter_kernel<CV>, dim3(s2.tiles, (unsigned)m), dim3(S2_THREADS), msm_sort2_lds_scatter(s2, g), stream,
                       d_scalars, cstart, tile_base, tile_hist, group_base, mid_ref, mid_lo, (uint32_t)n, col_stride, n_bases, s2, g);
    hipLaunchKernelGGL(msm2_fine_kernel, dim3(H), dim3(S2_THREADS), msm_sort2_lds_fine(), stream, cstart, mid_ref, mid_lo, sref,
                       offsets, s2.F, s2.lo_bits, s2.pack_shift);
  } else {
    const size_t lds = (size_t)g.B * 4;     // dynamic LDS limits were raised once per device by msm_kernel_setup
    const uint32_t tiles = (uint32_t)((n + ws.tile - 1) / ws.tile);
    const uint32_t sort_grid = msm_tile_grid(tiles, (uint32_t)m);
    hipLaunchKernelGGL(msm_digits_kernel<CV>, dim3(sort_grid), dim3(MSM_SORT_THREADS), lds, stream, d_scalars, gcounts,
                       tile_base, tile_hist, (uint32_t)n, col_stride, ws.tile, tiles, (uint32_t)m, g);
    if (ws.K <= SCAN_LDS_MAX) {
      uint32_t per = (uint32_t)((ws.K + 1023) / 1024);
      per |= 1u;                                 // odd stride: the per-thread LDS walks do not collide on banks
      hipLaunchKernelGGL(scan_lds_kernel, dim3(1), dim3(1024), ws.K * 4, stream, gcounts, offsets, (uint32_t)ws.K, per);
    } else {
      hipLaunchKernelGGL(msm_group_fold_kernel, dim3((unsigned)((ws.K + 255) / 256)), dim3(256), 0, stream, gcounts, counts, ws.K);
      hipLaunchKernelGGL(scan_reduce_kernel, dim3((unsigned)ws.nblk), dim3(256), 0, stream, counts, blocksums, ws.K);
      hipLaunchKernelGGL(scan_blocksums_kernel, dim3(1), dim3(1024), 0, stream, blocksums, (uint32_t)ws.nblk, misc + 1);
      hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)ws.nblk), dim3(256), 0, stream, counts, blocksums, offsets,
                         ws.K);
    }
    if (ws.staged) {
      // the keys pass shared out over the scatter's blocks (at most one key per thread; the staging area holds its LDS
      // at every bucket count this sort is chosen for -- the checks keep small geometries honest)
      const size_t per = (ws.K + (size_t)tiles * m - 1) / ((size_t)tiles * m);
      keys_merged = per <= MSM_SORT_THREADS && ws.stage_lds >= (size_t)4 * (MSM_SORT_THREADS + 1);
      MsmKeysArgs ka = keys_args;
      const uint32_t grid = sort_grid;
      if (keys_merged) ka.keys_per_block = (uint32_t)per;
      hipLaunchKernelGGL(msm_scatter_staged_kernel<CV>, dim3(grid), dim3(MSM_SORT_THREADS), ws.stage_lds, stream, d_scalars,
                         offsets, gcounts, tile_base, tile_hist, sref, (uint32_t)n, col_stride, n_bases, ws.tile, tiles,
                         (uint32_t)m, g, (uint32_t)(ws.tile * g.W), ka);
    } else
      hipLaunchKernelGGL(msm_scatter_kernel<CV>, dim3(sort_grid), dim3(MSM_SORT_THREADS), lds, stream, d_scalars, offsets,
                         gcounts, tile_base, sref, (uint32_t)n, col_stride, n_bases, ws.tile, tiles, (uint32_t)m, g);
  }
  if (!keys_merged)
    hipLaunchKernelGGL(msm_keys_kernel, dim3((unsigned)((ws.K + MSM_KEYS_THREADS - 1) / MSM_KEYS_THREADS)), dim3(MSM_KEYS_THREADS), 0,
                       stream, (const uint32_t*)offsets, ws.K, keys_args);
  // The roofline's start / stop events and the tail event (from the accumulate kernel's end on only small-grid kernels
  // run: other streams may fill the chip) ride on the kernel's own dispatch packet (hipExtLaunchKernelGGL): as separate
  // hipEventRecord calls each was a barrier packet of its own, ~6 us of stream time before and after the kernel
  // (profiles/r03_step_kernel_timeline.txt).
  hipEvent_t ev_end = ev_stop ? ev_stop : ev_tail;
  if (ev_start || ev_end)
    hipExtLaunchKernelGGL(msm_chunk_kernel<CV>, dim3((unsigned)((ws.nchunks + 255) / 256)), dim3(256), 0, stream, ev_start, ev_end,
                          0, table, (const U128* const*)d_tables, log_b, (const uint32_t*)sref, (const uint32_t*)chunk_first,
                          (const uint32_t*)offsets, ws.K, ws.T, bsum, head, tail);
  else
    hipLaunchKernelGGL(msm_chunk_kernel<CV>, dim3((unsigned)((ws.nchunks + 255) / 256)), dim3(256), 0, stream, table,
                       (const U128* const*)d_tables, log_b, sref, chunk_first, offsets, ws.K, ws.T, bsum, head, tail);
  if (ev_stop && ev_tail) (void)hipEventRecord(ev_tail, stream);
  hipLaunchKernelGGL(msm_hot_reduce_kernel<CV>, dim3(1024), dim3(64), 0, stream, offsets, ws.K, ws.T, hot_slot, hot_tasks,
                     misc, ws.max_tasks, head, tail, hot_part);
  const size_t fix_threads = ws.K << ws.log_g;
  hipLaunchKernelGGL(msm_fixup_kernel<CV>, dim3((unsigned)((fix_threads + 255) / 256)), dim3(256), 0, stream, offsets,
                     ws.K, ws.T, ws.log_g, bsum, head, tail, hot_slot, hot_part, xsum);
  uint32_t* part = (uint32_t*)(ws_base + ws.off_part);
  uint32_t* done = (uint32_t*)(ws_base + ws.off_done);
  hipLaunchKernelGGL(msm_rowcol_kernel<CV>, dim3(ws.rc, (unsigned)m), dim3(64 * msm_rowcol_waves(g.c - 1, ws.lb, m)), 0, stream,
                     xsum, rc, done, g.c - 1, ws.lb, (U128*)misc, (uint32_t)(ws.zero_bytes / 16));
  hipLaunchKernelGGL(msm_final_kernel<CV>, dim3(msm_final_blocks(g.c - 1, ws.lb), (unsigned)m), dim3(64), 0, stream, rc, part,
                     done, tree2, d_out_jac, g.c - 1, ws.lb);
  return hipGetLastError();
}

}  // namespace h2
