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
//    HBM: at registration the table T[w][i] = 2^(c*w) * P_i (affine) is built once.  Every
//    c-bit window of a scalar then selects a point of the same weight, so ALL windows share
//    one set of 2^(c-1) buckets, the per-window running sums shrink to a single one and the
//    255 serial doublings of the window combination disappear.
//  * Signed digits (|d| <= 2^(c-1)) halve the bucket count; the sign rides in bit 31 of the
//    sorted entry and negates y on the fly.
//  * digits -> histogram -> exclusive scan -> scatter gives, per (window, bucket) key, the
//    contiguous list of table indices to add; one thread accumulates one key in XYZZ
//    coordinates with mixed additions (8M + 2S each).  Keys whose list is longer than
//    MSM_HOT are left to a block-cooperative kernel (degenerate columns: all-ones witnesses).
//  * bucket weights: P_b = (b+1) * sum_w acc[w][b], then a tree sum.
#pragma once
#include "h2_curve.hpp"

namespace h2 {

constexpr uint32_t MSM_HOT = 2048;        // longer key lists go to the cooperative kernel
constexpr uint32_t MSM_SIGN = 0x80000000u;
constexpr uint32_t MSM_TREE_SEG = 2048;   // points summed by one block of msm_tree_sum_kernel

struct MsmGeom {
  uint32_t c;        // window bits
  uint32_t W;        // windows
  uint32_t B;        // buckets = 2^(c-1)
  uint32_t nbits;    // scalar field bits
};

inline MsmGeom msm_geometry(size_t n, uint32_t nbits) {
  uint32_t lg = 0;
  while (((size_t)1 << (lg + 1)) <= n) lg++;
  int c = (int)lg - 4;
  if (c < 6) c = 6;
  if (c > 20) c = 20;
  MsmGeom g;
  g.c = (uint32_t)c;
  g.nbits = nbits;
  g.W = (nbits + 1 + g.c - 1) / g.c;
  g.B = 1u << (g.c - 1);
  return g;
}

// ---- table build: T[w][i] = 2^(c*w) * P_i, affine ------------------------------------------
template <class CV>
__global__ void __launch_bounds__(256)
msm_table_kernel(const U128* __restrict__ bases, U128* __restrict__ table, uint32_t n, MsmGeom g) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  using B = typename CV::Base;
  Affine<CV> p = affine_load<CV>(bases + 4 * (size_t)i);
  Xyzz<CV> cur = xyzz_from_affine(p);
  for (uint32_t w = 0; w < g.W; w++) {
    Affine<CV> a = (w == 0) ? p : xyzz_to_affine(cur);
    U128* dst = table + 4 * ((size_t)w * n + i);
    fe_store<B>(dst, a.x);
    fe_store<B>(dst + 2, a.y);
    if (w + 1 < g.W) {
      for (uint32_t k = 0; k < g.c; k++) cur = xyzz_double(cur);
    }
  }
}

// ---- digits + histogram -------------------------------------------------------------------
// One signed-digit step: window w of the canonical scalar v (8 x u32), carry in/out.
// Returns 0 (no contribution) or |d| | sign<<31 with 1 <= |d| <= 2^(c-1).
// Invariant: sum_w d_w * 2^(c*w) = v, and the top window never carries out because
// W*c >= nbits + 1 (msm_geometry).
H2_HD uint32_t msm_digit_step(const uint32_t v[8], const MsmGeom& g, uint32_t w, uint32_t& carry) {
  const uint32_t mask = (1u << g.c) - 1, halfw = 1u << (g.c - 1);
  const uint32_t bit = w * g.c, limb = bit >> 5, off = bit & 31;
  uint32_t raw = 0;
  if (limb < 8) {
    uint64_t two = v[limb];
    if (limb + 1 < 8) two |= (uint64_t)v[limb + 1] << 32;
    raw = (uint32_t)(two >> off) & mask;
  }
  raw += carry;
  if (raw > halfw) {
    carry = 1;
    const uint32_t mag = (1u << g.c) - raw;  // digit = raw - 2^c <= 0 (0 when raw == 2^c)
    return mag ? (mag | MSM_SIGN) : 0u;
  }
  carry = 0;
  return raw;
}

// digits[(col*W + w)*n + i] = 0 (skip) or |d| | sign<<31 ;  counts[(col*W + w)*B + |d|-1]++
template <class CV>
__global__ void __launch_bounds__(256)
msm_digits_kernel(const U128* __restrict__ scalars, uint32_t* __restrict__ digits, uint32_t* __restrict__ counts,
                  uint32_t n, size_t col_stride /* elements */, MsmGeom g) {
  using S = typename CV::Scalar;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t col = blockIdx.y;
  if (i >= n) return;
  Fe<S> s = fe_from_mont(fe_load<S>(scalars + 2 * (col_stride * col + i)));
  uint32_t carry = 0;
  for (uint32_t w = 0; w < g.W; w++) {
    const uint32_t enc = msm_digit_step(s.v, g, w, carry);
    if (enc) atomicAdd(&counts[((size_t)col * g.W + w) * g.B + ((enc & ~MSM_SIGN) - 1)], 1u);
    digits[((size_t)col * g.W + w) * n + i] = enc;
  }
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
static __global__ void __launch_bounds__(1024) scan_blocksums_kernel(uint32_t* block_sums, uint32_t nb, uint32_t* total_out) {
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
// offsets[i] = exclusive prefix; cursor[i] = same (scatter cursors)
static __global__ void __launch_bounds__(256)
scan_apply_kernel(const uint32_t* in, const uint32_t* block_sums, uint32_t* offsets, uint32_t* cursor, size_t K) {
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
      cursor[idx] = run;
    }
    run += v[k];
  }
}

// ---- scatter: sorted[cursor[key]++] = table index | sign -------------------------------------
static __global__ void __launch_bounds__(256)
msm_scatter_kernel(const uint32_t* __restrict__ digits, uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted,
                   uint32_t n, uint32_t n_bases, MsmGeom g) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t cw = blockIdx.y;  // col*W + w
  if (i >= n) return;
  const uint32_t enc = digits[(size_t)cw * n + i];
  if (enc == 0) return;
  const uint32_t w = cw % g.W;
  const uint32_t mag = enc & ~MSM_SIGN;
  const uint32_t pos = atomicAdd(&cursor[(size_t)cw * g.B + (mag - 1)], 1u);
  sorted[pos] = (w * n_bases + i) | (enc & MSM_SIGN);
}

// ---- accumulate: one thread per (col, w, bucket) key ------------------------------------------
template <class CV>
__device__ __forceinline__ Affine<CV> msm_fetch(const U128* __restrict__ table, uint32_t entry) {
  Affine<CV> p = affine_load<CV>(table + 4 * (size_t)(entry & ~MSM_SIGN));
  if (entry & MSM_SIGN) p.y = fe_neg(p.y);
  return p;
}

template <class CV>
__global__ void __launch_bounds__(256)
msm_accumulate_kernel(const U128* __restrict__ table, const uint32_t* __restrict__ sorted,
                      const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
                      U128* __restrict__ acc, size_t K, uint32_t* __restrict__ hot_list, uint32_t* hot_count) {
  const size_t key = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (key >= K) return;
  const uint32_t cnt = counts[key];
  Xyzz<CV> a = Xyzz<CV>::identity();
  if (cnt > MSM_HOT) {
    const uint32_t slot = atomicAdd(hot_count, 1u);
    hot_list[slot] = (uint32_t)key;
  } else {
    const uint32_t* lst = sorted + offsets[key];
    for (uint32_t k = 0; k < cnt; k++) a = xyzz_add_affine(a, msm_fetch<CV>(table, lst[k]));
  }
  xyzz_store<CV>(acc + 8 * key, a);
}

// hot keys: one 256-thread block per key, strided partial sums then an LDS tree
template <class CV>
__global__ void __launch_bounds__(256)
msm_hot_kernel(const U128* __restrict__ table, const uint32_t* __restrict__ sorted,
               const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts, U128* __restrict__ acc,
               const uint32_t* __restrict__ hot_list, const uint32_t* __restrict__ hot_count) {
  __shared__ U128 sh[256 * 8];
  const uint32_t nhot = *hot_count;
  for (uint32_t h = blockIdx.x; h < nhot; h += gridDim.x) {
    const uint32_t key = hot_list[h];
    const uint32_t cnt = counts[key];
    const uint32_t* lst = sorted + offsets[key];
    Xyzz<CV> a = Xyzz<CV>::identity();
    for (uint32_t k = threadIdx.x; k < cnt; k += 256) a = xyzz_add_affine(a, msm_fetch<CV>(table, lst[k]));
    xyzz_store<CV>(sh + 8 * threadIdx.x, a);
    __syncthreads();
    for (uint32_t st = 128; st > 0; st >>= 1) {
      if (threadIdx.x < st) {
        Xyzz<CV> x = xyzz_load<CV>(sh + 8 * threadIdx.x), y = xyzz_load<CV>(sh + 8 * (threadIdx.x + st));
        xyzz_store<CV>(sh + 8 * threadIdx.x, xyzz_add(x, y));
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      Xyzz<CV> r = xyzz_load<CV>(sh);
      xyzz_store<CV>(acc + 8 * (size_t)key, r);
    }
    __syncthreads();
  }
}

// ---- bucket weights: P[col][b] = (b+1) * sum_w acc[col][w][b] ---------------------------------
template <class CV>
__global__ void __launch_bounds__(256)
msm_weight_kernel(const U128* __restrict__ acc, U128* __restrict__ weighted, MsmGeom g) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t col = blockIdx.y;
  if (b >= g.B) return;
  Xyzz<CV> x = Xyzz<CV>::identity();
  for (uint32_t w = 0; w < g.W; w++) x = xyzz_add(x, xyzz_load<CV>(acc + 8 * (((size_t)col * g.W + w) * g.B + b)));
  // (b+1) * x, MSB-first double-and-add
  Xyzz<CV> r = Xyzz<CV>::identity();
  if (!x.is_identity()) {
    const uint32_t k = b + 1;
    int top = 31 - __clz(k);
    r = x;
    for (int bit = top - 1; bit >= 0; bit--) {
      r = xyzz_double(r);
      if ((k >> bit) & 1) r = xyzz_add(r, x);
    }
  }
  xyzz_store<CV>(weighted + 8 * ((size_t)col * g.B + b), r);
}

// ---- tree sum: out[col][blockIdx.x] = sum of up to MSM_TREE_SEG points of in[col][...] ----------
template <class CV>
__global__ void __launch_bounds__(256)
msm_tree_sum_kernel(const U128* __restrict__ in, U128* __restrict__ out, uint32_t count /* per column */,
                    uint32_t out_per_col) {
  __shared__ U128 sh[256 * 8];
  const uint32_t col = blockIdx.y;
  const uint32_t base = blockIdx.x * MSM_TREE_SEG;
  Xyzz<CV> a = Xyzz<CV>::identity();
  for (uint32_t k = threadIdx.x; k < MSM_TREE_SEG; k += 256) {
    const uint32_t idx = base + k;
    if (idx < count) a = xyzz_add(a, xyzz_load<CV>(in + 8 * ((size_t)col * count + idx)));
  }
  xyzz_store<CV>(sh + 8 * threadIdx.x, a);
  __syncthreads();
  for (uint32_t st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) {
      Xyzz<CV> x = xyzz_load<CV>(sh + 8 * threadIdx.x), y = xyzz_load<CV>(sh + 8 * (threadIdx.x + st));
      xyzz_store<CV>(sh + 8 * threadIdx.x, xyzz_add(x, y));
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    Xyzz<CV> r = xyzz_load<CV>(sh);
    xyzz_store<CV>(out + 8 * ((size_t)col * out_per_col + blockIdx.x), r);
  }
}

// ---- finish: XYZZ -> Jacobian (m points) ---------------------------------------------------------
template <class CV>
__global__ void msm_to_jacobian_kernel(const U128* __restrict__ in, U128* __restrict__ out_jac, uint32_t m) {
  const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= m) return;
  using B = typename CV::Base;
  Xyzz<CV> p = xyzz_load<CV>(in + 8 * (size_t)col);
  Fe<B> x, y, z;
  xyzz_to_jacobian(p, x, y, z);
  fe_store<B>(out_jac + 6 * (size_t)col, x);
  fe_store<B>(out_jac + 6 * (size_t)col + 2, y);
  fe_store<B>(out_jac + 6 * (size_t)col + 4, z);
}
// XYZZ -> affine (m points), for h2_msm_batch's normalised output
template <class CV>
__global__ void msm_to_affine_kernel(const U128* __restrict__ in, U128* __restrict__ out_aff, uint32_t m) {
  const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= m) return;
  using B = typename CV::Base;
  Affine<CV> a = xyzz_to_affine(xyzz_load<CV>(in + 8 * (size_t)col));
  fe_store<B>(out_aff + 4 * (size_t)col, a.x);
  fe_store<B>(out_aff + 4 * (size_t)col + 2, a.y);
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

// ---- workspace layout -------------------------------------------------------------------------
struct MsmWorkspace {
  size_t K;             // keys = m * W * B
  size_t E;             // max entries = m * W * n
  size_t nblk;          // scan blocks
  uint32_t lvl1;        // partials per column after the first tree level
  size_t off_digits, off_counts, off_offsets, off_cursor, off_blocksums, off_sorted, off_hot, off_misc, off_acc,
      off_weighted, off_tree1, off_tree2, total;
};
inline size_t h2_align256(size_t x) { return (x + 255) & ~(size_t)255; }
inline MsmWorkspace msm_workspace(size_t n, size_t m, const MsmGeom& g) {
  MsmWorkspace ws{};
  ws.K = m * g.W * g.B;
  ws.E = m * g.W * n;
  ws.nblk = (ws.K + SCAN_BLOCK - 1) / SCAN_BLOCK;
  ws.lvl1 = (g.B + MSM_TREE_SEG - 1) / MSM_TREE_SEG;
  size_t o = 0;
  ws.off_digits = o; o = h2_align256(o + ws.E * 4);
  ws.off_counts = o; o = h2_align256(o + ws.K * 4);
  ws.off_offsets = o; o = h2_align256(o + ws.K * 4);
  ws.off_cursor = o; o = h2_align256(o + ws.K * 4);
  ws.off_blocksums = o; o = h2_align256(o + (ws.nblk + 1) * 4);
  ws.off_sorted = o; o = h2_align256(o + ws.E * 4);
  ws.off_hot = o; o = h2_align256(o + (ws.E / MSM_HOT + 16) * 4);
  ws.off_misc = o; o = h2_align256(o + 64);  // [0] = hot_count, [1] = total entries
  ws.off_acc = o; o = h2_align256(o + ws.K * 128);
  ws.off_weighted = o; o = h2_align256(o + m * g.B * 128);
  ws.off_tree1 = o; o = h2_align256(o + m * ws.lvl1 * 128);
  ws.off_tree2 = o; o = h2_align256(o + m * 128);
  ws.total = o;
  return ws;
}

// Enqueue m MSMs of n terms against `table` (built for n_bases points with geometry g).
// Result: m XYZZ points at ws_base + off_tree2.
template <class CV>
inline hipError_t msm_launch(const U128* table, uint32_t n_bases, const U128* d_scalars, size_t n, size_t m,
                             const MsmGeom& g, char* ws_base, const MsmWorkspace& ws, hipStream_t stream,
                             hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
  uint32_t* digits = (uint32_t*)(ws_base + ws.off_digits);
  uint32_t* counts = (uint32_t*)(ws_base + ws.off_counts);
  uint32_t* offsets = (uint32_t*)(ws_base + ws.off_offsets);
  uint32_t* cursor = (uint32_t*)(ws_base + ws.off_cursor);
  uint32_t* blocksums = (uint32_t*)(ws_base + ws.off_blocksums);
  uint32_t* sorted = (uint32_t*)(ws_base + ws.off_sorted);
  uint32_t* hot = (uint32_t*)(ws_base + ws.off_hot);
  uint32_t* misc = (uint32_t*)(ws_base + ws.off_misc);
  U128* acc = (U128*)(ws_base + ws.off_acc);
  U128* weighted = (U128*)(ws_base + ws.off_weighted);
  U128* tree1 = (U128*)(ws_base + ws.off_tree1);
  U128* tree2 = (U128*)(ws_base + ws.off_tree2);
  hipError_t e;
  if ((e = hipMemsetAsync(counts, 0, ws.K * 4, stream)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(misc, 0, 64, stream)) != hipSuccess) return e;
  const uint32_t nb_n = (uint32_t)((n + 255) / 256);
  hipLaunchKernelGGL(msm_digits_kernel<CV>, dim3(nb_n, (unsigned)m), dim3(256), 0, stream, d_scalars, digits, counts,
                     (uint32_t)n, n, g);
  hipLaunchKernelGGL(scan_reduce_kernel, dim3((unsigned)ws.nblk), dim3(256), 0, stream, counts, blocksums, ws.K);
  hipLaunchKernelGGL(scan_blocksums_kernel, dim3(1), dim3(1024), 0, stream, blocksums, (uint32_t)ws.nblk, misc + 1);
  hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)ws.nblk), dim3(256), 0, stream, counts, blocksums, offsets,
                     cursor, ws.K);
  hipLaunchKernelGGL(msm_scatter_kernel, dim3(nb_n, (unsigned)(m * g.W)), dim3(256), 0, stream, digits, cursor,
                     sorted, (uint32_t)n, n_bases, g);
  if (ev_start) (void)hipEventRecord(ev_start, stream);
  hipLaunchKernelGGL(msm_accumulate_kernel<CV>, dim3((unsigned)((ws.K + 255) / 256)), dim3(256), 0, stream, table,
                     sorted, offsets, counts, acc, ws.K, hot, misc);
  if (ev_stop) (void)hipEventRecord(ev_stop, stream);
  hipLaunchKernelGGL(msm_hot_kernel<CV>, dim3(256), dim3(256), 0, stream, table, sorted, offsets, counts, acc, hot,
                     misc);
  hipLaunchKernelGGL(msm_weight_kernel<CV>, dim3((g.B + 255) / 256, (unsigned)m), dim3(256), 0, stream, acc, weighted,
                     g);
  hipLaunchKernelGGL(msm_tree_sum_kernel<CV>, dim3(ws.lvl1, (unsigned)m), dim3(256), 0, stream, weighted, tree1, g.B,
                     ws.lvl1);
  hipLaunchKernelGGL(msm_tree_sum_kernel<CV>, dim3(1, (unsigned)m), dim3(256), 0, stream, tree1, tree2, ws.lvl1, 1u);
  return hipGetLastError();
}

}  // namespace h2
