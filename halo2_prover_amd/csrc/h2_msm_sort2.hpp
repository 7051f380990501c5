// h2_msm_sort2.hpp -- the MSM's bucket sort in two LDS-staged levels (included by h2_msm.hpp).
//
// Contract (the accumulate kernel's input, unchanged): offsets[key] for key = col * B + bucket (offsets[K] = E, the
// number of non-zero digits) and sorted_ref[E] = (w * n_bases + i) | sign, every bucket's entries contiguous.
//
// Why two levels.  With B = 2^15 buckets (n = 2^20) a tile of scalars has less than one entry per bucket, so whatever
// the first-generation scatter did in LDS its stores left as single 4-byte words, each a 32-byte write at the memory
// side (MI355X_MICROARCH.md, stores of each flavour): 266 us and 8x write amplification for 67 MB of entries, behind a
// 110 us count kernel whose 128 KB histogram allowed one 1024-thread block per CU on half the CUs
// (profiles/r03_msm_2e20_pallas_baseline_kernel_stats.csv).  Here every store leaves LDS in runs:
//   level 1  (msm2_count_kernel, msm2_coarse_scan_kernel, msm2_scatter_kernel)  the key's HIGH bits: Hc <= 1024 coarse
//            bins per column.  A tile's entries are placed in LDS in bin order and written so that lane j of a store
//            instruction holds staged entry j: the ~16 entries of a (tile, bin) run leave as one 64-byte run.  Entry =
//            the final 4-byte word in `mid_ref`; the key's low bits ride in its spare bits below the sign bit when
//            W * n_bases leaves room for them (a 2^20-point SRS: 24 + 5 bits), else in the byte array `mid_lo`.
//   level 2  (msm2_fine_kernel)  one block per coarse bin: its entries (contiguous in mid_*) are placed in LDS in
//            bucket order, a slab of S2_SLAB at a time, and written to sorted_ref with fully coalesced runs; the same
//            block writes the bin's slice of `offsets` (bin start + prefix of its F fine counts), so no global scan
//            over the K keys is needed.  A bin of any size is correct (more slabs); uniform scalars give bins of
//            about one slab.
// The two levels read the scalars twice (count, scatter: 32 bytes per term each) and move 5 + 5 + 4 bytes per entry.
#pragma once

namespace h2 {

constexpr uint32_t S2_THREADS = 1024;
constexpr uint32_t S2_MAX_HC = 1024;        // coarse bins per column
constexpr uint32_t S2_MAX_H = 8192;         // coarse bins per launch (m * Hc): the one-block scan's reach
constexpr uint32_t S2_STAGE = 16384;        // entries a level-1 tile stages in LDS (7 bytes each)
constexpr uint32_t S2_SLAB = 24576;         // entries a level-2 block stages at a time (5 bytes each)
constexpr uint32_t S2_MAX_F = 256;          // fine buckets per coarse bin

constexpr uint32_t S2_GROUP = 4;            // level-1 tiles counted by one block at most (one global atomic per bin for all of them)
struct Sort2Geom {
  uint32_t lo_bits;    // key bits resolved by level 2
  uint32_t F;          // fine buckets per coarse bin = 2^lo_bits
  uint32_t Hc;         // coarse bins per column = B / F
  uint32_t tile;       // scalars per level-1 tile
  uint32_t tiles;      // level-1 tiles per column
  uint32_t group;      // level-1 tiles per count block (<= S2_GROUP)
  uint32_t groups;     // count blocks per column = ceil(tiles / group)
  uint32_t pack_shift; // != 0: the key's low bits travel in bits [pack_shift, pack_shift + lo_bits) of the entry itself
                       // (free below the sign bit when W * n_bases < 2^pack_shift); 0: in the byte array mid_lo
};

inline Sort2Geom msm_sort2_geom(size_t n, const MsmGeom& g, size_t n_bases = 0) {
  Sort2Geom s{};
  uint32_t hc = g.B < S2_MAX_HC ? g.B : S2_MAX_HC;
  s.Hc = hc;
  s.F = g.B / hc;
  s.lo_bits = 0;
  while ((1u << s.lo_bits) < s.F) s.lo_bits++;
  size_t tile = (size_t)tune_int("H2_TUNE_S2_STAGE", (int)S2_STAGE) / g.W;
  if (tile > S2_THREADS) tile = S2_THREADS;
  if (tile > n) tile = n;
  if (tile < 1) tile = 1;
  s.tile = (uint32_t)tile;
  s.tiles = (uint32_t)((n + tile - 1) / tile);
  s.group = (uint32_t)tune_int("H2_TUNE_S2_GROUP", (int)S2_GROUP);
  if (s.group < 1 || s.group > S2_GROUP) s.group = S2_GROUP;
  s.groups = (s.tiles + s.group - 1) / s.group;
  s.pack_shift = 0;
  if (n_bases) {
    uint32_t bits = 0;
    while (((uint64_t)1 << bits) < (uint64_t)g.W * n_bases) bits++;
    if (bits + s.lo_bits <= 31 && s.lo_bits > 0) s.pack_shift = bits;
  }
  return s;
}
inline size_t msm_sort2_lds_scatter(const Sort2Geom& s, const MsmGeom& g) {
  const size_t cap = (size_t)s.tile * g.W;
  return (size_t)s.Hc * 8 + cap * 4 + cap * 2 + (s.pack_shift ? 0 : ((cap + 3) & ~(size_t)3)) + 64;
}
inline size_t msm_sort2_lds_fine() { return (size_t)S2_SLAB * 5 + 64; }

// ---- level 1, count: block (group, col) counts the entries per coarse bin of S2_GROUP consecutive tiles, one after the
// other: tile_cnt[tile][bin], tile_base[tile][bin] = entries of the group's earlier tiles in that bin, and ONE returning
// global atomic per non-empty (group, bin) hands the group its base inside the bin (group_base).  (One atomic per tile
// and bin -- 2^20 contended atomics for a 2^20-term column -- made this kernel 51 us.)  grid (groups, m), LDS 2 Hc words.
template <class CV>
__global__ void __launch_bounds__(S2_THREADS)
msm2_count_kernel(const U128* __restrict__ scalars, uint32_t* __restrict__ gcount, uint32_t* __restrict__ tile_base,
                  uint32_t* __restrict__ tile_cnt, uint32_t* __restrict__ group_base, uint32_t n,
                  size_t col_stride /* elements */, Sort2Geom s, MsmGeom g) {
  using S = typename CV::Scalar;
  extern __shared__ uint32_t s2_lds[];
  uint32_t* hist = s2_lds;
  uint32_t* run = s2_lds + s.Hc;
  const uint32_t col = blockIdx.y;
  for (uint32_t b = threadIdx.x; b < s.Hc; b += blockDim.x) {
    hist[b] = 0;
    run[b] = 0;
  }
  __syncthreads();
  // the scalars of all the group's tiles are requested up front (one per thread and tile)
  Fe<S> sc[S2_GROUP];
#pragma unroll
  for (uint32_t k = 0; k < S2_GROUP; k++) {
    const uint32_t tile = blockIdx.x * s.group + k;
    const uint32_t i = tile * s.tile + threadIdx.x;
    if (k < s.group && tile < s.tiles && threadIdx.x < s.tile && i < n) sc[k] = fe_load<S>(scalars + 2 * (col_stride * col + i));
    else sc[k] = Fe<S>::zero();
  }
#pragma unroll
  for (uint32_t k = 0; k < S2_GROUP; k++) {
    const uint32_t tile = blockIdx.x * s.group + k;
    if (k >= s.group || tile >= s.tiles) break;
    {
      MsmDigits dg(fe_from_mont(sc[k]).v);           // zero (no scalar here) has no digits
      for (uint32_t w = 0; w < g.W; w++) {
        const uint32_t enc = dg.next(g, w);
        if (enc) atomicAdd(&hist[((enc & ~MSM_SIGN) - 1) >> s.lo_bits], 1u);
      }
    }
    __syncthreads();
    const size_t row = ((size_t)col * s.tiles + tile) * s.Hc;
    for (uint32_t b = threadIdx.x; b < s.Hc; b += blockDim.x) {
      const uint32_t h = hist[b], r = run[b];
      tile_cnt[row + b] = h;
      tile_base[row + b] = r;
      run[b] = r + h;
      hist[b] = 0;
    }
    __syncthreads();
  }
  const size_t grow = ((size_t)col * s.groups + blockIdx.x) * s.Hc;
  uint32_t* gc = gcount + (size_t)col * s.Hc;
  for (uint32_t b = threadIdx.x; b < s.Hc; b += blockDim.x) {
    const uint32_t r = run[b];
    group_base[grow + b] = r ? atomicAdd(&gc[b], r) : 0u;
  }
}

// ---- level 1, scan: cstart[h] = exclusive prefix of gcount over the H = m * Hc coarse bins (H <= S2_MAX_H), one
// block; cstart[H] = E is also stored to offsets[K]
static __global__ void __launch_bounds__(1024)
msm2_coarse_scan_kernel(const uint32_t* __restrict__ gcount, uint32_t* __restrict__ cstart, uint32_t H,
                        uint32_t* __restrict__ offsets_last) {
  __shared__ uint32_t wave_sum[16];
  const uint32_t per = (H + 1023) / 1024;                  // <= 8
  const uint32_t lo = min(H, threadIdx.x * per), hi = min(H, lo + per);
  uint32_t v[S2_MAX_H / 1024], sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < S2_MAX_H / 1024; k++) {
    v[k] = lo + k < hi ? gcount[lo + k] : 0u;
    sum += v[k];
  }
  uint32_t incl = sum;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wave_sum[wave] = incl;
  __syncthreads();
  uint32_t run = incl - sum;
  for (uint32_t w = 0; w < wave; w++) run += wave_sum[w];
#pragma unroll
  for (uint32_t k = 0; k < S2_MAX_H / 1024; k++) {
    if (lo + k < hi) cstart[lo + k] = run;
    run += v[k];
  }
  if (threadIdx.x == 1023) {
    cstart[H] = run;
    *offsets_last = run;
  }
}

// ---- level 1, scatter: the tile's entries staged in LDS in coarse-bin order, written in that order.
// LDS: cur[Hc], delta[Hc], sref[cap] (u32), sbin[cap] (u16), slo[cap] (u8); cap = tile * W.  grid (tiles, m).
template <class CV>
__global__ void __launch_bounds__(S2_THREADS)
msm2_scatter_kernel(const U128* __restrict__ scalars, const uint32_t* __restrict__ cstart,
                    const uint32_t* __restrict__ tile_base, const uint32_t* __restrict__ tile_cnt,
                    const uint32_t* __restrict__ group_base, uint32_t* __restrict__ mid_ref, uint8_t* __restrict__ mid_lo,
                    uint32_t n, size_t col_stride, uint32_t n_bases, Sort2Geom s, MsmGeom g) {
  using S = typename CV::Scalar;
  extern __shared__ uint32_t s2_lds[];
  __shared__ uint32_t wave_sum[16];
  __shared__ uint32_t total_s;
  const uint32_t cap = s.tile * g.W;
  uint32_t* cur = s2_lds;
  uint32_t* delta = s2_lds + s.Hc;
  uint32_t* sref = s2_lds + 2 * (size_t)s.Hc;
  uint16_t* sbin = reinterpret_cast<uint16_t*>(sref + cap);
  uint8_t* slo = reinterpret_cast<uint8_t*>(sbin + cap);
  const uint32_t col = blockIdx.y, tile = blockIdx.x;
  const size_t row = ((size_t)col * s.tiles + tile) * s.Hc;
  const size_t grow = ((size_t)col * s.groups + tile / s.group) * s.Hc;
  // exclusive scan of the tile's counts over the bins: thread t owns `per` consecutive bins
  const uint32_t per = (s.Hc + blockDim.x - 1) / blockDim.x;
  const uint32_t b_lo = min(s.Hc, threadIdx.x * per), b_hi = min(s.Hc, b_lo + per);
  uint32_t sum = 0;
  for (uint32_t b = b_lo; b < b_hi; b++) sum += tile_cnt[row + b];
  uint32_t incl = sum;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wave_sum[wave] = incl;
  __syncthreads();
  uint32_t run = incl - sum;
  for (uint32_t w = 0; w < wave; w++) run += wave_sum[w];
  for (uint32_t b = b_lo; b < b_hi; b++) {
    cur[b] = run;
    delta[b] = cstart[(size_t)col * s.Hc + b] + group_base[grow + b] + tile_base[row + b] - run;   // modulo 2^32
    run += tile_cnt[row + b];
  }
  if (threadIdx.x == blockDim.x - 1) total_s = run;
  __syncthreads();
  const uint32_t lo = tile * s.tile, hi = min(lo + s.tile, n);
  const uint32_t lo_mask = s.F - 1;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    MsmDigits dg(fe_from_mont(fe_load<S>(scalars + 2 * (col_stride * col + i))).v);
    for (uint32_t w = 0; w < g.W; w++) {
      const uint32_t enc = dg.next(g, w);
      if (enc) {
        const uint32_t b = (enc & ~MSM_SIGN) - 1, hb = b >> s.lo_bits;
        const uint32_t pos = atomicAdd(&cur[hb], 1u);
        if (pos < cap) {                   // always: the counts come from the same digits (msm2_count_kernel)
          const uint32_t ref = (w * n_bases + i) | (enc & MSM_SIGN);
          sbin[pos] = (uint16_t)hb;
          if (s.pack_shift) {
            sref[pos] = ref | ((b & lo_mask) << s.pack_shift);
          } else {
            sref[pos] = ref;
            slo[pos] = (uint8_t)(b & lo_mask);
          }
        }
      }
    }
  }
  __syncthreads();
  const uint32_t total = min(total_s, cap);
  for (uint32_t j = threadIdx.x; j < total; j += blockDim.x) {
    const uint32_t at = delta[sbin[j]] + j;
    mid_ref[at] = sref[j];
    if (!s.pack_shift) mid_lo[at] = slo[j];
  }
}

// ---- level 2: one block per coarse bin h (grid H).  The bin's entries [cstart[h], cstart[h+1]) of mid_* go to the
// same range of sorted_ref in bucket order; offsets[h * F + f] = cstart[h] + (entries of the bin's buckets below f).
// LDS: out[S2_SLAB] (u32), fo[S2_SLAB] (u8).
static __global__ void __launch_bounds__(S2_THREADS)
msm2_fine_kernel(const uint32_t* __restrict__ cstart, const uint32_t* __restrict__ mid_ref,
                 const uint8_t* __restrict__ mid_lo, uint32_t* __restrict__ sorted_ref, uint32_t* __restrict__ offsets,
                 uint32_t F, uint32_t lo_bits, uint32_t pack_shift) {
  extern __shared__ uint32_t s2_lds[];
  __shared__ uint32_t cnt[S2_MAX_F];       // the bin's entries per fine bucket
  __shared__ uint32_t gpos[S2_MAX_F];      // where the next run of bucket f goes in sorted_ref
  __shared__ uint32_t lcnt[S2_MAX_F];      // the slab's entries per fine bucket
  __shared__ uint32_t lstart[S2_MAX_F];    // their exclusive prefix
  uint32_t* out = s2_lds;
  uint8_t* fo = reinterpret_cast<uint8_t*>(s2_lds + S2_SLAB);
  const uint32_t h = blockIdx.x;
  const uint32_t start = cstart[h], end = cstart[h + 1], size = end - start;
  const bool one_slab = size <= S2_SLAB;
  const uint32_t lo_mask = F - 1, clear = ~(pack_shift ? lo_mask << pack_shift : 0u);
  if (threadIdx.x < F) {
    cnt[threadIdx.x] = 0;
    lcnt[threadIdx.x] = 0;
  }
  __syncthreads();
  if (!one_slab) {
    // pass A over the whole bin: the fine counts (a single slab gets them from its own count below)
    for (uint32_t e = start + threadIdx.x; e < end; e += blockDim.x)
      atomicAdd(&cnt[pack_shift ? (mid_ref[e] >> pack_shift) & lo_mask : (uint32_t)mid_lo[e]], 1u);
    __syncthreads();
  }
  constexpr uint32_t PER = S2_SLAB / S2_THREADS;   // 24 entries per thread and slab
  for (uint32_t s0 = start; s0 < end || s0 == start; s0 += S2_SLAB) {
    const uint32_t s1 = min(end, s0 + S2_SLAB), ssize = s1 - s0;
    // the slab's entries into registers; their rank inside (slab, bucket) from a returning LDS atomic
    uint32_t ref[PER], lf[PER], rank[PER];
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
      const uint32_t e = s0 + threadIdx.x + k * S2_THREADS;
      lf[k] = 0xFFFFFFFFu;
      if (e < s1) {
        const uint32_t word = mid_ref[e];
        lf[k] = pack_shift ? (word >> pack_shift) & lo_mask : (uint32_t)mid_lo[e];
        ref[k] = word;
        rank[k] = atomicAdd(&lcnt[lf[k]], 1u);
      }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      // exclusive scan of lcnt over F <= 256 buckets by one wave (4 per lane)
      uint32_t v[4], sum = 0;
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        const uint32_t f = threadIdx.x * 4 + k;
        v[k] = f < F ? lcnt[f] : 0u;
        sum += v[k];
      }
      uint32_t incl = sum;
      for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d, 64);
        if (threadIdx.x >= d) incl += t;
      }
      uint32_t run = incl - sum;
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        const uint32_t f = threadIdx.x * 4 + k;
        if (f < F) {
          lstart[f] = run;
          if (s0 == start) {
            // first slab: the bin's bucket starts.  One slab: its counts are the bin's; else pass A's are
            if (one_slab) {
              gpos[f] = start + run;
              offsets[((size_t)h << lo_bits) + f] = start + run;
            }
          }
        }
        run += v[k];
      }
    }
    if (!one_slab && s0 == start && threadIdx.x >= 64 && threadIdx.x < 128) {
      // the bin's bucket starts from pass A's counts, by the second wave
      const uint32_t t = threadIdx.x - 64;
      uint32_t v[4], sum = 0;
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        const uint32_t f = t * 4 + k;
        v[k] = f < F ? cnt[f] : 0u;
        sum += v[k];
      }
      uint32_t incl = sum;
      for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t u = __shfl_up(incl, d, 64);
        if (t >= d) incl += u;
      }
      uint32_t run = incl - sum;
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        const uint32_t f = t * 4 + k;
        if (f < F) {
          gpos[f] = start + run;
          offsets[((size_t)h << lo_bits) + f] = start + run;
        }
        run += v[k];
      }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
      if (lf[k] != 0xFFFFFFFFu) {
        const uint32_t pos = lstart[lf[k]] + rank[k];
        out[pos] = ref[k];
        if (!pack_shift) fo[pos] = (uint8_t)lf[k];
      }
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < ssize; j += blockDim.x) {
      const uint32_t word = out[j];
      const uint32_t f = pack_shift ? (word >> pack_shift) & lo_mask : (uint32_t)fo[j];
      sorted_ref[gpos[f] + (j - lstart[f])] = word & clear;
    }
    __syncthreads();
    if (threadIdx.x < F) {
      gpos[threadIdx.x] += lcnt[threadIdx.x];
      lcnt[threadIdx.x] = 0;
    }
    __syncthreads();
    if (s1 >= end) break;
  }
}

}  // namespace h2
