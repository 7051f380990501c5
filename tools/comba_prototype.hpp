// prototype: Montgomery product-scanning (FIPS) with in-place v_mad_u64_u32 accumulation
#pragma once
namespace h2x {
using namespace h2;
// acc(64) += a*b ; cnt += carry-out
__device__ __forceinline__ void mac_vv(uint64_t& acc, uint32_t& cnt, uint32_t a, uint32_t b) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(cnt) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ void mac_vs(uint64_t& acc, uint32_t& cnt, uint32_t a, uint32_t b_const) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(cnt) : "v"(a), "s"(b_const) : "vcc");
}
__device__ __forceinline__ void add64(uint64_t& acc, uint32_t& cnt, uint64_t x) {
  uint64_t n = acc + x;
  cnt += (n < x) ? 1u : 0u;
  acc = n;
}
template <class FP, int J>
__device__ __forceinline__ void mac_modulus(uint64_t& acc, uint32_t& cnt, uint32_t m) {
  constexpr uint32_t pj = FP::P(J);
  if constexpr (pj == 0) {
  } else if constexpr (pj == 1) {
    add64(acc, cnt, (uint64_t)m);
  } else if constexpr ((pj & (pj - 1)) == 0) {
    constexpr int sh = __builtin_ctz(pj);
    add64(acc, cnt, (uint64_t)m << sh);
  } else {
    mac_vs(acc, cnt, m, pj);
  }
}
template <class FP, int K, int I>
__device__ __forceinline__ void col_ab(uint64_t& acc, uint32_t& cnt, const uint32_t* a, const uint32_t* b) {
  if constexpr (I <= (K < 8 ? K : 7)) {
    mac_vv(acc, cnt, a[I], b[K - I]);
    col_ab<FP, K, I + 1>(acc, cnt, a, b);
  }
}
template <class FP, int K, int I, int IEND>
__device__ __forceinline__ void col_mp(uint64_t& acc, uint32_t& cnt, const uint32_t* m) {
  if constexpr (I <= IEND) {
    mac_modulus<FP, K - I>(acc, cnt, m[I]);
    col_mp<FP, K, I + 1, IEND>(acc, cnt, m);
  }
}
template <class FP, int K>
__device__ __forceinline__ void columns(uint64_t& acc, uint32_t& cnt, const uint32_t* a, const uint32_t* b, uint32_t* m, uint32_t* t) {
  if constexpr (K < 16) {
    constexpr int i0 = K < 8 ? 0 : K - 7;
    col_ab<FP, K, i0>(acc, cnt, a, b);
    if constexpr (K < 8) {
      col_mp<FP, K, 0, K - 1>(acc, cnt, m);
      m[K] = (uint32_t)acc * FP::INV;
      mac_modulus<FP, 0>(acc, cnt, m[K]);
    } else {
      if constexpr (K < 15) col_mp<FP, K, K - 7, 7>(acc, cnt, m);
      t[K - 8] = (uint32_t)acc;
    }
    acc = (acc >> 32) | ((uint64_t)cnt << 32);
    cnt = 0;
    columns<FP, K + 1>(acc, cnt, a, b, m, t);
  }
}
template <class FP>
__device__ __forceinline__ Fe<FP> fe_mul_comba(const Fe<FP>& a, const Fe<FP>& b) {
  uint64_t acc = 0;
  uint32_t cnt = 0;
  uint32_t m[8], t[9];
  columns<FP, 0>(acc, cnt, a.v, b.v, m, t);
  // K = 15 handled: t[7]; remaining carry
  t[8] = (uint32_t)acc;
  Fe<FP> r;
  fe_reduce_once<FP>(r.v, t, t[8]);
  return r;
}
}  // namespace h2x
