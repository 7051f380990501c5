// h2_host.hpp -- host-side pieces of the product surface (h2_prover.hip): scalar field arithmetic, hex / byte
// conversions, Blake2b, the Fiat-Shamir transcript.
//
// The reference gets these from its dependencies (halo2curves 0.3.2 bn256::{Fr, Fq}; blake2b_simd through
// halo2_proofs::transcript::{Blake2bWrite, Blake2bRead}, used at /root/reference/circuits/src/utils.rs:79-80,
// 103-104,132,147); their behaviour on the proof stream is restated in SURVEY.md App. A.4-A.5.  The field code is
// the HOST instantiation of the same __host__ __device__ templates the kernels use (h2_field.hpp).
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "h2_field.hpp"

namespace h2 {

// ---- a field element on the host: Montgomery form inside, canonical integers at the edges -------------------
template <class FP>
struct HF {
  Fe<FP> v;
  HF() : v(Fe<FP>::zero()) {}
  explicit HF(const Fe<FP>& x) : v(x) {}
  static HF zero() { return HF(); }
  static HF one() { return HF(Fe<FP>::one()); }
  static HF from_u64(uint64_t x) {
    Fe<FP> t = Fe<FP>::zero();
    t.v[0] = (uint32_t)x;
    t.v[1] = (uint32_t)(x >> 32);
    return HF(fe_to_mont(t));
  }
  // 32 little-endian bytes holding any integer below 2^256 -> the element it reduces to
  static HF from_le_bytes_reduce(const uint8_t* b) {
    Fe<FP> t;
    memcpy(t.v, b, 32);
    // below 2^256 < 6p: a few conditional subtractions make it canonical
    for (int k = 0; k < 6; k++) {
      uint32_t r[8];
      fe_reduce_once<FP>(r, t.v, 0);
      memcpy(t.v, r, 32);
    }
    return HF(fe_to_mont(t));
  }
  // canonical only: false when the integer is >= p
  static bool from_le_bytes_canonical(const uint8_t* b, HF* out) {
    Fe<FP> t;
    memcpy(t.v, b, 32);
    for (int i = 7; i >= 0; i--) {
      if (t.v[i] < FP::P(i)) break;
      if (t.v[i] > FP::P(i) || i == 0) return false;
    }
    *out = HF(fe_to_mont(t));
    return true;
  }
  // 64 little-endian bytes (a 512-bit integer) mod p: Fr::from_bytes_wide / Fr::random / Challenge255
  static HF from_le_bytes_wide(const uint8_t* b) {
    HF lo = from_le_bytes_reduce(b), hi = from_le_bytes_reduce(b + 32);
    return lo + hi * HF(Fe<FP>::one()).mont_r();
  }
  // 2^256 mod p as an element: ONE holds R mod p as a plain integer, i.e. the Montgomery form of 1; the element
  // "R" has Montgomery form R^2 mod p
  HF mont_r() const {
    Fe<FP> r2;
    for (int i = 0; i < 8; i++) r2.v[i] = FP::R2(i);
    return HF(r2);
  }
  static HF from_hex(const char* s) {
    if (s[0] == '0' && (s[1] == 'x' || s[1] == 'X')) s += 2;
    uint8_t b[64];
    memset(b, 0, sizeof b);
    const size_t len = strlen(s);
    if (len > 128) throw std::invalid_argument("hex literal too long");
    for (size_t i = 0; i < len; i++) {
      const char c = s[len - 1 - i];
      int d = (c >= '0' && c <= '9') ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : (c >= 'A' && c <= 'F') ? c - 'A' + 10 : -1;
      if (d < 0) throw std::invalid_argument("bad hex digit");
      b[i / 2] |= (uint8_t)(d << (4 * (i & 1)));
    }
    return from_le_bytes_wide(b);
  }
  void to_le_bytes(uint8_t* out) const {
    const Fe<FP> c = fe_from_mont(v);
    memcpy(out, c.v, 32);
  }
  // the Montgomery limbs as they sit in device memory / across the C ABI
  void mont_limbs(uint64_t out[4]) const { memcpy(out, v.v, 32); }
  static HF from_mont_limbs(const void* p) {
    HF r;
    memcpy(r.v.v, p, 32);
    return r;
  }
  std::string hex64() const {  // 64 lower-case hex digits, big-endian: Rust's {:?} of a field element after "0x"
    uint8_t b[32];
    to_le_bytes(b);
    static const char* d = "0123456789abcdef";
    std::string s(64, '0');
    for (int i = 0; i < 32; i++) {
      s[63 - 2 * i] = d[b[i] & 15];
      s[62 - 2 * i] = d[b[i] >> 4];
    }
    return s;
  }
  bool is_zero() const { return v.is_zero(); }
  bool is_odd() const { return fe_from_mont(v).v[0] & 1; }
  bool operator==(const HF& o) const { return v == o.v; }
  bool operator!=(const HF& o) const { return !(v == o.v); }
  bool operator<(const HF& o) const {  // canonical integer order (BTreeSet<Fr> in the SHPLONK point sets)
    const Fe<FP> a = fe_from_mont(v), b = fe_from_mont(o.v);
    for (int i = 7; i >= 0; i--)
      if (a.v[i] != b.v[i]) return a.v[i] < b.v[i];
    return false;
  }
  HF operator+(const HF& o) const { return HF(fe_add(v, o.v)); }
  HF operator-(const HF& o) const { return HF(fe_sub(v, o.v)); }
  HF operator*(const HF& o) const { return HF(fe_mul(v, o.v)); }
  HF operator-() const { return HF(fe_neg(v)); }
  HF& operator+=(const HF& o) { v = fe_add(v, o.v); return *this; }
  HF& operator-=(const HF& o) { v = fe_sub(v, o.v); return *this; }
  HF& operator*=(const HF& o) { v = fe_mul(v, o.v); return *this; }
  HF sqr() const { return HF(fe_sqr(v)); }
  // 0 -> 0.  Binary extended Euclid on the 256-bit integer (a few microseconds; Fermat's 380 products were most of the
  // verifier's affine Miller loops): for the stored a R it yields a^-1 R^-1, two products with R^2 give a^-1 R.
  HF inv() const {
    typedef unsigned __int128 u128;
    auto load = [](const Fe<FP>& f, uint64_t o[4]) {
      for (int i = 0; i < 4; i++) o[i] = (uint64_t)f.v[2 * i] | ((uint64_t)f.v[2 * i + 1] << 32);
    };
    auto is_one = [](const uint64_t a[4]) { return a[0] == 1 && !(a[1] | a[2] | a[3]); };
    auto geq = [](const uint64_t a[4], const uint64_t b[4]) {
      for (int i = 3; i >= 0; i--)
        if (a[i] != b[i]) return a[i] > b[i];
      return true;
    };
    auto sub = [](uint64_t a[4], const uint64_t b[4]) {       // a -= b, returns the borrow
      u128 br = 0;
      for (int i = 0; i < 4; i++) {
        const u128 d = (u128)a[i] - b[i] - br;
        a[i] = (uint64_t)d;
        br = (d >> 64) & 1;
      }
      return (uint64_t)br;
    };
    auto add = [](uint64_t a[4], const uint64_t b[4]) {
      u128 c = 0;
      for (int i = 0; i < 4; i++) {
        c += (u128)a[i] + b[i];
        a[i] = (uint64_t)c;
        c >>= 64;
      }
    };
    auto shr1 = [](uint64_t a[4]) {
      for (int i = 0; i < 3; i++) a[i] = (a[i] >> 1) | (a[i + 1] << 63);
      a[3] >>= 1;
    };
    uint64_t p[4], u[4], w[4], x1[4] = {1, 0, 0, 0}, x2[4] = {0, 0, 0, 0};
    Fe<FP> pm;
    for (int i = 0; i < 8; i++) pm.v[i] = FP::P(i);
    load(pm, p);
    load(v, u);
    load(pm, w);
    if (!(u[0] | u[1] | u[2] | u[3])) return zero();
    auto halve = [&](uint64_t x[4]) {                          // x / 2 mod p (p < 2^255: x + p fits)
      if (x[0] & 1) add(x, p);
      shr1(x);
    };
    while (!is_one(u) && !is_one(w)) {
      while (!(u[0] & 1)) { shr1(u); halve(x1); }
      while (!(w[0] & 1)) { shr1(w); halve(x2); }
      if (geq(u, w)) {
        sub(u, w);
        if (sub(x1, x2)) add(x1, p);
      } else {
        sub(w, u);
        if (sub(x2, x1)) add(x2, p);
      }
    }
    const uint64_t* r = is_one(u) ? x1 : x2;
    Fe<FP> t, r2;
    for (int i = 0; i < 4; i++) {
      t.v[2 * i] = (uint32_t)r[i];
      t.v[2 * i + 1] = (uint32_t)(r[i] >> 32);
    }
    for (int i = 0; i < 8; i++) r2.v[i] = FP::R2(i);
    return HF(fe_mul(fe_mul(t, r2), r2));
  }
  HF pow_u64(uint64_t e) const { return HF(fe_pow_u64(v, e)); }
  // exponent as 4 x u64 little-endian limbs
  HF pow_limbs(const uint64_t e[4]) const {
    HF acc = one();
    for (int i = 255; i >= 0; i--) {
      acc = acc.sqr();
      if ((e[i >> 6] >> (i & 63)) & 1) acc *= *this;
    }
    return acc;
  }
};
using Fr = HF<BN254_FR>;
using Fq = HF<BN254_FQ>;

inline Fr fr_root_of_unity() {  // ROOT_OF_UNITY = 7^((r-1) / 2^28)
  Fe<BN254_FR> r;
  for (int i = 0; i < 8; i++) r.v[i] = BN254_FR::ROOT(i);
  return Fr(r);
}

// ---- Blake2b (RFC 7693), 64-byte digest, optional 16-byte personalisation, no key -------------------------------
class Blake2b {
 public:
  explicit Blake2b(const char* personal16 = nullptr, size_t outlen = 64) : outlen_(outlen) {
    static const uint64_t IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull,
                                   0xa54ff53a5f1d36f1ull, 0x510e527fade682d1ull, 0x9b05688c2b3e6c1full,
                                   0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    memcpy(h_, IV, sizeof h_);
    h_[0] ^= 0x01010000ull ^ (uint64_t)outlen;
    if (personal16) {
      uint64_t p[2];
      memcpy(p, personal16, 16);
      h_[6] ^= p[0];
      h_[7] ^= p[1];
    }
    t_[0] = t_[1] = 0;
    buflen_ = 0;
  }
  void update(const void* data, size_t len) {
    const uint8_t* in = (const uint8_t*)data;
    while (len) {
      if (buflen_ == 128) {  // a full buffer is compressed only when more input follows (the last block is special)
        t_[0] += 128;
        if (t_[0] < 128) t_[1]++;
        compress(false);
        buflen_ = 0;
      }
      const size_t take = len < 128 - buflen_ ? len : 128 - buflen_;
      memcpy(buf_ + buflen_, in, take);
      buflen_ += take;
      in += take;
      len -= take;
    }
  }
  // digest of everything so far; the object itself is left untouched (the transcript keeps absorbing)
  void digest(uint8_t* out) const {
    Blake2b c = *this;
    c.t_[0] += c.buflen_;
    if (c.t_[0] < c.buflen_) c.t_[1]++;
    memset(c.buf_ + c.buflen_, 0, 128 - c.buflen_);
    c.compress(true);
    memcpy(out, c.h_, outlen_);
  }

 private:
  static uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
  void compress(bool last) {
    static const uint8_t S[12][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
    static const uint64_t IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull,
                                   0xa54ff53a5f1d36f1ull, 0x510e527fade682d1ull, 0x9b05688c2b3e6c1full,
                                   0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    uint64_t m[16], v[16];
    memcpy(m, buf_, 128);
    for (int i = 0; i < 8; i++) {
      v[i] = h_[i];
      v[i + 8] = IV[i];
    }
    v[12] ^= t_[0];
    v[13] ^= t_[1];
    if (last) v[14] = ~v[14];
    auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
      v[a] = v[a] + v[b] + x; v[d] = rotr(v[d] ^ v[a], 32);
      v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 24);
      v[a] = v[a] + v[b] + y; v[d] = rotr(v[d] ^ v[a], 16);
      v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 63);
    };
    for (int r = 0; r < 12; r++) {
      const uint8_t* s = S[r];
      G(0, 4, 8, 12, m[s[0]], m[s[1]]);   G(1, 5, 9, 13, m[s[2]], m[s[3]]);
      G(2, 6, 10, 14, m[s[4]], m[s[5]]);  G(3, 7, 11, 15, m[s[6]], m[s[7]]);
      G(0, 5, 10, 15, m[s[8]], m[s[9]]);  G(1, 6, 11, 12, m[s[10]], m[s[11]]);
      G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
    for (int i = 0; i < 8; i++) h_[i] ^= v[i] ^ v[i + 8];
  }
  uint64_t h_[8], t_[2];
  uint8_t buf_[128];
  size_t buflen_, outlen_;
};

// ---- affine G1 point on the host: canonical-form coordinates in Fq, identity flag ----------------------------
struct G1 {
  Fq x, y;
  bool inf = true;
  bool operator==(const G1& o) const { return inf == o.inf && (inf || (x == o.x && y == o.y)); }
};

// Blake2bWrite / Blake2bRead with Challenge255: one running state, never reset;
// scalars are absorbed as 0x02 || canonical LE, points as 0x01 || x LE || y LE, a squeeze absorbs 0x00 and reduces
// the 64-byte digest of a CLONE of the state.  Wire encodings (App. A.5): scalar = 32 canonical LE bytes, point =
// x LE with the parity of y in bit 6 of the last byte.
class Transcript {
 public:
  Transcript() : st_("Halo2-Transcript") {}
  explicit Transcript(const uint8_t* proof, size_t len) : st_("Halo2-Transcript"), in_(proof), in_len_(len) {}
  void common_scalar(const Fr& s) {
    uint8_t b[33];
    b[0] = 2;
    s.to_le_bytes(b + 1);
    st_.update(b, 33);
  }
  void common_point(const G1& p) {
    uint8_t b[65];
    b[0] = 1;
    memset(b + 1, 0, 64);
    if (!p.inf) {
      p.x.to_le_bytes(b + 1);
      p.y.to_le_bytes(b + 33);
    }
    st_.update(b, 65);
  }
  void write_scalar(const Fr& s) {
    common_scalar(s);
    uint8_t b[32];
    s.to_le_bytes(b);
    out_.insert(out_.end(), b, b + 32);
  }
  void write_point(const G1& p) {
    common_point(p);
    uint8_t b[32];
    memset(b, 0, 32);
    if (!p.inf) {
      p.x.to_le_bytes(b);
      b[31] |= (uint8_t)((p.y.is_odd() ? 1 : 0) << 6);
    }
    out_.insert(out_.end(), b, b + 32);
  }
  Fr squeeze_challenge() {
    const uint8_t z = 0;
    st_.update(&z, 1);
    uint8_t d[64];
    st_.digest(d);
    return Fr::from_le_bytes_wide(d);
  }
  // reader side: false = malformed proof
  bool read_scalar(Fr* s) {
    if (in_pos_ + 32 > in_len_) return false;
    if (!Fr::from_le_bytes_canonical(in_ + in_pos_, s)) return false;
    in_pos_ += 32;
    common_scalar(*s);
    return true;
  }
  bool read_point(G1* p);   // h2_prover.hip (needs the square root in Fq)
  const std::vector<uint8_t>& bytes() const { return out_; }

 private:
  Blake2b st_;
  std::vector<uint8_t> out_;
  const uint8_t* in_ = nullptr;
  size_t in_len_ = 0, in_pos_ = 0;

 public:
  const uint8_t* take32() {
    if (in_pos_ + 32 > in_len_) return nullptr;
    const uint8_t* p = in_ + in_pos_;
    in_pos_ += 32;
    return p;
  }
};

}  // namespace h2
