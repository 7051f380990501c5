// h2_prover.hip -- the reference crate's product surface behind the C ABI (include/h2hip.h, "product surface"):
//   h2_setup / h2_generate_proof / h2_verify_proof / h2_simulate / h2_circuit_count
// = setup / wasm_generate_proof / wasm_verify_proof / wasm_simulate_circuit / get_circuit_count of
// /root/reference/circuits/src/wasm.rs:49,68,77,125,182, which call utils.rs:59-158 (generate_params, generate_keys,
// generate_proof[_with_instance], verify[_with_instance]) over halo2_proofs @6b43b6b's keygen_vk / keygen_pk /
// create_proof / verify_proof with KZG over BN254, GWC or SHPLONK openings and a Blake2b transcript (SURVEY.md
// App. A.4-A.8).  halo2_prover_amd/prover.py + verifier.py are the readable Python statement of the same thing and
// produce identical bytes; this file is the one a Rust or JS host links against.
//
// Orchestration is host C++; every column stays in HBM.  Per proof: 6 MSM phases (keygen's fixed + sigma columns,
// advice, permutation products, random polynomial, quotient pieces, opening witnesses) through msm_device_run, the
// Lagrange -> coefficient -> extended-coset transforms through ntt_enqueue, and between them ONE launch each for the
// permutation ratio, the whole quotient numerator (expr_kernel), all evaluations at x, each opening combination
// (h2_prover_kernels.hpp).  The host hashes the transcript, synthesises the (sparse) witness and draws the blinding
// scalars from the caller's RNG in the reference's order, so that under the same RNG stream the proof bytes are the
// reference's.
#include <sys/random.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <functional>
#include <set>
#include <thread>

#include "h2_circuits.hpp"
#include "h2_curve.hpp"
#include "h2_internal.hpp"
#include "h2_pairing.hpp"
#include "h2_poly.hpp"
#include "h2_prover_kernels.hpp"

using namespace h2;
using namespace h2::plonk;

namespace {

struct Fail {
  int status;
  std::string what;
};
[[noreturn]] void fail(int status, const std::string& what) { throw Fail{status, what}; }
void hip_ok(hipError_t e, const char* where) {
  if (e != hipSuccess) fail(H2_EDEVICE, std::string(where) + ": " + hipGetErrorString(e));
}
void st_ok(int rc, const char* where) {
  if (rc != H2_OK) fail(rc, where);
}

// phase timings on stderr when H2_TRACE is set (wall clock, the stream is NOT synchronised for the marks)
struct Trace {
  bool on;
  std::chrono::steady_clock::time_point t0, last;
  const char* what;
  explicit Trace(const char* w) : on(getenv("H2_TRACE") != nullptr), what(w) { t0 = last = std::chrono::steady_clock::now(); }
  void mark(const char* phase) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[h2 %s] %-28s %8.3f ms (+%.3f)\n", what, phase, std::chrono::duration<double, std::milli>(now - t0).count(),
            std::chrono::duration<double, std::milli>(now - last).count());
    last = now;
  }
};

// ---- the caller's RNG, consumed call by call exactly as the reference's RngCore is -------------------------------
struct Rng {
  h2_rng_fill_t fn;
  void* ctx;
  void fill(uint8_t* out, size_t n) {
    if (fn) {
      fn(ctx, out, n);
      return;
    }
    size_t got = 0;
    while (got < n) {
      const ssize_t r = getrandom(out + got, n - got, 0);
      if (r <= 0) fail(H2_EDEVICE, "getrandom failed");
      got += (size_t)r;
    }
  }
  // Fr::random(rng): eight next_u64 calls, the 512-bit integer reduced mod r (halo2curves' from_bytes_wide)
  Fr fr_random() {
    uint8_t b[64];
    for (int i = 0; i < 8; i++) fill(b + 8 * i, 8);
    return Fr::from_le_bytes_wide(b);
  }
};

// ---- the three JSON inputs (arithmetic_circuit.rs:39-45, collatz.rs:20-23, poseidon_circuit.rs:37-41) --------------
struct Json {
  std::map<std::string, std::string> scalars;              // "x": 6   or  "output": "0x.."
  std::map<std::string, std::vector<uint64_t>> arrays;     // "x": [1, 2]
  static uint64_t to_u64(const std::string& s) {
    if (s.empty()) fail(H2_EPROOF, "json: empty number");
    uint64_t v = 0;
    for (char c : s) {
      if (c < '0' || c > '9') fail(H2_EPROOF, "json: not an unsigned integer");
      if (v > (~0ull - (uint64_t)(c - '0')) / 10) fail(H2_EPROOF, "json: integer exceeds u64");
      v = v * 10 + (uint64_t)(c - '0');
    }
    return v;
  }
  explicit Json(const char* s) {
    if (!s) fail(H2_EINVAL, "json: null");
    const char* p = s;
    auto ws = [&] { while (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r') p++; };
    auto token = [&] {   // a bare number or a quoted string
      ws();
      std::string t;
      if (*p == '"') {
        p++;
        while (*p && *p != '"') t += *p++;
        if (*p != '"') fail(H2_EPROOF, "json: unterminated string");
        p++;
      } else {
        while ((*p >= '0' && *p <= '9') || *p == '-' || *p == '.') t += *p++;
      }
      return t;
    };
    ws();
    if (*p != '{') fail(H2_EPROOF, "json: expected an object");
    p++;
    for (;;) {
      ws();
      if (*p == '}') break;
      if (*p != '"') fail(H2_EPROOF, "json: expected a key");
      const std::string key = token();
      ws();
      if (*p != ':') fail(H2_EPROOF, "json: expected ':'");
      p++;
      ws();
      if (*p == '[') {
        p++;
        std::vector<uint64_t> arr;
        for (;;) {
          ws();
          if (*p == ']') { p++; break; }
          arr.push_back(to_u64(token()));
          ws();
          if (*p == ',') p++;
        }
        arrays[key] = arr;
      } else if (strncmp(p, "null", 4) == 0) {
        p += 4;
      } else {
        scalars[key] = token();
      }
      ws();
      if (*p == ',') p++;
      else if (*p != '}') fail(H2_EPROOF, "json: expected ',' or '}'");
    }
  }
  uint64_t u64(const std::string& k) const {
    auto it = scalars.find(k);
    if (it == scalars.end()) fail(H2_EPROOF, "json: missing field " + k);
    return to_u64(it->second);
  }
  const std::vector<uint64_t>& array(const std::string& k) const {
    auto it = arrays.find(k);
    if (it == arrays.end()) fail(H2_EPROOF, "json: missing array " + k);
    return it->second;
  }
};

// ---- device side plumbing ---------------------------------------------------------------------------------------------
using Col = U128*;    // a column of field elements in HBM (n or 2^extended_k of them)

struct Dev {
  DevCtx* c;
  hipStream_t s;
  const CurveOps* ops;
  std::vector<std::pair<void*, size_t>> live;                 // everything handed out, freed by the owner's destructor
  std::vector<std::vector<uint8_t>> staged;                   // host buffers of in-flight uploads (kept until sync)
  // cache of freed blocks keyed by (device, size): hipMalloc / hipFree synchronise the device, a proof needs ~60 buffers
  using BlockKey = std::pair<int, size_t>;
  static std::multimap<BlockKey, void*>& cache() {
    static auto* m = new std::multimap<BlockKey, void*>();   // never destroyed: keys cached until process exit release into it
    return *m;
  }
  explicit Dev(DevCtx* ctx) : c(ctx), s(ctx->stream), ops(ops_of(H2_BN254)) {}
  void* alloc(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    void* p = nullptr;
    auto it = cache().find(BlockKey{c->device, bytes});
    if (it != cache().end()) {
      p = it->second;
      cache().erase(it);
    } else {
      hipError_t e = hipMalloc(&p, bytes);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        fail(H2_ENOMEM, "hipMalloc");
      }
    }
    live.push_back({p, bytes});
    return p;
  }
  Col col(size_t elems) { return (Col)alloc(elems * 32); }
  // give everything back to the cache (the stream is in order: a later user of the block queues behind this one)
  void release_all() {
    for (auto& b : live) cache().insert({BlockKey{c->device, b.second}, b.first});
    live.clear();
  }
  void release(void* p) {
    for (size_t i = 0; i < live.size(); i++)
      if (live[i].first == p) {
        cache().insert({BlockKey{c->device, live[i].second}, p});
        live.erase(live.begin() + i);
        return;
      }
  }
  void sync() {
    hip_ok(hipStreamSynchronize(s), "hipStreamSynchronize");
    staged.clear();
  }
  void* upload(const void* data, size_t bytes) {
    staged.emplace_back((const uint8_t*)data, (const uint8_t*)data + bytes);
    void* d = alloc(bytes);
    hip_ok(hipMemcpyAsync(d, staged.back().data(), bytes, hipMemcpyHostToDevice, s), "hipMemcpyAsync(H2D)");
    return d;
  }
  Col upload_frs(const std::vector<Fr>& v) {
    std::vector<uint8_t> raw(v.size() * 32);
    for (size_t i = 0; i < v.size(); i++) memcpy(raw.data() + 32 * i, v[i].v.v, 32);
    return (Col)upload(raw.data(), raw.size());
  }
  std::vector<Fr> download_frs(const void* d, size_t count) {
    std::vector<uint8_t> raw(count * 32);
    hip_ok(hipMemcpyAsync(raw.data(), d, raw.size(), hipMemcpyDeviceToHost, s), "hipMemcpyAsync(D2H)");
    sync();
    std::vector<Fr> out(count);
    for (size_t i = 0; i < count; i++) out[i] = Fr::from_mont_limbs(raw.data() + 32 * i);
    return out;
  }
  void zero(void* p, size_t bytes) { hip_ok(hipMemsetAsync(p, 0, bytes, s), "hipMemsetAsync"); }
  void copy(void* dst, const void* src, size_t bytes) {
    hip_ok(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync(D2D)");
  }
  static void limbs(const Fr& f, uint64_t out[4]) { f.mont_limbs(out); }
  // m sparse columns -> m dense device columns (stride `stride` elements), zero elsewhere
  void fill_sparse(Col base, size_t stride, const std::vector<SparseCol>& cols) {
    zero(base, cols.size() * stride * 32);
    std::vector<pk::CellRef> refs;
    std::vector<Fr> vals;
    for (size_t j = 0; j < cols.size(); j++)
      for (auto& kv : cols[j]) {
        refs.push_back({(uint32_t)j, kv.first});
        vals.push_back(kv.second);
      }
    if (refs.empty()) return;
    const pk::CellRef* d_refs = (const pk::CellRef*)upload(refs.data(), refs.size() * sizeof(pk::CellRef));
    Col d_vals = upload_frs(vals);
    hipLaunchKernelGGL(pk::scatter_cells_kernel, dim3((unsigned)((refs.size() + 255) / 256)), dim3(256), 0, s, base, stride,
                       d_refs, d_vals, (uint32_t)refs.size());
    hip_ok(hipGetLastError(), "scatter_cells_kernel");
  }
  void ntt(Col a, size_t m, const Fr& omega, uint32_t log_n, const Fr* scale = nullptr, hipStream_t on = nullptr) {
    uint64_t w[4], sc[4];
    limbs(omega, w);
    if (scale) limbs(*scale, sc);
    st_ok(ntt_enqueue(*c, H2_BN254, a, m, w, log_n, on ? on : s, scale ? sc : nullptr), "ntt_enqueue");
  }
  // `on` waits for the accumulate kernel of the MSM enqueued last on this context (commit_begin)
  void wait_msm_tail(hipStream_t on) {
    if (c->tail_recorded && c->tail_wait) hip_ok(hipStreamWaitEvent(on, c->tail_wait, 0), "hipStreamWaitEvent(tail)");
  }
  // the second stream of this context (non-blocking) and three events to hand work back and forth
  hipStream_t side() {
    if (!c->side_stream) {
      hip_ok(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking), "hipStreamCreateWithFlags");
      for (auto& e : c->side_ev) hip_ok(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreateWithFlags");
    }
    return c->side_stream;
  }
  // `to` waits for everything enqueued on `from` so far
  void order(hipStream_t from, hipStream_t to, int ev) {
    side();
    hip_ok(hipEventRecord(c->side_ev[ev], from), "hipEventRecord");
    hip_ok(hipStreamWaitEvent(to, c->side_ev[ev], 0), "hipStreamWaitEvent");
  }
  void lincomb(Col out, uint32_t n, const std::vector<std::pair<Col, Fr>>& terms) {
    bool accumulate = false;
    for (size_t lo = 0; lo < terms.size(); lo += pk::LINCOMB_MAX) {
      pk::LincombArgs A{};
      A.count = (int)std::min<size_t>(pk::LINCOMB_MAX, terms.size() - lo);
      for (int j = 0; j < A.count; j++) {
        A.a[j] = terms[lo + j].first;
        A.c[j] = terms[lo + j].second.v;
      }
      A.unit_first = terms[lo].second == Fr::one();
      hipLaunchKernelGGL(pk::lincomb_kernel, dim3((n + 255) / 256), dim3(256), 0, s, A, out, n, accumulate ? 1 : 0);
      hip_ok(hipGetLastError(), "lincomb_kernel");
      accumulate = true;
    }
  }
  // q = (a - a(z)) / (X - z)
  void divide_linear(Col a, uint32_t n, const Fr& z, Col q) {
    uint64_t zl[4];
    limbs(z, zl);
    st_ok(arena_acquire(c->div_ws, (size_t)2 * DIV_MAX_CHUNKS * 32, s), "arena");
    hip_ok(ops->poly_divide_linear(a, n, zl, q, c->div_ws.p, s), "poly_divide_linear");
    st_ok(arena_release(c->div_ws, s), "arena");
  }
  void prefix_product(Col a, uint32_t n, Col out) {
    st_ok(arena_acquire(c->div_ws, (size_t)2 * DIV_MAX_CHUNKS * 32, s), "arena");
    hip_ok(ops->poly_prefix_product(a, n, out, c->div_ws.p, s), "poly_prefix_product");
    st_ok(arena_release(c->div_ws, s), "arena");
  }
  // mode 0: q_j = (a_j - a_j(z_j)) / (X - z_j); mode 1: out_j[i] = prod_{t < i} a_j[t] -- all jobs in one launch sequence
  void scan_batch(int mode, uint32_t n, const std::vector<Col>& in, const std::vector<Col>& out, const std::vector<Fr>& z) {
    for (size_t j0 = 0; j0 < in.size(); j0 += pk::SCAN_MAX_JOBS) {
      const size_t cnt = std::min<size_t>(pk::SCAN_MAX_JOBS, in.size() - j0);
      uint32_t L = (n + pk::SCAN_CHUNKS - 1) / pk::SCAN_CHUNKS;
      if (L < 16) L = 16;
      const uint32_t C = (n + L - 1) / L;
      pk::ScanBatch B{};
      for (size_t j = 0; j < cnt; j++) {
        B.a[j] = in[j0 + j];
        B.out[j] = out[j0 + j];
        if (mode == 0) {
          B.z[j] = z[j0 + j].v;
          B.w[j] = z[j0 + j].pow_u64(L).v;
        }
      }
      Col ws = col((size_t)2 * cnt * pk::SCAN_CHUNKS);
      Col H = ws, G = ws + 2 * cnt * (size_t)pk::SCAN_CHUNKS;
      hipLaunchKernelGGL(pk::scan_chunk_kernel, dim3((C + 63) / 64, (unsigned)cnt), dim3(64), 0, s, B, mode, n, L, C, H);
      hipLaunchKernelGGL(pk::scan_block_kernel, dim3((unsigned)cnt), dim3(pk::SCAN_BLOCK_THREADS), 0, s, B, mode, C, H, G);
      hipLaunchKernelGGL(pk::scan_apply_kernel, dim3((C + 63) / 64, (unsigned)cnt), dim3(64), 0, s, B, mode, n, L, C, G);
      hip_ok(hipGetLastError(), "scan_batch kernels");
      release(ws);
    }
  }
  // values of `jobs` = (polynomial, point) pairs, all polynomials of n coefficients
  std::vector<Fr> evaluate(const std::vector<std::pair<Col, Fr>>& jobs, uint32_t n) {
    if (jobs.empty()) return {};
    std::vector<pk::EvalJob> hj(jobs.size());
    for (size_t i = 0; i < jobs.size(); i++) {
      hj[i].poly = jobs[i].first;
      hj[i].point = jobs[i].second.v;
    }
    const pk::EvalJob* dj = (const pk::EvalJob*)upload(hj.data(), hj.size() * sizeof(pk::EvalJob));
    const uint32_t threads = (n + pk::EVAL_RUN - 1) / pk::EVAL_RUN;
    const uint32_t blocks = (threads + pk::EVAL_BLOCK - 1) / pk::EVAL_BLOCK;
    Col partial = col((size_t)blocks * jobs.size());
    Col out = col(jobs.size());
    hipLaunchKernelGGL(pk::poly_eval_partial_kernel, dim3(blocks, (unsigned)jobs.size()), dim3(pk::EVAL_BLOCK), 0, s, dj, n,
                       partial, blocks);
    hipLaunchKernelGGL(pk::poly_eval_final_kernel, dim3((unsigned)((jobs.size() + 63) / 64)), dim3(64), 0, s, partial, blocks,
                       out, (uint32_t)jobs.size());
    hip_ok(hipGetLastError(), "poly_eval kernels");
    std::vector<Fr> v = download_frs(out, jobs.size());
    release(partial);
    release(out);
    return v;
  }
  ~Dev() { release_all(); }
};

// ---- params: the SRS registered once per distinct byte string -----------------------------------------------------------
struct Params {
  uint32_t k = 0;
  uint64_t h_g = 0, h_gl = 0;       // bases handles (g, g_lagrange)
  G1 g0;
  bn::G2 g2, s_g2;
  std::array<uint8_t, 64> digest{};
  // [delta^j] commit_lagrange(w^i): the commitment of the IDENTITY permutation's column j -- a property of the SRS, not
  // of a circuit.  A circuit's sigma_j differs from it in the few cells its copy constraints move, so its
  // commitment is this point plus a sparse MSM (keygen would otherwise commit to 7 dense columns on every call)
  mutable std::vector<G1> sigma_identity;
};
std::vector<Params> g_params;       // small LRU: the UI keeps one SRS, tests a few

Fq fq_from_mont(const uint8_t* p) { return Fq::from_mont_limbs(p); }

G1 affine_from_raw(const uint8_t* p) {
  G1 g;
  g.x = fq_from_mont(p);
  g.y = fq_from_mont(p + 32);
  g.inf = g.x.is_zero() && g.y.is_zero();
  return g;
}

const Params& params_get(const uint8_t* bytes, size_t len) {
  if (!bytes || len < 4) fail(H2_EPROOF, "params: truncated");
  uint32_t k;
  memcpy(&k, bytes, 4);
  if (k > 28) fail(H2_EPROOF, "params: k out of range");
  const size_t n = (size_t)1 << k;
  if (len != 4 + 128 * n + 256) fail(H2_EPROOF, "params: wrong length for k");
  // cache key: multiply-xorshift lanes over every byte (eight interleaved lanes keep one core's multiplier busy:
  // 17 GB/s, 0.48 ms of every call at k = 16; Blake2b took 8 ms), the blob cut into four quarters hashed by four threads
  // (0.48 -> ~0.15 ms), finished through Blake2b -- a fingerprint against accidents, not against a caller attacking
  // itself (include/h2hip.h, "Trust")
  constexpr int PARTS = 4;
  uint64_t lane[PARTS][8];
  auto hash_part = [&](int part) {
    static const uint64_t seed[8] = {0x9E3779B97F4A7C15ull, 0xBF58476D1CE4E5B9ull, 0x94D049BB133111EBull, 0xD6E8FEB86659FD93ull,
                                     0xA0761D6478BD642Full, 0xE7037ED1A0B428DBull, 0x8EBC6AF09C88C6E3ull, 0x589965CC75374CC3ull};
    uint64_t* l = lane[part];
    for (int i = 0; i < 8; i++) l[i] = seed[i] + (uint64_t)part;
    const size_t blocks = len / 64, per = (blocks + PARTS - 1) / PARTS;
    const size_t b0 = std::min(blocks, per * part), b1 = std::min(blocks, b0 + per);
    const uint8_t* q = bytes + 64 * b0;
    for (size_t i = b0; i < b1; i++, q += 64) {
      uint64_t w[8];
      memcpy(w, q, 64);
      for (int k = 0; k < 8; k++) {
        l[k] = (l[k] ^ w[k]) * 0xFF51AFD7ED558CCDull;
        l[k] ^= l[k] >> 29;
      }
    }
    if (part == PARTS - 1)
      for (q = bytes + 64 * blocks; q < bytes + len; q++) l[0] = (l[0] ^ *q) * 0x100000001B3ull;
  };
  if (len >= (1u << 20)) {
    std::thread th[PARTS - 1];
    for (int t = 0; t < PARTS - 1; t++) th[t] = std::thread(hash_part, t + 1);
    hash_part(0);
    for (auto& t : th) t.join();
  } else {
    for (int t = 0; t < PARTS; t++) hash_part(t);
  }
  Blake2b h;
  h.update(lane, sizeof lane);
  h.update(&len, sizeof len);
  h.update(bytes, 4);
  h.update(bytes + len - 256, 256);
  std::array<uint8_t, 64> dg;
  h.digest(dg.data());
  for (size_t i = 0; i < g_params.size(); i++)
    if (g_params[i].digest == dg) {
      if (i) std::swap(g_params[i], g_params[0]);
      return g_params[0];
    }
  Params p;
  p.k = k;
  p.digest = dg;
  // the reference reads with SerdeFormat::RawBytes (wasm.rs:79-80): raw Montgomery limbs, 64 B per G1 point
  int rc = h2_bases_register(H2_BN254, (const uint64_t*)(bytes + 4), n, &p.h_g);
  if (rc == H2_EINVAL) fail(H2_EPROOF, "params: g holds a point that is not on the curve");
  st_ok(rc, "h2_bases_register(g)");
  rc = h2_bases_register(H2_BN254, (const uint64_t*)(bytes + 4 + 64 * n), n, &p.h_gl);
  if (rc != H2_OK) (void)h2_bases_release(p.h_g);
  if (rc == H2_EINVAL) fail(H2_EPROOF, "params: g_lagrange holds a point that is not on the curve");
  st_ok(rc, "h2_bases_register(g_lagrange)");
  p.g0 = affine_from_raw(bytes + 4);
  const uint8_t* t = bytes + 4 + 128 * n;
  p.g2 = bn::G2{{fq_from_mont(t), fq_from_mont(t + 32)}, {fq_from_mont(t + 64), fq_from_mont(t + 96)}, false};
  p.s_g2 = bn::G2{{fq_from_mont(t + 128), fq_from_mont(t + 160)}, {fq_from_mont(t + 192), fq_from_mont(t + 224)}, false};
  if (g_params.size() >= 4) {
    (void)h2_bases_release(g_params.back().h_g);
    (void)h2_bases_release(g_params.back().h_gl);
    g_params.pop_back();
  }
  g_params.insert(g_params.begin(), p);
  return g_params[0];
}

// ---- EvaluationDomain (halo2_proofs src/poly/domain.rs) ------------------------------------------------
struct Domain {
  uint32_t k, n, ext_k, en, qdeg;
  Fr omega, omega_inv, ext_omega, ext_omega_inv, zeta, zeta_inv, n_inv, en_inv;
  std::vector<Fr> t_evaluations;     // 1 / ((zeta w_ext^i)^n - 1), i < 2^(ext_k - k)
  Domain(uint32_t degree, uint32_t k_) : k(k_), n(1u << k_), qdeg(degree - 1) {
    ext_k = k;
    while ((1ull << ext_k) < (uint64_t)n * qdeg) ext_k++;
    if (ext_k > 28) fail(H2_EINVAL, "extended_k exceeds the field's two-adicity");
    en = 1u << ext_k;
    ext_omega = fr_root_of_unity();
    for (uint32_t i = ext_k; i < 28; i++) ext_omega = ext_omega.sqr();
    omega = ext_omega;
    for (uint32_t i = k; i < ext_k; i++) omega = omega.sqr();
    omega_inv = omega.inv();
    ext_omega_inv = ext_omega.inv();
    zeta = Fr::from_hex("0x30644e72e131a029048b6e193fd84104cc37a73fec2bc5e9b8ca0b2d36636f23");   // ZETA, as recalled (domain.py)
    zeta_inv = zeta.sqr();
    n_inv = Fr::from_u64(n).inv();
    en_inv = Fr::from_u64(en).inv();
    const uint32_t period = 1u << (ext_k - k);
    Fr cur = zeta;
    for (uint32_t i = 0; i < period; i++) {
      t_evaluations.push_back((cur.pow_u64(n) - Fr::one()).inv());
      cur *= ext_omega;
    }
  }
  Fr rotate(const Fr& x, int rot) const {
    const Fr w = rot >= 0 ? omega.pow_u64((uint64_t)rot) : omega_inv.pow_u64((uint64_t)(-rot));
    return x * w;
  }
};

// ---- the quotient numerator compiled to a straight-line program (h2_prover_kernels.hpp expr_kernel) ------------------------
struct ExprProgram {
  // nodes (hash-consed): 0 const, 1 column, 2 add, 3 sub, 4 mul
  struct Node {
    int op, a, b, col, rot, cidx;
    bool operator<(const Node& o) const {
      return std::tie(op, a, b, col, rot, cidx) < std::tie(o.op, o.a, o.b, o.col, o.rot, o.cidx);
    }
  };
  std::vector<Node> nodes;
  std::map<Node, int> index;
  std::vector<Fr> consts;              // constant table (proof-dependent entries are patched per proof)
  std::map<std::array<uint8_t, 32>, int> const_index;
  int intern(const Node& nd) {
    auto it = index.find(nd);
    if (it != index.end()) return it->second;
    nodes.push_back(nd);
    index[nd] = (int)nodes.size() - 1;
    return (int)nodes.size() - 1;
  }
  int constant(const Fr& v) {
    std::array<uint8_t, 32> key;
    memcpy(key.data(), v.v.v, 32);
    auto it = const_index.find(key);
    int ci;
    if (it == const_index.end()) {
      consts.push_back(v);
      ci = (int)consts.size() - 1;
      const_index[key] = ci;
    } else {
      ci = it->second;
    }
    return intern({0, -1, -1, -1, 0, ci});
  }
  // a slot of the constant table whose value is set later (challenges): never merged with another constant
  int variable(int* slot_out) {
    consts.push_back(Fr::zero());
    *slot_out = (int)consts.size() - 1;
    return intern({0, -1, -1, -1, 0, *slot_out});
  }
  int column(int col, int rot) { return intern({1, -1, -1, col, rot, -1}); }
  int add(int a, int b) { return intern({2, std::min(a, b), std::max(a, b), -1, 0, -1}); }
  int sub(int a, int b) { return intern({3, a, b, -1, 0, -1}); }
  int mul(int a, int b) { return intern({4, std::min(a, b), std::max(a, b), -1, 0, -1}); }

  std::vector<pk::XInstr> code;
  uint32_t nslots = 0, nreduce = 0;      // LDS slots; products with one inserted to keep magnitudes bounded
  // emit instructions for `root` (every arithmetic node it depends on, in order), slots reused after the last use.
  // Order: depth first, the operand that needs more live values first (Sethi-Ullman numbering; shared nodes count as
  // computed).  A result that the NEXT instruction consumes is handed over in a register (operand kind X_PREV), and a
  // result with no other use is never stored: a slot costs 32 bytes of LDS per row and the slots of 128 rows decide
  // how many blocks share a CU (h2_prover_kernels.hpp, expr_kernel).
  void compile(int root) {
    (void)constant(Fr::one());                       // the reducing product's operand: interned before the node tables are sized
    std::vector<int> need(nodes.size(), -1);
    std::function<int(int)> su = [&](int id) -> int {
      if (need[id] >= 0) return need[id];
      const Node& nd = nodes[id];
      if (nd.op < 2) return need[id] = 0;
      const int na = su(nd.a), nb = su(nd.b);
      return need[id] = std::max(1, na == nb ? na + 1 : std::max(na, nb));
    };
    su(root);
    std::vector<int> order;
    std::vector<char> seen(nodes.size(), 0);
    std::function<void(int)> visit = [&](int id) {
      if (seen[id]) return;
      seen[id] = 1;
      const Node& nd = nodes[id];
      if (nd.op >= 2) {
        if (need[nd.b] > need[nd.a]) {
          visit(nd.b);
          visit(nd.a);
        } else {
          visit(nd.a);
          visit(nd.b);
        }
        order.push_back(id);
      }
    };
    visit(root);
    if (order.empty()) fail(H2_EINVAL, "empty quotient program");
    std::vector<int> at(nodes.size(), -1);           // instruction index of a node
    for (size_t t = 0; t < order.size(); t++) at[order[t]] = (int)t;
    std::vector<int> last_use(nodes.size(), -1);
    std::vector<char> wants_slot(nodes.size(), 0);   // some use is not the very next instruction
    for (size_t t = 0; t < order.size(); t++)
      for (int src : {nodes[order[t]].a, nodes[order[t]].b}) {
        last_use[src] = (int)t;
        if (nodes[src].op >= 2 && at[src] + 1 != (int)t) wants_slot[src] = 1;
      }
    std::vector<int> slot_of(nodes.size(), -1);
    std::vector<uint32_t> free_slots;
    // magnitudes in units of p (expr_kernel's header): constants are canonical, columns below EXPR_COLUMN_BOUND, a
    // product of a and b below a b / 128 + 1 (p^2 / 2^261 < p / 128); a sum that would pass EXPR_VALUE_BOUND is
    // multiplied by one straight away
    std::vector<double> bound(nodes.size(), 0.0);
    for (size_t id = 0; id < nodes.size(); id++)
      if (nodes[id].op == 0) bound[id] = 1.0;
      else if (nodes[id].op == 1) bound[id] = pk::EXPR_COLUMN_BOUND;
    const uint32_t one_operand = pk::X_CONST | (uint32_t)nodes[constant(Fr::one())].cidx;
    for (size_t t = 0; t < order.size(); t++) {
      const Node& nd = nodes[order[t]];
      auto operand = [&](int id) -> uint32_t {
        const Node& o = nodes[id];
        if (o.op == 0) return pk::X_CONST | (uint32_t)o.cidx;
        if (o.op == 1) return pk::X_COL | ((uint32_t)o.col << 8) | (uint32_t)(o.rot + 128);
        if (at[id] + 1 == (int)t) return pk::X_PREV;
        return pk::X_SLOT | (uint32_t)slot_of[id];
      };
      const uint32_t a = operand(nd.a), b = operand(nd.b);
      // operands dying here free their slots before the destination is chosen (the kernel reads both first)
      for (int src : {nd.a, nd.b})
        if (nodes[src].op >= 2 && last_use[src] == (int)t && slot_of[src] >= 0) {
          free_slots.push_back((uint32_t)slot_of[src]);
          slot_of[src] = -2;
        }
      uint32_t dst = pk::X_NO_STORE;
      if (wants_slot[order[t]]) {
        if (!free_slots.empty()) {
          dst = free_slots.back();
          free_slots.pop_back();
        } else {
          dst = nslots++;
        }
        slot_of[order[t]] = (int)dst;
      }
      double bd = nd.op == 4 ? bound[nd.a] * bound[nd.b] / 128.0 + 1.0 : bound[nd.a] + bound[nd.b];
      if (nd.op != 4 && bd > pk::EXPR_VALUE_BOUND) {
        code.push_back({((uint32_t)(nd.op - 2) << 24) | pk::X_NO_STORE, a, b});
        code.push_back({(2u << 24) | dst, pk::X_PREV, one_operand});
        bd = bd / 128.0 + 1.0;
        nreduce++;
      } else {
        code.push_back({((uint32_t)(nd.op - 2) << 24) | dst, a, b});
      }
      bound[order[t]] = bd;
    }
    if (nslots == 0) nslots = 1;
  }
};

// pseudo-columns of the program beyond the circuit's own: indices into the pointer table handed to the kernel
struct ColumnMap {
  int advice0, fixed0, instance0, sigma0, z0, l0, l_last, l_blind, xcol, tinv, count;
};

// what depends on the domain alone -- (k, constraint degree, blinding factors) -- and not on a circuit's columns or an
// SRS: w^i, the polynomial X and the three Lagrange-basis combinations l_0, l_last, l_blind on the extended coset,
// 1 / (X^n - 1).  Kept for the life of the process (a handful of shapes), shared by every key of that shape.
struct DomainKit {
  int device;
  uint32_t k, ext_k;
  int bf;
  std::unique_ptr<Dev> dev;
  Col omega_col = nullptr, xcol_ext = nullptr, basis_ext = nullptr, tinv = nullptr;
};
std::vector<std::unique_ptr<DomainKit>> g_kits;

struct ProvingKey {
  std::unique_ptr<Circuit> circuit;
  std::unique_ptr<Domain> dom;
  const Params* params = nullptr;
  std::unique_ptr<Dev> dev;                 // owns the key's device buffers
  int bf = 0;
  std::vector<std::vector<int>> sets;       // permutation columns, d - 2 per grand product
  Col fixed_values = nullptr, sigma_values = nullptr, fixed_polys = nullptr, sigma_polys = nullptr;
  Col fixed_ext = nullptr, sigma_ext = nullptr, basis_ext = nullptr /* l0, l_last, l_blind */, xcol_ext = nullptr;
  Col tinv = nullptr, omega_col = nullptr;
  std::vector<G1> fixed_commitments, sigma_commitments;
  Fr transcript_repr;
  ExprProgram prog;
  ColumnMap cmap{};
  int c_beta = -1, c_gamma = -1, c_y = -1;
  std::vector<int> c_beta_delta;            // beta * delta^j, one per permutation column
  const pk::XInstr* d_code = nullptr;
  size_t nf() const { return (size_t)circuit->num_fixed; }
  size_t np() const { return circuit->permutation_columns.size(); }
  // the key's transforms may still be running on the second stream when a key that is not kept dies (verify with
  // the key cache off): its blocks go back to the cache only once that stream is idle
  ~ProvingKey() {
    if (dev && dev->c->side_stream) (void)hipStreamSynchronize(dev->c->side_stream);
  }
};

Fr fr_delta() {   // DELTA = 7^(2^28): generator of the coset structure of the permutation argument
  Fr d = Fr::from_u64(7);
  for (int i = 0; i < 28; i++) d = d.sqr();
  return d;
}

// m columns of n scalars -> m commitments (affine, canonical coordinates)
// host group law on G1 (the same XYZZ templates as the kernels, host instantiation): a handful of operations per keygen
using HX = Xyzz<BN254_CURVE>;
HX hx_of(const G1& p) {
  if (p.inf) return HX::identity();
  return xyzz_from_affine(Affine<BN254_CURVE>{p.x.v, p.y.v});
}
std::vector<G1> hx_to_affine(const std::vector<HX>& pts) {       // one shared inversion
  std::vector<Fq> z(pts.size()), pre(pts.size());
  Fq acc = Fq::one();
  for (size_t j = 0; j < pts.size(); j++) {
    z[j] = Fq(pts[j].zzz);
    pre[j] = acc;
    if (!pts[j].is_identity()) acc *= z[j];
  }
  Fq inv = acc.inv();
  std::vector<G1> out(pts.size());
  for (size_t j = pts.size(); j-- > 0;) {
    if (pts[j].is_identity()) continue;
    const Fq zi3 = inv * pre[j];                                  // 1 / zzz
    inv *= z[j];
    const Fq zi = zi3 * Fq(pts[j].zz);                            // zz / zzz = 1 / z
    out[j].x = Fq(pts[j].x) * zi.sqr();
    out[j].y = Fq(pts[j].y) * zi3;
    out[j].inf = false;
  }
  return out;
}
HX hx_mul(const Fr& k, const HX& p) {
  uint8_t kb[32];
  k.to_le_bytes(kb);
  HX r = HX::identity();
  for (int i = 255; i >= 0; i--) {
    r = xyzz_double(r);
    if ((kb[i >> 3] >> (i & 7)) & 1) r = xyzz_add(r, p);
  }
  return r;
}

// `split` < m: columns [0, split) commit against g_lagrange and [split, m) against g IN THE SAME LAUNCH (commitments
// that do not wait for each other: the permutation products and the RNG-drawn random polynomial)
// Jacobian -> affine on the host: m inversions folded into one (a one-thread device kernel took 0.35 ms per phase)
std::vector<G1> jacobian_to_affine_host(const std::vector<uint8_t>& raw, size_t m) {
  std::vector<Fq> zs(m), pre(m);
  Fq acc = Fq::one();
  for (size_t j = 0; j < m; j++) {
    zs[j] = Fq::from_mont_limbs(raw.data() + 96 * j + 64);
    pre[j] = acc;
    if (!zs[j].is_zero()) acc *= zs[j];
  }
  Fq inv = acc.inv();
  std::vector<G1> pts(m);
  for (size_t j = m; j-- > 0;) {
    if (zs[j].is_zero()) continue;                     // identity
    const Fq zi = inv * pre[j], zi2 = zi.sqr();
    inv *= zs[j];
    pts[j].x = Fq::from_mont_limbs(raw.data() + 96 * j) * zi2;
    pts[j].y = Fq::from_mont_limbs(raw.data() + 96 * j + 32) * zi2 * zi;
    pts[j].inf = false;
  }
  return pts;
}

// The MSM of a commit phase is enqueued by commit_begin and read back by commit_finish: what is queued between the two
// on the second stream behind Dev::wait_msm_tail starts when the accumulate kernel of that MSM has finished, i.e. runs
// beside the MSM's small-grid tail instead of competing with its sort and accumulate kernels (started at once, the
// advice transforms made the sort kernels of the advice commitment three times slower: 140 against 48 us).
struct PendingCommit {
  void* out = nullptr;
  size_t m = 0;
};
// With several contexts (h2_init_devices) a commit phase is spread over them by POINT RANGE (SURVEY.md section 8(e),
// as sharded.msm_phase_device does across ranks): context g commits rows / bases [n g / G, n (g+1) / G) of EVERY column
// of the phase against its own replica of the table, so phases of m = 1 .. 5 columns use every GPU.  The other
// contexts' shares of the columns travel device to device (peer copies, cnt * 32 bytes per column), their G x m partial
// sums (96 bytes each) come back the same way and are added on the prover's device (points_sum_kernel): the same group
// elements as the one-device commitment, hence the same proof bytes.  Transforms are NOT spread: a column would cross
// xGMI twice (2 x 16 MiB for an extended column at k = 16, ~0.5 ms) for ~60 us of butterflies.
uint64_t g_sharded_commits = 0;
size_t g_shard_min_rows = 1024;            // per context; below this the copies and the extra launches cost more than they save
PendingCommit commit_begin(Dev& d, const Params& P, Col cols, uint32_t n, size_t m, bool lagrange, size_t split = ~(size_t)0) {
  auto it = g_h2.bases.find(lagrange ? P.h_gl : P.h_g);
  if (it == g_h2.bases.end()) fail(H2_EHANDLE, "params bases released");
  PendingCommit pc;
  pc.m = m;
  pc.out = d.alloc(m * 96);
  d.c->tail_wanted = true;                   // the MSM records an event behind its accumulate kernel (Dev::wait_msm_tail)
  std::vector<const BasesEntry*> per;
  const BasesEntry* be = &it->second;
  if (split < m) {
    auto ig = g_h2.bases.find(P.h_g), il = g_h2.bases.find(P.h_gl);
    if (ig == g_h2.bases.end() || il == g_h2.bases.end()) fail(H2_EHANDLE, "params bases released");
    per.resize(m);
    for (size_t j = 0; j < m; j++) per[j] = j < split ? &il->second : &ig->second;
    be = &il->second;
  }
  const BasesEntry* const* perp = per.empty() ? nullptr : per.data();
  const size_t G = g_h2.ctx.size();
  if (G == 1 || (size_t)n < g_shard_min_rows * G) {
    st_ok(msm_device_run(*d.c, H2_BN254, *be, cols, 0, n, n, m, pc.out, false, d.s, perp), "msm_device_run");
    return pc;
  }
  g_sharded_commits++;
  const size_t self = ctx_index(d.c);
  char* partials = (char*)d.alloc(G * m * 96);
  auto event_of = [](DevCtx& c) {
    if (!c.shard_ev) hip_ok(hipEventCreateWithFlags(&c.shard_ev, hipEventDisableTiming), "hipEventCreateWithFlags");
    return c.shard_ev;
  };
  hip_ok(hipEventRecord(event_of(*d.c), d.s), "hipEventRecord");            // the columns are final from here on
  size_t slot = 1;
  for (size_t g = 0; g < G; g++) {
    if (g == self) continue;
    DevCtx& cg = g_h2.ctx[g];
    DeviceGuard dg(cg.device);
    const size_t lo = (size_t)n * slot / G, hi = (size_t)n * (slot + 1) / G, cnt = hi - lo;
    const size_t res_off = (m * cnt * 32 + 255) & ~(size_t)255;
    st_ok(arena_acquire(cg.stage, res_off + m * 96, cg.stream), "arena");
    hip_ok(hipStreamWaitEvent(cg.stream, d.c->shard_ev, 0), "hipStreamWaitEvent");
    for (size_t j = 0; j < m; j++)
      hip_ok(hipMemcpyPeerAsync((char*)cg.stage.p + j * cnt * 32, cg.device, (const char*)cols + (j * (size_t)n + lo) * 32,
                                d.c->device, cnt * 32, cg.stream), "hipMemcpyPeerAsync(columns)");
    void* d_res = (char*)cg.stage.p + res_off;
    st_ok(msm_device_run(cg, H2_BN254, *be, cg.stage.p, lo, cnt, cnt, m, d_res, false, cg.stream, perp), "msm_device_run");
    hip_ok(hipMemcpyPeerAsync(partials + slot * m * 96, d.c->device, d_res, cg.device, m * 96, cg.stream),
           "hipMemcpyPeerAsync(partials)");
    hip_ok(hipEventRecord(event_of(cg), cg.stream), "hipEventRecord");
    st_ok(arena_release(cg.stage, cg.stream), "arena");
    slot++;
  }
  // this context's share: rows [0, n / G)
  st_ok(msm_device_run(*d.c, H2_BN254, *be, cols, 0, (size_t)n / G, n, m, partials, false, d.s, perp), "msm_device_run");
  for (size_t g = 0; g < G; g++)
    if (g != self) hip_ok(hipStreamWaitEvent(d.s, g_h2.ctx[g].shard_ev, 0), "hipStreamWaitEvent");
  hip_ok(d.ops->points_sum(partials, pc.out, (uint32_t)G, (uint32_t)m, d.s), "points_sum");
  d.release(partials);
  return pc;
}
std::vector<G1> commit_finish(Dev& d, PendingCommit& pc) {
  std::vector<uint8_t> raw(pc.m * 96);
  hip_ok(hipMemcpyAsync(raw.data(), pc.out, raw.size(), hipMemcpyDeviceToHost, d.s), "hipMemcpyAsync(D2H)");
  d.sync();
  d.release(pc.out);
  pc.out = nullptr;
  return jacobian_to_affine_host(raw, pc.m);
}
std::vector<G1> commit(Dev& d, const Params& P, Col cols, uint32_t n, size_t m, bool lagrange, size_t split = ~(size_t)0) {
  PendingCommit pc = commit_begin(d, P, cols, n, m, lagrange, split);
  return commit_finish(d, pc);
}

// coeff (m columns of n, stride n) -> extended-coset evaluations (m columns of en)
void coeff_to_extended(Dev& d, const Domain& D, Col in, size_t m, Col out, hipStream_t on = nullptr) {
  hipLaunchKernelGGL(pk::coset_extend_kernel, dim3((D.en + 255) / 256, (unsigned)m), dim3(256), 0, on ? on : d.s, in,
                     (size_t)D.n, out, D.n, D.en, D.zeta.v, D.zeta.sqr().v);
  hip_ok(hipGetLastError(), "coset_extend_kernel");
  d.ntt(out, m, D.ext_omega, D.ext_k, nullptr, on);
}

const DomainKit& domain_kit(const Domain& D, int bf, DevCtx* ctx) {
  for (auto& kp : g_kits)
    if (kp->device == ctx->device && kp->k == D.k && kp->ext_k == D.ext_k && kp->bf == bf) return *kp;
  auto kit = std::make_unique<DomainKit>();
  kit->device = ctx->device;
  kit->k = D.k;
  kit->ext_k = D.ext_k;
  kit->bf = bf;
  kit->dev = std::make_unique<Dev>(ctx);
  Dev& d = *kit->dev;
  const uint32_t n = D.n;
  uint64_t w[4];
  // a ones column = the NTT of (1, 0, 0, ...) (every evaluation of the constant polynomial 1 is 1); then a[i] *= w^i
  kit->omega_col = d.col(n);
  {
    std::vector<SparseCol> c1(1);
    c1[0][0] = Fr::one();
    d.fill_sparse(kit->omega_col, n, c1);
    d.ntt(kit->omega_col, 1, D.omega, D.k);
    Dev::limbs(D.omega, w);
    hip_ok(d.ops->poly_powers(kit->omega_col, n, 1, w, d.s), "poly_powers");
  }
  // l_0, l_last, l_blind: Lagrange -> coefficients -> extended coset
  {
    Col basis = d.col(3 * (size_t)n);
    std::vector<SparseCol> b(3);
    b[0][0] = Fr::one();
    b[1][n - bf - 1] = Fr::one();
    for (uint32_t r = n - bf; r < n; r++) b[2][r] = Fr::one();
    d.fill_sparse(basis, n, b);
    d.ntt(basis, 3, D.omega_inv, D.k, &D.n_inv);
    kit->basis_ext = d.col(3 * (size_t)D.en);
    hipLaunchKernelGGL(pk::coset_extend_kernel, dim3((D.en + 255) / 256, 3u), dim3(256), 0, d.s, basis, (size_t)n,
                       kit->basis_ext, n, D.en, D.zeta.v, D.zeta.sqr().v);
    hip_ok(hipGetLastError(), "coset_extend_kernel");
    d.ntt(kit->basis_ext, 3, D.ext_omega, D.ext_k);
    d.release(basis);
  }
  // the polynomial X on the coset: zeta w_ext^i
  kit->xcol_ext = d.col(D.en);
  {
    std::vector<SparseCol> c1(1);
    c1[0][0] = D.zeta;
    d.fill_sparse(kit->xcol_ext, D.en, c1);
    d.ntt(kit->xcol_ext, 1, D.ext_omega, D.ext_k);                  // (zeta, zeta, ...)
    Dev::limbs(D.ext_omega, w);
    hip_ok(d.ops->poly_powers(kit->xcol_ext, D.en, 1, w, d.s), "poly_powers");
  }
  kit->tinv = d.upload_frs(D.t_evaluations);
  d.sync();
  if (g_kits.size() >= 8) g_kits.erase(g_kits.begin());
  g_kits.push_back(std::move(kit));
  return *g_kits.back();
}

// the permutation columns, d - 2 per grand product (halo2's chunking), and the blinding rows: host data of a key
void key_shape(ProvingKey& K) {
  const Circuit& C = *K.circuit;
  K.bf = C.blinding_factors();
  const int chunk = C.degree - 2;
  for (size_t s = 0; s < C.permutation_columns.size(); s += chunk) {
    std::vector<int> set;
    for (size_t j = s; j < std::min(s + chunk, C.permutation_columns.size()); j++) set.push_back((int)j);
    K.sets.push_back(set);
  }
}

// every gate, the permutation argument, the y-fold and the division by X^n - 1 as one program (host only)
void build_quotient_program(ProvingKey& K) {
  const Circuit& C = *K.circuit;
  const size_t np = K.np();
  ColumnMap& M = K.cmap;
  int next = 0;
  M.advice0 = next; next += C.num_advice;
  M.fixed0 = next; next += C.num_fixed;
  M.instance0 = next; next += C.num_instance;
  M.sigma0 = next; next += (int)np;
  M.z0 = next; next += (int)K.sets.size();
  M.l0 = next++; M.l_last = next++; M.l_blind = next++; M.xcol = next++; M.tinv = next++;
  M.count = next;
  ExprProgram& X = K.prog;
  std::function<int(const E&)> build = [&](const E& e) -> int {
    switch (e->kind) {
      case Expr::Const: return X.constant(e->c);
      case Expr::Advice: return X.column(M.advice0 + e->col, e->rot);
      case Expr::Fixed: return X.column(M.fixed0 + e->col, e->rot);
      case Expr::Instance: return X.column(M.instance0 + e->col, e->rot);
      case Expr::Neg: return X.sub(X.constant(Fr::zero()), build(e->a));
      case Expr::Sum:
        if (e->b->kind == Expr::Neg) return X.sub(build(e->a), build(e->b->a));
        return X.add(build(e->a), build(e->b));
      case Expr::Prod: return X.mul(build(e->a), build(e->b));
      case Expr::Scaled: return X.mul(build(e->a), X.constant(e->c));
    }
    return -1;
  };
  const int v_y = X.variable(&K.c_y), v_beta = X.variable(&K.c_beta), v_gamma = X.variable(&K.c_gamma);
  std::vector<int> v_bd(np);
  K.c_beta_delta.resize(np);
  for (size_t j = 0; j < np; j++) v_bd[j] = X.variable(&K.c_beta_delta[j]);
  const int one = X.constant(Fr::one());
  std::vector<int> terms;
  for (auto& g : C.gates) terms.push_back(build(g));
  auto colref = [&](const ColRef& cr) {
    return X.column((cr.first == ADVICE ? M.advice0 : cr.first == FIXED ? M.fixed0 : M.instance0) + cr.second, 0);
  };
  if (!K.sets.empty()) {
    const int l0 = X.column(M.l0, 0), l_last = X.column(M.l_last, 0), l_blind = X.column(M.l_blind, 0);
    const int nsets = (int)K.sets.size();
    auto z = [&](int i, int rot) { return X.column(M.z0 + i, rot); };
    terms.push_back(X.mul(l0, X.sub(one, z(0, 0))));
    terms.push_back(X.mul(l_last, X.sub(X.mul(z(nsets - 1, 0), z(nsets - 1, 0)), z(nsets - 1, 0))));
    for (int i = 1; i < nsets; i++) terms.push_back(X.mul(l0, X.sub(z(i, 0), z(i - 1, -(K.bf + 1)))));
    const int l_active = X.sub(X.sub(one, l_last), l_blind);
    const int xc = X.column(M.xcol, 0);
    for (int i = 0; i < nsets; i++) {
      int left = z(i, 1), right = z(i, 0);
      for (int j : K.sets[i]) {
        const int v = colref(C.permutation_columns[j]);
        left = X.mul(left, X.add(X.add(v, X.mul(X.column(M.sigma0 + j, 0), v_beta)), v_gamma));
        right = X.mul(right, X.add(X.add(v, X.mul(xc, v_bd[j])), v_gamma));
      }
      terms.push_back(X.mul(l_active, X.sub(left, right)));
    }
  }
  int numer = terms[0];
  for (size_t t = 1; t < terms.size(); t++) numer = X.add(X.mul(numer, v_y), terms[t]);
  const int root = X.mul(numer, X.column(M.tinv, 0));
  X.compile(root);
}

std::unique_ptr<ProvingKey> keygen(const Params& P, std::unique_ptr<Circuit> circuit, DevCtx* ctx) {
  Trace trace("keygen");
  auto pkp = std::make_unique<ProvingKey>();
  ProvingKey& K = *pkp;
  K.params = &P;
  K.circuit = std::move(circuit);
  const Circuit& C = *K.circuit;
  K.dom = std::make_unique<Domain>((uint32_t)C.degree, P.k);
  const Domain& D = *K.dom;
  K.dev = std::make_unique<Dev>(ctx);
  Dev& d = *K.dev;
  const uint32_t n = D.n;
  key_shape(K);
  if ((uint32_t)(K.bf + 1) >= n) fail(H2_EINVAL, "k too small for this circuit");
  const size_t nf = K.nf(), np = K.np();
  // fixed columns; the minimum rows a circuit needs are checked against n here
  std::vector<SparseCol> fixed = C.synthesize_fixed();
  for (auto& col : fixed)
    if (!col.empty() && col.rbegin()->first >= n - (uint32_t)(K.bf + 1)) fail(H2_EINVAL, "k too small for this circuit");
  Col lag = d.col((nf + np) * n);            // fixed then sigma, Lagrange form
  K.fixed_values = lag;
  K.sigma_values = lag + 2 * nf * (size_t)n;
  d.fill_sparse(K.fixed_values, n, fixed);
  // omega_col[i] = w^i (domain kit); sigma_j[i] = delta^j w^i except on the cells the copy constraints permute
  const DomainKit& kit = domain_kit(D, K.bf, ctx);
  K.omega_col = kit.omega_col;
  K.basis_ext = kit.basis_ext;
  K.xcol_ext = kit.xcol_ext;
  K.tinv = kit.tinv;
  const Fr delta = fr_delta();
  std::vector<std::pair<std::pair<int, uint32_t>, Fr>> moved;     // (permutation column, row) -> its sigma value
  {
    std::vector<std::pair<Col, Fr>> t(1);
    Fr dj = Fr::one();
    for (size_t j = 0; j < np; j++) {
      t[0] = {K.omega_col, dj};
      d.lincomb(K.sigma_values + 2 * j * (size_t)n, n, t);
      dj *= delta;
    }
    auto mapping = permutation_mapping(C);
    std::vector<pk::CellRef> refs;
    std::vector<Fr> vals;
    for (auto& kv : mapping) {
      if (kv.first == kv.second) continue;
      if (kv.first.second >= n || kv.second.second >= n) fail(H2_EINVAL, "k too small for this circuit");
      refs.push_back({(uint32_t)kv.first.first, kv.first.second});
      vals.push_back(delta.pow_u64((uint64_t)kv.second.first) * D.omega.pow_u64(kv.second.second));
      moved.push_back({kv.first, vals.back()});
    }
    if (!refs.empty()) {
      const pk::CellRef* d_refs = (const pk::CellRef*)d.upload(refs.data(), refs.size() * sizeof(pk::CellRef));
      Col d_vals = d.upload_frs(vals);
      hipLaunchKernelGGL(pk::scatter_cells_kernel, dim3((unsigned)((refs.size() + 255) / 256)), dim3(256), 0, d.s,
                         K.sigma_values, (size_t)n, d_refs, d_vals, (uint32_t)refs.size());
      hip_ok(hipGetLastError(), "scatter_cells_kernel");
    }
  }
  // commitments of the fixed + sigma columns, their coefficient and extended forms.  sigma_j = (identity permutation's
  // column j) + (the cells the copy constraints move): the first part's commitment is [delta^j] commit(w^i), kept
  // with the SRS; the second is sparse.  One MSM launch over nf + np SPARSE columns instead of np dense ones.
  // coefficient and extended forms of the fixed + sigma columns on the second stream, beside the commitment below and
  // the first phases of the proof (create_proof takes the second stream back before it needs them)
  Col polys = d.col((nf + np) * n);
  K.fixed_polys = polys;
  K.sigma_polys = polys + 2 * nf * (size_t)n;
  Col ext = d.col((nf + np) * (size_t)D.en);
  K.fixed_ext = ext;
  K.sigma_ext = ext + 2 * nf * (size_t)D.en;
  {
    hipStream_t side = d.side();
    d.order(d.s, side, 0);
    hip_ok(hipMemcpyAsync(polys, lag, (nf + np) * (size_t)n * 32, hipMemcpyDeviceToDevice, side), "hipMemcpyAsync(D2D)");
    d.ntt(polys, nf + np, D.omega_inv, D.k, &D.n_inv, side);
    coeff_to_extended(d, D, polys, nf + np, ext, side);
  }
  trace.mark("columns built");
  if (P.sigma_identity.size() < np) {
    const G1 c_omega = commit(d, P, K.omega_col, n, 1, true)[0];
    std::vector<HX> pts;
    HX cur = hx_of(c_omega);
    for (size_t j = 0; j < std::max<size_t>(np, 8); j++) {
      pts.push_back(cur);
      cur = hx_mul(delta, cur);
    }
    P.sigma_identity = hx_to_affine(pts);
    trace.mark("identity permutation committed");
  }
  std::vector<G1> commits;
  {
    Col sparse = d.col((nf + np) * (size_t)n);
    d.copy(sparse, K.fixed_values, nf * (size_t)n * 32);
    std::vector<SparseCol> diff(np);
    for (auto& kv : moved) {
      const int j = kv.first.first;
      diff[j][kv.first.second] = kv.second - delta.pow_u64((uint64_t)j) * D.omega.pow_u64(kv.first.second);
    }
    d.fill_sparse(sparse + 2 * nf * (size_t)n, n, diff);
    commits = commit(d, P, sparse, n, nf + np, true);
    d.release(sparse);
    std::vector<HX> sums;
    for (size_t j = 0; j < np; j++) sums.push_back(xyzz_add(hx_of(P.sigma_identity[j]), hx_of(commits[nf + j])));
    const std::vector<G1> sig = hx_to_affine(sums);
    for (size_t j = 0; j < np; j++) commits[nf + j] = sig[j];
  }
  trace.mark("fixed + sigma committed");
  K.fixed_commitments.assign(commits.begin(), commits.begin() + nf);
  K.sigma_commitments.assign(commits.begin() + nf, commits.end());
  // vk digest
  const std::string s = vk_debug_string(C, D.k, D.ext_k, D.omega, K.fixed_commitments, K.sigma_commitments);
  K.transcript_repr = vk_transcript_repr(s);
  trace.mark("vk digest");

  build_quotient_program(K);
  ExprProgram& X = K.prog;
  K.d_code = (const pk::XInstr*)d.upload(X.code.data(), X.code.size() * sizeof(pk::XInstr));
  trace.mark("program compiled");     // no synchronisation here: create_proof queues behind keygen's kernels on the same
                                      // stream, and the staged host copies of the uploads live as long as the key
  if (trace.on) {
    size_t nmul = 0, ncol = 0;
    for (auto& ins : X.code) {
      if ((ins.op_dst >> 24) == 2) nmul++;
      if ((ins.a & (3u << 30)) == pk::X_COL) ncol++;
      if ((ins.b & (3u << 30)) == pk::X_COL) ncol++;
    }
    fprintf(stderr, "[h2 keygen] quotient program: %zu instructions (%zu products), %zu column reads, %u slots, %zu constants\n",
            X.code.size(), nmul, ncol, X.nslots, X.consts.size());
  }
  return pkp;
}

// ---- create_proof ------------------------------------------------------------------------------
struct Query {
  Fr point;
  Col poly;
  Fr eval;
};

void shplonk_open(Transcript& tr, Dev& d, const Params& P, uint32_t n, const std::vector<Query>& queries);

std::vector<uint8_t> create_proof(ProvingKey& K, const Circuit& C, const std::vector<Fr>& public_input, Rng& rng, bool shplonk) {
  Trace trace("prove");
  const Domain& D = *K.dom;
  const Params& P = *K.params;
  Dev d(K.dev->c);
  const uint32_t n = D.n, en = D.en;
  const int bf = K.bf, deg = C.degree;
  const size_t na = (size_t)C.num_advice, ni = (size_t)C.num_instance, nz = K.sets.size(), nf = K.nf(), np = K.np();
  Transcript tr;
  tr.common_scalar(K.transcript_repr);
  if (public_input.size() > n - (uint32_t)(bf + 1)) fail(H2_EINVAL, "instance too long");
  if (!ni && !public_input.empty()) fail(H2_EINVAL, "circuit has no instance column");
  // Lagrange columns of this proof: advice | instance | z
  // ... | random polynomial (coefficient form; it sits behind the z columns so that both share one MSM launch)
  Col lag = d.col((na + ni + nz + 1) * (size_t)n);
  Col advice_values = lag, instance_values = lag + 2 * na * (size_t)n, z_values = lag + 2 * (na + ni) * (size_t)n;
  Col random_poly = lag + 2 * (na + ni + nz) * (size_t)n;
  if (ni) {
    std::vector<SparseCol> inst(1);
    for (size_t i = 0; i < public_input.size(); i++) inst[0][(uint32_t)i] = public_input[i];
    d.fill_sparse(instance_values, n, inst);
    for (auto& v : public_input) tr.common_scalar(v);
  }
  // advice: synthesize, blind the last bf + 1 rows, commit
  {
    std::vector<SparseCol> adv = C.synthesize_advice();
    for (auto& col : adv) {
      if (!col.empty() && col.rbegin()->first >= n - (uint32_t)(bf + 1)) fail(H2_EINVAL, "k too small for this circuit");
      for (uint32_t row = n - (bf + 1); row < n; row++) col[row] = rng.fr_random();
    }
    for (size_t j = 0; j < adv.size(); j++) (void)rng.fr_random();       // the Blind of each commitment (unused by KZG)
    d.fill_sparse(advice_values, n, adv);
  }
  trace.mark("witness uploaded");
  // Lagrange -> coefficient -> extended coset of the advice and instance columns: they wait for no challenge, so they
  // run on the context's second stream beside the commit phases (whose MSM tails leave the chip mostly idle); the
  // permutation products' follow once those exist; the main stream takes everything back before the quotient
  Col polys = d.col((na + ni + nz) * (size_t)n);
  Col ext = d.col((na + ni + nz) * (size_t)en);
  hipStream_t side = d.side();
  {
    PendingCommit pc = commit_begin(d, P, advice_values, n, na, true);
    d.wait_msm_tail(side);                     // (the tail event lies behind the witness upload on the main stream)
    hip_ok(hipMemcpyAsync(polys, lag, (na + ni) * (size_t)n * 32, hipMemcpyDeviceToDevice, side), "hipMemcpyAsync(D2D)");
    d.ntt(polys, na + ni, D.omega_inv, D.k, &D.n_inv, side);
    coeff_to_extended(d, D, polys, na + ni, ext, side);
    for (auto& pt : commit_finish(d, pc)) tr.write_point(pt);
  }
  trace.mark("advice committed");
  const Fr theta = tr.squeeze_challenge(), beta = tr.squeeze_challenge(), gamma = tr.squeeze_challenge();
  (void)theta;

  // permutation grand products
  const Fr delta = fr_delta();
  auto values_of = [&](const ColRef& cr) -> Col {
    if (cr.first == ADVICE) return advice_values + 2 * (size_t)cr.second * n;
    if (cr.first == FIXED) return K.fixed_values + 2 * (size_t)cr.second * n;
    return instance_values + 2 * (size_t)cr.second * n;
  };
  if (nz) {
    // every set's ratio column in ONE launch sequence (they only need beta and gamma), then one read-back of the
    // sets' last usable products: z_i = (prod_{s < i} tail_s) * prefix_i, blinding rows drawn set by set
    Col ratio = d.col(nz * (size_t)n);
    for (size_t s0 = 0; s0 < nz; s0 += pk::PERM_MAX_SETS) {
      const size_t cnt = std::min<size_t>(pk::PERM_MAX_SETS, nz - s0);
      pk::PermBatch B{};
      for (size_t q = 0; q < cnt; q++) {
        pk::PermArgs& A = B.set[q];
        const size_t si = s0 + q;
        A.ncols = (int)K.sets[si].size();
        if (A.ncols > pk::PERM_MAX_COLS) fail(H2_EINVAL, "permutation set too wide");
        for (int t = 0; t < A.ncols; t++) {
          const int j = K.sets[si][t];
          A.value[t] = values_of(C.permutation_columns[j]);
          A.sigma[t] = K.sigma_values + 2 * (size_t)j * n;
          A.beta_delta[t] = (beta * delta.pow_u64((uint64_t)j)).v;
        }
        A.beta = beta.v;
        A.gamma = gamma.v;
      }
      hipLaunchKernelGGL(pk::perm_ratio_kernel, dim3((n / pk::PERM_RUN + 255) / 256 + 1, (unsigned)cnt), dim3(256), 0, d.s, B,
                         K.omega_col, ratio + 2 * s0 * (size_t)n, n);
      hip_ok(hipGetLastError(), "perm_ratio_kernel");
    }
    {
      std::vector<Col> in, out;
      for (size_t si = 0; si < nz; si++) {
        in.push_back(ratio + 2 * si * (size_t)n);
        out.push_back(z_values + 2 * si * (size_t)n);
      }
      d.scan_batch(1, n, in, out, {});
    }
    // the products over the usable rows (row n - bf - 1 of every prefix column): one gather, one copy back
    Col tails_d = d.col(nz);
    for (size_t si = 0; si < nz; si++) d.copy(tails_d + 2 * si, z_values + 2 * (si * (size_t)n + (n - bf - 1)), 32);
    const std::vector<Fr> tails = d.download_frs(tails_d, nz);
    Fr last_z = Fr::one();
    for (size_t si = 0; si < nz; si++) {
      Col z = z_values + 2 * si * (size_t)n;
      if (!(last_z == Fr::one())) {
        hipLaunchKernelGGL(pk::scale_range_kernel, dim3((n + 255) / 256), dim3(256), 0, d.s, z, 0u, n, last_z.v);
        hip_ok(hipGetLastError(), "scale_range_kernel");
      }
      std::vector<Fr> blind(bf);
      for (int t = 0; t < bf; t++) blind[t] = rng.fr_random();
      Col d_blind = d.upload_frs(blind);
      d.copy(z + 2 * (size_t)(n - bf), d_blind, (size_t)bf * 32);
      last_z = last_z * tails[si];
      (void)rng.fr_random();
    }
    d.release(ratio);
  }
  trace.mark("grand products built");
  // the vanishing argument's random polynomial: one ChaCha20 seed, n sequential draws.  It comes from the RNG, not
  // from the transcript, so its commitment (over g) is computed in the launch of the permutation products' (over
  // g_lagrange); the transcript still receives the points in the reference's order
  {
    uint8_t seed[32];
    rng.fill(seed, 32);
    uint32_t key[8];
    memcpy(key, seed, 32);
    hip_ok(d.ops->chacha20_scalars(random_poly, n, 0, key, d.s), "chacha20_scalars");
    (void)rng.fr_random();
  }
  {
    PendingCommit pc = commit_begin(d, P, z_values, n, nz + 1, true, nz);
    if (nz) {
      d.wait_msm_tail(side);                   // the z columns (blinding rows included) are final before that MSM
      Col zp = polys + 2 * (na + ni) * (size_t)n;
      hip_ok(hipMemcpyAsync(zp, z_values, nz * (size_t)n * 32, hipMemcpyDeviceToDevice, side), "hipMemcpyAsync(D2D)");
      d.ntt(zp, nz, D.omega_inv, D.k, &D.n_inv, side);
      coeff_to_extended(d, D, zp, nz, ext + 2 * (na + ni) * (size_t)en, side);
    }
    for (auto& pt : commit_finish(d, pc)) tr.write_point(pt);
  }
  trace.mark("grand products + random poly committed");

  Col advice_polys = polys, z_polys = polys + 2 * (na + ni) * (size_t)n;
  const Fr y = tr.squeeze_challenge();
  d.order(side, d.s, 2);                       // every coefficient and extended form is needed from here on

  // the quotient: one program over every extended column
  Col h_ext = d.col(en);
  {
    const ColumnMap& M = K.cmap;
    std::vector<const U128*> ptrs(M.count);
    std::vector<uint32_t> masks(M.count, en - 1);
    for (size_t j = 0; j < na; j++) ptrs[M.advice0 + j] = ext + 2 * j * (size_t)en;
    for (size_t j = 0; j < nf; j++) ptrs[M.fixed0 + j] = K.fixed_ext + 2 * j * (size_t)en;
    for (size_t j = 0; j < ni; j++) ptrs[M.instance0 + j] = ext + 2 * (na + j) * (size_t)en;
    for (size_t j = 0; j < np; j++) ptrs[M.sigma0 + j] = K.sigma_ext + 2 * j * (size_t)en;
    for (size_t j = 0; j < nz; j++) ptrs[M.z0 + j] = ext + 2 * (na + ni + j) * (size_t)en;
    ptrs[M.l0] = K.basis_ext;
    ptrs[M.l_last] = K.basis_ext + 2 * (size_t)en;
    ptrs[M.l_blind] = K.basis_ext + 4 * (size_t)en;
    ptrs[M.xcol] = K.xcol_ext;
    ptrs[M.tinv] = K.tinv;
    masks[M.tinv] = (1u << (D.ext_k - D.k)) - 1;
    std::vector<Fr> consts = K.prog.consts;
    consts[K.c_y] = y;
    consts[K.c_beta] = beta;
    consts[K.c_gamma] = gamma;
    for (size_t j = 0; j < np; j++) consts[K.c_beta_delta[j]] = beta * delta.pow_u64((uint64_t)j);
    const U128* const* d_ptrs = (const U128* const*)d.upload(ptrs.data(), ptrs.size() * sizeof(void*));
    const uint32_t* d_masks = (const uint32_t*)d.upload(masks.data(), masks.size() * 4);
    for (auto& c : consts)
      for (int t = 0; t < 5; t++) c = c + c;            // c 2^256 -> c 2^261: the kernel's working form (R' = 2^261)
    Col d_consts = d.upload_frs(consts);
    const size_t lds_slots = K.prog.nslots > (uint32_t)pk::EXPR_REG_SLOTS ? K.prog.nslots - pk::EXPR_REG_SLOTS : 1;
    const size_t lds = lds_slots * 9 * pk::EXPR_BLOCK * 4;
    if (lds > 160 * 1024) fail(H2_EINVAL, "quotient program needs too many live values");
    hipLaunchKernelGGL(pk::expr_kernel, dim3((en + pk::EXPR_BLOCK - 1) / pk::EXPR_BLOCK), dim3(pk::EXPR_BLOCK), lds, d.s, K.d_code,
                       (uint32_t)K.prog.code.size(), d_ptrs, d_masks, d_consts, h_ext, en / n, en);
    hip_ok(hipGetLastError(), "expr_kernel");
  }
  d.release(ext);
  // extended -> coefficients: inverse NTT with 1 / 2^ext_k, un-shift the coset, keep n (deg - 1) coefficients
  d.ntt(h_ext, 1, D.ext_omega_inv, D.ext_k, &D.en_inv);
  const uint32_t hlen = n * (uint32_t)(deg - 1);
  hipLaunchKernelGGL(pk::coset_shrink_kernel, dim3((hlen + 255) / 256), dim3(256), 0, d.s, h_ext, hlen, D.zeta_inv.v,
                     D.zeta_inv.sqr().v);
  hip_ok(hipGetLastError(), "coset_shrink_kernel");
  Col h_pieces = h_ext;                       // deg - 1 pieces of n coefficients, contiguous
  trace.mark("quotient enqueued");
  for (auto& pt : commit(d, P, h_pieces, n, (size_t)(deg - 1), false)) tr.write_point(pt);
  trace.mark("quotient committed");
  for (int t = 0; t < deg - 1; t++) (void)rng.fr_random();

  // evaluations at x
  const Fr x = tr.squeeze_challenge();
  const Fr x_next = D.rotate(x, 1), x_last = D.rotate(x, -(bf + 1));
  std::vector<std::pair<Col, Fr>> wanted;
  for (auto& q : C.advice_queries) wanted.push_back({advice_polys + 2 * (size_t)q.first * n, D.rotate(x, q.second)});
  for (auto& q : C.fixed_queries) wanted.push_back({K.fixed_polys + 2 * (size_t)q.first * n, D.rotate(x, q.second)});
  wanted.push_back({random_poly, x});
  for (size_t j = 0; j < np; j++) wanted.push_back({K.sigma_polys + 2 * j * (size_t)n, x});
  for (size_t i = 0; i < nz; i++) {
    Col z = z_polys + 2 * i * (size_t)n;
    wanted.push_back({z, x});
    wanted.push_back({z, x_next});
    if (i + 1 < nz) wanted.push_back({z, x_last});
  }
  // h(X) = sum_i x^(n i) h_i(X), opened at x too
  Col h_poly = d.col(n);
  {
    const Fr xn = x.pow_u64(n);
    std::vector<std::pair<Col, Fr>> t;
    Fr p = Fr::one();
    for (int i = 0; i < deg - 1; i++) {
      t.push_back({h_pieces + 2 * (size_t)i * n, p});
      p *= xn;
    }
    d.lincomb(h_poly, n, t);
  }
  std::vector<std::pair<Col, Fr>> jobs = wanted;
  if (shplonk) jobs.push_back({h_poly, x});
  const std::vector<Fr> evals = d.evaluate(jobs, n);
  for (size_t i = 0; i < wanted.size(); i++) tr.write_scalar(evals[i]);
  trace.mark("evaluations");
  std::map<std::pair<Col, std::array<uint8_t, 32>>, Fr> eval_of;
  auto key_of = [](Col c, const Fr& pt) {
    std::array<uint8_t, 32> b;
    memcpy(b.data(), pt.v.v, 32);
    return std::make_pair(c, b);
  };
  for (size_t i = 0; i < jobs.size(); i++) eval_of[key_of(jobs[i].first, jobs[i].second)] = evals[i];

  // the opening queries in the prover's batching order
  std::vector<Query> queries;
  auto ask = [&](Col c, const Fr& pt) {
    auto it = eval_of.find(key_of(c, pt));
    queries.push_back({pt, c, it == eval_of.end() ? Fr::zero() : it->second});
  };
  for (auto& q : C.advice_queries) ask(advice_polys + 2 * (size_t)q.first * n, D.rotate(x, q.second));
  for (size_t i = 0; i < nz; i++) {
    ask(z_polys + 2 * i * (size_t)n, x);
    ask(z_polys + 2 * i * (size_t)n, x_next);
  }
  for (size_t i = nz >= 2 ? nz - 1 : 0; i-- > 0;) ask(z_polys + 2 * i * (size_t)n, x_last);
  for (auto& q : C.fixed_queries) ask(K.fixed_polys + 2 * (size_t)q.first * n, D.rotate(x, q.second));
  for (size_t j = 0; j < np; j++) ask(K.sigma_polys + 2 * j * (size_t)n, x);
  ask(h_poly, x);
  ask(random_poly, x);
  if (shplonk) {
    shplonk_open(tr, d, P, n, queries);
    return tr.bytes();
  }
  // GWC: one witness polynomial per distinct point, the v-power combination of everything opened there
  const Fr v = tr.squeeze_challenge();
  std::vector<Fr> points;
  for (auto& q : queries)
    if (std::find(points.begin(), points.end(), q.point) == points.end()) points.push_back(q.point);
  Col witnesses = d.col(points.size() * (size_t)n);
  Col accs = d.col(points.size() * (size_t)n);
  {
    std::vector<Col> in, out;
    for (size_t pi = 0; pi < points.size(); pi++) {
      std::vector<std::pair<Col, Fr>> t;
      Fr vp = Fr::one();
      for (auto& q : queries)
        if (q.point == points[pi]) {
          t.push_back({q.poly, vp});
          vp *= v;
        }
      d.lincomb(accs + 2 * pi * (size_t)n, n, t);
      in.push_back(accs + 2 * pi * (size_t)n);
      out.push_back(witnesses + 2 * pi * (size_t)n);
    }
    d.scan_batch(0, n, in, out, points);          // the four synthetic divisions side by side
  }
  for (auto& pt : commit(d, P, witnesses, n, points.size(), false)) tr.write_point(pt);
  trace.mark("openings committed");
  return tr.bytes();
}

// coefficients of the polynomial of degree < |points| through (points[i], values[i])
std::vector<Fr> interpolate(const std::vector<Fr>& points, const std::vector<Fr>& values) {
  std::vector<Fr> out(points.size(), Fr::zero());
  for (size_t i = 0; i < points.size(); i++) {
    std::vector<Fr> term(1, Fr::one());
    Fr den = Fr::one();
    for (size_t j = 0; j < points.size(); j++) {
      if (j == i) continue;
      std::vector<Fr> nt(term.size() + 1, Fr::zero());
      for (size_t dg = 0; dg < term.size(); dg++) {
        nt[dg] -= term[dg] * points[j];
        nt[dg + 1] += term[dg];
      }
      term = nt;
      den *= points[i] - points[j];
    }
    const Fr scale = values[i] * den.inv();
    for (size_t dg = 0; dg < term.size(); dg++) out[dg] += term[dg] * scale;
  }
  return out;
}
Fr horner(const std::vector<Fr>& c, const Fr& x) {
  Fr acc = Fr::zero();
  for (size_t i = c.size(); i-- > 0;) acc = acc * x + c[i];
  return acc;
}

// the rotation sets of SHPLONK: polynomials in first-appearance order with their (sorted) point sets, grouped by set
struct ShplonkSets {
  struct Member {
    Col poly;
    std::map<std::array<uint8_t, 32>, Fr> evals;    // by point
    int commitment = -1;                             // verifier side: index into its commitment list
  };
  struct Group {
    std::vector<Fr> points;                          // sorted
    std::vector<Member> members;
  };
  std::vector<Group> groups;
  std::vector<Fr> super;                             // sorted union
};
std::array<uint8_t, 32> fr_key(const Fr& f) {
  std::array<uint8_t, 32> b;
  memcpy(b.data(), f.v.v, 32);
  return b;
}
// `ids[i]` identifies the polynomial of query i (prover: its device pointer; verifier: a commitment index)
ShplonkSets shplonk_sets(const std::vector<Fr>& pts, const std::vector<Fr>& evs, const std::vector<uintptr_t>& ids) {
  struct Poly {
    uintptr_t id;
    std::vector<Fr> points;
    std::map<std::array<uint8_t, 32>, Fr> evals;
  };
  std::vector<Poly> polys;
  for (size_t i = 0; i < pts.size(); i++) {
    Poly* p = nullptr;
    for (auto& q : polys)
      if (q.id == ids[i]) p = &q;
    if (!p) {
      polys.push_back({ids[i], {}, {}});
      p = &polys.back();
    }
    if (std::find(p->points.begin(), p->points.end(), pts[i]) == p->points.end()) {
      p->points.push_back(pts[i]);
      p->evals[fr_key(pts[i])] = evs[i];
    }
  }
  ShplonkSets S;
  std::set<std::array<uint8_t, 32>> seen;
  for (auto& p : polys) {
    std::vector<Fr> pset = p.points;
    std::sort(pset.begin(), pset.end());
    ShplonkSets::Group* g = nullptr;
    for (auto& gg : S.groups)
      if (gg.points == pset) g = &gg;
    if (!g) {
      S.groups.push_back({pset, {}});
      g = &S.groups.back();
    }
    ShplonkSets::Member m;
    m.poly = (Col)p.id;
    m.commitment = (int)p.id;
    m.evals = p.evals;
    g->members.push_back(m);
    for (auto& pt : pset)
      if (seen.insert(fr_key(pt)).second) S.super.push_back(pt);
  }
  std::sort(S.super.begin(), S.super.end());
  return S;
}

// ProverSHPLONK::create_proof (prover.py _shplonk_open)
void shplonk_open(Transcript& tr, Dev& d, const Params& P, uint32_t n, const std::vector<Query>& queries) {
  const Fr y = tr.squeeze_challenge(), v = tr.squeeze_challenge();
  std::vector<Fr> pts, evs;
  std::vector<uintptr_t> ids;
  for (auto& q : queries) {
    pts.push_back(q.point);
    evs.push_back(q.eval);
    ids.push_back((uintptr_t)q.poly);
  }
  const ShplonkSets S = shplonk_sets(pts, evs, ids);
  Col h = d.col(n), acc = d.col(n), quo = d.col(n);
  std::vector<std::vector<std::vector<Fr>>> rems(S.groups.size());
  Fr vp = Fr::one();
  for (size_t gi = 0; gi < S.groups.size(); gi++) {
    const auto& G = S.groups[gi];
    std::vector<std::pair<Col, Fr>> t;
    std::vector<Fr> rsum(G.points.size(), Fr::zero());
    Fr yp = Fr::one();
    for (auto& m : G.members) {
      std::vector<Fr> vals;
      for (auto& pt : G.points) vals.push_back(m.evals.at(fr_key(pt)));
      const std::vector<Fr> r = interpolate(G.points, vals);
      rems[gi].push_back(r);
      t.push_back({m.poly, yp});
      for (size_t i = 0; i < r.size(); i++) rsum[i] += yp * r[i];
      yp *= y;
    }
    d.lincomb(acc, n, t);
    Col d_r = d.upload_frs(rsum);
    hipLaunchKernelGGL(pk::sub_prefix_kernel, dim3(1), dim3(64), 0, d.s, acc, d_r, (uint32_t)rsum.size());
    hip_ok(hipGetLastError(), "sub_prefix_kernel");
    // divide by prod (X - p): one synthetic division per point, ping-ponging two buffers (the quotient keeps the
    // column's length, its top coefficients are zero)
    Col src = acc, dst = quo;
    for (auto& pt : G.points) {
      d.divide_linear(src, n, pt, dst);
      std::swap(src, dst);
    }
    d.lincomb(h, n, gi == 0 ? std::vector<std::pair<Col, Fr>>{{src, vp}}
                            : std::vector<std::pair<Col, Fr>>{{h, Fr::one()}, {src, vp}});
    vp *= v;
  }
  tr.write_point(commit(d, P, h, n, 1, false)[0]);
  const Fr u = tr.squeeze_challenge();
  Fr zt = Fr::one();
  for (auto& pt : S.super) zt *= u - pt;
  // L(X) = sum_i v^i z_i (sum_j y^j (f_ij(X) - r_ij(u))) - Z_T(u) h(X), then / (X - u) / z_0
  std::vector<std::pair<Col, Fr>> lt;
  Fr lconst = Fr::zero(), z0 = Fr::zero();
  vp = Fr::one();
  for (size_t gi = 0; gi < S.groups.size(); gi++) {
    const auto& G = S.groups[gi];
    Fr z_i = Fr::one();
    for (auto& pt : S.super)
      if (std::find(G.points.begin(), G.points.end(), pt) == G.points.end()) z_i *= u - pt;
    if (gi == 0) z0 = z_i;
    Fr yp = Fr::one();
    for (size_t mi = 0; mi < G.members.size(); mi++) {
      lt.push_back({G.members[mi].poly, vp * z_i * yp});
      lconst += vp * z_i * yp * horner(rems[gi][mi], u);
      yp *= y;
    }
    vp *= v;
  }
  lt.push_back({h, -zt});
  d.lincomb(acc, n, lt);
  {
    std::vector<Fr> c0(1, lconst);
    Col d_c = d.upload_frs(c0);
    hipLaunchKernelGGL(pk::sub_prefix_kernel, dim3(1), dim3(64), 0, d.s, acc, d_c, 1u);
    hip_ok(hipGetLastError(), "sub_prefix_kernel");
  }
  d.divide_linear(acc, n, u, quo);
  hipLaunchKernelGGL(pk::scale_range_kernel, dim3((n + 255) / 256), dim3(256), 0, d.s, quo, 0u, n, z0.inv().v);
  hip_ok(hipGetLastError(), "scale_range_kernel");
  tr.write_point(commit(d, P, quo, n, 1, false)[0]);
}

// ---- verify_proof --------------------------------------------------------------------------------------------------------------
struct MsmTerms {
  std::vector<std::pair<Fr, G1>> t;
  void append(const Fr& s, const G1& p) { t.push_back({s, p}); }
  void scale(const Fr& f) {
    for (auto& x : t) x.first *= f;
  }
  void add(const MsmTerms& o) { t.insert(t.end(), o.t.begin(), o.t.end()); }
};

// the group elements of up to four MsmTerms, on the GPU side by side: equal points merged, the rest through the table-free
// small MSM (msm_small_kernel: one quad per term)
std::vector<G1> msm_eval(Dev& d, const std::vector<const MsmTerms*>& jobs) {
  const size_t count = jobs.size();
  std::vector<G1> out(count);
  std::vector<std::vector<uint8_t>> pts(count), sc(count);
  std::vector<const void*> d_pts, d_sc;
  std::vector<uint32_t> ms, live;
  size_t mmax = 0;
  for (size_t j = 0; j < count; j++) {
    std::vector<std::pair<G1, Fr>> merged;
    for (auto& st : jobs[j]->t) {
      if (st.second.inf || st.first.is_zero()) continue;
      bool found = false;
      for (auto& mg : merged)
        if (mg.first == st.second) {
          mg.second += st.first;
          found = true;
        }
      if (!found) merged.push_back({st.second, st.first});
    }
    if (merged.empty()) continue;
    const size_t m = merged.size();
    pts[j].resize(m * 64);
    sc[j].resize(m * 32);
    for (size_t i = 0; i < m; i++) {
      memcpy(pts[j].data() + 64 * i, merged[i].first.x.v.v, 32);
      memcpy(pts[j].data() + 64 * i + 32, merged[i].first.y.v.v, 32);
      memcpy(sc[j].data() + 32 * i, merged[i].second.v.v, 32);
    }
    d_pts.push_back(d.upload(pts[j].data(), pts[j].size()));
    d_sc.push_back(d.upload(sc[j].data(), sc[j].size()));
    ms.push_back((uint32_t)m);
    live.push_back((uint32_t)j);
    mmax = std::max(mmax, m);
  }
  if (live.empty()) return out;
  const size_t blocks = (mmax + 15) / 16, nl = live.size();
  Col work = d.col((nl * blocks * 144 + 4 * nl + 31) / 32), d_out = d.col(3 * nl);
  hip_ok(ops_of(H2_BN254)->msm_small(d_pts.data(), d_sc.data(), ms.data(), (uint32_t)nl, work, d_out, d.s), "msm_small");
  std::vector<uint64_t> jac(12 * nl);
  hip_ok(hipMemcpyAsync(jac.data(), d_out, 96 * nl, hipMemcpyDeviceToHost, d.s), "hipMemcpyAsync(D2H)");
  d.sync();
  d.release(work);
  d.release(d_out);
  for (size_t q = 0; q < nl; q++) {
    const uint64_t* J = jac.data() + 12 * q;
    const Fq X = Fq::from_mont_limbs(J), Y = Fq::from_mont_limbs(J + 4), Z = Fq::from_mont_limbs(J + 8);
    if (Z.is_zero()) continue;
    const Fq zi = Z.inv(), zi2 = zi.sqr();
    G1& g = out[live[q]];
    g.x = X * zi2;
    g.y = Y * zi2 * zi;
    g.inf = false;
  }
  return out;
}

bool verify_proof(ProvingKey& K, const uint8_t* proof, size_t proof_len, const std::vector<Fr>& instance, bool shplonk) {
  const Circuit& C = *K.circuit;
  const Domain& D = *K.dom;
  const Params& P = *K.params;
  const uint32_t n = D.n;
  const int bf = K.bf, deg = C.degree;
  if (instance.size() > n - (uint32_t)(bf + 1)) return false;
  if (!C.num_instance && !instance.empty()) return false;
  Trace trace("verify");
  Transcript tr(proof, proof_len);
  tr.common_scalar(K.transcript_repr);
  for (auto& v : instance) tr.common_scalar(v);
  std::vector<G1> advice_c(C.num_advice), z_c(K.sets.size()), h_c(deg - 1);
  for (auto& p : advice_c)
    if (!tr.read_point(&p)) return false;
  const Fr theta = tr.squeeze_challenge(), beta = tr.squeeze_challenge(), gamma = tr.squeeze_challenge();
  (void)theta;
  for (auto& p : z_c)
    if (!tr.read_point(&p)) return false;
  G1 random_c;
  if (!tr.read_point(&random_c)) return false;
  const Fr y = tr.squeeze_challenge();
  for (auto& p : h_c)
    if (!tr.read_point(&p)) return false;
  const Fr x = tr.squeeze_challenge();
  const Fr xn = x.pow_u64(n);
  if (xn == Fr::one()) return false;
  auto lagrange_at = [&](int row) {       // L_row(x) = w^row (x^n - 1) / (n (x - w^row)), row taken mod n
    const Fr wi = row >= 0 ? D.omega.pow_u64((uint64_t)row) : D.omega_inv.pow_u64((uint64_t)(-row));
    return wi * (xn - Fr::one()) * D.n_inv * (x - wi).inv();
  };
  std::vector<Fr> inst_evals;
  for (auto& q : C.instance_queries) {
    Fr acc = Fr::zero();
    for (size_t i = 0; i < instance.size(); i++) acc += instance[i] * lagrange_at((int)i - q.second);
    inst_evals.push_back(acc);
  }
  std::vector<Fr> adv_evals(C.advice_queries.size()), fix_evals(C.fixed_queries.size()), sigma_evals(K.np());
  for (auto& e : adv_evals)
    if (!tr.read_scalar(&e)) return false;
  for (auto& e : fix_evals)
    if (!tr.read_scalar(&e)) return false;
  Fr random_eval;
  if (!tr.read_scalar(&random_eval)) return false;
  for (auto& e : sigma_evals)
    if (!tr.read_scalar(&e)) return false;
  struct ZE { Fr ev, next, last; };
  std::vector<ZE> z_evals(K.sets.size());
  for (size_t i = 0; i < K.sets.size(); i++) {
    if (!tr.read_scalar(&z_evals[i].ev) || !tr.read_scalar(&z_evals[i].next)) return false;
    if (i + 1 < K.sets.size() && !tr.read_scalar(&z_evals[i].last)) return false;
  }
  // the vanishing argument
  const Fr l_last = lagrange_at(-(bf + 1)), l_0 = lagrange_at(0);
  Fr l_blind = Fr::zero();
  for (int r = -bf; r < 0; r++) l_blind += lagrange_at(r);
  std::vector<Fr> exprs;
  for (auto& g : C.gates) exprs.push_back(expr_eval(g, adv_evals, fix_evals, inst_evals));
  const Fr delta = fr_delta();
  if (!K.sets.empty()) {
    const size_t ns = K.sets.size();
    exprs.push_back(l_0 * (Fr::one() - z_evals[0].ev));
    exprs.push_back(l_last * (z_evals[ns - 1].ev.sqr() - z_evals[ns - 1].ev));
    for (size_t i = 1; i < ns; i++) exprs.push_back(l_0 * (z_evals[i].ev - z_evals[i - 1].last));
    auto column_eval = [&](const ColRef& cr) {
      const auto& qs = cr.first == ADVICE ? C.advice_queries : cr.first == FIXED ? C.fixed_queries : C.instance_queries;
      const auto& ev = cr.first == ADVICE ? adv_evals : cr.first == FIXED ? fix_evals : inst_evals;
      for (size_t i = 0; i < qs.size(); i++)
        if (qs[i].first == cr.second && qs[i].second == 0) return ev[i];
      fail(H2_EINVAL, "permutation column is not queried at the current rotation");
    };
    for (size_t i = 0; i < ns; i++) {
      Fr left = z_evals[i].next, right = z_evals[i].ev;
      for (int j : K.sets[i]) {
        const Fr v = column_eval(C.permutation_columns[j]);
        left *= v + beta * sigma_evals[j] + gamma;
        right *= v + delta.pow_u64((uint64_t)j) * beta * x + gamma;
      }
      exprs.push_back((left - right) * (Fr::one() - (l_last + l_blind)));
    }
  }
  Fr folded = Fr::zero();
  for (auto& e : exprs) folded = folded * y + e;
  const Fr expected_h = folded * (xn - Fr::one()).inv();
  MsmTerms h_msm;
  for (size_t i = h_c.size(); i-- > 0;) {
    h_msm.scale(xn);
    h_msm.append(Fr::one(), h_c[i]);
  }
  // the opening queries, in the prover's batching order; commitment -1 = the h combination
  struct VQ { Fr point; int id; Fr eval; };
  std::vector<G1> commitments;
  std::vector<VQ> queries;
  auto reg = [&](const G1& c) { commitments.push_back(c); return (int)commitments.size() - 1; };
  std::vector<int> id_adv, id_z, id_fix, id_sig;
  for (auto& c : advice_c) id_adv.push_back(reg(c));
  for (auto& c : z_c) id_z.push_back(reg(c));
  for (auto& c : K.fixed_commitments) id_fix.push_back(reg(c));
  for (auto& c : K.sigma_commitments) id_sig.push_back(reg(c));
  const int id_random = reg(random_c);
  const int id_h = (int)commitments.size();        // not a single point: h_msm
  const Fr x_next = D.rotate(x, 1), x_last = D.rotate(x, -(bf + 1));
  for (size_t qi = 0; qi < C.advice_queries.size(); qi++)
    queries.push_back({D.rotate(x, C.advice_queries[qi].second), id_adv[C.advice_queries[qi].first], adv_evals[qi]});
  for (size_t i = 0; i < K.sets.size(); i++) {
    queries.push_back({x, id_z[i], z_evals[i].ev});
    queries.push_back({x_next, id_z[i], z_evals[i].next});
  }
  for (size_t i = K.sets.size() >= 2 ? K.sets.size() - 1 : 0; i-- > 0;) queries.push_back({x_last, id_z[i], z_evals[i].last});
  for (size_t qi = 0; qi < C.fixed_queries.size(); qi++)
    queries.push_back({D.rotate(x, C.fixed_queries[qi].second), id_fix[C.fixed_queries[qi].first], fix_evals[qi]});
  for (size_t j = 0; j < K.np(); j++) queries.push_back({x, id_sig[j], sigma_evals[j]});
  queries.push_back({x, id_h, expected_h});
  queries.push_back({x, id_random, random_eval});
  auto as_msm = [&](int id, const Fr& factor) {
    MsmTerms m;
    if (id == id_h) {
      m = h_msm;
      m.scale(factor);
    } else {
      m.append(factor, commitments[id]);
    }
    return m;
  };
  MsmTerms left, right;
  if (!shplonk) {
    // VerifierGWC (halo2_proofs src/poly/kzg/multiopen/gwc/verifier.rs)
    const Fr v = tr.squeeze_challenge();
    std::vector<Fr> points;
    for (auto& q : queries)
      if (std::find(points.begin(), points.end(), q.point) == points.end()) points.push_back(q.point);
    std::vector<G1> ws(points.size());
    for (auto& w : ws)
      if (!tr.read_point(&w)) return false;
    const Fr u = tr.squeeze_challenge();
    Fr eval_multi = Fr::zero(), up = Fr::one();
    for (size_t pi = 0; pi < points.size(); pi++) {
      Fr vp = Fr::one(), ev = Fr::zero();
      MsmTerms batch;
      for (auto& q : queries)
        if (q.point == points[pi]) {
          batch.add(as_msm(q.id, vp));
          ev += vp * q.eval;
          vp *= v;
        }
      batch.scale(up);
      right.add(batch);
      eval_multi += up * ev;
      right.append(up * points[pi], ws[pi]);
      left.append(up, ws[pi]);
      up *= u;
    }
    right.append(-eval_multi, P.g0);
  } else {
    // VerifierSHPLONK (src/poly/kzg/multiopen/shplonk/verifier.rs)
    std::vector<Fr> pts, evs;
    std::vector<uintptr_t> ids;
    for (auto& q : queries) {
      pts.push_back(q.point);
      evs.push_back(q.eval);
      ids.push_back((uintptr_t)q.id);
    }
    const ShplonkSets S = shplonk_sets(pts, evs, ids);
    const Fr y_ch = tr.squeeze_challenge(), v = tr.squeeze_challenge();
    G1 h1, h2;
    if (!tr.read_point(&h1)) return false;
    const Fr u = tr.squeeze_challenge();
    if (!tr.read_point(&h2)) return false;
    MsmTerms outer;
    Fr r_outer = Fr::zero(), vp = Fr::one(), z_0 = Fr::one(), z_0_diff_inv = Fr::one();
    for (size_t gi = 0; gi < S.groups.size(); gi++) {
      const auto& G = S.groups[gi];
      Fr z_diff = Fr::one();
      for (auto& pt : S.super)
        if (std::find(G.points.begin(), G.points.end(), pt) == G.points.end()) z_diff *= u - pt;
      if (gi == 0) {
        for (auto& pt : G.points) z_0 *= u - pt;
        if (z_diff.is_zero()) return false;
        z_0_diff_inv = z_diff.inv();
        z_diff = Fr::one();
      } else {
        z_diff *= z_0_diff_inv;
      }
      MsmTerms inner;
      Fr r_inner = Fr::zero(), yp = Fr::one();
      for (auto& m : G.members) {
        std::vector<Fr> vals;
        for (auto& pt : G.points) vals.push_back(m.evals.at(fr_key(pt)));
        r_inner += yp * horner(interpolate(G.points, vals), u);
        inner.add(as_msm(m.commitment, yp));
        yp *= y_ch;
      }
      inner.scale(vp * z_diff);
      outer.add(inner);
      r_outer += vp * r_inner * z_diff;
      vp *= v;
    }
    outer.append(-r_outer, P.g0);
    outer.append(-z_0, h1);
    outer.append(u, h2);
    left.append(Fr::one(), h2);
    right.add(outer);
  }
  if (!bn::g2_on_curve(P.g2) || !bn::g2_on_curve(P.s_g2)) return false;
  trace.mark("transcript replayed");
  Dev d(K.dev->c);
  const std::vector<G1> lr = msm_eval(d, {&left, &right});
  const G1& L = lr[0];
  const G1& Rr = lr[1];
  trace.mark("two MSMs");
  bn::G2 neg_g2 = P.g2;
  neg_g2.y = -neg_g2.y;
  const bool ok = bn::pairing_check({{L, P.s_g2}, {Rr, neg_g2}});
  trace.mark("pairing check");
  return ok;
}

// ---- the circuits of wasm.rs by index ------------------------------------------------------------------------------------------
struct Job {
  std::unique_ptr<Circuit> circuit;
  std::vector<Fr> public_input;
  bool shplonk = false;
};
// prove side: circuit with witness, public inputs as the reference passes them (wasm.rs:84-117)
Job job_for_proof(const Json& js, int idx) {
  Job j;
  if (idx == 0) {
    auto c = std::make_unique<CollatzCircuit>();
    c->set_sequence(js.array("x"));
    j.circuit = std::move(c);
    j.shplonk = true;
  } else if (idx == 1) {
    auto c = std::make_unique<ArithmeticCircuit>();
    c->x = Fr::from_u64(js.u64("x"));
    c->y = Fr::from_u64(js.u64("y"));
    c->constant = Fr::from_u64(js.u64("constant"));
    c->has_witness = true;
    j.public_input = {Fr::from_u64(js.u64("constant")), Fr::from_u64(js.u64("z"))};     // wasm.rs:93-94
    j.circuit = std::move(c);
  } else {
    auto c = std::make_unique<PoseidonCircuit>();
    const auto& x = js.array("x");
    if (x.size() != 2) fail(H2_EPROOF, "poseidon: x must hold two values");
    c->message[0] = Fr::from_u64(x[0]);
    c->message[1] = Fr::from_u64(x[1]);
    c->has_witness = true;
    auto it = js.scalars.find("output");
    if (it == js.scalars.end()) fail(H2_EPROOF, "poseidon: missing output");
    j.public_input = {Fr::from_hex(it->second.c_str())};                                   // wasm.rs:116 hex_to_fr(output)
    j.circuit = std::move(c);
  }
  return j;
}
// verify side: the empty circuit, public inputs recomputed (wasm.rs:128-168)
Job job_for_verify(const Json& js, int idx) {
  Job j;
  if (idx == 0) {
    j.circuit = std::make_unique<CollatzCircuit>();
    j.shplonk = true;
  } else if (idx == 1) {
    j.circuit = std::make_unique<ArithmeticCircuit>();
    j.public_input = {Fr::from_u64(js.u64("constant")), Fr::from_u64(js.u64("z"))};
  } else {
    auto c = std::make_unique<PoseidonCircuit>();
    const auto& x = js.array("x");
    if (x.size() != 2) fail(H2_EPROOF, "poseidon: x must hold two values");
    c->message[0] = Fr::from_u64(x[0]);
    c->message[1] = Fr::from_u64(x[1]);
    j.public_input = {c->output()};
    j.circuit = std::move(c);
  }
  return j;
}

// [k] pt on the twist (affine, host): the [s]G2 of ParamsKZG::new
bn::G2 g2_mul(const Fr& k, const bn::G2& pt) {
  uint8_t kb[32];
  k.to_le_bytes(kb);
  bn::G2 r;   // identity
  auto add = [](const bn::G2& a, const bn::G2& b) {
    if (a.inf) return b;
    if (b.inf) return a;
    bn::F2 lam;
    if (a.x == b.x) {
      if (!(a.y == b.y) || a.y.is_zero()) return bn::G2{};
      lam = bn::scale(bn::sqr(a.x), Fq::from_u64(3)) * bn::inv(bn::scale(a.y, Fq::from_u64(2)));
    } else {
      lam = (b.y - a.y) * bn::inv(b.x - a.x);
    }
    bn::G2 o;
    o.x = bn::sqr(lam) - a.x - b.x;
    o.y = lam * (a.x - o.x) - a.y;
    o.inf = false;
    return o;
  };
  for (int i = 255; i >= 0; i--) {
    r = add(r, r);
    if ((kb[i >> 3] >> (i & 7)) & 1) r = add(r, pt);
  }
  return r;
}

// proving keys kept between calls (the reference rebuilds them on every prove and verify, wasm.rs:86,95,114,132;
// a key depends only on the params and the circuit's fixed columns): keyed by params digest and circuit index
struct KeyEntry {
  std::array<uint8_t, 64> digest;
  int circuit;
  int device;                         // a key's columns live on one GPU
  std::unique_ptr<ProvingKey> key;
};
std::vector<KeyEntry> g_keys;
bool g_key_cache = true;

std::unique_ptr<Circuit> empty_circuit(int idx) {
  if (idx == 0) return std::make_unique<CollatzCircuit>();
  if (idx == 1) return std::make_unique<ArithmeticCircuit>();
  return std::make_unique<PoseidonCircuit>();
}
int circuit_slot(int idx) { return idx == 0 ? 0 : idx == 1 ? 1 : 2; }

// the key for (params, circuit): cached or built now; `owner` keeps a freshly built uncached key alive
ProvingKey& key_for(const Params& P, int idx, DevCtx* ctx, std::unique_ptr<ProvingKey>& owner) {
  const int slot = circuit_slot(idx);
  if (g_key_cache) {
    for (auto& e : g_keys)
      if (e.circuit == slot && e.device == ctx->device && e.digest == P.digest) {
        e.key->params = &P;       // the params list may have been reordered since
        return *e.key;
      }
  }
  owner = keygen(P, empty_circuit(slot), ctx);
  if (!g_key_cache) return *owner;
  if (g_keys.size() >= 6) g_keys.erase(g_keys.begin());
  g_keys.push_back({P.digest, slot, ctx->device, std::move(owner)});
  return *g_keys.back().key;
}

DevCtx* the_ctx() {
  if (!g_h2.ready) fail(H2_ENOTINIT, "h2_init has not been called");
  DevCtx* c = ctx_current();
  if (!c) fail(H2_EINVAL, "no h2 context on the current HIP device");
  return c;
}

template <class F>
int guarded(F&& body) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  try {
    return body();
  } catch (const Fail& f) {
    g_h2.last_error = f.what;
    return f.status;
  } catch (const std::exception& e) {
    g_h2.last_error = e.what();
    return H2_EPROOF;
  }
}

int emit(const std::vector<uint8_t>& data, uint8_t* out, size_t cap, size_t* out_len) {
  if (out_len) *out_len = data.size();
  if (!out || cap < data.size()) return H2_EINVAL;      // *out_len says how much is needed
  memcpy(out, data.data(), data.size());
  return H2_OK;
}

}  // namespace

// Blake2bRead::read_point: 32 bytes, x little-endian with the parity of y in bit 6 and the identity flag in bit 7
bool h2::Transcript::read_point(G1* p) {
  const uint8_t* src = take32();
  if (!src) return false;
  uint8_t b[32];
  memcpy(b, src, 32);
  const int sign = (b[31] >> 6) & 1, inf = (b[31] >> 7) & 1;
  b[31] &= 0x3F;
  Fq x;
  if (!Fq::from_le_bytes_canonical(b, &x)) return false;
  G1 g;
  if (inf || (x.is_zero() && sign == 0)) {
    // the reference's Blake2bRead::read_point absorbs the point through common_point, which refuses the point at
    // infinity ("cannot write points at infinity to the transcript"): verify_proof returns Err for such a proof
    return false;
  } else {
    const Fq y2 = x * x * x + Fq::from_u64(3);
    // q = 3 mod 4: a square root is y2^((q + 1) / 4)
    uint64_t e[4];
    for (int i = 0; i < 4; i++) e[i] = (uint64_t)BN254_FQ::P(2 * i) | ((uint64_t)BN254_FQ::P(2 * i + 1) << 32);
    e[0] += 1;                                            // no carry: the low word of q ends in ...47
    for (int i = 0; i < 4; i++) e[i] = (e[i] >> 2) | (i < 3 ? e[i + 1] << 62 : 0);
    Fq y = y2.pow_limbs(e);
    if (!(y * y == y2)) return false;
    if ((y.is_odd() ? 1 : 0) != sign) y = -y;
    g.x = x;
    g.y = y;
    g.inf = false;
  }
  common_point(g);
  *p = g;
  return true;
}

extern "C" {

int h2_circuit_count(void) { return 3; }      // wasm.rs:182

int h2_simulate(const char* json, int circuit, char* out, size_t cap, size_t* out_len) {
  return guarded([&]() -> int {
    std::string r;
    if (circuit == 0) {
      r = "N/A";                                                    // collatz.rs:248-250
    } else {
      const Json js(json);
      if (circuit == 1) {
        // (x * x) * (y * y) + constant in u64 arithmetic (arithmetic_circuit.rs:298-301; the reference panics on overflow)
        const unsigned __int128 x = js.u64("x"), y = js.u64("y"), c = js.u64("constant");
        const unsigned __int128 xx = x * x, yy = y * y;
        if (xx >> 64 || yy >> 64) fail(H2_EPROOF, "u64 overflow");
        const unsigned __int128 p = xx * yy;
        if (p >> 64 || (p + c) >> 64) fail(H2_EPROOF, "u64 overflow");
        r = std::to_string((uint64_t)(p + c));
      } else {
        PoseidonCircuit c;
        const auto& x = js.array("x");
        if (x.size() != 2) fail(H2_EPROOF, "poseidon: x must hold two values");
        c.message[0] = Fr::from_u64(x[0]);
        c.message[1] = Fr::from_u64(x[1]);
        r = "0x" + c.output().hex64();                               // format!("{:?}", Fr)
      }
    }
    if (out_len) *out_len = r.size();
    if (!out || cap < r.size() + 1) return H2_EINVAL;
    memcpy(out, r.c_str(), r.size() + 1);
    return H2_OK;
  });
}

int h2_setup(uint32_t k, h2_rng_fill_t rng_fn, void* rng_ctx, uint8_t* out, size_t cap, size_t* out_len) {
  return guarded([&]() -> int {
    if (k < 1 || k > 24) return H2_EINVAL;
    const size_t n = (size_t)1 << k, total = 4 + 128 * n + 256;
    if (out_len) *out_len = total;
    if (!out || cap < total) return H2_EINVAL;
    DevCtx* ctx = the_ctx();
    Dev d(ctx);
    Rng rng{rng_fn, rng_ctx};
    const Fr s = rng.fr_random();
    // g[i] = [s^i] G
    Col g = d.col(2 * n), gl = d.col(2 * n);
    uint64_t sl[4];
    Dev::limbs(s, sl);
    hip_ok(d.ops->srs_powers(g, sl, (uint32_t)n, d.s), "srs_powers");
    // g_lagrange[i] = [L_i(s)] G with L_i(s) = w^i (s^n - 1) / (n (s - w^i))
    Domain D(3, k);
    const Fr sn = s.pow_u64(n);
    Col lag = d.col(n);
    if (sn == Fr::one()) {             // s on the domain (never in practice): L_i(s) is an indicator
      std::vector<SparseCol> ind(1);
      Fr w = Fr::one();
      for (uint32_t i = 0; i < n; i++) {
        if (w == s) ind[0][i] = Fr::one();
        w *= D.omega;
      }
      d.fill_sparse(lag, n, ind);
    } else {
      // ws = w^i ; den = s - w^i ; lag = ws / den * (s^n - 1) / n
      Col ws = d.col(n), den = d.col(n);
      std::vector<SparseCol> c1(1);
      c1[0][0] = Fr::one();
      d.fill_sparse(ws, n, c1);
      d.ntt(ws, 1, D.omega, k);                                     // ones
      d.copy(den, ws, n * 32);
      uint64_t w[4];
      Dev::limbs(D.omega, w);
      hip_ok(d.ops->poly_powers(ws, n, 1, w, d.s), "poly_powers");
      d.lincomb(den, (uint32_t)n, {{den, s}, {ws, -Fr::one()}});
      hip_ok(d.ops->poly_inverse(den, n, d.s), "poly_inverse");
      hip_ok(d.ops->poly_pointwise(den, ws, n, 2, d.s), "poly_pointwise");
      d.lincomb(lag, (uint32_t)n, {{den, (sn - Fr::one()) * D.n_inv}});
    }
    hip_ok(d.ops->fixed_base_mul(gl, lag, (uint32_t)n, d.s), "fixed_base_mul");
    memcpy(out, &k, 4);
    hip_ok(hipMemcpyAsync(out + 4, g, 64 * n, hipMemcpyDeviceToHost, d.s), "D2H");
    hip_ok(hipMemcpyAsync(out + 4 + 64 * n, gl, 64 * n, hipMemcpyDeviceToHost, d.s), "D2H");
    // g2 and [s] g2
    bn::G2 g2;
    g2.x = {Fq::from_hex("0x1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed"),
            Fq::from_hex("0x198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2")};
    g2.y = {Fq::from_hex("0x12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa"),
            Fq::from_hex("0x090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b")};
    g2.inf = false;
    const bn::G2 s_g2 = g2_mul(s, g2);
    uint8_t* t = out + 4 + 128 * n;
    const Fq* parts[8] = {&g2.x.a, &g2.x.b, &g2.y.a, &g2.y.b, &s_g2.x.a, &s_g2.x.b, &s_g2.y.a, &s_g2.y.b};
    for (int i = 0; i < 8; i++) memcpy(t + 32 * i, parts[i]->v.v, 32);
    d.sync();
    return H2_OK;
  });
}

int h2_generate_proof(const uint8_t* params, size_t params_len, const char* json, int circuit, h2_rng_fill_t rng_fn,
                      void* rng_ctx, uint8_t* out, size_t cap, size_t* out_len) {
  return guarded([&]() -> int {
    Trace trace("generate_proof");
    DevCtx* ctx = the_ctx();
    const Params& P = params_get(params, params_len);
    trace.mark("params");
    const Json js(json);
    Job job = job_for_proof(js, circuit);
    trace.mark("job");
    // the key comes from the EMPTY circuit (wasm.rs:86,95,114 rebuild it on every call; here it is kept, see
    // h2_key_cache), the witness from the JSON
    std::unique_ptr<ProvingKey> owner;
    ProvingKey& K = key_for(P, circuit, ctx, owner);
    trace.mark("key");
    Rng rng{rng_fn, rng_ctx};
    const std::vector<uint8_t> proof = create_proof(K, *job.circuit, job.public_input, rng, job.shplonk);
    trace.mark("create_proof");
    return emit(proof, out, cap, out_len);
  });
}

int h2_verify_proof(const uint8_t* params, size_t params_len, const uint8_t* proof, size_t proof_len, const char* json,
                    int circuit, int* ok) {
  if (ok) *ok = 0;
  return guarded([&]() -> int {
    if (!ok || (!proof && proof_len)) return H2_EINVAL;
    DevCtx* ctx = the_ctx();
    const Params& P = params_get(params, params_len);
    const Json js(json);
    Job job = job_for_verify(js, circuit);
    std::unique_ptr<ProvingKey> owner;
    ProvingKey& K = key_for(P, circuit, ctx, owner);
    *ok = verify_proof(K, proof, proof_len, job.public_input, job.shplonk) ? 1 : 0;
    return H2_OK;
  });
}

// forget the resident SRS tables (the next call with any params blob parses and registers it again)
int h2_params_cache_clear(void) {
  return guarded([&]() -> int {
    g_keys.clear();                                   // keys point into the params list
    for (auto& p : g_params) {
      (void)h2_bases_release(p.h_g);
      (void)h2_bases_release(p.h_gl);
    }
    g_params.clear();
    return H2_OK;
  });
}

// called by h2_shutdown: drop keys, params and the cached device blocks
void h2_prover_shutdown(void) {
  g_keys.clear();
  g_kits.clear();
  for (auto& p : g_params) {
    (void)h2_bases_release(p.h_g);
    (void)h2_bases_release(p.h_gl);
  }
  g_params.clear();
  for (auto& kv : Dev::cache()) {
    DeviceGuard dg(kv.first.first);
    (void)hipFree(kv.second);
  }
  Dev::cache().clear();
}

// keep proving keys between calls (default) or rebuild them on every call as the reference does; returns the old setting
// commit phases that were spread over more than one context since the library was loaded (tests)
uint64_t h2_selftest_sharded_commits(void) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  return g_sharded_commits;
}

// test hook: rows per context from which a commit phase is spread over the contexts (0 restores the default)
int h2_selftest_set_shard_min_rows(size_t rows) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  g_shard_min_rows = rows ? rows : 1024;
  return H2_OK;
}

int h2_key_cache(int enable) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  const int old = g_key_cache ? 1 : 0;
  g_key_cache = enable != 0;
  if (!g_key_cache) g_keys.clear();
  return old;
}

// host-side pieces exposed for the CPU tests (no GPU needed): the vk digest of a circuit for given commitments, the
// Poseidon constants, Blake2b, the pairing
int h2_selftest_host(int what, const uint8_t* in, size_t in_len, uint8_t* out, size_t cap, size_t* out_len) {
  return guarded([&]() -> int {
    std::vector<uint8_t> r;
    if (what == 0) {                       // Blake2b-512 with the transcript personalisation
      Blake2b h("Halo2-Transcript");
      h.update(in, in_len);
      r.resize(64);
      h.digest(r.data());
    } else if (what == 1) {                // Poseidon constants: 68 x 3 round constants, mds, minv (canonical LE)
      const PoseidonConstants& pc = poseidon_constants();
      auto put = [&](const Fr& f) {
        uint8_t b[32];
        f.to_le_bytes(b);
        r.insert(r.end(), b, b + 32);
      };
      for (auto& row : pc.rcs)
        for (auto& v : row) put(v);
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) put(pc.mds[i][j]);
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) put(pc.minv[i][j]);
    } else if (what == 2 || what == 3 || what == 4) {   // vk Debug string of circuit (what - 2) for k = in[0] and the given
      const int idx = what - 2;                         // commitments (64-byte canonical x || y each, zero = identity)
      std::unique_ptr<Circuit> c;
      if (idx == 0) c = std::make_unique<CollatzCircuit>();
      else if (idx == 1) c = std::make_unique<ArithmeticCircuit>();
      else c = std::make_unique<PoseidonCircuit>();
      if (in_len < 1) return H2_EINVAL;
      const uint32_t k = in[0];
      Domain D((uint32_t)c->degree, k);
      const size_t nfx = (size_t)c->num_fixed, nsg = c->permutation_columns.size();
      if (in_len != 1 + 64 * (nfx + nsg)) return H2_EINVAL;
      auto pt = [&](size_t i) {
        G1 g;
        const uint8_t* p = in + 1 + 64 * i;
        Fq::from_le_bytes_canonical(p, &g.x);
        Fq::from_le_bytes_canonical(p + 32, &g.y);
        g.inf = g.x.is_zero() && g.y.is_zero();
        return g;
      };
      std::vector<G1> fc, sc;
      for (size_t i = 0; i < nfx; i++) fc.push_back(pt(i));
      for (size_t i = 0; i < nsg; i++) sc.push_back(pt(nfx + i));
      const std::string s = vk_debug_string(*c, k, D.ext_k, D.omega, fc, sc);
      const Fr repr = vk_transcript_repr(s);
      r.resize(32);
      repr.to_le_bytes(r.data());
      r.insert(r.end(), s.begin(), s.end());
    } else if (what == 6) {                // quotient program of circuit in[0]: u32 x 6 = instructions, products, column
      if (in_len != 1) return H2_EINVAL;   // reads, LDS slots, constants, inserted reductions; then the code (12 bytes each)
      ProvingKey K;
      if (in[0] == 0) K.circuit = std::make_unique<CollatzCircuit>();
      else if (in[0] == 1) K.circuit = std::make_unique<ArithmeticCircuit>();
      else K.circuit = std::make_unique<PoseidonCircuit>();
      key_shape(K);
      build_quotient_program(K);
      uint32_t st[6] = {(uint32_t)K.prog.code.size(), 0, 0, K.prog.nslots, (uint32_t)K.prog.consts.size(), K.prog.nreduce};
      for (auto& ins : K.prog.code) {
        if ((ins.op_dst >> 24) == 2) st[1]++;
        if ((ins.a & (3u << 30)) == pk::X_COL) st[2]++;
        if ((ins.b & (3u << 30)) == pk::X_COL) st[2]++;
      }
      r.resize(24 + K.prog.code.size() * sizeof(pk::XInstr));
      memcpy(r.data(), st, 24);
      memcpy(r.data() + 24, K.prog.code.data(), K.prog.code.size() * sizeof(pk::XInstr));
    } else if (what == 5) {                // pairing check on two (G1, G2) pairs: 2 x (64 + 128) canonical bytes -> 1 byte
      if (in_len != 2 * 192) return H2_EINVAL;
      std::vector<std::pair<G1, bn::G2>> pairs;
      for (int i = 0; i < 2; i++) {
        const uint8_t* p = in + 192 * i;
        G1 g;
        Fq::from_le_bytes_canonical(p, &g.x);
        Fq::from_le_bytes_canonical(p + 32, &g.y);
        g.inf = g.x.is_zero() && g.y.is_zero();
        bn::G2 q;
        Fq::from_le_bytes_canonical(p + 64, &q.x.a);
        Fq::from_le_bytes_canonical(p + 96, &q.x.b);
        Fq::from_le_bytes_canonical(p + 128, &q.y.a);
        Fq::from_le_bytes_canonical(p + 160, &q.y.b);
        q.inf = q.x.is_zero() && q.y.is_zero();
        if (!bn::g2_on_curve(q)) return H2_EPROOF;
        pairs.push_back({g, q});
      }
      r.push_back(bn::pairing_check(pairs) ? 1 : 0);
    } else {
      return H2_EINVAL;
    }
    return emit(r, out, cap, out_len);
  });
}

}  // extern "C"
