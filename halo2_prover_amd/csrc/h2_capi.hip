// h2_capi.hip -- implementation of include/h2hip.h (the drop-in C ABI).
//
// One context per process, bound to one GPU.  Every entry point validates its arguments,
// takes the context mutex, enqueues on the context stream (or the caller's) and reports
// errors as h2_status_t values: nothing throws or aborts across the ABI
// (the reference's panics -- /root/reference/circuits/src/utils.rs:91,120 `.expect(..)`,
// best_multiexp's assert_eq!(coeffs.len(), bases.len()) -- become H2_EINVAL here).
#include "h2_internal.hpp"

#include <cstdio>
#include <cstring>

#include "h2_host.hpp"
#include "h2_ntt.hpp"
#include "h2_poly.hpp"

using namespace h2;

namespace {
// m Jacobian points (96 B, Montgomery limbs) -> m affine points (64 B, identity = (0, 0)) on the host, the m inversions
// folded into one: the one-thread-per-point device kernel this replaces took 0.35 ms of pure latency per call
template <class FP>
void jac_to_affine_host(const uint8_t* jac, size_t m, uint8_t* out) {
  using H = HF<FP>;
  std::vector<H> z(m), pre(m);
  H acc = H::one();
  for (size_t j = 0; j < m; j++) {
    z[j] = H::from_mont_limbs(jac + 96 * j + 64);
    pre[j] = acc;
    if (!z[j].is_zero()) acc *= z[j];
  }
  H inv = acc.inv();
  for (size_t j = m; j-- > 0;) {
    uint8_t* o = out + 64 * j;
    if (z[j].is_zero()) {
      memset(o, 0, 64);
      continue;
    }
    const H zi = inv * pre[j], zi2 = zi.sqr();
    inv *= z[j];
    const H x = H::from_mont_limbs(jac + 96 * j) * zi2, y = H::from_mont_limbs(jac + 96 * j + 32) * zi2 * zi;
    memcpy(o, x.v.v, 32);
    memcpy(o + 32, y.v.v, 32);
  }
}
void jac_to_affine_host(int curve, const uint8_t* jac, size_t m, uint8_t* out) {
  if (curve == H2_BN254) jac_to_affine_host<BN254_FQ>(jac, m, out);
  else if (curve == H2_PALLAS) jac_to_affine_host<PASTA_FP>(jac, m, out);
  else jac_to_affine_host<PASTA_FQ>(jac, m, out);
}
}  // namespace

namespace h2 {

Global g_h2;
std::recursive_mutex g_h2_mu;

int dev_fail(hipError_t e, const char* where) {
  char buf[256];
  snprintf(buf, sizeof buf, "%s: %s", where, hipGetErrorString(e));
  g_h2.last_error = buf;
  return H2_EDEVICE;
}

bool curve_ok(int c) { return c == H2_BN254 || c == H2_PALLAS || c == H2_VESTA; }

const CurveOps* ops_of(int curve) {
  switch (curve) {
    case H2_BN254: return curve_ops_bn254();
    case H2_PALLAS: return curve_ops_pallas();
    case H2_VESTA: return curve_ops_vesta();
  }
  return nullptr;
}

DevCtx* ctx_current() {
  if (!g_h2.ready || g_h2.ctx.empty()) return nullptr;
  if (g_h2.ctx.size() == 1) return &g_h2.ctx[0];
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  for (auto& c : g_h2.ctx)
    if (c.device == dev) return &c;
  return nullptr;
}
size_t ctx_index(const DevCtx* c) { return (size_t)(c - &g_h2.ctx[0]); }

// ---- arenas ----------------------------------------------------------------------------------
static uint64_t g_arena_growths = 0, g_arena_waits = 0;
int arena_acquire(Arena& a, size_t want, hipStream_t s) {
  if (a.bytes < want) {
    g_arena_growths++;
    if (a.p) {
      // rare: a larger call than any before.  Work enqueued on other streams may still use the arena.
      H2_TRY(hipDeviceSynchronize());
      (void)hipFree(a.p);
      a.p = nullptr;
      a.bytes = 0;
      a.used = false;
    }
    a.clean_bytes = 0;
    const size_t sz = want + (want >> 3);  // head-room so slightly larger calls do not reallocate
    hipError_t e = hipMalloc(&a.p, sz);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      g_h2.last_error = std::string("hipMalloc: ") + hipGetErrorString(e);
      return H2_ENOMEM;
    }
    a.bytes = sz;
  }
  if (!a.ev) H2_TRY(hipEventCreateWithFlags(&a.ev, hipEventDisableTiming));
  if (a.used && a.last != s) {                                         // order behind the previous user
    g_arena_waits++;
    H2_TRY(hipStreamWaitEvent(s, a.ev, 0));
  }
  return H2_OK;
}
int arena_release(Arena& a, hipStream_t s) {
  H2_TRY(hipEventRecord(a.ev, s));
  a.last = s;
  a.used = true;
  return H2_OK;
}
static void arena_free(Arena& a) {
  if (a.p) (void)hipFree(a.p);
  if (a.ev) (void)hipEventDestroy(a.ev);
  a = Arena{};
}

// ---- bases ---------------------------------------------------------------------------------
// d_affine lives on context `src`; the table is built there and copied to every other context
static int register_device(DevCtx& src, int curve, const void* d_affine, size_t n, uint64_t* handle_out) {
  if (!curve_ok(curve) || !d_affine || !handle_out || n == 0) return H2_EINVAL;
  const CurveOps* ops = ops_of(curve);
  MsmGeom g = msm_geometry(n, ops->scalar_bits);
  if ((uint64_t)g.W * n >= (1ull << 31)) return H2_EINVAL;
  BasesEntry be{};
  be.curve = curve;
  be.n = n;
  be.geom = g;
  be.table_bytes = (size_t)g.W * n * 64;
  be.table.assign(g_h2.ctx.size(), nullptr);
  auto fail = [&](int rc) {
    for (size_t i = 0; i < be.table.size(); i++)
      if (be.table[i]) {
        DeviceGuard dg(g_h2.ctx[i].device);
        (void)hipFree(be.table[i]);
      }
    return rc;
  };
  for (size_t i = 0; i < g_h2.ctx.size(); i++) {
    DeviceGuard dg(g_h2.ctx[i].device);
    hipError_t e = hipMalloc(&be.table[i], be.table_bytes);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      be.table[i] = nullptr;
      g_h2.last_error = std::string("hipMalloc(table): ") + hipGetErrorString(e);
      return fail(H2_ENOMEM);
    }
  }
  const size_t si = ctx_index(&src);
  {
    DeviceGuard dg(src.device);
    // the points are checked while the table is built: a counter in the (otherwise idle) scan arena
    int rc = arena_acquire(src.div_ws, (size_t)2 * DIV_MAX_CHUNKS * 32, src.stream);
    if (rc != H2_OK) return fail(rc);
    uint32_t* d_bad = (uint32_t*)src.div_ws.p;
    uint32_t bad = 0;
    // the table kernel's scratch (80 bytes per table entry) is the MSM workspace, idle while bases are being registered
    Arena& table_ws = src.msm_ws.of(src.stream);
    table_ws.clean_bytes = 0;                      // the table kernel's scratch overwrites what an MSM left zeroed
    rc = arena_acquire(table_ws, be.table_bytes / 64 * MSM_TABLE_SCRATCH, src.stream);
    if (rc != H2_OK) return fail(rc);
    hipError_t e = hipMemsetAsync(d_bad, 0, 4, src.stream);
    if (e == hipSuccess) e = ops->table_build(d_affine, be.table[si], table_ws.p, (uint32_t)n, g, d_bad, src.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, src.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(src.stream);
    if (e != hipSuccess) return fail(dev_fail(e, "msm_table_kernel"));
    (void)arena_release(src.div_ws, src.stream);
    (void)arena_release(table_ws, src.stream);
    if (bad) {
      g_h2.last_error = "bases: " + std::to_string(bad) + " point(s) not on the curve";
      return fail(H2_EINVAL);
    }
    for (size_t i = 0; i < g_h2.ctx.size(); i++) {
      if (i == si) continue;
      e = hipMemcpyPeer(be.table[i], g_h2.ctx[i].device, be.table[si], src.device, be.table_bytes);
      if (e != hipSuccess) return fail(dev_fail(e, "hipMemcpyPeer(table)"));
    }
  }
  uint64_t h = g_h2.next_handle++;
  g_h2.bases[h] = be;
  *handle_out = h;
  return H2_OK;
}

// ---- MSM -----------------------------------------------------------------------------------
int msm_common_checks(int curve, uint64_t handle, size_t first, size_t n, size_t m, const BasesEntry** be) {
  if (!g_h2.ready) return H2_ENOTINIT;
  if (!curve_ok(curve) || m == 0) return H2_EINVAL;
  auto it = g_h2.bases.find(handle);
  if (it == g_h2.bases.end()) return H2_EHANDLE;
  if (it->second.curve != curve) return H2_EINVAL;
  if (first > it->second.n || n > it->second.n - first) return H2_EINVAL;  // best_multiexp: assert_eq!(coeffs.len(), bases.len())
  *be = &it->second;
  return H2_OK;
}

// Columns per launch: the sort indexes its m * W * n entries with 32 bits, so a wide batch of long columns
// (2^24 rows x 8 columns) goes through in groups of columns, one after the other on the same stream and workspace.
static uint64_t g_msm_max_entries = (1ull << 31) - 1;   // lowered only by h2_selftest_set_msm_max_entries (tests)
// guard mode (tests only, h2_selftest_msm_guard): the workspace is laid out with a red zone behind every region, filled
// with a pattern before each launch sequence and inspected after it
static bool g_msm_guard = false, g_msm_guard_poke = false, g_sort2_pack = true;
static uint64_t g_guard_launches = 0, g_guard_violations = 0;
static std::string g_guard_first;
static size_t msm_cols_per_launch(const MsmGeom& geom, size_t n) {
  const uint64_t per_col = (uint64_t)geom.W * n;
  const uint64_t by_entries = g_msm_max_entries / per_col;
  uint64_t by_keys = ((1ull << 31) - 1) / geom.B;
  // wide windows go through the two-level sort, whose one-block scan of the coarse bins holds S2_MAX_H of them: rather
  // than fall back to the one-level sort's scattered stores (which windows beyond 16 bits cannot use at all), a wider
  // batch runs in groups of that many columns
  if (geom.B > 4096) {
    const uint64_t by_bins = S2_MAX_H / msm_sort2_geom(n, geom).Hc;
    if (by_bins >= 1 && by_bins < by_keys) by_keys = by_bins;
  }
  return (size_t)(by_entries < by_keys ? by_entries : by_keys);   // 0: a single column is already too long
}

int msm_device_run(DevCtx& c, int curve, const BasesEntry& be, const void* d_scalars, size_t first_base, size_t n,
                   size_t col_stride, size_t m, void* d_out, bool affine_out, hipStream_t stream,
                   const BasesEntry* const* per_column) {
  const size_t group = msm_cols_per_launch(be.geom, n);
  if (group == 0) return H2_EINVAL;
  // columns with their own bases: one launch only (they must share the registered length, hence the geometry)
  const void* col_tables[MSM_MAX_MULTI];
  if (per_column) {
    if (m > MSM_MAX_MULTI || m > group) return H2_EINVAL;
    for (size_t j = 0; j < m; j++) {
      if (per_column[j]->n != be.n || per_column[j]->curve != be.curve) return H2_EINVAL;
      col_tables[j] = (const char*)per_column[j]->table[ctx_index(&c)] + first_base * 64;
    }
  }
  const size_t out_sz = affine_out ? 64 : 96;
  const CurveOps* ops = ops_of(curve);
  // the table rows of bases first_base ...: entries are w * n_bases + i relative to this pointer
  const char* table = (const char*)be.table[ctx_index(&c)] + first_base * 64;
  for (size_t j0 = 0; j0 < m; j0 += group) {
    const size_t mm = m - j0 < group ? m - j0 : group;
    MsmWorkspace ws = msm_workspace(n, mm, be.geom, g_msm_guard ? 256u : 0u, be.n, g_sort2_pack);
    if (ws.E >= (1ull << 31) || ws.K >= (1ull << 31)) return H2_EINVAL;
    Arena& A = c.msm_ws.of(stream);
    int rc = arena_acquire(A, ws.total, stream);
    if (rc != H2_OK) return rc;
    // every kernel's index range against the region it indexes, before anything is enqueued.  (n_bases is the
    // REGISTERED length whatever the range: a sorted entry is w * be.n + i relative to the table row of first_base)
    if (const char* broken = msm_check(ws, be.geom, n, mm, col_stride, (uint32_t)be.n, A.bytes)) {
      g_h2.last_error = std::string("msm launch geometry: ") + broken;
      return H2_EDEVICE;
    }
    if (g_msm_guard) H2_TRY(hipMemsetAsync(A.p, MSM_GUARD_BYTE, ws.total, stream));
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (g_h2.profiling) {
      if (c.prof_used == c.prof_events.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) c.prof_events.push_back({a, b});
      }
      if (c.prof_used < c.prof_events.size()) {
        ev0 = c.prof_events[c.prof_used].first;
        ev1 = c.prof_events[c.prof_used].second;
        c.prof_used++;
        c.prof_alg_bytes += (double)mm * (double)n * 96.0 + (double)mm * 96.0;  // SURVEY.md 8(d) bytes_msm
      }
    }
    if (c.tail_wanted && !c.tail_event) H2_TRY(hipEventCreateWithFlags(&c.tail_event, hipEventDisableTiming));
    void* dst = (char*)d_out + j0 * out_sz;
    // the previous launch sequence on this workspace left its counter region zero: no memset when this one's fits in it
    const bool zeroed = !g_msm_guard && A.clean_bytes >= ws.zero_bytes && A.clean_off == ws.off_misc;
    A.clean_bytes = 0;                             // (until this sequence is enqueued whole)
    hipError_t e = ops->msm_launch(table, per_column ? col_tables : nullptr, (uint32_t)be.n,
                                   (const char*)d_scalars + j0 * col_stride * 32, n, col_stride,
                                   mm, be.geom, (char*)A.p, ws, stream, ev0, ev1, (c.tail_wanted && !ev1) ? c.tail_event : nullptr,
                                   affine_out ? nullptr : dst, zeroed);
    c.tail_recorded = c.tail_wanted;
    c.tail_wait = ev1 ? ev1 : c.tail_event;
    if (e != hipSuccess) return dev_fail(e, "msm_launch");
    if (!g_msm_guard) {
      A.clean_off = ws.off_misc;
      A.clean_bytes = ws.zero_bytes;
    }
    if (g_msm_guard) {
      uint32_t* d_bad = nullptr;
      std::vector<uint32_t> bad(ws.n_regions, 0);
      H2_TRY(hipMalloc(&d_bad, ws.n_regions * 4));
      H2_TRY(hipMemsetAsync(d_bad, 0, ws.n_regions * 4, stream));
      if (g_msm_guard_poke)        // the checker's own test: one byte just behind the second region
        H2_TRY(hipMemsetAsync((char*)A.p + ws.regions[1].off + ws.regions[1].bytes, 0, 1, stream));
      hipLaunchKernelGGL(msm_guard_check_kernel, dim3(ws.n_regions), dim3(256), 0, stream, (const uint8_t*)A.p, ws, d_bad);
      H2_TRY(hipMemcpyAsync(bad.data(), d_bad, ws.n_regions * 4, hipMemcpyDeviceToHost, stream));
      H2_TRY(hipStreamSynchronize(stream));
      (void)hipFree(d_bad);
      g_guard_launches++;
      for (uint32_t r = 0; r < ws.n_regions; r++)
        if (bad[r]) {
          if (!g_guard_violations)
            g_guard_first = std::string(ws.regions[r].name) + ": " + std::to_string(bad[r]) + " byte(s) behind the region, n=" +
                            std::to_string(n) + " m=" + std::to_string(mm) + (ws.sort2 ? " two-level sort" : ws.staged ? " staged scatter" : " direct scatter");
          g_guard_violations++;
        }
    }
    if (affine_out) {
      e = ops->to_affine((char*)A.p + ws.off_tree2, dst, (uint32_t)mm, stream);
      if (e != hipSuccess) return dev_fail(e, "msm finish kernel");
    }
    rc = arena_release(A, stream);
    if (rc != H2_OK) return rc;
  }
  return H2_OK;
}

// ---- NTT -----------------------------------------------------------------------------------
static int get_twiddles(DevCtx& c, const CurveOps* ops, const uint64_t omega[4], uint32_t log_n, const uint64_t* scale,
                        const void** out) {
  for (auto& t : c.twiddles) {
    if (t.field == ops->scalar_field_id && t.log_n == log_n && memcmp(t.omega, omega, 32) == 0 && t.scaled == (scale != nullptr) &&
        (!scale || memcmp(t.scale, scale, 32) == 0)) {
      t.stamp = ++c.stamp;
      *out = t.tw;
      return H2_OK;
    }
  }
  if (c.twiddles.size() >= 16) {  // evict the least recently used table
    size_t victim = 0;
    for (size_t i = 1; i < c.twiddles.size(); i++)
      if (c.twiddles[i].stamp < c.twiddles[victim].stamp) victim = i;
    H2_TRY(hipDeviceSynchronize());
    (void)hipFree(c.twiddles[victim].tw);
    c.twiddles.erase(c.twiddles.begin() + victim);
  }
  TwiddleEntry te{};
  te.field = ops->scalar_field_id;
  te.log_n = log_n;
  memcpy(te.omega, omega, 32);
  te.scaled = scale != nullptr;
  if (scale) memcpy(te.scale, scale, 32);
  hipError_t e = hipMalloc(&te.tw, ops->ntt_table_bytes(log_n));
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return H2_ENOMEM;
  }
  // built on the library's stream and finished before anybody uses it: a table is shared by every later caller,
  // whatever stream they bring (once per (field, omega, log n, constant))
  e = ops->ntt_twiddles(te.tw, omega, log_n, c.stream, scale);
  if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
  if (e != hipSuccess) {
    (void)hipFree(te.tw);
    return dev_fail(e, "ntt_build_tables");
  }
  te.stamp = ++c.stamp;
  c.twiddles.push_back(te);
  *out = te.tw;
  return H2_OK;
}

int ntt_enqueue(DevCtx& c, int curve, void* d_a, size_t m, const uint64_t omega[4], uint32_t log_n, hipStream_t stream,
                const uint64_t* scale) {
  const CurveOps* ops = ops_of(curve);
  if (!ops) return H2_EINVAL;
  const void* tw = nullptr;
  // a constant that rides in the inter-pass twiddles needs tables built with it
  int rc = get_twiddles(c, ops, omega, log_n, (scale && ops->ntt_scale_in_table(log_n)) ? scale : nullptr, &tw);
  if (rc != H2_OK) return rc;
  NttPlan pl = ntt_make_plan(log_n);
  void* scratch = nullptr;
  Arena* A = nullptr;
  if (pl.npass > 1) {
    A = &c.ntt_ws.of(stream);
    rc = arena_acquire(*A, m * ((size_t)32 << log_n), stream);
    if (rc != H2_OK) return rc;
    scratch = A->p;
  }
  hipError_t e = ops->ntt_launch(d_a, scratch, tw, log_n, m, stream, scale);
  if (e != hipSuccess) return dev_fail(e, "ntt_launch");
  if (A) return arena_release(*A, stream);
  return H2_OK;
}

}  // namespace h2

namespace {

// every *_device entry point: the context of the current device, and the stream the work goes to
struct Call {
  DevCtx* c = nullptr;
  hipStream_t stream = nullptr;
  int rc = H2_OK;
  Call(void* stream_) {
    if (!g_h2.ready) { rc = H2_ENOTINIT; return; }
    c = ctx_current();
    if (!c) { rc = H2_EINVAL; g_h2.last_error = "no h2 context on the current HIP device"; return; }
    stream = stream_ ? (hipStream_t)stream_ : c->stream;
  }
};

int init_devices(int n, const int* ids) {
  if (g_h2.ready) {
    if ((size_t)n != g_h2.ctx.size()) return H2_EINVAL;
    for (int i = 0; i < n; i++)
      if (g_h2.ctx[i].device != ids[i]) return H2_EINVAL;
    return H2_OK;
  }
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) {
    (void)hipGetLastError();
    g_h2.last_error = "no HIP device available";
    return H2_EDEVICE;
  }
  for (int i = 0; i < n; i++)
    if (ids[i] < 0 || ids[i] >= count) return H2_EINVAL;
  int prev = 0;
  (void)hipGetDevice(&prev);
  g_h2.ctx.assign((size_t)n, DevCtx{});
  for (int i = 0; i < n; i++) {
    DevCtx& c = g_h2.ctx[i];
    c.device = ids[i];
    H2_TRY(hipSetDevice(ids[i]));
    // a BLOCKING stream on purpose: callers that pass stream = NULL (e.g. PyTorch's legacy default stream)
    // get work that is ordered against the null stream, so their own copies / kernels see finished results
    H2_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamDefault));
    for (int cv = 0; cv < 3; cv++) H2_TRY(ops_of(cv)->kernel_setup());
  }
  // one context: stay on its device (h2_init(device) has always left the process there); several: back to where we were
  H2_TRY(hipSetDevice(n == 1 ? ids[0] : prev));
  g_h2.ready = true;
  return H2_OK;
}

}  // namespace

extern "C" {

int h2_version(void) { return 1001; }

const char* h2_strerror(int s) {
  switch (s) {
    case H2_OK: return "ok";
    case H2_EINVAL: return "invalid argument (length / log_n mismatch, null pointer or unknown curve)";
    case H2_ENOMEM: return "out of memory";
    case H2_EDEVICE: return "HIP device error";
    case H2_EHANDLE: return "unknown bases handle";
    case H2_ENOTINIT: return "h2_init has not been called";
    case H2_EPROOF: return "proof or input rejected";
  }
  return "unknown status";
}

const char* h2_last_device_error(void) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  static thread_local std::string copy;
  copy = g_h2.last_error;
  return copy.c_str();
}

int h2_init(int device) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  return init_devices(1, &device);
}

int h2_init_devices(int n_devices, const int* device_ids) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  if (n_devices <= 0 || n_devices > 64 || !device_ids) return H2_EINVAL;
  return init_devices(n_devices, device_ids);
}

int h2_device_count(void) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  return g_h2.ready ? (int)g_h2.ctx.size() : H2_ENOTINIT;
}

void h2_prover_shutdown(void);   // h2_prover.hip: keys, params and cached blocks of the product surface

int h2_shutdown(void) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  if (!g_h2.ready) return H2_OK;
  h2_prover_shutdown();
  for (size_t i = 0; i < g_h2.ctx.size(); i++) {
    DevCtx& c = g_h2.ctx[i];
    DeviceGuard dg(c.device);
    (void)hipDeviceSynchronize();
    for (auto& kv : g_h2.bases)
      if (kv.second.table[i]) (void)hipFree(kv.second.table[i]);
    for (auto& t : c.twiddles) (void)hipFree(t.tw);
    for (Arena& a : c.msm_ws.slot) arena_free(a);
    for (Arena& a : c.ntt_ws.slot) arena_free(a);
    arena_free(c.stage);
    arena_free(c.div_ws);
    for (auto& pe : c.prof_events) {
      (void)hipEventDestroy(pe.first);
      (void)hipEventDestroy(pe.second);
    }
    if (c.shard_ev) (void)hipEventDestroy(c.shard_ev);
    if (c.side_stream) (void)hipStreamDestroy(c.side_stream);
    for (auto& e : c.side_ev)
      if (e) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(c.stream);
  }
  g_h2.bases.clear();
  g_h2.ctx.clear();
  g_h2.ready = false;
  return H2_OK;
}

int h2_bases_register_device(h2_curve_t curve, const void* d_affine, size_t n, uint64_t* handle_out) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  Call k(nullptr);
  if (k.rc != H2_OK) return k.rc;
  // the caller's copy / kernel that produced d_affine may still be in flight on another stream
  H2_TRY(hipDeviceSynchronize());
  return register_device(*k.c, (int)curve, d_affine, n, handle_out);
}

int h2_bases_register(h2_curve_t curve, const uint64_t* affine, size_t n, uint64_t* handle_out) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  if (!g_h2.ready) return H2_ENOTINIT;
  if (!curve_ok((int)curve) || !affine || !handle_out || n == 0) return H2_EINVAL;
  DevCtx& c = g_h2.ctx[0];
  DeviceGuard dg(c.device);
  int rc = arena_acquire(c.stage, n * 64, c.stream);
  if (rc != H2_OK) return rc;
  H2_TRY(hipMemcpyAsync(c.stage.p, affine, n * 64, hipMemcpyHostToDevice, c.stream));
  rc = register_device(c, (int)curve, c.stage.p, n, handle_out);   // synchronises c.stream
  return rc;
}

int h2_bases_release(uint64_t handle) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  if (!g_h2.ready) return H2_ENOTINIT;
  auto it = g_h2.bases.find(handle);
  if (it == g_h2.bases.end()) return H2_EHANDLE;
  for (size_t i = 0; i < g_h2.ctx.size(); i++) {
    DeviceGuard dg(g_h2.ctx[i].device);
    (void)hipDeviceSynchronize();     // launches on caller streams may still read the table
    (void)hipFree(it->second.table[i]);
  }
  g_h2.bases.erase(it);
  return H2_OK;
}

int64_t h2_bases_len(uint64_t handle) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  if (!g_h2.ready) return H2_ENOTINIT;
  auto it = g_h2.bases.find(handle);
  if (it == g_h2.bases.end()) return H2_EHANDLE;
  return (int64_t)it->second.n;
}

int h2_msm_plan(uint64_t handle, h2_msm_plan_t* out) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  if (!g_h2.ready) return H2_ENOTINIT;
  if (!out) return H2_EINVAL;
  auto it = g_h2.bases.find(handle);
  if (it == g_h2.bases.end()) return H2_EHANDLE;
  out->window_bits = it->second.geom.c;
  out->windows = it->second.geom.W;
  out->buckets = it->second.geom.B;
  out->table_bytes = it->second.table_bytes;
  return H2_OK;
}

int h2_msm_device_range(h2_curve_t curve, uint64_t handle, const void* d_scalars, size_t first_base, size_t n,
                        size_t col_stride, size_t m, void* d_out_jac, void* stream_) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  const BasesEntry* be = nullptr;
  int rc = msm_common_checks((int)curve, handle, first_base, n, m, &be);
  if (rc != H2_OK) return rc;
  if (!d_out_jac || (n && !d_scalars) || (m > 1 && col_stride < n)) return H2_EINVAL;
  Call k(stream_);
  if (k.rc != H2_OK) return k.rc;
  if (n == 0) {
    H2_TRY(hipMemsetAsync(d_out_jac, 0, m * 96, k.stream));
    return H2_OK;
  }
  return msm_device_run(*k.c, (int)curve, *be, d_scalars, first_base, n, col_stride, m, d_out_jac, false, k.stream);
}

int h2_msm_device_multi(h2_curve_t curve, const uint64_t* handles, const void* d_scalars, size_t first_base, size_t n,
                        size_t col_stride, size_t m, void* d_out_jac, void* stream_) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  if (!handles || m == 0 || m > MSM_MAX_MULTI) return H2_EINVAL;
  const BasesEntry* bes[MSM_MAX_MULTI];
  for (size_t j = 0; j < m; j++) {
    int rc = msm_common_checks((int)curve, handles[j], first_base, n, m, &bes[j]);
    if (rc != H2_OK) return rc;
  }
  if (!d_out_jac || (n && !d_scalars) || (m > 1 && col_stride < n)) return H2_EINVAL;
  Call k(stream_);
  if (k.rc != H2_OK) return k.rc;
  if (n == 0) {
    H2_TRY(hipMemsetAsync(d_out_jac, 0, m * 96, k.stream));
    return H2_OK;
  }
  return msm_device_run(*k.c, (int)curve, *bes[0], d_scalars, first_base, n, col_stride, m, d_out_jac, false, k.stream, bes);
}

int h2_msm_device(h2_curve_t curve, uint64_t handle, const void* d_scalars, size_t n, size_t m, void* d_out_jac,
                  void* stream_) {
  return h2_msm_device_range(curve, handle, d_scalars, 0, n, n, m, d_out_jac, stream_);
}

int h2_stream_wait_msm_tail(void* stream_) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  Call k(stream_);
  if (k.rc != H2_OK) return k.rc;
  if (k.c->tail_recorded && k.c->tail_wait) H2_TRY(hipStreamWaitEvent(k.stream, k.c->tail_wait, 0));
  k.c->tail_wanted = true;          // from now on every MSM of this context marks the end of its accumulate kernel
  return H2_OK;
}

int h2_points_sum_device(h2_curve_t curve, const void* d_in_jac, size_t groups, size_t count, void* d_out_jac,
                         void* stream_) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  Call k(stream_);
  if (k.rc != H2_OK) return k.rc;
  if (!curve_ok((int)curve) || !d_in_jac || !d_out_jac || groups == 0 || groups > (1u << 20) || count > (1u << 24))
    return H2_EINVAL;
  if (count == 0) return H2_OK;
  hipError_t e = ops_of((int)curve)->points_sum(d_in_jac, d_out_jac, (uint32_t)groups, (uint32_t)count, k.stream);
  if (e != hipSuccess) return dev_fail(e, "points_sum_kernel");
  return H2_OK;
}

// Host-pointer MSMs.  With several contexts (h2_init_devices) a batch is sharded by column, column j -> context
// j mod G, and a single long MSM by contiguous point range with the G partial sums added on context 0
// (SURVEY.md section 8(e)); every context works on its own stream, the host waits once at the end.
static int msm_host(h2_curve_t curve, uint64_t handle, const uint64_t* const* cols, size_t n, size_t m, uint64_t* out,
                    bool affine_out) {
  const BasesEntry* be = nullptr;
  int rc = msm_common_checks((int)curve, handle, 0, n, m, &be);
  if (rc != H2_OK) return rc;
  if (!out || !cols) return H2_EINVAL;
  const size_t out_sz = affine_out ? 64 : 96;
  if (n == 0) {
    memset(out, 0, m * out_sz);
    return H2_OK;
  }
  for (size_t j = 0; j < m; j++)
    if (!cols[j]) return H2_EINVAL;
  const size_t G = g_h2.ctx.size();
  const size_t col_bytes = n * 32;
  if (m == 1 && G > 1 && n >= 4096 * G) {
    // point-range split of one MSM: context g takes bases [lo_g, hi_g)
    std::vector<uint64_t> partial(G * 12);
    for (size_t g = 0; g < G; g++) {
      DevCtx& c = g_h2.ctx[g];
      DeviceGuard dg(c.device);
      const size_t lo = n * g / G, hi = n * (g + 1) / G, cnt = hi - lo;
      const size_t res_off = h2_align256(cnt * 32);
      rc = arena_acquire(c.stage, res_off + 96, c.stream);
      if (rc != H2_OK) return rc;
      H2_TRY(hipMemcpyAsync(c.stage.p, (const char*)cols[0] + lo * 32, cnt * 32, hipMemcpyHostToDevice, c.stream));
      void* d_res = (char*)c.stage.p + res_off;
      rc = msm_device_run(c, (int)curve, *be, c.stage.p, lo, cnt, cnt, 1, d_res, false, c.stream);
      if (rc != H2_OK) return rc;
      H2_TRY(hipMemcpyAsync(&partial[12 * g], d_res, 96, hipMemcpyDeviceToHost, c.stream));
    }
    for (size_t g = 0; g < G; g++) {
      DeviceGuard dg(g_h2.ctx[g].device);
      H2_TRY(hipStreamSynchronize(g_h2.ctx[g].stream));
    }
    // add the G partial sums on context 0 (they are 96 bytes each)
    DevCtx& c = g_h2.ctx[0];
    DeviceGuard dg(c.device);
    const size_t res_off = h2_align256(G * 96);
    rc = arena_acquire(c.stage, res_off + 96, c.stream);
    if (rc != H2_OK) return rc;
    H2_TRY(hipMemcpyAsync(c.stage.p, partial.data(), G * 96, hipMemcpyHostToDevice, c.stream));
    void* d_res = (char*)c.stage.p + res_off;
    hipError_t e = ops_of((int)curve)->points_sum(c.stage.p, d_res, (uint32_t)G, 1, c.stream);
    if (e != hipSuccess) return dev_fail(e, "points_sum_kernel");
    if (affine_out) return H2_EINVAL;   // not reached: h2_msm asks for Jacobian
    H2_TRY(hipMemcpyAsync(out, d_res, 96, hipMemcpyDeviceToHost, c.stream));
    H2_TRY(hipStreamSynchronize(c.stream));
    return H2_OK;
  }
  // column sharding: context g takes columns g, g + G, ...; the results come back as Jacobian points and are
  // normalised on the host when the caller wants affine ones
  std::vector<uint8_t> jac(affine_out ? m * 96 : 0);
  uint8_t* dst = affine_out ? jac.data() : (uint8_t*)out;
  for (size_t g = 0; g < G && g < m; g++) {
    DevCtx& c = g_h2.ctx[g];
    DeviceGuard dg(c.device);
    const size_t mine = (m - g + G - 1) / G;
    const size_t res_off = h2_align256(mine * col_bytes);
    rc = arena_acquire(c.stage, res_off + mine * 96, c.stream);
    if (rc != H2_OK) return rc;
    for (size_t i = 0; i < mine; i++)
      H2_TRY(hipMemcpyAsync((char*)c.stage.p + i * col_bytes, cols[g + i * G], col_bytes, hipMemcpyHostToDevice,
                            c.stream));
    void* d_res = (char*)c.stage.p + res_off;
    rc = msm_device_run(c, (int)curve, *be, c.stage.p, 0, n, n, mine, d_res, false, c.stream);
    if (rc != H2_OK) return rc;
    if (G == 1) {
      H2_TRY(hipMemcpyAsync(dst, d_res, mine * 96, hipMemcpyDeviceToHost, c.stream));
    } else {
      for (size_t i = 0; i < mine; i++)
        H2_TRY(hipMemcpyAsync(dst + (g + i * G) * 96, (char*)d_res + i * 96, 96, hipMemcpyDeviceToHost, c.stream));
    }
  }
  for (size_t g = 0; g < G && g < m; g++) {
    DeviceGuard dg(g_h2.ctx[g].device);
    H2_TRY(hipStreamSynchronize(g_h2.ctx[g].stream));
  }
  if (affine_out) jac_to_affine_host((int)curve, jac.data(), m, (uint8_t*)out);
  return H2_OK;
}

int h2_msm(h2_curve_t curve, uint64_t handle, const uint64_t* scalars, size_t n, uint64_t out_jac[12]) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  if (n && !scalars) return H2_EINVAL;
  const uint64_t* cols[1] = {scalars ? scalars : (const uint64_t*)out_jac};
  return msm_host(curve, handle, cols, n, 1, out_jac, false);
}

int h2_msm_batch(h2_curve_t curve, uint64_t handle, const uint64_t* const* scalars, size_t n, size_t m,
                 uint64_t* out_affine) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  return msm_host(curve, handle, scalars, n, m, out_affine, true);
}

int h2_srs_generate(h2_curve_t curve, const uint64_t s[4], size_t n, void* d_out_affine, void* stream_) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  Call k(stream_);
  if (k.rc != H2_OK) return k.rc;
  const CurveOps* ops = ops_of((int)curve);
  if (!ops || !s || !d_out_affine || n == 0 || n >= (1ull << 32)) return H2_EINVAL;
  hipError_t e = ops->srs_powers(d_out_affine, s, (uint32_t)n, k.stream);
  if (e != hipSuccess) return dev_fail(e, "srs_powers_kernel");
  return H2_OK;
}

int h2_fixed_base_mul(h2_curve_t curve, const void* d_scalars, size_t n, void* d_out_affine, void* stream_) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  Call k(stream_);
  if (k.rc != H2_OK) return k.rc;
  const CurveOps* ops = ops_of((int)curve);
  if (!ops || !d_scalars || !d_out_affine || n == 0 || n >= (1ull << 32)) return H2_EINVAL;
  hipError_t e = ops->fixed_base_mul(d_out_affine, d_scalars, (uint32_t)n, k.stream);
  if (e != hipSuccess) return dev_fail(e, "fixed_base_mul_kernel");
  return H2_OK;
}

int h2_profile_enable(int on) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  g_h2.profiling = on != 0;
  for (auto& c : g_h2.ctx) {
    c.prof_used = 0;
    c.prof_alg_bytes = 0;
  }
  return H2_OK;
}

int h2_profile_read(h2_profile_t* out) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  if (!out) return H2_EINVAL;
  double ms = 0, bytes = 0;
  uint64_t launches = 0;
  for (auto& c : g_h2.ctx) {
    DeviceGuard dg(c.device);
    for (size_t i = 0; i < c.prof_used; i++) {
      float t = 0;
      H2_TRY(hipEventSynchronize(c.prof_events[i].second));
      H2_TRY(hipEventElapsedTime(&t, c.prof_events[i].first, c.prof_events[i].second));
      ms += t;
    }
    launches += c.prof_used;
    bytes += c.prof_alg_bytes;
    c.prof_used = 0;
    c.prof_alg_bytes = 0;
  }
  out->launches = launches;
  out->kernel_ms = ms;
  out->algorithmic_bytes = bytes;
  return H2_OK;
}

int h2_ntt_device(h2_curve_t curve, void* d_a, size_t m, const uint64_t omega[4], uint32_t log_n, void* stream_) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  Call k(stream_);
  if (k.rc != H2_OK) return k.rc;
  if (!curve_ok((int)curve) || !d_a || !omega || m == 0 || log_n > 30) return H2_EINVAL;
  if (log_n == 0) return H2_OK;
  return ntt_enqueue(*k.c, (int)curve, d_a, m, omega, log_n, k.stream);
}

int h2_ntt_scaled_device(h2_curve_t curve, void* d_a, size_t m, const uint64_t omega[4], uint32_t log_n,
                         const uint64_t scale[4], void* stream_) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  Call k(stream_);
  if (k.rc != H2_OK) return k.rc;
  if (!curve_ok((int)curve) || !d_a || !omega || !scale || m == 0 || log_n > 30) return H2_EINVAL;
  if (log_n == 0) {
    hipError_t e = ops_of((int)curve)->poly_scale(d_a, m, scale, k.stream);
    if (e != hipSuccess) return dev_fail(e, "poly_scale_kernel");
    return H2_OK;
  }
  return ntt_enqueue(*k.c, (int)curve, d_a, m, omega, log_n, k.stream, scale);
}

int h2_poly_scale_device(h2_curve_t curve, void* d_a, size_t n, size_t m, const uint64_t c[4], void* stream_) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  Call k(stream_);
  if (k.rc != H2_OK) return k.rc;
  if (!curve_ok((int)curve) || !d_a || !c) return H2_EINVAL;
  if (n * m == 0) return H2_OK;
  hipError_t e = ops_of((int)curve)->poly_scale(d_a, n * m, c, k.stream);
  if (e != hipSuccess) return dev_fail(e, "poly_scale_kernel");
  return H2_OK;
}

int h2_poly_coset_device(h2_curve_t curve, void* d_a, size_t n, size_t m, const uint64_t g[4], void* stream_) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  Call k(stream_);
  if (k.rc != H2_OK) return k.rc;
  if (!curve_ok((int)curve) || !d_a || !g) return H2_EINVAL;
  if (n * m == 0) return H2_OK;
  hipError_t e = ops_of((int)curve)->poly_powers(d_a, n, m, g, k.stream);
  if (e != hipSuccess) return dev_fail(e, "poly_powers_kernel");
  return H2_OK;
}

int h2_poly_mul_periodic_device(h2_curve_t curve, void* d_a, size_t n, size_t m, const void* d_t, size_t period,
                                void* stream_) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  Call k(stream_);
  if (k.rc != H2_OK) return k.rc;
  if (!curve_ok((int)curve) || !d_a || !d_t || period == 0 || (period & (period - 1))) return H2_EINVAL;
  if (n * m == 0) return H2_OK;
  hipError_t e = ops_of((int)curve)->poly_mul_periodic(d_a, n * m, d_t, period, k.stream);
  if (e != hipSuccess) return dev_fail(e, "poly_mul_periodic_kernel");
  return H2_OK;
}

int h2_poly_inverse_device(h2_curve_t curve, void* d_a, size_t n, void* stream_) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  Call k(stream_);
  if (k.rc != H2_OK) return k.rc;
  if (!curve_ok((int)curve) || !d_a) return H2_EINVAL;
  if (n == 0) return H2_OK;
  hipError_t e = ops_of((int)curve)->poly_inverse(d_a, n, k.stream);
  if (e != hipSuccess) return dev_fail(e, "poly_inverse_kernel");
  return H2_OK;
}

int h2_poly_divide_linear_device(h2_curve_t curve, const void* d_a, size_t n, const uint64_t z[4], void* d_q,
                                 void* stream_) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  Call k(stream_);
  if (k.rc != H2_OK) return k.rc;
  if (!curve_ok((int)curve) || !d_a || !d_q || !z || d_a == d_q) return H2_EINVAL;
  if (n == 0) return H2_OK;
  int rc = arena_acquire(k.c->div_ws, (size_t)2 * DIV_MAX_CHUNKS * 32, k.stream);
  if (rc != H2_OK) return rc;
  hipError_t e = ops_of((int)curve)->poly_divide_linear(d_a, n, z, d_q, k.c->div_ws.p, k.stream);
  if (e != hipSuccess) return dev_fail(e, "poly_divide kernels");
  return arena_release(k.c->div_ws, k.stream);
}

int h2_poly_prefix_product_device(h2_curve_t curve, const void* d_a, size_t n, void* d_out, void* stream_) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  Call k(stream_);
  if (k.rc != H2_OK) return k.rc;
  if (!curve_ok((int)curve) || !d_a || !d_out) return H2_EINVAL;
  if (n == 0) return H2_OK;
  int rc = arena_acquire(k.c->div_ws, (size_t)2 * DIV_MAX_CHUNKS * 32, k.stream);
  if (rc != H2_OK) return rc;
  hipError_t e = ops_of((int)curve)->poly_prefix_product(d_a, n, d_out, k.c->div_ws.p, k.stream);
  if (e != hipSuccess) return dev_fail(e, "poly_prefix kernels");
  return arena_release(k.c->div_ws, k.stream);
}

int h2_chacha20_scalars_device(h2_curve_t curve, const uint8_t seed[32], uint64_t first_block, size_t n, void* d_out,
                               void* stream_) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  Call k(stream_);
  if (k.rc != H2_OK) return k.rc;
  if (!curve_ok((int)curve) || !seed || !d_out) return H2_EINVAL;
  if (n == 0) return H2_OK;
  uint32_t key[8];
  for (int i = 0; i < 8; i++)
    key[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) | ((uint32_t)seed[4 * i + 2] << 16) |
             ((uint32_t)seed[4 * i + 3] << 24);
  hipError_t e = ops_of((int)curve)->chacha20_scalars(d_out, n, first_block, key, k.stream);
  if (e != hipSuccess) return dev_fail(e, "chacha20_scalars_kernel");
  return H2_OK;
}

int h2_poly_pointwise_device(h2_curve_t curve, int op, void* d_a, const void* d_b, size_t n, void* stream_) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  Call k(stream_);
  if (k.rc != H2_OK) return k.rc;
  if (!curve_ok((int)curve) || !d_a || !d_b || op < 0 || op > 2) return H2_EINVAL;
  if (n == 0) return H2_OK;
  hipError_t e = ops_of((int)curve)->poly_pointwise(d_a, d_b, n, op, k.stream);
  if (e != hipSuccess) return dev_fail(e, "poly_pointwise_kernel");
  return H2_OK;
}

// host columns; with several contexts column j is transformed by context j mod G (a single NTT is not split:
// "replicas only")
int h2_ntt_batch(h2_curve_t curve, uint64_t* const* cols, size_t m, const uint64_t omega[4], uint32_t log_n) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  if (!g_h2.ready) return H2_ENOTINIT;
  if (!curve_ok((int)curve) || !cols || !omega || m == 0 || log_n > 30) return H2_EINVAL;
  for (size_t j = 0; j < m; j++)
    if (!cols[j]) return H2_EINVAL;
  if (log_n == 0) return H2_OK;
  const size_t col_bytes = (size_t)32 << log_n;
  const size_t G = g_h2.ctx.size();
  for (size_t g = 0; g < G && g < m; g++) {
    DevCtx& c = g_h2.ctx[g];
    DeviceGuard dg(c.device);
    const size_t mine = (m - g + G - 1) / G;
    int rc = arena_acquire(c.stage, mine * col_bytes, c.stream);
    if (rc != H2_OK) return rc;
    for (size_t i = 0; i < mine; i++)
      H2_TRY(hipMemcpyAsync((char*)c.stage.p + i * col_bytes, cols[g + i * G], col_bytes, hipMemcpyHostToDevice,
                            c.stream));
    rc = ntt_enqueue(c, (int)curve, c.stage.p, mine, omega, log_n, c.stream);
    if (rc != H2_OK) return rc;
    for (size_t i = 0; i < mine; i++)
      H2_TRY(hipMemcpyAsync(cols[g + i * G], (char*)c.stage.p + i * col_bytes, col_bytes, hipMemcpyDeviceToHost,
                            c.stream));
  }
  for (size_t g = 0; g < G && g < m; g++) {
    DeviceGuard dg(g_h2.ctx[g].device);
    H2_TRY(hipStreamSynchronize(g_h2.ctx[g].stream));
  }
  return H2_OK;
}

// best_fft over group elements (FftGroup for C::Curve): n = 2^log_n Jacobian points in place, natural order, unscaled
int h2_fft_group_device(h2_curve_t curve, void* d_points_jac, const uint64_t omega[4], uint32_t log_n, void* stream_) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  Call k(stream_);
  if (k.rc != H2_OK) return k.rc;
  if (!curve_ok((int)curve) || !d_points_jac || !omega || log_n > 26) return H2_EINVAL;
  if (log_n == 0) return H2_OK;
  const CurveOps* ops = ops_of((int)curve);
  Arena& A = k.c->msm_ws.of(k.stream);
  A.clean_bytes = 0;
  int rc = arena_acquire(A, ops->group_fft_scratch(log_n), k.stream);      // the MSM workspace, idle here
  if (rc != H2_OK) return rc;
  hipError_t e = ops->group_fft(d_points_jac, d_points_jac, A.p, omega, log_n, k.stream);
  if (e != hipSuccess) return dev_fail(e, "group fft kernels");
  return arena_release(A, k.stream);
}

int h2_fft_group(h2_curve_t curve, uint64_t* points_jac, const uint64_t omega[4], uint32_t log_n) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  if (!g_h2.ready) return H2_ENOTINIT;
  if (!curve_ok((int)curve) || !points_jac || !omega || log_n > 26) return H2_EINVAL;
  if (log_n == 0) return H2_OK;
  DevCtx& c = g_h2.ctx[0];
  DeviceGuard dg(c.device);
  const size_t bytes = ((size_t)96) << log_n;
  int rc = arena_acquire(c.stage, bytes, c.stream);
  if (rc != H2_OK) return rc;
  H2_TRY(hipMemcpyAsync(c.stage.p, points_jac, bytes, hipMemcpyHostToDevice, c.stream));
  rc = h2_fft_group_device(curve, c.stage.p, omega, log_n, c.stream);
  if (rc != H2_OK) return rc;
  H2_TRY(hipMemcpyAsync(points_jac, c.stage.p, bytes, hipMemcpyDeviceToHost, c.stream));
  H2_TRY(hipStreamSynchronize(c.stream));
  return arena_release(c.stage, c.stream);
}

int h2_ntt(h2_curve_t curve, uint64_t* a, const uint64_t omega[4], uint32_t log_n) {
  uint64_t* cols[1] = {a};
  return h2_ntt_batch(curve, cols, 1, omega, log_n);
}

}  // extern "C"

// ---- host self-test hooks (include/h2hip_selftest.h) ------------------------------------------
#include "../../include/h2hip_selftest.h"
extern "C" int h2_selftest_field_op(int field, int op, const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {
  if (!a || !b || !out) return H2_EINVAL;
  int rc = -1;
  switch (field) {
    case 0: rc = curve_ops_bn254()->selftest_field(0, op, a, b, out); break;   // bn254 Fq
    case 1: rc = curve_ops_bn254()->selftest_field(1, op, a, b, out); break;   // bn254 Fr
    case 2: rc = curve_ops_pallas()->selftest_field(0, op, a, b, out); break;  // pasta Fp
    case 3: rc = curve_ops_pallas()->selftest_field(1, op, a, b, out); break;  // pasta Fq
  }
  return rc == 0 ? H2_OK : H2_EINVAL;
}
extern "C" int h2_selftest_curve_op(int curve, int op, const uint64_t p[8], const uint64_t q[8], uint64_t out[8]) {
  const CurveOps* ops = ops_of(curve);
  if (!ops || !p || !q || !out) return H2_EINVAL;
  return ops->selftest_curve(op, p, q, out) == 0 ? H2_OK : H2_EINVAL;
}
extern "C" int h2_selftest_digits(int curve, const uint64_t scalar[4], size_t n_for_geometry, uint32_t* out,
                                  uint32_t cap) {
  const CurveOps* ops = ops_of(curve);
  if (!ops || !scalar || !out) return H2_EINVAL;
  return ops->selftest_digits(scalar, n_for_geometry, out, cap);
}
// n element pairs through the DEVICE instantiation (one kernel launch); host pointers in and out
extern "C" int h2_selftest_curve_op_device(int curve, int op, const uint64_t* p, const uint64_t* q, uint64_t* out,
                                           size_t n) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  if (!g_h2.ready) return H2_ENOTINIT;
  const CurveOps* ops = ops_of(curve);
  if (!ops || !p || !q || !out || n == 0 || n > (1u << 20)) return H2_EINVAL;
  DevCtx& g_ctx = g_h2.ctx[0];
  DeviceGuard dg(g_ctx.device);
  int rc = arena_acquire(g_ctx.stage, 3 * n * 64, g_ctx.stream);
  if (rc != H2_OK) return rc;
  char* d = (char*)g_ctx.stage.p;
  H2_TRY(hipMemcpyAsync(d, p, n * 64, hipMemcpyHostToDevice, g_ctx.stream));
  H2_TRY(hipMemcpyAsync(d + n * 64, q, n * 64, hipMemcpyHostToDevice, g_ctx.stream));
  hipError_t e = ops->selftest_curve_device(op, d, d + n * 64, d + 2 * n * 64, (uint32_t)n, g_ctx.stream);
  if (e != hipSuccess) return dev_fail(e, "selftest_curve_kernel");
  H2_TRY(hipMemcpyAsync(out, d + 2 * n * 64, n * 64, hipMemcpyDeviceToHost, g_ctx.stream));
  H2_TRY(hipStreamSynchronize(g_ctx.stream));
  return H2_OK;
}

extern "C" int h2_selftest_field_op_device(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out,
                                           size_t n) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  if (!g_h2.ready) return H2_ENOTINIT;
  if (!a || !b || !out || n == 0 || n > (1u << 24) || field < 0 || field > 3) return H2_EINVAL;
  const CurveOps* ops = field < 2 ? curve_ops_bn254() : curve_ops_pallas();
  const int which = field & 1;
  DevCtx& g_ctx = g_h2.ctx[0];
  DeviceGuard dg(g_ctx.device);
  int rc = arena_acquire(g_ctx.stage, 3 * n * 32, g_ctx.stream);
  if (rc != H2_OK) return rc;
  char* d = (char*)g_ctx.stage.p;
  H2_TRY(hipMemcpyAsync(d, a, n * 32, hipMemcpyHostToDevice, g_ctx.stream));
  H2_TRY(hipMemcpyAsync(d + n * 32, b, n * 32, hipMemcpyHostToDevice, g_ctx.stream));
  hipError_t e = ops->selftest_field_device(which, op, d, d + n * 32, d + 2 * n * 32, (uint32_t)n, g_ctx.stream);
  if (e != hipSuccess) return dev_fail(e, "selftest_field_kernel");
  H2_TRY(hipMemcpyAsync(out, d + 2 * n * 32, n * 32, hipMemcpyDeviceToHost, g_ctx.stream));
  H2_TRY(hipStreamSynchronize(g_ctx.stream));
  return H2_OK;
}

// test hooks around the MSM workspace (include/h2hip_selftest.h)
extern "C" int h2_selftest_msm_guard(int on) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  h2::g_msm_guard = on != 0;
  h2::g_msm_guard_poke = on == 2;
  h2::g_sort2_pack = on != 3;            // guard(3): the two-level sort keeps the low key bits in the side array
  h2::g_guard_launches = h2::g_guard_violations = 0;
  h2::g_guard_first.clear();
  return H2_OK;
}
extern "C" int h2_selftest_msm_guard_report(uint64_t out[2], char* first, size_t cap) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  if (!out) return H2_EINVAL;
  out[0] = h2::g_guard_launches;
  out[1] = h2::g_guard_violations;
  if (first && cap) {
    snprintf(first, cap, "%s", h2::g_guard_first.c_str());
  }
  return H2_OK;
}
// host only: lay out the workspace of an (n_bases, n, m, col_stride) launch as msm_device_run would and run the bounds
// proof on it; out[0..7] = window bits, windows, buckets, tile, staged, two-level sort, entries per thread, regions.
// Returns H2_OK, or H2_EINVAL with the violated condition in h2_last_device_error().
extern "C" int h2_selftest_msm_check(int curve, size_t n_bases, size_t n, size_t m, size_t col_stride, int guard, uint64_t out[8]) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  const CurveOps* ops = ops_of(curve);
  if (!ops || n == 0 || m == 0 || n > n_bases) return H2_EINVAL;
  const MsmGeom g = msm_geometry(n_bases, ops->scalar_bits);
  // as msm_device_run: a batch wider than one launch takes runs in column groups; the first (widest) group is checked
  const size_t group = msm_cols_per_launch(g, n);
  if (group == 0) return H2_EINVAL;
  if (m > group) {
    if (col_stride < n) { g_h2.last_error = "msm launch geometry: col_stride >= n"; return H2_EINVAL; }
    m = group;
  }
  const MsmWorkspace ws = msm_workspace(n, m, g, guard ? 256u : 0u, n_bases);
  if (out) {
    out[0] = g.c; out[1] = g.W; out[2] = g.B; out[3] = ws.sort2 ? ws.s2.tile : ws.tile;
    out[4] = ws.staged; out[5] = ws.sort2; out[6] = ws.T; out[7] = ws.n_regions;
  }
  if (const char* broken = msm_check(ws, g, n, m, col_stride, (uint32_t)n_bases, ws.total)) {
    g_h2.last_error = std::string("msm launch geometry: ") + broken;
    return H2_EINVAL;
  }
  return H2_OK;
}
// scratch arenas of the current context: out = {allocations (first use or growth), cross-stream hand-overs (event waits),
// MSM slots taken over, NTT slots taken over}
extern "C" int h2_selftest_arena_stats(uint64_t out[4]) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  DevCtx* c = g_h2.ready ? ctx_current() : nullptr;
  if (!c || !out) return H2_EINVAL;
  out[0] = g_arena_growths;
  out[1] = g_arena_waits;
  out[2] = c->msm_ws.takeovers;
  out[3] = c->ntt_ws.takeovers;
  return H2_OK;
}
// host only: the sort's block -> (column, tile) mapping for `tiles` tiles per column and m columns: every block of
// the grid is either dead or maps to a (column < m, tile < tiles) pair that no other block takes, and all pairs are taken
extern "C" int h2_selftest_msm_tiles(uint32_t tiles, uint32_t m) {
  if (tiles == 0 || m == 0 || (uint64_t)tiles * m > (1u << 24)) return H2_EINVAL;
  const uint32_t grid = msm_tile_grid(tiles, m);
  std::vector<uint8_t> seen((size_t)tiles * m, 0);
  size_t live = 0;
  for (uint32_t b = 0; b < grid; b++) {
    const MsmTileId t = msm_tile_id_of(b, tiles, m);
    if (!t.live) continue;
    if (t.col >= m || t.tile >= tiles || t.group >= MSM_XCDS || seen[(size_t)t.col * tiles + t.tile]) return H2_EINVAL;
    seen[(size_t)t.col * tiles + t.tile] = 1;
    live++;
  }
  return live == (size_t)tiles * m ? H2_OK : H2_EINVAL;
}

// test hook: lower the sort's entry limit so that the grouped-columns path is reached at small sizes (0 = default)
extern "C" int h2_selftest_set_msm_max_entries(uint64_t limit) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  h2::g_msm_max_entries = (limit > 0 && limit < (1ull << 31) - 1) ? limit : (1ull << 31) - 1;
  return H2_OK;
}

// measured integer ceiling: dependent 9 x 29-bit Montgomery products of `curve`'s base field at `waves_per_simd`
// resident waves per SIMD on every CU of the current device
extern "C" int h2_selftest_modmul_rate(int curve, int waves_per_simd, int iters, double* modmul_per_s) {
  std::lock_guard<std::recursive_mutex> lk(g_h2_mu);
  if (!g_h2.ready) return H2_ENOTINIT;
  const CurveOps* ops = ops_of(curve);
  if (!ops || !modmul_per_s || waves_per_simd < 1 || waves_per_simd > 8 || iters < 1 || iters > (1 << 20)) return H2_EINVAL;
  DevCtx* c = ctx_current();
  if (!c) return H2_EINVAL;
  hipDeviceProp_t prop;
  H2_TRY(hipGetDeviceProperties(&prop, c->device));
  hipError_t e = ops->modmul_rate(prop.multiProcessorCount * waves_per_simd, iters, c->stream, modmul_per_s);
  if (e != hipSuccess) return dev_fail(e, "modmul_rate_kernel");
  return H2_OK;
}
