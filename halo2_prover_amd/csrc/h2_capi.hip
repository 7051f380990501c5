// h2_capi.hip -- implementation of include/h2hip.h (the drop-in C ABI).
//
// One context per process, bound to one GPU.  Every entry point validates its arguments,
// takes the context mutex, enqueues on the context stream (or the caller's) and reports
// errors as h2_status_t values: nothing throws or aborts across the ABI
// (the reference's panics -- /root/reference/circuits/src/utils.rs:91,120 `.expect(..)`,
// best_multiexp's assert_eq!(coeffs.len(), bases.len()) -- become H2_EINVAL here).
#include "../../include/h2hip.h"

#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "h2_curve_ops.hpp"
#include "h2_msm.hpp"
#include "h2_ntt.hpp"
#include "h2_poly.hpp"

using namespace h2;

namespace {

struct BasesEntry {
  int curve;
  size_t n;
  MsmGeom geom;
  void* table;  // W * n affine points
  size_t table_bytes;
};

struct TwiddleEntry {
  int field;
  uint32_t log_n;
  uint64_t omega[4];
  void* tw;
  uint64_t stamp;
};

struct Context {
  bool ready = false;
  int device = -1;
  hipStream_t stream = nullptr;
  void* ws = nullptr;       // workspace arena (MSM scratch / NTT ping-pong)
  size_t ws_bytes = 0;
  void* stage = nullptr;    // device staging for host-pointer entry points
  size_t stage_bytes = 0;
  void* div_ws = nullptr;   // chunk values of h2_poly_divide_linear_device (2 * 1024 elements)
  std::map<uint64_t, BasesEntry> bases;
  uint64_t next_handle = 1;
  std::vector<TwiddleEntry> twiddles;
  uint64_t stamp = 0;
  // kernel timing for the roofline (h2_profile_*): event pairs around the bucket-accumulate kernel
  bool profiling = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  size_t prof_used = 0;
  double prof_alg_bytes = 0;
  std::string last_error;
};

Context g_ctx;
std::mutex g_mu;

int dev_fail(hipError_t e, const char* where) {
  char buf[256];
  snprintf(buf, sizeof buf, "%s: %s", where, hipGetErrorString(e));
  g_ctx.last_error = buf;
  return H2_EDEVICE;
}
#define H2_TRY(call)                                  \
  do {                                                \
    hipError_t _e = (call);                           \
    if (_e != hipSuccess) return dev_fail(_e, #call); \
  } while (0)

int ensure_arena(void** p, size_t* have, size_t want) {
  if (*have >= want) return H2_OK;
  if (*p) {
    // rare: a larger call than any before.  Work enqueued on caller streams may still use the arena.
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return dev_fail(e, "hipDeviceSynchronize");
    (void)hipFree(*p);
    *p = nullptr;
    *have = 0;
  }
  size_t sz = want + (want >> 3);  // head-room so slightly larger calls do not reallocate
  hipError_t e = hipMalloc(p, sz);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    g_ctx.last_error = std::string("hipMalloc: ") + hipGetErrorString(e);
    return H2_ENOMEM;
  }
  *have = sz;
  return H2_OK;
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

// ---- bases ---------------------------------------------------------------------------------
int register_device(int curve, const void* d_affine, size_t n, uint64_t* handle_out) {
  if (!g_ctx.ready) return H2_ENOTINIT;
  if (!curve_ok(curve) || !d_affine || !handle_out || n == 0) return H2_EINVAL;
  const CurveOps* ops = ops_of(curve);
  MsmGeom g = msm_geometry(n, ops->scalar_bits);
  if ((uint64_t)g.W * n >= (1ull << 31)) return H2_EINVAL;
  BasesEntry be{};
  be.curve = curve;
  be.n = n;
  be.geom = g;
  be.table_bytes = (size_t)g.W * n * 64;
  hipError_t e = hipMalloc(&be.table, be.table_bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    g_ctx.last_error = std::string("hipMalloc(table): ") + hipGetErrorString(e);
    return H2_ENOMEM;
  }
  e = ops->table_build(d_affine, be.table, (uint32_t)n, g, g_ctx.stream);
  if (e == hipSuccess) e = hipStreamSynchronize(g_ctx.stream);
  if (e != hipSuccess) {
    (void)hipFree(be.table);
    return dev_fail(e, "msm_table_kernel");
  }
  uint64_t h = g_ctx.next_handle++;
  g_ctx.bases[h] = be;
  *handle_out = h;
  return H2_OK;
}

// ---- MSM -----------------------------------------------------------------------------------
// enqueue; the m XYZZ results land at ws + off_tree2
int msm_enqueue(int curve, const BasesEntry& be, const void* d_scalars, size_t n, size_t m, hipStream_t stream,
                MsmWorkspace* ws_out) {
  MsmWorkspace ws = msm_workspace(n, m, be.geom);
  if (ws.E >= (1ull << 31) || ws.K >= (1ull << 31)) return H2_EINVAL;
  int rc = ensure_arena(&g_ctx.ws, &g_ctx.ws_bytes, ws.total);
  if (rc != H2_OK) return rc;
  *ws_out = ws;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (g_ctx.profiling) {
    if (g_ctx.prof_used == g_ctx.prof_events.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) g_ctx.prof_events.push_back({a, b});
    }
    if (g_ctx.prof_used < g_ctx.prof_events.size()) {
      ev0 = g_ctx.prof_events[g_ctx.prof_used].first;
      ev1 = g_ctx.prof_events[g_ctx.prof_used].second;
      g_ctx.prof_used++;
      g_ctx.prof_alg_bytes += (double)m * (double)n * 96.0 + (double)m * 96.0;  // SURVEY.md 8(d) bytes_msm
    }
  }
  hipError_t e = ops_of(curve)->msm_launch(be.table, (uint32_t)be.n, d_scalars, n, m, be.geom, (char*)g_ctx.ws, ws,
                                          stream, ev0, ev1);
  if (e != hipSuccess) return dev_fail(e, "msm_launch");
  return H2_OK;
}

int msm_common_checks(int curve, uint64_t handle, size_t n, size_t m, const BasesEntry** be) {
  if (!g_ctx.ready) return H2_ENOTINIT;
  if (!curve_ok(curve) || m == 0) return H2_EINVAL;
  auto it = g_ctx.bases.find(handle);
  if (it == g_ctx.bases.end()) return H2_EHANDLE;
  if (it->second.curve != curve) return H2_EINVAL;
  if (n > it->second.n) return H2_EINVAL;  // best_multiexp: assert_eq!(coeffs.len(), bases.len())
  *be = &it->second;
  return H2_OK;
}

// ---- NTT -----------------------------------------------------------------------------------
int get_twiddles(const CurveOps* ops, const uint64_t omega[4], uint32_t log_n, hipStream_t stream,
                 const void** out) {
  for (auto& t : g_ctx.twiddles) {
    if (t.field == ops->scalar_field_id && t.log_n == log_n && memcmp(t.omega, omega, 32) == 0) {
      t.stamp = ++g_ctx.stamp;
      *out = t.tw;
      return H2_OK;
    }
  }
  if (g_ctx.twiddles.size() >= 16) {  // evict the least recently used table
    size_t victim = 0;
    for (size_t i = 1; i < g_ctx.twiddles.size(); i++)
      if (g_ctx.twiddles[i].stamp < g_ctx.twiddles[victim].stamp) victim = i;
    H2_TRY(hipDeviceSynchronize());
    (void)hipFree(g_ctx.twiddles[victim].tw);
    g_ctx.twiddles.erase(g_ctx.twiddles.begin() + victim);
  }
  TwiddleEntry te{};
  te.field = ops->scalar_field_id;
  te.log_n = log_n;
  memcpy(te.omega, omega, 32);
  const size_t bytes = (((size_t)1 << log_n) / 2) * 32;
  hipError_t e = hipMalloc(&te.tw, bytes < 64 ? 64 : bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return H2_ENOMEM;
  }
  e = ops->ntt_twiddles(te.tw, omega, log_n, stream);
  if (e != hipSuccess) {
    (void)hipFree(te.tw);
    return dev_fail(e, "ntt_build_twiddles");
  }
  te.stamp = ++g_ctx.stamp;
  g_ctx.twiddles.push_back(te);
  *out = te.tw;
  return H2_OK;
}

int ntt_enqueue(int curve, void* d_a, size_t m, const uint64_t omega[4], uint32_t log_n, hipStream_t stream,
                const uint64_t* scale = nullptr) {
  const CurveOps* ops = ops_of(curve);
  if (!ops) return H2_EINVAL;
  const void* tw = nullptr;
  int rc = get_twiddles(ops, omega, log_n, stream, &tw);
  if (rc != H2_OK) return rc;
  NttPlan pl = ntt_make_plan(log_n);
  void* scratch = nullptr;
  if (pl.npass > 1) {
    rc = ensure_arena(&g_ctx.ws, &g_ctx.ws_bytes, m * ((size_t)32 << log_n));
    if (rc != H2_OK) return rc;
    scratch = g_ctx.ws;
  }
  hipError_t e = ops->ntt_launch(d_a, scratch, tw, log_n, m, stream, scale);
  if (e != hipSuccess) return dev_fail(e, "ntt_launch");
  return H2_OK;
}

}  // namespace

extern "C" {

int h2_version(void) { return 1000; }

const char* h2_strerror(int s) {
  switch (s) {
    case H2_OK: return "ok";
    case H2_EINVAL: return "invalid argument (length / log_n mismatch, null pointer or unknown curve)";
    case H2_ENOMEM: return "out of memory";
    case H2_EDEVICE: return "HIP device error";
    case H2_EHANDLE: return "unknown bases handle";
    case H2_ENOTINIT: return "h2_init has not been called";
  }
  return "unknown status";
}

const char* h2_last_device_error(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  static thread_local std::string copy;
  copy = g_ctx.last_error;
  return copy.c_str();
}

int h2_init(int device) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_ctx.ready) return g_ctx.device == device ? H2_OK : H2_EINVAL;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) {
    (void)hipGetLastError();
    g_ctx.last_error = "no HIP device available";
    return H2_EDEVICE;
  }
  if (device < 0 || device >= count) return H2_EINVAL;
  H2_TRY(hipSetDevice(device));
  // a BLOCKING stream on purpose: callers that pass stream = NULL (e.g. PyTorch's legacy default stream)
  // get work that is ordered against the null stream, so their own copies / kernels see finished results
  H2_TRY(hipStreamCreateWithFlags(&g_ctx.stream, hipStreamDefault));
  g_ctx.device = device;
  g_ctx.ready = true;
  return H2_OK;
}

int h2_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx.ready) return H2_OK;
  (void)hipStreamSynchronize(g_ctx.stream);
  for (auto& kv : g_ctx.bases) (void)hipFree(kv.second.table);
  g_ctx.bases.clear();
  for (auto& t : g_ctx.twiddles) (void)hipFree(t.tw);
  g_ctx.twiddles.clear();
  if (g_ctx.ws) (void)hipFree(g_ctx.ws);
  if (g_ctx.stage) (void)hipFree(g_ctx.stage);
  if (g_ctx.div_ws) (void)hipFree(g_ctx.div_ws);
  g_ctx.ws = g_ctx.stage = g_ctx.div_ws = nullptr;
  g_ctx.ws_bytes = g_ctx.stage_bytes = 0;
  (void)hipStreamDestroy(g_ctx.stream);
  g_ctx.stream = nullptr;
  g_ctx.ready = false;
  g_ctx.device = -1;
  return H2_OK;
}

int h2_bases_register_device(h2_curve_t curve, const void* d_affine, size_t n, uint64_t* handle_out) {
  std::lock_guard<std::mutex> lk(g_mu);
  return register_device((int)curve, d_affine, n, handle_out);
}

int h2_bases_register(h2_curve_t curve, const uint64_t* affine, size_t n, uint64_t* handle_out) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx.ready) return H2_ENOTINIT;
  if (!curve_ok((int)curve) || !affine || !handle_out || n == 0) return H2_EINVAL;
  int rc = ensure_arena(&g_ctx.stage, &g_ctx.stage_bytes, n * 64);
  if (rc != H2_OK) return rc;
  H2_TRY(hipMemcpyAsync(g_ctx.stage, affine, n * 64, hipMemcpyHostToDevice, g_ctx.stream));
  return register_device((int)curve, g_ctx.stage, n, handle_out);
}

int h2_bases_release(uint64_t handle) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx.ready) return H2_ENOTINIT;
  auto it = g_ctx.bases.find(handle);
  if (it == g_ctx.bases.end()) return H2_EHANDLE;
  (void)hipStreamSynchronize(g_ctx.stream);
  (void)hipFree(it->second.table);
  g_ctx.bases.erase(it);
  return H2_OK;
}

int64_t h2_bases_len(uint64_t handle) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx.ready) return H2_ENOTINIT;
  auto it = g_ctx.bases.find(handle);
  if (it == g_ctx.bases.end()) return H2_EHANDLE;
  return (int64_t)it->second.n;
}

int h2_msm_plan(uint64_t handle, h2_msm_plan_t* out) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx.ready) return H2_ENOTINIT;
  if (!out) return H2_EINVAL;
  auto it = g_ctx.bases.find(handle);
  if (it == g_ctx.bases.end()) return H2_EHANDLE;
  out->window_bits = it->second.geom.c;
  out->windows = it->second.geom.W;
  out->buckets = it->second.geom.B;
  out->table_bytes = it->second.table_bytes;
  return H2_OK;
}

// Columns per launch: the sort indexes its m * W * n entries with 32 bits, so a wide batch of long columns
// (2^24 rows x 8 columns) goes through in groups of columns, one after the other on the same stream and workspace.
// H2_MSM_MAX_ENTRIES lowers the limit (tests use it to reach the grouped path at small sizes).
static size_t msm_cols_per_launch(const BasesEntry& be, size_t n) {
  uint64_t limit = (1ull << 31) - 1;
  if (const char* ov = getenv("H2_MSM_MAX_ENTRIES")) {
    const uint64_t v = strtoull(ov, nullptr, 10);
    if (v > 0 && v < limit) limit = v;
  }
  const uint64_t per_col = (uint64_t)be.geom.W * n;
  uint64_t by_entries = limit / per_col;
  const uint64_t by_keys = ((1ull << 31) - 1) / be.geom.B;
  return (size_t)(by_entries < by_keys ? by_entries : by_keys);   // 0: a single column is already too long
}

// m columns at d_scalars (column stride n) -> m results at d_out (96-byte Jacobian or 64-byte affine), enqueued
static int msm_device_run(int curve, const BasesEntry& be, const void* d_scalars, size_t n, size_t m, void* d_out,
                          bool affine_out, hipStream_t stream) {
  const size_t group = msm_cols_per_launch(be, n);
  if (group == 0) return H2_EINVAL;
  const size_t out_sz = affine_out ? 64 : 96;
  const CurveOps* ops = ops_of(curve);
  for (size_t j0 = 0; j0 < m; j0 += group) {
    const size_t mm = m - j0 < group ? m - j0 : group;
    MsmWorkspace ws;
    int rc = msm_enqueue(curve, be, (const char*)d_scalars + j0 * n * 32, n, mm, stream, &ws);
    if (rc != H2_OK) return rc;
    const void* src = (char*)g_ctx.ws + ws.off_tree2;
    void* dst = (char*)d_out + j0 * out_sz;
    hipError_t e = affine_out ? ops->to_affine(src, dst, (uint32_t)mm, stream)
                              : ops->to_jacobian(src, dst, (uint32_t)mm, stream);
    if (e != hipSuccess) return dev_fail(e, "msm finish kernel");
  }
  return H2_OK;
}

int h2_msm_device(h2_curve_t curve, uint64_t handle, const void* d_scalars, size_t n, size_t m, void* d_out_jac,
                  void* stream_) {
  std::lock_guard<std::mutex> lk(g_mu);
  const BasesEntry* be = nullptr;
  int rc = msm_common_checks((int)curve, handle, n, m, &be);
  if (rc != H2_OK) return rc;
  if (!d_out_jac || (n && !d_scalars)) return H2_EINVAL;
  hipStream_t stream = stream_ ? (hipStream_t)stream_ : g_ctx.stream;
  if (n == 0) {
    H2_TRY(hipMemsetAsync(d_out_jac, 0, m * 96, stream));
    return H2_OK;
  }
  return msm_device_run((int)curve, *be, d_scalars, n, m, d_out_jac, false, stream);
}

static int msm_host(h2_curve_t curve, uint64_t handle, const uint64_t* const* cols, size_t n, size_t m, uint64_t* out,
                    bool affine_out) {
  const BasesEntry* be = nullptr;
  int rc = msm_common_checks((int)curve, handle, n, m, &be);
  if (rc != H2_OK) return rc;
  if (!out || !cols) return H2_EINVAL;
  const size_t out_sz = affine_out ? 64 : 96;
  if (n == 0) {
    memset(out, 0, m * out_sz);
    return H2_OK;
  }
  for (size_t j = 0; j < m; j++)
    if (!cols[j]) return H2_EINVAL;
  const size_t col_bytes = n * 32;
  const size_t res_off = h2_align256(m * col_bytes);
  rc = ensure_arena(&g_ctx.stage, &g_ctx.stage_bytes, res_off + m * 96);
  if (rc != H2_OK) return rc;
  for (size_t j = 0; j < m; j++)
    H2_TRY(hipMemcpyAsync((char*)g_ctx.stage + j * col_bytes, cols[j], col_bytes, hipMemcpyHostToDevice,
                          g_ctx.stream));
  void* d_res = (char*)g_ctx.stage + res_off;
  rc = msm_device_run((int)curve, *be, g_ctx.stage, n, m, d_res, affine_out, g_ctx.stream);
  if (rc != H2_OK) return rc;
  H2_TRY(hipMemcpyAsync(out, d_res, m * out_sz, hipMemcpyDeviceToHost, g_ctx.stream));
  H2_TRY(hipStreamSynchronize(g_ctx.stream));
  return H2_OK;
}

int h2_msm(h2_curve_t curve, uint64_t handle, const uint64_t* scalars, size_t n, uint64_t out_jac[12]) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (n && !scalars) return H2_EINVAL;
  const uint64_t* cols[1] = {scalars ? scalars : (const uint64_t*)out_jac};
  return msm_host(curve, handle, cols, n, 1, out_jac, false);
}

int h2_msm_batch(h2_curve_t curve, uint64_t handle, const uint64_t* const* scalars, size_t n, size_t m,
                 uint64_t* out_affine) {
  std::lock_guard<std::mutex> lk(g_mu);
  return msm_host(curve, handle, scalars, n, m, out_affine, true);
}

int h2_srs_generate(h2_curve_t curve, const uint64_t s[4], size_t n, void* d_out_affine, void* stream_) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx.ready) return H2_ENOTINIT;
  const CurveOps* ops = ops_of((int)curve);
  if (!ops || !s || !d_out_affine || n == 0 || n >= (1ull << 32)) return H2_EINVAL;
  hipStream_t stream = stream_ ? (hipStream_t)stream_ : g_ctx.stream;
  hipError_t e = ops->srs_powers(d_out_affine, s, (uint32_t)n, stream);
  if (e != hipSuccess) return dev_fail(e, "srs_powers_kernel");
  return H2_OK;
}

int h2_fixed_base_mul(h2_curve_t curve, const void* d_scalars, size_t n, void* d_out_affine, void* stream_) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx.ready) return H2_ENOTINIT;
  const CurveOps* ops = ops_of((int)curve);
  if (!ops || !d_scalars || !d_out_affine || n == 0 || n >= (1ull << 32)) return H2_EINVAL;
  hipStream_t stream = stream_ ? (hipStream_t)stream_ : g_ctx.stream;
  hipError_t e = ops->fixed_base_mul(d_out_affine, d_scalars, (uint32_t)n, stream);
  if (e != hipSuccess) return dev_fail(e, "fixed_base_mul_kernel");
  return H2_OK;
}

int h2_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_ctx.profiling = on != 0;
  g_ctx.prof_used = 0;
  g_ctx.prof_alg_bytes = 0;
  return H2_OK;
}

int h2_profile_read(h2_profile_t* out) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!out) return H2_EINVAL;
  double ms = 0;
  for (size_t i = 0; i < g_ctx.prof_used; i++) {
    float t = 0;
    H2_TRY(hipEventSynchronize(g_ctx.prof_events[i].second));
    H2_TRY(hipEventElapsedTime(&t, g_ctx.prof_events[i].first, g_ctx.prof_events[i].second));
    ms += t;
  }
  out->launches = g_ctx.prof_used;
  out->kernel_ms = ms;
  out->algorithmic_bytes = g_ctx.prof_alg_bytes;
  g_ctx.prof_used = 0;
  g_ctx.prof_alg_bytes = 0;
  return H2_OK;
}

int h2_ntt_device(h2_curve_t curve, void* d_a, size_t m, const uint64_t omega[4], uint32_t log_n, void* stream_) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx.ready) return H2_ENOTINIT;
  if (!curve_ok((int)curve) || !d_a || !omega || m == 0 || log_n > 30) return H2_EINVAL;
  hipStream_t stream = stream_ ? (hipStream_t)stream_ : g_ctx.stream;
  if (log_n == 0) return H2_OK;
  return ntt_enqueue((int)curve, d_a, m, omega, log_n, stream);
}

int h2_ntt_scaled_device(h2_curve_t curve, void* d_a, size_t m, const uint64_t omega[4], uint32_t log_n,
                         const uint64_t scale[4], void* stream_) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx.ready) return H2_ENOTINIT;
  if (!curve_ok((int)curve) || !d_a || !omega || !scale || m == 0 || log_n > 30) return H2_EINVAL;
  hipStream_t stream = stream_ ? (hipStream_t)stream_ : g_ctx.stream;
  if (log_n == 0) {
    hipError_t e = ops_of((int)curve)->poly_scale(d_a, m, scale, stream);
    if (e != hipSuccess) return dev_fail(e, "poly_scale_kernel");
    return H2_OK;
  }
  return ntt_enqueue((int)curve, d_a, m, omega, log_n, stream, scale);
}

int h2_poly_scale_device(h2_curve_t curve, void* d_a, size_t n, size_t m, const uint64_t c[4], void* stream_) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx.ready) return H2_ENOTINIT;
  if (!curve_ok((int)curve) || !d_a || !c) return H2_EINVAL;
  if (n * m == 0) return H2_OK;
  hipError_t e = ops_of((int)curve)->poly_scale(d_a, n * m, c, stream_ ? (hipStream_t)stream_ : g_ctx.stream);
  if (e != hipSuccess) return dev_fail(e, "poly_scale_kernel");
  return H2_OK;
}

int h2_poly_coset_device(h2_curve_t curve, void* d_a, size_t n, size_t m, const uint64_t g[4], void* stream_) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx.ready) return H2_ENOTINIT;
  if (!curve_ok((int)curve) || !d_a || !g) return H2_EINVAL;
  if (n * m == 0) return H2_OK;
  hipError_t e = ops_of((int)curve)->poly_powers(d_a, n, m, g, stream_ ? (hipStream_t)stream_ : g_ctx.stream);
  if (e != hipSuccess) return dev_fail(e, "poly_powers_kernel");
  return H2_OK;
}

int h2_poly_mul_periodic_device(h2_curve_t curve, void* d_a, size_t n, size_t m, const void* d_t, size_t period,
                                void* stream_) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx.ready) return H2_ENOTINIT;
  if (!curve_ok((int)curve) || !d_a || !d_t || period == 0 || (period & (period - 1))) return H2_EINVAL;
  if (n * m == 0) return H2_OK;
  hipError_t e = ops_of((int)curve)->poly_mul_periodic(d_a, n * m, d_t, period,
                                                     stream_ ? (hipStream_t)stream_ : g_ctx.stream);
  if (e != hipSuccess) return dev_fail(e, "poly_mul_periodic_kernel");
  return H2_OK;
}

int h2_poly_inverse_device(h2_curve_t curve, void* d_a, size_t n, void* stream_) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx.ready) return H2_ENOTINIT;
  if (!curve_ok((int)curve) || !d_a) return H2_EINVAL;
  if (n == 0) return H2_OK;
  hipError_t e = ops_of((int)curve)->poly_inverse(d_a, n, stream_ ? (hipStream_t)stream_ : g_ctx.stream);
  if (e != hipSuccess) return dev_fail(e, "poly_inverse_kernel");
  return H2_OK;
}

int h2_poly_divide_linear_device(h2_curve_t curve, const void* d_a, size_t n, const uint64_t z[4], void* d_q,
                                 void* stream_) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx.ready) return H2_ENOTINIT;
  if (!curve_ok((int)curve) || !d_a || !d_q || !z || d_a == d_q) return H2_EINVAL;
  if (n == 0) return H2_OK;
  if (!g_ctx.div_ws) H2_TRY(hipMalloc(&g_ctx.div_ws, (size_t)2 * DIV_MAX_CHUNKS * 32));
  hipError_t e = ops_of((int)curve)->poly_divide_linear(d_a, n, z, d_q, g_ctx.div_ws,
                                                        stream_ ? (hipStream_t)stream_ : g_ctx.stream);
  if (e != hipSuccess) return dev_fail(e, "poly_divide kernels");
  return H2_OK;
}

int h2_poly_prefix_product_device(h2_curve_t curve, const void* d_a, size_t n, void* d_out, void* stream_) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx.ready) return H2_ENOTINIT;
  if (!curve_ok((int)curve) || !d_a || !d_out) return H2_EINVAL;
  if (n == 0) return H2_OK;
  if (!g_ctx.div_ws) H2_TRY(hipMalloc(&g_ctx.div_ws, (size_t)2 * DIV_MAX_CHUNKS * 32));
  hipError_t e = ops_of((int)curve)->poly_prefix_product(d_a, n, d_out, g_ctx.div_ws,
                                                         stream_ ? (hipStream_t)stream_ : g_ctx.stream);
  if (e != hipSuccess) return dev_fail(e, "poly_prefix kernels");
  return H2_OK;
}

int h2_chacha20_scalars_device(h2_curve_t curve, const uint8_t seed[32], uint64_t first_block, size_t n, void* d_out,
                               void* stream_) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx.ready) return H2_ENOTINIT;
  if (!curve_ok((int)curve) || !seed || !d_out) return H2_EINVAL;
  if (n == 0) return H2_OK;
  uint32_t key[8];
  for (int i = 0; i < 8; i++)
    key[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) | ((uint32_t)seed[4 * i + 2] << 16) |
             ((uint32_t)seed[4 * i + 3] << 24);
  hipError_t e = ops_of((int)curve)->chacha20_scalars(d_out, n, first_block, key,
                                                      stream_ ? (hipStream_t)stream_ : g_ctx.stream);
  if (e != hipSuccess) return dev_fail(e, "chacha20_scalars_kernel");
  return H2_OK;
}

int h2_poly_pointwise_device(h2_curve_t curve, int op, void* d_a, const void* d_b, size_t n, void* stream_) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx.ready) return H2_ENOTINIT;
  if (!curve_ok((int)curve) || !d_a || !d_b || op < 0 || op > 2) return H2_EINVAL;
  if (n == 0) return H2_OK;
  hipError_t e = ops_of((int)curve)->poly_pointwise(d_a, d_b, n, op, stream_ ? (hipStream_t)stream_ : g_ctx.stream);
  if (e != hipSuccess) return dev_fail(e, "poly_pointwise_kernel");
  return H2_OK;
}

int h2_ntt_batch(h2_curve_t curve, uint64_t* const* cols, size_t m, const uint64_t omega[4], uint32_t log_n) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx.ready) return H2_ENOTINIT;
  if (!curve_ok((int)curve) || !cols || !omega || m == 0 || log_n > 30) return H2_EINVAL;
  for (size_t j = 0; j < m; j++)
    if (!cols[j]) return H2_EINVAL;
  if (log_n == 0) return H2_OK;
  const size_t col_bytes = (size_t)32 << log_n;
  int rc = ensure_arena(&g_ctx.stage, &g_ctx.stage_bytes, m * col_bytes);
  if (rc != H2_OK) return rc;
  for (size_t j = 0; j < m; j++)
    H2_TRY(hipMemcpyAsync((char*)g_ctx.stage + j * col_bytes, cols[j], col_bytes, hipMemcpyHostToDevice,
                          g_ctx.stream));
  rc = ntt_enqueue((int)curve, g_ctx.stage, m, omega, log_n, g_ctx.stream);
  if (rc != H2_OK) return rc;
  for (size_t j = 0; j < m; j++)
    H2_TRY(hipMemcpyAsync(cols[j], (char*)g_ctx.stage + j * col_bytes, col_bytes, hipMemcpyDeviceToHost,
                          g_ctx.stream));
  H2_TRY(hipStreamSynchronize(g_ctx.stream));
  return H2_OK;
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
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx.ready) return H2_ENOTINIT;
  const CurveOps* ops = ops_of(curve);
  if (!ops || !p || !q || !out || n == 0 || n > (1u << 20)) return H2_EINVAL;
  int rc = ensure_arena(&g_ctx.stage, &g_ctx.stage_bytes, 3 * n * 64);
  if (rc != H2_OK) return rc;
  char* d = (char*)g_ctx.stage;
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
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx.ready) return H2_ENOTINIT;
  if (!a || !b || !out || n == 0 || n > (1u << 24) || field < 0 || field > 3) return H2_EINVAL;
  const CurveOps* ops = field < 2 ? curve_ops_bn254() : curve_ops_pallas();
  const int which = field & 1;
  int rc = ensure_arena(&g_ctx.stage, &g_ctx.stage_bytes, 3 * n * 32);
  if (rc != H2_OK) return rc;
  char* d = (char*)g_ctx.stage;
  H2_TRY(hipMemcpyAsync(d, a, n * 32, hipMemcpyHostToDevice, g_ctx.stream));
  H2_TRY(hipMemcpyAsync(d + n * 32, b, n * 32, hipMemcpyHostToDevice, g_ctx.stream));
  hipError_t e = ops->selftest_field_device(which, op, d, d + n * 32, d + 2 * n * 32, (uint32_t)n, g_ctx.stream);
  if (e != hipSuccess) return dev_fail(e, "selftest_field_kernel");
  H2_TRY(hipMemcpyAsync(out, d + 2 * n * 32, n * 32, hipMemcpyDeviceToHost, g_ctx.stream));
  H2_TRY(hipStreamSynchronize(g_ctx.stream));
  return H2_OK;
}
