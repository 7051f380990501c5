// h2_internal.hpp -- state shared by the translation units behind the C ABI (h2_capi.hip and the C++ prover).
//
// One DevCtx per GPU the process drives (h2_init: one; h2_init_devices: several).  Every scratch arena remembers the
// stream that used it last and an event recorded behind that use, so a call on ANOTHER stream first waits for it:
// callers may hand any stream to the *_device entry points (include/h2hip.h, threading paragraph).  The MSM and the
// NTT have separate arenas, so an MSM on one stream and an NTT on another do overlap.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/h2hip.h"
#include "h2_curve_ops.hpp"
#include "h2_msm.hpp"

namespace h2 {

struct Arena {
  void* p = nullptr;
  size_t bytes = 0;
  hipEvent_t ev = nullptr;        // recorded after the last enqueue that used the arena
  hipStream_t last = nullptr;
  bool used = false;
  // MSM workspace: [clean_off, clean_off + clean_bytes) is zero when the work enqueued so far has run -- the previous MSM
  // launch sequence zeroed its counter region again on its way out (msm_rowcol_kernel) -- so the next one skips its memset
  size_t clean_off = 0, clean_bytes = 0;
};

// Scratch that a launch sequence owns from its first kernel to its last (the MSM workspace, the NTT's second buffer): one
// arena per stream for up to H2_ARENA_SLOTS streams, so that launch sequences enqueued on different streams run side by
// side (two proofs in flight on one GPU: bench.py's `two_steps_in_flight`).  A further stream takes over the slot that
// has been idle longest; arena_acquire orders it behind that slot's previous user (an event wait), as it ordered every
// stream behind every other before round 3.
constexpr int H2_ARENA_SLOTS = 4;
struct ArenaSet {
  Arena slot[H2_ARENA_SLOTS];
  hipStream_t owner[H2_ARENA_SLOTS] = {};
  bool owned[H2_ARENA_SLOTS] = {};
  uint64_t used_at[H2_ARENA_SLOTS] = {};
  uint64_t clock = 0, takeovers = 0;
  Arena& of(hipStream_t s) {
    int pick = -1;
    for (int i = 0; i < H2_ARENA_SLOTS && pick < 0; i++)
      if (owned[i] && owner[i] == s) pick = i;
    for (int i = 0; i < H2_ARENA_SLOTS && pick < 0; i++)
      if (!owned[i]) pick = i;
    if (pick < 0) {
      pick = 0;
      for (int i = 1; i < H2_ARENA_SLOTS; i++)
        if (used_at[i] < used_at[pick]) pick = i;
      takeovers++;
    }
    owned[pick] = true;
    owner[pick] = s;
    used_at[pick] = ++clock;
    return slot[pick];
  }
};

struct TwiddleEntry {
  int field;
  uint32_t log_n;
  uint64_t omega[4];
  bool scaled;            // the inter-pass twiddles carry `scale` (a two-pass scaled transform)
  uint64_t scale[4];
  void* tw;
  uint64_t stamp;
};

struct DevCtx {
  int device = -1;
  hipStream_t stream = nullptr;   // the library's own (blocking) stream on this device
  hipStream_t side_stream = nullptr;   // the C++ prover's second stream (transforms beside the MSM tails), created on demand
  hipEvent_t side_ev[3] = {nullptr, nullptr, nullptr};
  ArenaSet msm_ws, ntt_ws;
  Arena stage, div_ws;
  hipEvent_t shard_ev = nullptr;     // C++ prover: this context's share of a commit phase is done / the columns are final
  hipEvent_t tail_event = nullptr;   // recorded behind the accumulate kernel of the latest MSM (h2_stream_wait_msm_tail)
  hipEvent_t tail_wait = nullptr;    // the event to wait on for that: tail_event, or the profiling stop event of the
                                     // launch when profiling records one at the same place (an event record costs ~6 us
                                     // of stream time: profiles/r03 step timeline)
  bool tail_recorded = false, tail_wanted = false;   // the event costs ~5 us per MSM: recorded once somebody asked
  std::vector<TwiddleEntry> twiddles;
  uint64_t stamp = 0;
  // kernel timing for the roofline (h2_profile_*): event pairs around the bucket-accumulate kernel
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  size_t prof_used = 0;
  double prof_alg_bytes = 0;
};

struct BasesEntry {
  int curve;
  size_t n;
  MsmGeom geom;
  std::vector<void*> table;       // per context: W * n affine points in the working form
  size_t table_bytes;
};

struct Global {
  bool ready = false;
  std::vector<DevCtx> ctx;
  std::map<uint64_t, BasesEntry> bases;
  uint64_t next_handle = 1;
  bool profiling = false;
  std::string last_error;
};

extern Global g_h2;
extern std::recursive_mutex g_h2_mu;

// the process's current HIP device is switched for the lifetime of the guard
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = hipSetDevice(device) == hipSuccess;
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

int dev_fail(hipError_t e, const char* where);
const CurveOps* ops_of(int curve);
bool curve_ok(int c);
// context of the calling thread's current HIP device (the only context when there is just one); null if none
DevCtx* ctx_current();
size_t ctx_index(const DevCtx* c);

// arenas: acquire = grow if needed + order `s` behind the previous user; release = record the event behind this use
int arena_acquire(Arena& a, size_t want, hipStream_t s);
int arena_release(Arena& a, hipStream_t s);

// enqueue m MSMs (columns of n scalars, col_stride elements apart, bases first_base ... first_base + n - 1 of the
// registered vector) -> m Jacobian (96 B) or affine (64 B) points at d_out; all on `stream`
// `per_column` (optional, m <= MSM_MAX_MULTI entries of the same length as `be`): column j commits against
// per_column[j] instead of `be` -- the columns of one launch may use different bases
int msm_device_run(DevCtx& c, int curve, const BasesEntry& be, const void* d_scalars, size_t first_base, size_t n,
                   size_t col_stride, size_t m, void* d_out, bool affine_out, hipStream_t stream,
                   const BasesEntry* const* per_column = nullptr);
int ntt_enqueue(DevCtx& c, int curve, void* d_a, size_t m, const uint64_t omega[4], uint32_t log_n, hipStream_t stream,
                const uint64_t* scale = nullptr);
int msm_common_checks(int curve, uint64_t handle, size_t first, size_t n, size_t m, const BasesEntry** be);

#define H2_TRY(call)                                        \
  do {                                                      \
    hipError_t _e = (call);                                 \
    if (_e != hipSuccess) return ::h2::dev_fail(_e, #call); \
  } while (0)

}  // namespace h2
