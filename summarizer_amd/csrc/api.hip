// Error reporting, version, device probe and the per-kernel event profiler of libsumk.so.
#include "sumk_internal.h"
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <vector>

namespace sumk {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
  return SUMK_ERR_HIP;
}

// ---- profiler: event pairs recorded on the launch stream around tagged kernels
struct ProfTag {
  std::vector<hipEvent_t> start, stop;  // pooled, reused after each read
  size_t used = 0;
  double total_ms = 0.0;
  int64_t launches = 0;
};
static ProfTag g_tags[SUMK_PROF_NTAGS];
static unsigned g_prof_mask = 0u;   // bit t set: launches tagged t are bracketed with events
static std::mutex g_prof_mu;

void prof_begin(int tag, hipStream_t s) {
  if (tag < 0 || tag >= SUMK_PROF_NTAGS || !(g_prof_mask & (1u << tag))) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfTag& t = g_tags[tag];
  if (t.used == t.start.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    t.start.push_back(a); t.stop.push_back(b);
  }
  (void)hipEventRecord(t.start[t.used], s);
}

void prof_end(int tag, hipStream_t s) {
  if (tag < 0 || tag >= SUMK_PROF_NTAGS || !(g_prof_mask & (1u << tag))) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfTag& t = g_tags[tag];
  if (t.used >= t.stop.size()) return;
  (void)hipEventRecord(t.stop[t.used], s);
  t.used++;
}

static void prof_drain(ProfTag& t) {
  for (size_t i = 0; i < t.used; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(t.stop[i]) == hipSuccess && hipEventElapsedTime(&ms, t.start[i], t.stop[i]) == hipSuccess) {
      t.total_ms += ms; t.launches++;
    }
  }
  t.used = 0;
}

}  // namespace sumk

extern "C" const char* sumk_last_error(void) { return sumk::g_err; }
extern "C" int sumk_version(void) { return 100; }
extern "C" int sumk_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { sumk::hip_fail(e, "hipGetDeviceCount"); return SUMK_ERR_HIP; }
  return n;
}
extern "C" int sumk_prof_enable(int32_t on) {
  std::lock_guard<std::mutex> lk(sumk::g_prof_mu);
  sumk::g_prof_mask = (unsigned)on;
  return SUMK_OK;
}
extern "C" int sumk_prof_read(int32_t tag, double* total_ms, int64_t* launches, int32_t reset) {
  using namespace sumk;
  SUMK_ARG(tag >= 0 && tag < SUMK_PROF_NTAGS, "prof: bad tag %d", tag);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfTag& t = g_tags[tag];
  prof_drain(t);
  if (total_ms) *total_ms = t.total_ms;
  if (launches) *launches = t.launches;
  if (reset) { t.total_ms = 0.0; t.launches = 0; }
  return SUMK_OK;
}
