// Error plumbing, version, and the per-family event-timing facility.
#include "common.h"

#include <stdarg.h>

#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

void fcd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* fcd_last_error_string(void) { return g_err; }
extern "C" int fcd_version(void) { return 100; }

// ---------------------------------------------------------------------------
namespace {
struct Pending {
  int fam;
  hipEvent_t e0, e1;
};
std::mutex g_mu;
bool g_prof_on = false;
std::vector<Pending> g_pending;
std::vector<hipEvent_t> g_free_events;
double g_ms[FCD_K_COUNT], g_launches[FCD_K_COUNT], g_flops[FCD_K_COUNT], g_bytes[FCD_K_COUNT];
const char* kNames[FCD_K_COUNT] = {"conv_igemm_fwd", "conv_igemm_dgrad", "conv_wgrad", "pack_weights",
                                   "norm_act",       "pool_resize",      "loss",       "optim",
                                   "misc", "conv_wino_fwd", "conv_wino_dgrad", "wino_gemm", "wino_transform"};

hipEvent_t get_event() {
  if (!g_free_events.empty()) {
    hipEvent_t e = g_free_events.back();
    g_free_events.pop_back();
    return e;
  }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}
}  // namespace

FcdProfScope::FcdProfScope(int family, hipStream_t stream, double flops, double bytes)
    : fam(family), st(stream), on(false) {
  // Every launching entry point opens one of these scopes first.  hipGetLastError() is per-thread
  // and sticky: drop whatever an earlier, unrelated HIP user of this thread left behind (e.g. a
  // benign device probe of the host framework), so that FCD_LAUNCH_CHECK reports OUR launches only.
  (void)hipGetLastError();
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_mu);
  on = true;
  e0 = get_event();
  e1 = get_event();
  g_launches[fam] += 1;
  g_flops[fam] += flops;
  g_bytes[fam] += bytes;
  hipEventRecord(e0, st);
}

FcdProfScope::~FcdProfScope() {
  if (!on) return;
  std::lock_guard<std::mutex> lk(g_mu);
  hipEventRecord(e1, st);
  g_pending.push_back({fam, e0, e1});
}

extern "C" void fcd_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_prof_on = on != 0;
}
extern "C" int fcd_prof_families(void) { return FCD_K_COUNT; }
extern "C" const char* fcd_prof_family_name(int f) { return (f >= 0 && f < FCD_K_COUNT) ? kNames[f] : "?"; }

extern "C" int fcd_prof_read(double* out, int reset) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& p : g_pending) {
    hipEventSynchronize(p.e1);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) g_ms[p.fam] += ms;
    g_free_events.push_back(p.e0);
    g_free_events.push_back(p.e1);
  }
  g_pending.clear();
  for (int f = 0; f < FCD_K_COUNT; ++f) {
    out[f * 4 + 0] = g_ms[f];
    out[f * 4 + 1] = g_launches[f];
    out[f * 4 + 2] = g_flops[f];
    out[f * 4 + 3] = g_bytes[f];
    if (reset) g_ms[f] = g_launches[f] = g_flops[f] = g_bytes[f] = 0.0;
  }
  return FCD_OK;
}
