// Error plumbing, version, and the per-family event-timing facility.
#include "common.h"

#include <stdarg.h>

#include <mutex>
#include <string>
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
#include "build/build_hash.h"
extern "C" const char* fcd_build_hash(void) { return FCD_BUILD_HASH; }

// --------------------------------------------------------------------------- run-time switches (switches.h)
#include <stdlib.h>

int g_fcd_switch[FCD_SW_COUNT];
namespace {
struct SwitchRow { const char* name; int def; const char* help; };
const SwitchRow kSwitches[FCD_SW_COUNT] = {
#define FCD_SW_ROW(NAME, DEF, HELP) {#NAME, DEF, HELP},
    FCD_SWITCH_TABLE(FCD_SW_ROW)
#undef FCD_SW_ROW
};
// value of a switch after range rules that used to live next to each getenv
int switch_clean(int id, int v) {
  switch (id) {
    case FCD_SW_WINO: return (v == 0 || v == 2 || v == 4) ? v : 4;
    case FCD_SW_WINO_SPLIT: return v < 0 ? 1 : (v > 2 ? 1 : v);
    case FCD_SW_THIN_MFMA: return v < 0 ? 0 : v;
    case FCD_SW_WGRAD_WGS: return v < 1 ? 512 : v;
    case FCD_SW_WINO_WG_WGS: return v < 0 ? 0 : v;
    case FCD_SW_WINO2_WAVES: return (v == 1 || v == 4) ? v : 8;
    case FCD_SW_WINO2_KS: return (v == 1 || v == 3) ? v : 2;
    case FCD_SW_D_POOL: return (v == 1 || v == 2) ? v : 0;
    default: return v;
  }
}
int switch_parse(int id, const char* e) {
  if (id == FCD_SW_D_POOL) {       // historical spellings
    if (!strcmp(e, "fused")) return 0;
    if (!strcmp(e, "diff")) return 1;
    if (!strcmp(e, "pooled")) return 2;
  }
  return atoi(e);
}
// the environment is read HERE, once, when the library is mapped -- and nowhere else
__attribute__((constructor)) void fcd_switches_from_environment() {
  char var[64];
  for (int i = 0; i < FCD_SW_COUNT; ++i) {
    snprintf(var, sizeof(var), "FCD_%s", kSwitches[i].name);
    const char* e = getenv(var);
    g_fcd_switch[i] = switch_clean(i, (e && e[0]) ? switch_parse(i, e) : kSwitches[i].def);
  }
}
int switch_index(const char* name) {
  if (!name) return -1;
  if (!strncmp(name, "FCD_", 4)) name += 4;
  for (int i = 0; i < FCD_SW_COUNT; ++i)
    if (!strcmp(name, kSwitches[i].name)) return i;
  return -1;
}
}  // namespace

extern "C" int fcd_switch_count(void) { return FCD_SW_COUNT; }
extern "C" const char* fcd_switch_name(int i) { return (i >= 0 && i < FCD_SW_COUNT) ? kSwitches[i].name : nullptr; }
extern "C" const char* fcd_switch_help(int i) { return (i >= 0 && i < FCD_SW_COUNT) ? kSwitches[i].help : nullptr; }
extern "C" int fcd_switch_default(int i) { return (i >= 0 && i < FCD_SW_COUNT) ? kSwitches[i].def : 0; }
// current value by name ("WGRAD_SPLIT" or "FCD_WGRAD_SPLIT"); FCD_ERR_INVALID for an unknown name (no switch is negative)
extern "C" int fcd_switch_get(const char* name) {
  const int i = switch_index(name);
  FCD_CHECK_ARG(i >= 0, "fcd_switch_get: unknown switch '%s'", name ? name : "(null)");
  return g_fcd_switch[i];
}
// set by name, returns the previous value (value < 0: restore the default).  Not synchronised with launches in flight on other
// threads: switches are for A/B runs and tests.
extern "C" int fcd_switch_set(const char* name, int value) {
  const int i = switch_index(name);
  FCD_CHECK_ARG(i >= 0, "fcd_switch_set: unknown switch '%s'", name ? name : "(null)");
  const int old = g_fcd_switch[i];
  g_fcd_switch[i] = switch_clean(i, value < 0 ? kSwitches[i].def : value);
  return old;
}

// ---------------------------------------------------------------------------
namespace {
struct Pending {
  int fam;
  hipEvent_t e0, e1;
  int detail;   // index into g_detail, -1: none
};
struct Detail {
  int fam;
  std::string tag;
  double flops, bytes, ms;
};
std::mutex g_mu;
bool g_prof_on = false;
bool g_detail_on = false;
std::vector<Detail> g_detail;
std::vector<Pending> g_pending;
std::vector<hipEvent_t> g_free_events;
double g_ms[FCD_K_COUNT], g_launches[FCD_K_COUNT], g_flops[FCD_K_COUNT], g_bytes[FCD_K_COUNT];
const char* kNames[FCD_K_COUNT] = {"conv_igemm_fwd", "conv_igemm_dgrad", "conv_wgrad", "pack_weights",
                                   "norm_act",       "pool_resize",      "loss",       "optim",
                                   "misc", "conv_wino_fwd", "conv_wino_dgrad", "wino_gemm", "wino_transform",
                                   "conv_wgrad_wino", "conv_wino2_fwd", "conv_wino2_dgrad", "wino_gemm_bf16x6", "conv_wgrad_bf16x6"};

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

const char* fcd_prof_tagf(const char* fmt, ...) {
  if (!g_detail_on) return nullptr;
  static thread_local char buf[192];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return buf;
}

FcdProfScope::FcdProfScope(int family, hipStream_t stream, double flops, double bytes, const char* tag)
    : fam(family), st(stream), on(false), detail_idx(-1) {
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
  if (g_detail_on) {
    g_detail.push_back({fam, tag ? tag : "", flops, bytes, 0.0});
    detail_idx = (int)g_detail.size() - 1;
  }
  hipEventRecord(e0, st);
}

FcdProfScope::~FcdProfScope() {
  if (!on) return;
  std::lock_guard<std::mutex> lk(g_mu);
  hipEventRecord(e1, st);
  g_pending.push_back({fam, e0, e1, detail_idx});
}

// 0 off, 1 per-family totals, 2 totals + a per-launch log (family, tag, ms, flops, bytes: fcd_prof_detail_read)
extern "C" void fcd_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_prof_on = on != 0;
  g_detail_on = on == 2;
}
extern "C" int fcd_prof_families(void) { return FCD_K_COUNT; }
extern "C" const char* fcd_prof_family_name(int f) { return (f >= 0 && f < FCD_K_COUNT) ? kNames[f] : "?"; }

extern "C" int fcd_prof_read(double* out, int reset) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& p : g_pending) {
    hipEventSynchronize(p.e1);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
      g_ms[p.fam] += ms;
      if (p.detail >= 0 && p.detail < (int)g_detail.size()) g_detail[p.detail].ms = ms;
    }
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

// Per-launch log as text, one line per scope: "family\ttag\tms\tflops\tbytes\n".  Call fcd_prof_read() first (it
// resolves the events).  Returns the number of bytes needed (incl. NUL); copies at most cap - 1; clears the log
// when ``reset``.
extern "C" int64_t fcd_prof_detail_read(char* buf, int64_t cap, int reset) {
  std::lock_guard<std::mutex> lk(g_mu);
  std::string out;
  char line[320];
  for (auto& d : g_detail) {
    snprintf(line, sizeof(line), "%s\t%s\t%.6f\t%.6g\t%.6g\n", kNames[d.fam], d.tag.c_str(), d.ms, d.flops, d.bytes);
    out += line;
  }
  if (buf && cap > 0) {
    const size_t n = std::min<size_t>(out.size(), (size_t)cap - 1);
    memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  if (reset) g_detail.clear();
  return (int64_t)out.size() + 1;
}
