#include "cn_profile.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cn_common.h"

namespace {
struct Rec { hipEvent_t a, b; int kind; double flops; char desc[96]; };
char g_desc[96] = {0};
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t g_pending = nullptr;

hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

bool cn_prof_on() { return g_on; }

// Shape tag of the next recorded launch (only formatted while profiling; dumped when CN_PROF_DUMP names a file).
void cn_prof_desc(const char* fmt, ...) {
  if (!g_on) return;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_desc, sizeof(g_desc), fmt, ap);
  va_end(ap);
}

void cn_prof_before(hipStream_t stream) {
  if (!g_on) return;
  g_pending = get_event();
  (void)hipEventRecord(g_pending, stream);
}

void cn_prof_after(hipStream_t stream, int kind, double flops) {
  if (!g_on || g_pending == nullptr) return;
  hipEvent_t b = get_event();
  (void)hipEventRecord(b, stream);
  Rec r = {g_pending, b, kind, flops, {0}};
  snprintf(r.desc, sizeof(r.desc), "%s", g_desc);
  g_desc[0] = 0;
  g_recs.push_back(r);
  g_pending = nullptr;
}

// Start recording (not thread-safe; one profiling client per process).
extern "C" int cn_profile_begin(void) {
  g_on = true;
  return CN_OK;
}

// Stop, synchronise the recorded events and reduce: out[kind][3] = {milliseconds, flops, launches}.
extern "C" int cn_profile_end(double* out) {
  g_on = false;
  for (int i = 0; i < CN_PROF_KINDS * 3; ++i) out[i] = 0.0;
  const char* dump = getenv("CN_PROF_DUMP");
  FILE* df = dump ? fopen(dump, "a") : nullptr;
  for (auto& r : g_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) != hipSuccess) return CN_ERR_LAUNCH;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return CN_ERR_LAUNCH;
    out[r.kind * 3 + 0] += ms;
    out[r.kind * 3 + 1] += r.flops;
    out[r.kind * 3 + 2] += 1.0;
    if (df) fprintf(df, "%d\t%s\t%.3f\t%.0f\n", r.kind, r.desc, ms * 1e3, r.flops);
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  if (df) fclose(df);
  g_recs.clear();
  return CN_OK;
}
