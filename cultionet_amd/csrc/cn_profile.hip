#include "cn_profile.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#include "cn_common.h"

namespace {
struct Rec { hipEvent_t a, b; int kind; double flops, bytes; char desc[96]; char name[64]; };
struct Agg { char name[64]; double ms, flops, launches, bytes; };
char g_desc[96] = {0};
char g_name[64] = {0};
double g_bytes = 0.0;   // algorithmic bytes of the next recorded launch (cn_prof_bytes)
char g_filter[64] = {0};  // non-empty: record only launches whose cn_prof_name equals it
std::vector<Agg> g_aggs;  // per kernel name, filled by cn_profile_end, sorted by time
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

long g_cn_launches = 0;

// Kernel launches issued by the library since the last reset (reset != 0: return the count and zero it).
extern "C" long cn_launch_count(int reset) {
  return reset ? __atomic_exchange_n(&g_cn_launches, 0L, __ATOMIC_RELAXED) : __atomic_load_n(&g_cn_launches, __ATOMIC_RELAXED);
}

bool cn_prof_on() { return g_on; }

// Shape tag of the next recorded launch (only formatted while profiling; dumped when CN_PROF_DUMP names a file).
void cn_prof_desc(const char* fmt, ...) {
  if (!g_on) return;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_desc, sizeof(g_desc), fmt, ap);
  va_end(ap);
}

void cn_prof_name(const char* fmt, ...) {
  if (!g_on) return;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_name, sizeof(g_name), fmt, ap);
  va_end(ap);
}

// Algorithmic HBM bytes of the next recorded launch: every operand read once, every result written once (true,
// unpadded dims) -- what roofline.traffic (PMC counters) is held against.
void cn_prof_bytes(double bytes) {
  if (g_on) g_bytes = bytes;
}

void cn_prof_before(hipStream_t stream) {
  if (!g_on) return;
  // filtered window: bracket only launches of the named kernel (a few per step) so that the events themselves --
  // two markers on the launch stream per recorded kernel -- do not perturb the region being timed
  if (g_filter[0] != 0 && strncmp(g_filter, g_name, sizeof(g_name)) != 0) { g_desc[0] = 0; g_name[0] = 0; g_bytes = 0.0; return; }
  g_pending = get_event();
  (void)hipEventRecord(g_pending, stream);
}

void cn_prof_after(hipStream_t stream, int kind, double flops) {
  if (!g_on || g_pending == nullptr) return;
  hipEvent_t b = get_event();
  (void)hipEventRecord(b, stream);
  Rec r = {g_pending, b, kind, flops, g_bytes, {0}, {0}};
  g_bytes = 0.0;
  snprintf(r.desc, sizeof(r.desc), "%s", g_desc);
  snprintf(r.name, sizeof(r.name), "%s", g_name);
  g_desc[0] = 0;
  g_name[0] = 0;
  g_recs.push_back(r);
  g_pending = nullptr;
}

// Start recording (not thread-safe; one profiling client per process).
extern "C" int cn_profile_begin(void) {
  g_on = true;
  return CN_OK;
}

// Restrict the NEXT windows to one kernel (exact rocprof-style name as reported by cn_profile_top); "" or NULL: all.
extern "C" int cn_profile_set_filter(const char* name) {
  snprintf(g_filter, sizeof(g_filter), "%s", name ? name : "");
  return CN_OK;
}

// Stop, synchronise the recorded events and reduce: out[kind][3] = {milliseconds, flops, launches}.
extern "C" int cn_profile_end(double* out) {
  g_on = false;
  for (int i = 0; i < CN_PROF_KINDS * 3; ++i) out[i] = 0.0;
  g_aggs.clear();
  const char* dump = getenv("CN_PROF_DUMP");
  FILE* df = dump ? fopen(dump, "a") : nullptr;
  for (auto& r : g_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) != hipSuccess) return CN_ERR_LAUNCH;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return CN_ERR_LAUNCH;
    out[r.kind * 3 + 0] += ms;
    out[r.kind * 3 + 1] += r.flops;
    out[r.kind * 3 + 2] += 1.0;
    {
      size_t i = 0;
      for (; i < g_aggs.size(); ++i)
        if (strncmp(g_aggs[i].name, r.name, sizeof(r.name)) == 0) break;
      if (i == g_aggs.size()) {
        Agg a = {{0}, 0.0, 0.0, 0.0, 0.0};
        snprintf(a.name, sizeof(a.name), "%s", r.name);
        g_aggs.push_back(a);
      }
      g_aggs[i].ms += ms; g_aggs[i].flops += r.flops; g_aggs[i].launches += 1.0; g_aggs[i].bytes += r.bytes;
    }
    if (df) fprintf(df, "%d\t%s\t%.3f\t%.0f\n", r.kind, r.desc, ms * 1e3, r.flops);
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  if (df) fclose(df);
  g_recs.clear();
  std::sort(g_aggs.begin(), g_aggs.end(), [](const Agg& a, const Agg& b) { return a.ms > b.ms; });
  return CN_OK;
}

// After cn_profile_end: the rank-th kernel by total time (rocprof-style name) and its {ms, flops, launches}.
// Returns the number of distinct kernels recorded (rank >= that: outputs untouched).
extern "C" int cn_profile_top(int rank, char* name_out, int cap, double* out) {
  if (rank >= 0 && rank < (int)g_aggs.size() && name_out != nullptr && cap > 0) {
    snprintf(name_out, (size_t)cap, "%s", g_aggs[rank].name);
    out[0] = g_aggs[rank].ms; out[1] = g_aggs[rank].flops; out[2] = g_aggs[rank].launches;
  }
  return (int)g_aggs.size();
}

// After cn_profile_end: algorithmic HBM bytes (cn_prof_bytes at the launch sites: operands read once + results written
// once, true dims) summed over the recorded launches of the rank-th kernel; 0 where a launch site states none.
extern "C" double cn_profile_top_bytes(int rank) {
  return (rank >= 0 && rank < (int)g_aggs.size()) ? g_aggs[rank].bytes : 0.0;
}
