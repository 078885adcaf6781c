// Native executor of recorded launch plans (cultionet_amd/replay.py).
//
// A plan is the C-ABI call list of one forward / training step, recorded once and replayed with constant arguments.
// Replaying it from Python costs ~12 us per entry (ctypes marshalling of ~20 arguments, torch's event / stream wrappers):
// 7.5 ms for the 463 launches + ~150 stream operations of the reference-default training step (hidden 64, batch 4,
// 16-mixed), whose GPU work is ~7 ms -- host-bound even when replayed. cn_plan_run walks the same list in C: every entry
// is a trampoline index + its argument slots (tools/gen_plan_trampolines.py generates one trampoline per entry point of
// include/cultionet_hip.h into cn_plan_gen.inc), or an event record / stream wait on raw HIP handles.
// The reference has no counterpart (torch eager + Lightning's loop: /root/reference/src/cultionet/model.py:273-314).
#include <cstring>
#include "cn_common.h"
#include "../../include/cultionet_hip.h"

#define CN_PLAN_SLOTS 28

struct CnPlanOp {   // 232 bytes; the host builds arrays of these (replay.py: numpy uint64 [n][29])
  int kind;         // 0: C-ABI call fn(a...)   1: hipEventRecord(a[0] event, a[1] stream)   2: hipStreamWaitEvent(a[0] stream, a[1] event)
  int fn;           // kind 0: index from cn_plan_fn_index
  uint64_t a[CN_PLAN_SLOTS];
};
static_assert(sizeof(CnPlanOp) == 8 + 8 * CN_PLAN_SLOTS, "CnPlanOp layout");

struct CnPlanFn { const char* name; long (*call)(const uint64_t*); int nargs; };

static inline float cn_plan_f32(uint64_t v) {
  const uint32_t lo = (uint32_t)v;
  float f;
  memcpy(&f, &lo, 4);
  return f;
}

#include "cn_plan_gen.inc"

static constexpr int CN_PLAN_NFN = (int)(sizeof(cn_plan_fns) / sizeof(cn_plan_fns[0]));

// Index of an entry point for CnPlanOp.fn (-1: unknown). *nargs_out (nullable) receives its parameter count.
extern "C" int cn_plan_fn_index(const char* name, int* nargs_out) {
  if (name == nullptr) return -1;
  for (int i = 0; i < CN_PLAN_NFN; ++i)
    if (strcmp(cn_plan_fns[i].name, name) == 0) {
      if (nargs_out != nullptr) *nargs_out = cn_plan_fns[i].nargs;
      return i;
    }
  return -1;
}

// Run n plan entries in order. Stops at the first failing entry: returns its status (a CN_ERR_* of the entry point, or
// CN_ERR_LAUNCH for a failed HIP event / stream call, CN_ERR_ARG for a malformed entry) with its index in *failed.
extern "C" int cn_plan_run(const void* ops, int n, int* failed) {
  const CnPlanOp* p = (const CnPlanOp*)ops;
  if (p == nullptr && n > 0) return CN_ERR_ARG;
  for (int i = 0; i < n; ++i) {
    int rc = CN_OK;
    switch (p[i].kind) {
      case 0:
        if (p[i].fn < 0 || p[i].fn >= CN_PLAN_NFN) rc = CN_ERR_ARG;
        else rc = (int)cn_plan_fns[p[i].fn].call(p[i].a);
        break;
      case 1:
        rc = hipEventRecord((hipEvent_t)(uintptr_t)p[i].a[0], (hipStream_t)(uintptr_t)p[i].a[1]) == hipSuccess ? CN_OK : CN_ERR_LAUNCH;
        break;
      case 2:
        rc = hipStreamWaitEvent((hipStream_t)(uintptr_t)p[i].a[0], (hipEvent_t)(uintptr_t)p[i].a[1], 0) == hipSuccess ? CN_OK : CN_ERR_LAUNCH;
        break;
      default:
        rc = CN_ERR_ARG;
    }
    if (rc != CN_OK) {
      if (failed != nullptr) *failed = i;
      return rc;
    }
  }
  return CN_OK;
}
