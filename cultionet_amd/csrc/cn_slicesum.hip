// One launch for all pending weight-gradient slice sums (see cn_slicesum.h).
#include "cn_slicesum.h"

#include <cstddef>
#include <cstring>

static thread_local CnSliceSum* cn_ss_sink = nullptr;
static thread_local int cn_ss_cap = 0, cn_ss_count = 0;
static thread_local const float* cn_ss_lo = nullptr;  // only sums into [lo, hi) are deferred (the flat gradient buffer)
static thread_local const float* cn_ss_hi = nullptr;

bool cn_ss_push(const CnSliceSum* js, int n) {
  if (cn_ss_sink == nullptr || n < 1 || cn_ss_count + n > cn_ss_cap) return false;
  for (int k = 0; k < n; ++k) {
    if (js[k].nslices <= 0 || js[k].n <= 0) return false;
    if (js[k].dw < cn_ss_lo || js[k].dw + js[k].n > cn_ss_hi) return false;  // a temporary dW is consumed at once
    // two pending records must never accumulate into the same dW from different blocks of one launch
    for (int i = 0; i < cn_ss_count; ++i)
      if (cn_ss_sink[i].dw == js[k].dw) return false;
  }
  for (int k = 0; k < n; ++k) {
    // chunk0 (the record's last word) belongs to cn_slice_sums_run. A table upload of the previous step may still be
    // in flight when a plan is replayed back to back: an identical record writes NOTHING, a different one is stored by
    // one assignment of a complete record -- an upload never sees a record with a zeroed chunk0.
    CnSliceSum* slot = cn_ss_sink + cn_ss_count++;
    if (memcmp(slot, &js[k], offsetof(CnSliceSum, chunk0)) == 0) continue;
    CnSliceSum r = js[k];
    r.chunk0 = slot->chunk0;
    *slot = r;
  }
  return true;
}

// Weight-gradient launches of THIS thread whose dW lies inside [dw_lo, dw_lo + dw_floats) -- the flat gradient buffer --
// append their slice sums to host_table (capacity records of 64 bytes; pinned host memory if cn_slice_sums_run is to
// upload it) from now on. The count restarts at 0.
extern "C" int cn_slice_sums_begin(void* host_table, int capacity, const float* dw_lo, long dw_floats) {
  if (host_table == nullptr || capacity < 1 || dw_lo == nullptr || dw_floats < 1) return CN_ERR_ARG;
  cn_ss_sink = (CnSliceSum*)host_table;
  cn_ss_cap = capacity;
  cn_ss_count = 0;
  cn_ss_lo = dw_lo;
  cn_ss_hi = dw_lo + dw_floats;
  return CN_OK;
}

// Records appended since cn_slice_sums_begin (-1: no sink registered).
extern "C" int cn_slice_sums_count(void) { return cn_ss_sink ? cn_ss_count : -1; }

// Stop deferring (records not yet run are the caller's to run).
extern "C" int cn_slice_sums_end(void) {
  cn_ss_sink = nullptr;
  cn_ss_cap = cn_ss_count = 0;
  cn_ss_lo = cn_ss_hi = nullptr;
  return CN_OK;
}

__global__ __launch_bounds__(256) void cn_slice_sums_kernel(const CnSliceSum* __restrict__ tab, int n) {
  __shared__ float red[4][64];
  // binary search: last record whose chunk0 <= blockIdx.x (wave-uniform)
  int lo = 0, hi = n - 1;
  const int b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].chunk0 <= b) lo = mid; else hi = mid - 1;
  }
  const CnSliceSum j = tab[lo];
  const int c = b - j.chunk0;
  if (j.kind == 0) {
    if (j.nslices <= 128) {
      cn_ss_chunk_flat(j.part, j.dw, j.slice_stride, j.nslices, j.n, (long)c * CN_SS_CHUNK0);
    } else {
      cn_ss_chunk_waves(j.part, j.dw, j.slice_stride, j.nslices, j.n, (long)c * CN_SS_CHUNK1, red,
                        [](long i) { return i; }, [](long i) { return i; });
    }
  } else {
    const int T = j.T, CP = j.CP, CQ = j.CQ;
    const long CPp = j.CPp, CQp = j.CQp;
    cn_ss_chunk_waves(j.part, j.dw, j.slice_stride, j.nslices, j.n, (long)c * CN_SS_CHUNK1, red,
                      [=](long i) {  // i = (t*CP + cp)*CQ + cq
                        const int cq = (int)(i % CQ);
                        const long r = i / CQ;
                        const int cp = (int)(r % CP), t = (int)(r / CP);
                        return ((long)t * CPp + cp) * CQp + cq;
                      },
                      [=](long i) {
                        const int cq = (int)(i % CQ);
                        const long r = i / CQ;
                        const int cp = (int)(r % CP), t = (int)(r / CP);
                        return ((long)cp * CQ + cq) * T + t;
                      });
  }
}

// Sum records [first, first + n) of host_table in ONE launch. Deals the launch's blocks to the records (writes each
// record's first block into host_table). dev_table: device copy of the table (capacity * 64 bytes), refreshed from
// host_table for that range when upload != 0 (hipMemcpyAsync on `stream`: host_table must be pinned and that range must
// not be rewritten with different contents until the copy has run); with upload == 0 the caller asserts that the device
// copy already holds exactly these records, dealt over exactly this range.
extern "C" int cn_slice_sums_run(void* host_table, void* dev_table, int first, int n, int upload, void* stream) {
  if (n <= 0) return CN_OK;
  if (host_table == nullptr || dev_table == nullptr || first < 0) return CN_ERR_ARG;
  CnSliceSum* h = (CnSliceSum*)host_table + first;
  CnSliceSum* d = (CnSliceSum*)dev_table + first;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    h[i].chunk0 = blocks;
    blocks += cn_ss_chunks(h[i]);
  }
  if (blocks <= 0) return CN_OK;
  if (upload &&
      hipMemcpyAsync(d, h, (size_t)n * sizeof(CnSliceSum), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess)
    return CN_ERR_LAUNCH;
  CN_LAUNCH(cn_slice_sums_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d, n);
  return cn_check_launch();
}
