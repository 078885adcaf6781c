// Device-side "last block done" reductions shared by the statistics / parameter-gradient kernels (gfx950).
#pragma once
#include "cn_common.h"

// ---- two-level last-block reduction of per-block rows ----------------------------------------------------------------
// Every block of a launch has stored one row of W floats (agent-scope stores: cn_t2_store). Blocks are grouped by 16
// consecutive indices; the LAST ARRIVING block of a group sums its group's rows (block order, fp64) into a group row,
// and the last arriving GROUP FINISHER sums the group rows (group order) into `tot` (LDS, W doubles) and returns true:
// one launch instead of "partial + finalize", two load latencies + two atomics deep, and bit-reproducible (the order
// of every sum is fixed by indices, never by arrival). Counters: ngroups + 1 ints, zero on entry, left zero on exit.
// Memory model (MI355X_MICROARCH.md, inter-workgroup visibility): rows / group rows are written write-through and
// read with agent-scope loads, each storing wave drains its stores before the block's ticket is drawn.
#define CN_T2_GROUP 16
struct CnTicket2 {
  float* rows;    // [nblk][W]
  double* grows;  // [ceil(nblk / 16)][W]
  int* counters;  // [ceil(nblk / 16) + 1]
  int W, nblk;
};
#define CN_T2_COUNTERS 64  // ints per ticket domain (<= 1008 blocks); ALWAYS at a fixed place of the caller's workspace:
                           // a buffer shared by calls of different shapes must never see row data where another
                           // shape keeps its counters
static inline long cn_t2_body_floats(int nblk, int W) {
  const long ng = (nblk + CN_T2_GROUP - 1) / CN_T2_GROUP;
  return (long)nblk * W + ng * W * 2 + 16;
}
// counters: CN_T2_COUNTERS zero ints; body: cn_t2_body_floats(nblk, W) floats of scratch (8-byte aligned)
static inline CnTicket2 cn_t2_carve(int* counters, float* body, int nblk, int W) {
  const long ng = (nblk + CN_T2_GROUP - 1) / CN_T2_GROUP;
  CnTicket2 t;
  t.counters = counters;
  t.grows = reinterpret_cast<double*>(body);
  t.rows = body + ng * W * 2;
  t.W = W; t.nblk = nblk;
  return t;
}
__device__ __forceinline__ void cn_t2_store(const CnTicket2& t, int blk, int col, float v) {
  __hip_atomic_store(t.rows + (long)blk * t.W + col, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Any blockDim.x (a multiple of 64). s_flag: one int of LDS. Returns true in exactly one block of the launch -- the
// one whose threads have just called finish(col, total) for every column (block-strided over the threads).
template <typename F>
__device__ __forceinline__ bool cn_t2_reduce_fn(const CnTicket2& t, int blk, int* s_flag, F&& finish) {
  const int tid = threadIdx.x;
  const int ng = (t.nblk + CN_T2_GROUP - 1) / CN_T2_GROUP;
  const int grp = blk / CN_T2_GROUP;
  const int g0 = grp * CN_T2_GROUP;
  const int gn = g0 + CN_T2_GROUP <= t.nblk ? CN_T2_GROUP : t.nblk - g0;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const int k = __hip_atomic_fetch_add(t.counters + grp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_flag = (k == gn - 1);
    if (k == gn - 1) __hip_atomic_store(t.counters + grp, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!*s_flag) return false;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int col = tid; col < t.W; col += (int)blockDim.x) {
    float v[CN_T2_GROUP];
#pragma unroll
    for (int i = 0; i < CN_T2_GROUP; ++i)  // clamped, never predicated loads: all in flight together
      v[i] = __hip_atomic_load(t.rows + (long)(g0 + (i < gn ? i : gn - 1)) * t.W + col, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < CN_T2_GROUP; ++i) s += i < gn ? (double)v[i] : 0.0;
    __hip_atomic_store(t.grows + (long)grp * t.W + col, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const int k = __hip_atomic_fetch_add(t.counters + ng, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_flag = (k == ng - 1);
    if (k == ng - 1) __hip_atomic_store(t.counters + ng, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!*s_flag) return false;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int col = tid; col < t.W; col += (int)blockDim.x) {
    double s = 0.0;
    // (32 group rows per batch: the 512-block launches -- 32 groups -- finish in ONE load round trip instead of two; the
    // order of the sum is unchanged)
    constexpr int NB = 2 * CN_T2_GROUP;
    for (int k0 = 0; k0 < ng; k0 += NB) {
      double v[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i)
        v[i] = __hip_atomic_load(t.grows + (long)(k0 + i < ng ? k0 + i : ng - 1) * t.W + col, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int i = 0; i < NB; ++i) s += k0 + i < ng ? v[i] : 0.0;
    }
    finish(col, s);
  }
  return true;
}

// The same with the column totals left in `tot` (LDS, W doubles), visible to the whole block on return.
__device__ __forceinline__ bool cn_t2_reduce(const CnTicket2& t, int blk, int* s_flag, double* tot) {
  const bool last = cn_t2_reduce_fn(t, blk, s_flag, [&](int col, double s) { tot[col] = s; });
  if (last) __syncthreads();
  return last;
}

// ---- head of the grouped-BatchNorm workspace (cn_bn_group_workspace_floats_bf16) --------------------------------------
// ONE zero-filled buffer per (device, stream) serves calls of every shape, so every ticket counter lives at a FIXED
// place in its head: [CN_BNWS_FIN_INTS finalize tickets][CN_BNWS_T2_DOMAINS x CN_T2_COUNTERS statistics-pass tickets]
// [CN_BNWS_CONV_DOMAINS x CN_T2_COUNTERS tickets of the convolution launches that finish their own statistics].
#define CN_BNWS_FIN_INTS 512
#define CN_BNWS_T2_DOMAINS 4
#define CN_BNWS_CONV_DOMAINS 64
#define CN_BNWS_CONV_OFF (CN_BNWS_FIN_INTS + CN_BNWS_T2_DOMAINS * CN_T2_COUNTERS)
#define CN_BNWS_HEAD_INTS (CN_BNWS_CONV_OFF + CN_BNWS_CONV_DOMAINS * CN_T2_COUNTERS)
#define CN_BNWS_CONV_MAX_TILES ((CN_T2_COUNTERS - 1) * CN_T2_GROUP)  // 1008 tiles per (group, cout block)
