// Deferred sums of weight-gradient partial slices (gfx950).
//
// A many-split weight gradient stores one partial dW slice per pixel split into scratch and needs
//     dW += sum_s slice[s]
// before anything reads dW -- which is the optimizer (or the gradient bucket's all-reduce), not the backward pass.
// Launched right behind every contraction these sums were 34 (fp32) / 68 (bf16) small dependent kernels per step
// (1.27 ms of the bf16 step). While a sink is registered on the calling thread (cn_slice_sums_begin) the
// weight-gradient entry points append a CnSliceSum record instead of launching their reduction; cn_slice_sums_run sums
// ALL pending records in ONE launch whose blocks are dealt to (record, output chunk) pairs, so the work is balanced
// whatever the mix of layer sizes. Every output is summed by the same threads in the same order as the immediate
// kernels (deterministic, and bit-identical to the immediate path for the non-atomic cases).
// Reference semantics: plain autograd accumulation of conv weight gradients
// (/root/reference/src/cultionet/nn/modules/convolution.py:71-120).
#pragma once
#include "cn_common.h"

struct CnSliceSum {        // 64 bytes; the engine treats the table as opaque memory
  const float* part;       // first slice
  float* dw;               // destination, accumulated (+=)
  long slice_stride;       // floats between consecutive slices
  long n;                  // outputs
  int nslices;
  int kind;                // 0: dw[i] += sum part[s][i]   1: bf16 path, part is [T][CPp][CQp], dw is [CP][CQ][T]
  int T, CP, CQ;           // kind 1
  int CPp, CQp;            // kind 1: padded slice dimensions
  int chunk0;              // first block of this record in the batched launch (cn_slice_sums_plan)
};
static_assert(sizeof(CnSliceSum) == 64, "CnSliceSum is a 64-byte record");

#define CN_SS_CHUNK0 256   // outputs per block, kind 0
#define CN_SS_CHUNK1 64    // outputs per block, kind 1 (and kind 0 with more than 128 slices)

static inline int cn_ss_chunks(const CnSliceSum& j) {
  const int per = (j.kind == 0 && j.nslices <= 128) ? CN_SS_CHUNK0 : CN_SS_CHUNK1;
  return (int)((j.n + per - 1) / per);
}

// host side: true if ALL n records were taken by the thread's sink (the caller then skips its own reduce launch)
bool cn_ss_push(const CnSliceSum* js, int n);

// ---- device: one chunk of one record ----------------------------------------------------------------------------
// kind 0, <= 128 slices: thread = output, even / odd slices in two accumulators (the order of cn_wgrad_reduce_kernel
// with gridDim.y == 1)
__device__ __forceinline__ void cn_ss_chunk_flat(const float* __restrict__ part, float* __restrict__ dw, long stride,
                                                 int nslices, long n, long i0) {
  const long i = i0 + threadIdx.x;
  if (i >= n) return;
  float s0 = 0.f, s1 = 0.f;
  int k = 0;
  for (; k + 1 < nslices; k += 2) {
    s0 += part[(long)k * stride + i];
    s1 += part[(long)(k + 1) * stride + i];
  }
  if (k < nslices) s0 += part[(long)k * stride + i];
  dw[i] += s0 + s1;
}

// 64 outputs per block, the slices dealt over the block's 4 waves with 4 independent loads in flight per thread,
// combined through LDS (the order of cn_bwgrad_reduce_kernel). src(i) / dst(i) map an output to its slice / dW offset.
template <typename SrcFn, typename DstFn>
__device__ __forceinline__ void cn_ss_chunk_waves(const float* __restrict__ part, float* __restrict__ dw, long slice,
                                                  int nslices, long n, long i0, float (*red)[64], SrcFn src, DstFn dst) {
  const int lane = threadIdx.x & 63, kg = threadIdx.x >> 6;
  const long i = i0 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < n) {
    const float* p = part + src(i);
    int k = kg;
    for (; k + 12 < nslices; k += 16) {
      s0 += p[k * slice];
      s1 += p[(k + 4) * slice];
      s2 += p[(k + 8) * slice];
      s3 += p[(k + 12) * slice];
    }
    for (; k < nslices; k += 4) s0 += p[k * slice];
  }
  red[kg][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (kg == 0 && i < n) dw[dst(i)] += (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
  __syncthreads();
}
