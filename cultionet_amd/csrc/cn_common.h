// Shared helpers for the cultionet_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CN_OK 0
#define CN_ERR_ARG (-1)      // invalid argument / unsupported shape
#define CN_ERR_LAUNCH (-2)   // hipGetLastError() != hipSuccess after a launch
#define CN_ERR_LDS (-3)      // tile does not fit the 160 KiB LDS budget

#define CN_WAVE 64

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int cn_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? CN_OK : CN_ERR_LAUNCH;
}

static inline int cn_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- wave / block reductions (64-wide wavefronts) -------------------------
template <typename T>
__device__ __forceinline__ T cn_wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ float cn_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Block-wide sum for blockDim.x == NT (multiple of 64). `scratch` holds NT/64 values.
template <typename T, int NT>
__device__ __forceinline__ T cn_block_sum(T v, T* scratch) {
  v = cn_wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  T r = 0;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) r += scratch[i];
  return r;
}

__device__ __forceinline__ float cn_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float cn_silu(float x) { return x * cn_sigmoid(x); }
// d/dx silu(x) = s(x) * (1 + x * (1 - s(x)))
__device__ __forceinline__ float cn_silu_grad(float x) {
  const float s = cn_sigmoid(x);
  return s * (1.0f + x * (1.0f - s));
}
