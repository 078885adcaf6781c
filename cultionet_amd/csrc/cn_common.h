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
typedef float f32x2 __attribute__((ext_vector_type(2)));

static inline int cn_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? CN_OK : CN_ERR_LAUNCH;
}

// Every kernel launch of the library goes through CN_LAUNCH: a process-wide launch counter (cn_launch_count, read by
// bench.py as kernel launches per step) in front of hipLaunchKernelGGL.
extern long g_cn_launches;
#define CN_LAUNCH(...) do { __atomic_fetch_add(&g_cn_launches, 1L, __ATOMIC_RELAXED); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

// Dropout seeds: `seed` is a launch argument (constant inside a recorded launch plan); `step` (nullable) points at a
// device word the host bumps once per training step (cn_rng_advance_u64), so that a REPLAYED plan draws fresh masks
// every step and the eager step -- same word, same bump -- draws identical ones.
__device__ __forceinline__ unsigned long long cn_step_seed(unsigned long long seed, const unsigned long long* step) {
  return step != nullptr ? seed + (*step) * 0xD1B54A32D192ED03ull : seed;
}

static inline int cn_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- XCD-aware block order -------------------------------------------------------------------------------
// Workgroups are dealt round-robin to the 8 XCDs, each with its own 4 MiB L2. Kernels whose neighbouring blocks
// share data (halo rows, weight tiles, pixel chunks) are launched as a 1-D grid of cn_xcd_grid(total) blocks and
// decode their logical (x, y, z) here, so that CONSECUTIVE logical blocks run on the SAME XCD and hit its L2.
static inline unsigned cn_xcd_grid(long total) { return (unsigned)(((total + 7) / 8) * 8); }

__device__ __forceinline__ bool cn_xcd_block(int gx, int gy, int total, int& bx, int& by, int& bz) {
  const int lin = blockIdx.x;
  const int per = (total + 7) >> 3;
  const int l = (lin & 7) * per + (lin >> 3);
  if (l >= total) return false;
  bx = l % gx;
  const int r = l / gx;
  by = r % gy;
  bz = r / gy;
  return true;
}

// ---- wave / block reductions (64-wide wavefronts) -------------------------
template <typename T>
__device__ __forceinline__ T cn_wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Wave sum through DPP (one VALU op per step, no LDS traffic): row-wise inclusive scan, then the row totals are
// carried across rows; the wave total ends up in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float cn_dpp_add(float v) {
  const int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
  return v + __int_as_float(r);
}
__device__ __forceinline__ float cn_wave_sum_to_lane63(float v) {
  v = cn_dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = cn_dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = cn_dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = cn_dpp_add<0x118, 0xf>(v);  // row_shr:8  -> lane 15 of each row holds the row total
  v = cn_dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v = cn_dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
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
