// Micro-benchmark: how well does ONE wave per SIMD (and two, three) keep the matrix pipe busy with cn_bconv_kernel's step
// loop, and what does rotating the LDS fragment reads buy?
//   batch:  per 32-channel step: 8 ds_read_b128 (all pixel fragments of the step), then 8 MFMAs     (the shipped loop)
//   rotate: the 4 fragments of k16 half 0 of step s+1 are read right after the 4 MFMAs of half 0 of step s have issued
//           (same registers), likewise half 1: every read has ~4 MFMAs (128 cycles) of cover instead of none.
// Weight fragments: global loads one step ahead in both variants. Occupancy is set through the dynamic LDS size.
// Build: hipcc --offload-arch=gfx950 -O3 -o loop_rotate_bench loop_rotate_bench.hip ; run: ./loop_rotate_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int toff_of(int it) {
  const int tap = it % 9;
  return ((tap / 3) * 27 + (tap % 3)) * 144 + ((it / 9) & 1) * 64;
}

template <int ROT>
__global__ __launch_bounds__(256, 3) void k(const u32x4* __restrict__ w, const u32x4* __restrict__ img, float* out,
                                           int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int q = tid; q < 189 * 8; q += 256) {
    const int p = q >> 3, c = q & 7;
    *reinterpret_cast<u32x4*>(lds + p * 144 + c * 16) = img[(blockIdx.x % 64) * 189 * 8 + q];
  }
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const u32x4* wp = w + (size_t)wid * 64 + lane;
  const int r = lane & 31, h = lane >> 5;
  int pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) pb[i] = ((i * 32 + r) % 125 / 25 * 27 + (i * 32 + r) % 25) * 144 + h * 16;
  u32x4 a0v = wp[0], a1v = wp[256];
  bf16x8 xa[4], xb[4];
  if (ROT) {
    const int t0 = toff_of(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xa[i] = *reinterpret_cast<const bf16x8*>(lds + pb[i] + t0);
      xb[i] = *reinterpret_cast<const bf16x8*>(lds + pb[i] + t0 + 32);
    }
  }
#pragma unroll 2
  for (int it = 0; it < iters; ++it) {
    const bf16x8 a0 = __builtin_bit_cast(bf16x8, a0v), a1 = __builtin_bit_cast(bf16x8, a1v);
    const u32x4 n0 = wp[(size_t)((it + 1) % 512) * 512];
    const u32x4 n1 = wp[(size_t)((it + 1) % 512) * 512 + 256];
    if (!ROT) {
      const int toff = toff_of(it);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xa[i] = *reinterpret_cast<const bf16x8*>(lds + pb[i] + toff);
        xb[i] = *reinterpret_cast<const bf16x8*>(lds + pb[i] + toff + 32);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, xa[i], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, xb[i], acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      const int tn = toff_of(it + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, xa[i], acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) xa[i] = *reinterpret_cast<const bf16x8*>(lds + pb[i] + tn);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, xb[i], acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) xb[i] = *reinterpret_cast<const bf16x8*>(lds + pb[i] + tn + 32);
      __builtin_amdgcn_sched_barrier(0);
    }
    a0v = n0; a1v = n1;
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[(size_t)blockIdx.x * 256 + tid] = s;
}

int main() {
  const int iters = 2000;
  std::vector<unsigned short> hw(512 * 512 * 8 + 4096), himg(64 * 189 * 64);
  srand(1);
  auto rb = []() { float f = (rand() / (float)RAND_MAX - 0.5f); union { float f; unsigned u; } cv; cv.f = f; return (unsigned short)(cv.u >> 16); };
  for (auto& v : hw) v = rb();
  for (auto& v : himg) v = rb();
  u32x4 *w, *img; float* out;
  hipMalloc(&w, hw.size() * 2 + (1 << 22)); hipMalloc(&img, himg.size() * 2); hipMalloc(&out, (size_t)768 * 8 * 256 * 4);
  hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(img, himg.data(), himg.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int occ : {1, 2, 3}) {
    const size_t shmem = occ == 3 ? 189 * 144 + 256 : (occ == 2 ? 70 * 1024 : 120 * 1024);
    const int blocks = 256 * occ * 4;
    for (int rot : {0, 1, 0, 1}) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int l = 0; l < 10; ++l) {
          if (rot) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), shmem, 0, w, img, out, iters);
          else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), shmem, 0, w, img, out, iters);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      const double flop = 10.0 * blocks * 4 * iters * 8 * 2.0 * 32 * 32 * 16;
      printf("blocks/CU %d  %s: %.3f ms per launch, %.1f TFLOP/s\n", occ, rot ? "rotate" : "batch ", best / 10,
             flop / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}
