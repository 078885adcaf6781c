// Micro-benchmark: does v_mfma_f32_16x16x32_bf16 sustain more FLOP/s than v_mfma_f32_32x32x16_bf16 inside a loop shaped like
// cn_bconv_kernel's step loop (per 32-k step and wave: 8 ds_read_b128 pixel fragments from a padded LDS image, 2 weight
// fragments straight from global memory, 8 x 32x32x16 or 16 x 16x16x32 MFMAs; 3 blocks of 4 waves per CU; random data)?
// MI355X_MICROARCH.md 'DVFS give-back' (7) reports 1.12-1.15x for bare loops: the chip holds a higher clock on the smaller
// shape. Build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape_bench mfma_shape_bench.hip ; run: ./mfma_shape_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256, 3) void k(const u32x4* __restrict__ w, const u32x4* __restrict__ img, float* out,
                                           int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // stage a 189-pixel x 64-channel image, pitch 144 B (as the conv kernel)
  for (int q = tid; q < 189 * 8; q += 256) {
    const int p = q >> 3, c = q & 7;
    *reinterpret_cast<u32x4*>(lds + p * 144 + c * 16) = img[(blockIdx.x % 64) * 189 * 8 + q];
  }
  __syncthreads();
  f32x16 acc32[4];
  f32x4 acc16[16];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) acc16[i][j] = 0.f;
  const u32x4* wp = w + (size_t)wid * 64 + lane;
  for (int it = 0; it < iters; ++it) {
    const int tap = it % 9;
    const int toff = ((tap / 3) * 27 + (tap % 3)) * 144 + ((it / 9) & 1) * 64;
    const u32x4 a0v = wp[(size_t)(it % 512) * 512];
    const u32x4 a1v = wp[(size_t)(it % 512) * 512 + 256];
    const bf16x8 a0 = __builtin_bit_cast(bf16x8, a0v), a1 = __builtin_bit_cast(bf16x8, a1v);
    if (SHAPE == 32) {
      const int r = lane & 31, h = lane >> 5;
      bf16x8 x[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int base = ((i * 32 + r) % 125 / 25 * 27 + (i * 32 + r) % 25) * 144 + h * 16 + toff;
        x[i] = *reinterpret_cast<const bf16x8*>(lds + base);
        x[4 + i] = *reinterpret_cast<const bf16x8*>(lds + base + 32);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) acc32[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, x[i], acc32[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc32[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, x[4 + i], acc32[i], 0, 0, 0);
    } else {
      const int r = lane & 15, kb = lane >> 4;
      bf16x8 x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int m = i * 16 + r;
        const int base = (m % 125 / 25 * 27 + m % 25) * 144 + kb * 16 + toff;
        x[i] = *reinterpret_cast<const bf16x8*>(lds + base);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc16[2 * i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, x[i], acc16[2 * i], 0, 0, 0);
        acc16[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, x[i], acc16[2 * i + 1], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc32[i][j];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) s += acc16[i][j];
  out[(size_t)blockIdx.x * 256 + tid] = s;
}

int main() {
  const int blocks = 768 * 4, iters = 2000;
  std::vector<unsigned short> hw(512 * 512 * 8 + 4096), himg(64 * 189 * 64);
  srand(1);
  auto rb = []() { float f = (rand() / (float)RAND_MAX - 0.5f); union { float f; unsigned u; } cv; cv.f = f; return (unsigned short)(cv.u >> 16); };
  for (auto& v : hw) v = rb();
  for (auto& v : himg) v = rb();
  u32x4 *w, *img; float* out;
  hipMalloc(&w, hw.size() * 2 + (1 << 22)); hipMalloc(&img, himg.size() * 2); hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(img, himg.data(), himg.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t shmem = 189 * 144 + 256;
  for (int shape : {32, 16, 32, 16}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int l = 0; l < 20; ++l) {
        if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(256), shmem, 0, w, img, out, iters);
        else hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), shmem, 0, w, img, out, iters);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flop = 20.0 * blocks * 4 * iters * 8 * 2.0 * 32 * 32 * 16;
      printf("shape %dx%d: %.3f ms per launch, %.1f TFLOP/s\n", shape, shape, ms / 20, flop / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
