// bf16 helpers shared by the mixed-precision (bf16 NHWC) kernels. gfx950 only.
#pragma once
#include "cn_common.h"

typedef unsigned short bf16_t;  // storage type of a bfloat16 element in HBM / LDS
typedef short bf16x8 __attribute__((ext_vector_type(8)));  // MFMA A/B fragment: 8 bf16 = 4 VGPRs
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float cn_bf16_to_f32(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ float cn_bf16_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float cn_bf16_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }

// round-to-nearest-even, NaN-preserving (plain casts: hipcc emits v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ unsigned cn_pack_bf16(float lo, float hi) {
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 f = {lo, hi};
  const b2 r = __builtin_convertvector(f, b2);
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ bf16_t cn_f32_to_bf16(float v) {
  const __bf16 r = (__bf16)v;
  return __builtin_bit_cast(bf16_t, r);
}

// 8 consecutive bf16 (16 bytes) <-> 8 floats
__device__ __forceinline__ void cn_unpack8(const u32x4 v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = cn_bf16_lo(v[i]);
    f[2 * i + 1] = cn_bf16_hi(v[i]);
  }
}
__device__ __forceinline__ u32x4 cn_pack8(const float* f) {
  u32x4 o = {cn_pack_bf16(f[0], f[1]), cn_pack_bf16(f[2], f[3]), cn_pack_bf16(f[4], f[5]), cn_pack_bf16(f[6], f[7])};
  return o;
}
