// Streaming ops of the mixed-precision path on bf16 NHWC activations (gfx950, HBM/L2-bound): neighborhood attention
// core, bilinear resize, slice copies / adds, and the layout+precision converters at the edges of the bf16 region.
//
// Reference ops: natten NeighborhoodAttention2D core (nn/modules/convolution.py:341-350; natten 0.17.1 semantics
// restated in oracle/na2d_ref.py), F.interpolate(bilinear, align_corners=True) (nn/functional.py:72-81), torch.cat /
// residual adds (nn/modules/unet_parts.py:700-760).
#include "cn_bf16.h"

// ------------------------------------------------------------------------------------------------------------
// layout / precision converters
// ------------------------------------------------------------------------------------------------------------
// src f32 [B][C][HW] (batch stride sbs) -> dst bf16 [B*HW][Cpad] rows (ld), channels >= C zero-filled up to Cpad.
__global__ __launch_bounds__(256) void cn_f32nchw_to_bf16nhwc_kernel(const float* __restrict__ src, long sbs,
                                                                    bf16_t* __restrict__ dst, long ld, int C,
                                                                    int Cpad, int HW) {
  __shared__ float tile[64][65];
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * 64;
  for (int c0 = 0; c0 < Cpad; c0 += 64) {
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
      const int c = i >> 6, p = i & 63;
      tile[c][p] = (c0 + c < C && p0 + p < HW) ? src[b * sbs + (long)(c0 + c) * HW + p0 + p] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {
      const int p = i >> 5, c2 = (i & 31) * 2;
      if (p0 + p < HW && c0 + c2 < Cpad)
        *reinterpret_cast<unsigned*>(dst + ((long)b * HW + p0 + p) * ld + c0 + c2) =
            cn_pack_bf16(tile[c2][p], tile[c2 + 1][p]);
    }
  }
}

// src bf16 [B*HW][>=C] rows (ld) -> dst f32 [B][C][HW] (batch stride dbs), (+)=.
__global__ __launch_bounds__(256) void cn_bf16nhwc_to_f32nchw_kernel(const bf16_t* __restrict__ src, long ld,
                                                                    float* __restrict__ dst, long dbs, int C, int HW,
                                                                    int accumulate) {
  __shared__ float tile[64][65];
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * 64;
  for (int c0 = 0; c0 < C; c0 += 64) {
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
      const int p = i >> 6, c = i & 63;
      tile[c][p] = (c0 + c < C && p0 + p < HW) ? cn_bf16_to_f32(src[((long)b * HW + p0 + p) * ld + c0 + c]) : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
      const int c = i >> 6, p = i & 63;
      if (c0 + c < C && p0 + p < HW) {
        float* o = dst + b * dbs + (long)(c0 + c) * HW + p0 + p;
        *o = accumulate ? *o + tile[c][p] : tile[c][p];
      }
    }
  }
}

extern "C" int cn_convert_f32nchw_to_bf16nhwc(const float* src, long sbs, void* dst, long ld, int B, int C, int Cpad,
                                              int HW, void* stream) {
  if (B <= 0 || HW <= 0) return CN_OK;
  if (Cpad < C || (Cpad & 1) || ld < Cpad) return CN_ERR_ARG;
  CN_LAUNCH(cn_f32nchw_to_bf16nhwc_kernel, dim3((HW + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, src, sbs,
                     (bf16_t*)dst, ld, C, Cpad, HW);
  return cn_check_launch();
}

extern "C" int cn_convert_bf16nhwc_to_f32nchw(const void* src, long ld, float* dst, long dbs, int B, int C, int HW,
                                              int accumulate, void* stream) {
  if (B <= 0 || HW <= 0) return CN_OK;
  CN_LAUNCH(cn_bf16nhwc_to_f32nchw_kernel, dim3((HW + 63) / 64, B), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src, ld, dst, dbs, C, HW, accumulate);
  return cn_check_launch();
}

// ------------------------------------------------------------------------------------------------------------
// slice copy / add / fill on [P][C] rows (C % 8 == 0): dst (+)= src ; dst = a + c ; fill
// ------------------------------------------------------------------------------------------------------------
// IDX = int when the piece count fits 31 bits (one 32-bit division per piece instead of a 64-bit one); two pieces
// per iteration so that every lane has 2-4 independent 16-byte loads in flight.
template <typename IDX>
__global__ __launch_bounds__(256) void cn_bcopy_kernel(const bf16_t* __restrict__ a, long lda,
                                                      const bf16_t* __restrict__ c, long ldc, bf16_t* __restrict__ d,
                                                      long ldd, long P, int C8, int mode) {
  const IDX n = (IDX)(P * C8);
  const IDX stride = (IDX)gridDim.x * 256;
  auto one = [&](IDX i, const bf16_t*& pa, const bf16_t*& pc, bf16_t*& pd) {
    const IDX p = i / (IDX)C8;
    const int cg = (int)(i - p * (IDX)C8);
    pa = a + (long)p * lda + cg * 8;
    pd = d + (long)p * ldd + cg * 8;
    pc = mode == 1 ? pd : (mode == 2 ? c + (long)p * ldc + cg * 8 : nullptr);
  };
  auto fin = [&](const u32x4& ar, const u32x4& cr, bf16_t* pd) {
    if (mode != 0) {  // 1: d += a ; 2: d = a + c
      float av[8], cv[8];
      cn_unpack8(ar, av);
      cn_unpack8(cr, cv);
#pragma unroll
      for (int j = 0; j < 8; ++j) av[j] += cv[j];
      *reinterpret_cast<u32x4*>(pd) = cn_pack8(av);
    } else {
      *reinterpret_cast<u32x4*>(pd) = ar;
    }
  };
  const u32x4 z4 = {0u, 0u, 0u, 0u};
  IDX i = (IDX)blockIdx.x * 256 + threadIdx.x;
  for (; i + stride < n; i += 2 * stride) {
    const bf16_t *pa0, *pc0, *pa1, *pc1;
    bf16_t *pd0, *pd1;
    one(i, pa0, pc0, pd0);
    one(i + stride, pa1, pc1, pd1);
    const u32x4 a0 = *reinterpret_cast<const u32x4*>(pa0), a1 = *reinterpret_cast<const u32x4*>(pa1);
    const u32x4 c0 = mode != 0 ? *reinterpret_cast<const u32x4*>(pc0) : z4;
    const u32x4 c1 = mode != 0 ? *reinterpret_cast<const u32x4*>(pc1) : z4;
    fin(a0, c0, pd0);
    fin(a1, c1, pd1);
  }
  if (i < n) {
    const bf16_t *pa0, *pc0;
    bf16_t* pd0;
    one(i, pa0, pc0, pd0);
    const u32x4 a0 = *reinterpret_cast<const u32x4*>(pa0);
    const u32x4 c0 = mode != 0 ? *reinterpret_cast<const u32x4*>(pc0) : z4;
    fin(a0, c0, pd0);
  }
}

static void bcopy_launch(const bf16_t* a, long lda, const bf16_t* c, long ldc, bf16_t* d, long ldd, long P, int C8,
                         int mode, hipStream_t stream) {
  const long n = P * C8;
  long blocks = (n + 511) / 512;  // two pieces per thread and iteration
  if (blocks > 16384) blocks = 16384;
  if (blocks < 1) blocks = 1;
  if (n + 2 * blocks * 256 < (1L << 31))
    CN_LAUNCH(cn_bcopy_kernel<int>, dim3((unsigned)blocks), dim3(256), 0, stream, a, lda, c, ldc, d, ldd, P, C8,
                       mode);
  else
    CN_LAUNCH(cn_bcopy_kernel<long>, dim3((unsigned)blocks), dim3(256), 0, stream, a, lda, c, ldc, d, ldd, P,
                       C8, mode);
}

static inline unsigned bops_blocks(long n) {
  long b = (n + 255) / 256;
  return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

extern "C" int cn_copy_bf16(const void* src, long lds_, void* dst, long ldd, long P, int C, int accumulate,
                            void* stream) {
  if (P <= 0 || C <= 0) return CN_OK;
  if (C & 7) return CN_ERR_ARG;
  bcopy_launch((const bf16_t*)src, lds_, nullptr, 0L, (bf16_t*)dst, ldd, P, C >> 3, accumulate ? 1 : 0,
               (hipStream_t)stream);
  return cn_check_launch();
}

extern "C" int cn_add_bf16(const void* a, long lda, const void* c, long ldc, void* dst, long ldd, long P, int C,
                           void* stream) {
  if (P <= 0 || C <= 0) return CN_OK;
  if (C & 7) return CN_ERR_ARG;
  bcopy_launch((const bf16_t*)a, lda, (const bf16_t*)c, ldc, (bf16_t*)dst, ldd, P, C >> 3, 2, (hipStream_t)stream);
  return cn_check_launch();
}

__global__ __launch_bounds__(256) void cn_bfill_kernel(bf16_t* __restrict__ d, long ldd, long P, int C8) {
  const long n = P * C8;
  const u32x4 z = {0u, 0u, 0u, 0u};
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long p = i / C8;
    *reinterpret_cast<u32x4*>(d + p * ldd + (i - p * C8) * 8) = z;
  }
}

extern "C" int cn_zero_bf16(void* dst, long ldd, long P, int C, void* stream) {
  if (P <= 0 || C <= 0) return CN_OK;
  if (C & 7) return CN_ERR_ARG;
  CN_LAUNCH(cn_bfill_kernel, dim3(bops_blocks(P * (C >> 3))), dim3(256), 0, (hipStream_t)stream, (bf16_t*)dst,
                     ldd, P, C >> 3);
  return cn_check_launch();
}

// ------------------------------------------------------------------------------------------------------------
// bilinear resize, align_corners=True (ATen fp32 index math; weights fp32, data bf16)
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bbl_src(int o, float scale, int in_size, int& i0, int& i1, float& l1) {
#pragma clang fp contract(off)
  const float src = scale * (float)o;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = src - i0;
}

__global__ __launch_bounds__(256) void cn_bbilinear_fwd_kernel(const bf16_t* __restrict__ x, long ldx,
                                                              bf16_t* __restrict__ y, long ldy, int B, int C8, int Hi,
                                                              int Wi, int Ho, int Wo, float sh, float sw) {
  const long n = (long)B * Ho * Wo * C8;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long po = i / C8;
    const int cg = (int)(i - po * C8);
    const int ox = (int)(po % Wo);
    const long t = po / Wo;
    const int oy = (int)(t % Ho);
    const long b = t / Ho;
    int y0, y1, x0, x1;
    float ly, lx;
    bbl_src(oy, sh, Hi, y0, y1, ly);
    bbl_src(ox, sw, Wi, x0, x1, lx);
    const bf16_t* xb = x + (b * Hi * Wi) * ldx + cg * 8;
    float v00[8], v01[8], v10[8], v11[8], o[8];
    cn_unpack8(*reinterpret_cast<const u32x4*>(xb + ((long)y0 * Wi + x0) * ldx), v00);
    cn_unpack8(*reinterpret_cast<const u32x4*>(xb + ((long)y0 * Wi + x1) * ldx), v01);
    cn_unpack8(*reinterpret_cast<const u32x4*>(xb + ((long)y1 * Wi + x0) * ldx), v10);
    cn_unpack8(*reinterpret_cast<const u32x4*>(xb + ((long)y1 * Wi + x1) * ldx), v11);
    const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = hy * (hx * v00[j] + lx * v01[j]) + ly * (hx * v10[j] + lx * v11[j]);
    *reinterpret_cast<u32x4*>(y + po * ldy + cg * 8) = cn_pack8(o);
  }
}

// candidate output rows / columns reading input index i (<= 4 for resizes that shrink by less than 2x)
__device__ __forceinline__ int bbl_candidates(int i, int in_size, int out_size, float scale, float inv_scale, int* idx,
                                             float* wgt) {
  int lo = (int)floorf((i - 1) * inv_scale) - 1, hi = (int)ceilf((i + 1) * inv_scale) + 1;
  if (scale == 0.f) { lo = 0; hi = out_size - 1; }
  lo = max(lo, 0);
  hi = min(hi, out_size - 1);
  // The outputs reading input index i are CONSECUTIVE (the source coordinate is monotonic; a zero weight can only be
  // the first output of the run, whose source falls exactly on i - 1): the search only counts them and notes the first,
  // the <= 4 weights are then recomputed with static register indices. (Storing idx[n] / wgt[n] from inside the search
  // loop through an if-chain on n lost candidate 1 whenever a fourth one was found -- resizes growing by 1.5x..2x.)
  int n = 0, first = 0;
#pragma unroll 1
  for (int o = lo; o <= hi; ++o) {
    int i0, i1; float l1;
    bbl_src(o, scale, in_size, i0, i1, l1);
    float w = 0.f;
    if (i0 == i) w += 1.f - l1;
    if (i1 == i) w += l1;
    if (w != 0.f) {
      if (n == 0) first = o;
      ++n;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool live = k < n;
    const int o = live ? first + k : first;
    int i0, i1; float l1;
    bbl_src(o, scale, in_size, i0, i1, l1);
    float w = 0.f;
    if (i0 == i) w += 1.f - l1;
    if (i1 == i) w += l1;
    idx[k] = o;
    wgt[k] = live ? w : 0.f;
  }
  return n;
}

// adjoint in gather form (deterministic): every input pixel sums the output pixels that read it
__global__ __launch_bounds__(256) void cn_bbilinear_bwd_kernel(const bf16_t* __restrict__ dy, long lddy,
                                                              bf16_t* __restrict__ dx, long lddx, int B, int C8,
                                                              int Hi, int Wi, int Ho, int Wo, float sh, float sw,
                                                              float inv_sh, float inv_sw, int accumulate) {
  const long n = (long)B * Hi * Wi * C8;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long pi = i / C8;
    const int cg = (int)(i - pi * C8);
    const int ix = (int)(pi % Wi);
    const long t = pi / Wi;
    const int iy = (int)(t % Hi);
    const long b = t / Hi;
    int oyv[4] = {0, 0, 0, 0}, oxv[4] = {0, 0, 0, 0};
    float wyv[4] = {0.f, 0.f, 0.f, 0.f}, wxv[4] = {0.f, 0.f, 0.f, 0.f};
    const int ny = bbl_candidates(iy, Hi, Ho, sh, inv_sh, oyv, wyv);
    const int nx = bbl_candidates(ix, Wi, Wo, sw, inv_sw, oxv, wxv);
    const bf16_t* db = dy + (b * Ho * Wo) * lddy + cg * 8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (ny <= 4 && nx <= 4) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          const float w = wyv[k] * wxv[l];
          if (w != 0.f) {
            float v[8];
            cn_unpack8(*reinterpret_cast<const u32x4*>(db + ((long)oyv[k] * Wo + oxv[l]) * lddy), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += w * v[j];
          }
        }
    } else {  // general (strong down-scaling): scan the candidate window
      int oy_lo = (int)floorf((iy - 1) * inv_sh) - 1, oy_hi = (int)ceilf((iy + 1) * inv_sh) + 1;
      int ox_lo = (int)floorf((ix - 1) * inv_sw) - 1, ox_hi = (int)ceilf((ix + 1) * inv_sw) + 1;
      if (sh == 0.f) { oy_lo = 0; oy_hi = Ho - 1; }
      if (sw == 0.f) { ox_lo = 0; ox_hi = Wo - 1; }
      oy_lo = max(oy_lo, 0); oy_hi = min(oy_hi, Ho - 1);
      ox_lo = max(ox_lo, 0); ox_hi = min(ox_hi, Wo - 1);
      for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        int y0, y1; float ly;
        bbl_src(oy, sh, Hi, y0, y1, ly);
        float wy = 0.f;
        if (y0 == iy) wy += 1.f - ly;
        if (y1 == iy) wy += ly;
        if (wy == 0.f) continue;
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
          int x0, x1; float lx;
          bbl_src(ox, sw, Wi, x0, x1, lx);
          float wx = 0.f;
          if (x0 == ix) wx += 1.f - lx;
          if (x1 == ix) wx += lx;
          if (wx == 0.f) continue;
          float v[8];
          cn_unpack8(*reinterpret_cast<const u32x4*>(db + ((long)oy * Wo + ox) * lddy), v);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += wy * wx * v[j];
        }
      }
    }
    bf16_t* o = dx + pi * lddx + cg * 8;
    if (accumulate) {
      float ov[8];
      cn_unpack8(*reinterpret_cast<const u32x4*>(o), ov);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += ov[j];
    }
    *reinterpret_cast<u32x4*>(o) = cn_pack8(acc);
  }
}

// Row-wise variants (the ones the launchers pick): one block per (image, row), 32-bit index math only, and for the
// adjoint the candidate lists of the row (computed once) and of every column (LDS table built by the first Wi threads)
// instead of two scalar search loops per 16-byte piece. Same candidate order and the same fp32 expressions as the
// element-wise kernels above, so results are identical to them.
#define BBL_MAXC 12   // candidates per axis held in the tables (up-scaling by up to ~5x); more => element-wise kernel
#define BBL_MAXW 256  // widest input row the column table covers (25 KB of LDS)

__global__ __launch_bounds__(256) void cn_bbilinear_fwd_rows_kernel(const bf16_t* __restrict__ x, long ldx,
                                                                   bf16_t* __restrict__ y, long ldy, int C8, int Hi,
                                                                   int Wi, int Ho, int Wo, float sh, float sw) {
  const int oy = blockIdx.x, b = blockIdx.y;
  int y0, y1;
  float ly;
  bbl_src(oy, sh, Hi, y0, y1, ly);
  const float hy = 1.f - ly;
  const bf16_t* r0 = x + ((long)b * Hi + y0) * Wi * ldx;
  const bf16_t* r1 = x + ((long)b * Hi + y1) * Wi * ldx;
  bf16_t* yo = y + ((long)b * Ho + oy) * Wo * ldy;
  const int n = Wo * C8;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int ox = i / C8, cg = i - ox * C8;
    int x0, x1;
    float lx;
    bbl_src(ox, sw, Wi, x0, x1, lx);
    float v00[8], v01[8], v10[8], v11[8], o[8];
    cn_unpack8(*reinterpret_cast<const u32x4*>(r0 + (long)x0 * ldx + cg * 8), v00);
    cn_unpack8(*reinterpret_cast<const u32x4*>(r0 + (long)x1 * ldx + cg * 8), v01);
    cn_unpack8(*reinterpret_cast<const u32x4*>(r1 + (long)x0 * ldx + cg * 8), v10);
    cn_unpack8(*reinterpret_cast<const u32x4*>(r1 + (long)x1 * ldx + cg * 8), v11);
    const float hx = 1.f - lx;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = hy * (hx * v00[j] + lx * v01[j]) + ly * (hx * v10[j] + lx * v11[j]);
    *reinterpret_cast<u32x4*>(yo + (long)ox * ldy + cg * 8) = cn_pack8(o);
  }
}

// all output indices reading input index i, in increasing order; returns the count (may exceed BBL_MAXC: caller checks)
__device__ __forceinline__ int bbl_candidates_n(int i, int in_size, int out_size, float scale, float inv_scale,
                                                int* idx, float* wgt) {
  int lo = (int)floorf((i - 1) * inv_scale) - 1, hi = (int)ceilf((i + 1) * inv_scale) + 1;
  if (scale == 0.f) { lo = 0; hi = out_size - 1; }
  lo = max(lo, 0);
  hi = min(hi, out_size - 1);
  int n = 0;
#pragma unroll 1
  for (int o = lo; o <= hi; ++o) {
    int i0, i1; float l1;
    bbl_src(o, scale, in_size, i0, i1, l1);
    float w = 0.f;
    if (i0 == i) w += 1.f - l1;
    if (i1 == i) w += l1;
    if (w != 0.f) {
      if (n < BBL_MAXC) { idx[n] = o; wgt[n] = w; }
      ++n;
    }
  }
  return n;
}

__global__ __launch_bounds__(256) void cn_bbilinear_bwd_rows_kernel(const bf16_t* __restrict__ dy, long lddy,
                                                                   bf16_t* __restrict__ dx, long lddx, int C8, int Hi,
                                                                   int Wi, int Ho, int Wo, float sh, float sw,
                                                                   float inv_sh, float inv_sw, int accumulate) {
  __shared__ int xn[BBL_MAXW];
  __shared__ int xi[BBL_MAXW * BBL_MAXC];
  __shared__ float xw[BBL_MAXW * BBL_MAXC];
  __shared__ int yn, yi[BBL_MAXC];
  __shared__ float yw[BBL_MAXC];
  const int iy = blockIdx.x, b = blockIdx.y;
  for (int ix = threadIdx.x; ix < Wi; ix += 256)
    xn[ix] = bbl_candidates_n(ix, Wi, Wo, sw, inv_sw, xi + ix * BBL_MAXC, xw + ix * BBL_MAXC);
  if (threadIdx.x == 255) yn = bbl_candidates_n(iy, Hi, Ho, sh, inv_sh, yi, yw);
  __syncthreads();
  const int ny = min(yn, BBL_MAXC);
  const bf16_t* db = dy + (long)b * Ho * Wo * lddy;
  bf16_t* dxr = dx + ((long)b * Hi + iy) * Wi * lddx;
  const int n = Wi * C8;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int ix = i / C8, cg = i - ix * C8;
    const int nx = min(xn[ix], BBL_MAXC);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (nx > 0) {
      // The first three column candidates of a row are loaded together (a serial load -> FMA chain per candidate left
      // one L2 round trip exposed per 16 bytes); absent ones alias candidate 0 with weight 0, which leaves the sum
      // unchanged. Resizes by less than 2x have at most three candidates per axis.
      const int* xip = xi + ix * BBL_MAXC;
      const float* xwp = xw + ix * BBL_MAXC;
      const long o0 = (long)xip[0] * lddy;
      const long o1 = nx > 1 ? (long)xip[1] * lddy : o0;
      const long o2 = nx > 2 ? (long)xip[2] * lddy : o0;
      const float wx0 = xwp[0], wx1 = nx > 1 ? xwp[1] : 0.f, wx2 = nx > 2 ? xwp[2] : 0.f;
#pragma unroll 1
      for (int k = 0; k < ny; ++k) {
        const bf16_t* row = db + (long)yi[k] * Wo * lddy + cg * 8;
        const float wy = yw[k];
        const u32x4 r0 = *reinterpret_cast<const u32x4*>(row + o0);
        const u32x4 r1 = *reinterpret_cast<const u32x4*>(row + o1);
        const u32x4 r2 = *reinterpret_cast<const u32x4*>(row + o2);
        float v[8];
        const float w0 = wy * wx0, w1 = wy * wx1, w2 = wy * wx2;
        cn_unpack8(r0, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += w0 * v[j];
        cn_unpack8(r1, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += w1 * v[j];
        cn_unpack8(r2, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += w2 * v[j];
#pragma unroll 1
        for (int l = 3; l < nx; ++l) {
          const float w = wy * xwp[l];
          cn_unpack8(*reinterpret_cast<const u32x4*>(row + (long)xip[l] * lddy), v);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += w * v[j];
        }
      }
    }
    bf16_t* o = dxr + (long)ix * lddx + cg * 8;
    if (accumulate) {
      float ov[8];
      cn_unpack8(*reinterpret_cast<const u32x4*>(o), ov);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += ov[j];
    }
    *reinterpret_cast<u32x4*>(o) = cn_pack8(acc);
  }
}

// largest number of output indices that read one input index (host restatement of the device search)
static int bbl_max_candidates(int in_size, int out_size) {
  if (out_size <= 1) return out_size;
  if (in_size <= 1) return out_size;
  // output o reads floor(o * scale) and its successor: an input index is read by the outputs of two consecutive
  // unit intervals of o * scale
  const double scale = (double)(in_size - 1) / (double)(out_size - 1);
  return (int)(2.0 / scale) + 2;
}

static inline float bbl_scale(int in_size, int out_size) {
  return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
}

extern "C" int cn_bilinear_fwd_bf16(const void* x, long ldx, void* y, long ldy, int B, int C, int Hi, int Wi, int Ho,
                                    int Wo, void* stream) {
  if (B <= 0 || C <= 0) return CN_OK;
  if (C & 7) return CN_ERR_ARG;
  if (Ho <= 0 || Wo <= 0 || Hi <= 0 || Wi <= 0) return CN_OK;
  if (B <= 65535 && (long)Wo * (C >> 3) < (1L << 30))
    CN_LAUNCH(cn_bbilinear_fwd_rows_kernel, dim3(Ho, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       ldx, (bf16_t*)y, ldy, C >> 3, Hi, Wi, Ho, Wo, bbl_scale(Hi, Ho), bbl_scale(Wi, Wo));
  else
    CN_LAUNCH(cn_bbilinear_fwd_kernel, dim3(bops_blocks((long)B * Ho * Wo * (C >> 3))), dim3(256), 0,
                       (hipStream_t)stream, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, B, C >> 3, Hi, Wi, Ho, Wo,
                       bbl_scale(Hi, Ho), bbl_scale(Wi, Wo));
  return cn_check_launch();
}

extern "C" int cn_bilinear_bwd_bf16(const void* dy, long lddy, void* dx, long lddx, int B, int C, int Hi, int Wi,
                                    int Ho, int Wo, int accumulate, void* stream) {
  if (B <= 0 || C <= 0) return CN_OK;
  if (C & 7) return CN_ERR_ARG;
  const float sh = bbl_scale(Hi, Ho), sw = bbl_scale(Wi, Wo);
  if (Ho <= 0 || Wo <= 0 || Hi <= 0 || Wi <= 0) return CN_OK;
  const float ish = sh > 0.f ? 1.f / sh : 0.f, isw = sw > 0.f ? 1.f / sw : 0.f;
  if (B <= 65535 && Wi <= BBL_MAXW && (long)Wi * (C >> 3) < (1L << 30) && bbl_max_candidates(Hi, Ho) <= BBL_MAXC &&
      bbl_max_candidates(Wi, Wo) <= BBL_MAXC)
    CN_LAUNCH(cn_bbilinear_bwd_rows_kernel, dim3(Hi, B), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, C >> 3, Hi, Wi, Ho, Wo, sh, sw, ish, isw, accumulate);
  else
    CN_LAUNCH(cn_bbilinear_bwd_kernel, dim3(bops_blocks((long)B * Hi * Wi * (C >> 3))), dim3(256), 0,
                       (hipStream_t)stream, (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, B, C >> 3, Hi, Wi, Ho, Wo, sh,
                       sw, ish, isw, accumulate);
  return cn_check_launch();
}

// ------------------------------------------------------------------------------------------------------------
// neighborhood attention core: qkv bf16 [B][H][W][3C] (channel = which*C + head*D + d), out bf16 [B][H][W][C];
// attn / dattn fp32 [B][heads][9][H][W] (saved probabilities / dS scratch). One lane per (pixel, head), head fastest.
// ------------------------------------------------------------------------------------------------------------
#define NAB_K 3
#define NAB_KK 9

__device__ __forceinline__ int nab_window_start(int i, int len, int dil) {
  if (dil <= 1) return max(i - 1, 0) + ((i + 1 >= len) ? (len - i - 2) : 0);
  const int ni = i - dil;
  if (ni < 0) return i % dil;
  if (i + dil >= len) {
    const int imodd = i % dil;
    const int a = (len / dil) * dil;
    const int b = len - a;
    if (imodd < b) return len - b + imodd - 2 * dil;
    return a + imodd - NAB_K * dil;
  }
  return ni;
}

// attn_drop (nn.Dropout on the soft-maxed logits): keep/(1-p) factor of tap t from the same counter hash as the fp32
// kernels (cn_na2d.hip na_keep), recomputed in forward and backward; 1.0 when dropout is off.
__device__ __forceinline__ unsigned long long nab_splitmix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ float nab_keep(unsigned long long thresh, float scale, unsigned long long seed, long bh,
                                          int t, int HW, int p) {
  if (thresh == 0ull) return 1.0f;
  const unsigned long long i = ((unsigned long long)bh * NAB_KK + t) * (unsigned long long)HW + p;
  return nab_splitmix64(seed + i) >= thresh ? scale : 0.f;
}
static inline unsigned long long nab_thresh(float p) {
  if (!(p > 0.f)) return 0ull;
  const double t = (double)p * 18446744073709551616.0;  // p * 2^64
  return t >= 18446744073709551615.0 ? ~0ull : (unsigned long long)t;
}

// ---- dropout on bf16 NHWC activations ---------------------------------------------------------------------------
// nn.Dropout2d after the encoder blocks (convolution.py:495,511) and natten's proj_drop, on the mixed-precision path.
// The SAME counter-based masks as cn_dropout_f32 (cn_pointwise.hip): channelwise: one decision per (b, c) from
// splitmix64(seed + b*C + c); elementwise: splitmix64(seed + (b*C + c)*HW + pixel). y = x * keep / (1 - p); the same
// entry point serves backward (x := dy, accumulate into dx).
__global__ __launch_bounds__(256) void cn_bdropout_kernel(const bf16_t* __restrict__ x, long ldx, bf16_t* __restrict__ y,
                                                         long ldy, long P, int C, int HW, unsigned long long thresh,
                                                         float scale, unsigned long long seed_, const unsigned long long* __restrict__ step, int channelwise,
                                                         int accumulate) {
  const unsigned long long seed = cn_step_seed(seed_, step);
  const int groups = C >> 3;
  const long n = P * groups;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
    const long row = i / groups;
    const int c0 = (int)(i - row * groups) << 3;
    const long b = row / HW;
    const int pix = (int)(row - b * HW);
    float v[8];
    cn_unpack8(*reinterpret_cast<const u32x4*>(x + row * ldx + c0), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned long long plane = (unsigned long long)b * C + c0 + j;
      const unsigned long long ctr = channelwise ? plane : plane * (unsigned long long)HW + pix;
      v[j] *= nab_splitmix64(seed + ctr) >= thresh ? scale : 0.f;
    }
    bf16_t* yp = y + row * ldy + c0;
    if (accumulate) {
      float o[8];
      cn_unpack8(*reinterpret_cast<const u32x4*>(yp), o);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += o[j];
    }
    *reinterpret_cast<u32x4*>(yp) = cn_pack8(v);
  }
}

extern "C" int cn_dropout_bf16(const void* x, long ldx, void* y, long ldy, int B, int C, int HW, float p,
                               unsigned long long seed, const unsigned long long* step, int channelwise, int accumulate,
                               void* stream) {
  if (B <= 0 || C <= 0 || HW <= 0) return CN_OK;
  if (!(p >= 0.f && p < 1.f) || (C & 7)) return CN_ERR_ARG;
  const long P = (long)B * HW;
  const long n = P * (C >> 3);
  long blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  CN_LAUNCH(cn_bdropout_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx,
            (bf16_t*)y, ldy, P, C, HW, nab_thresh(p), 1.0f / (1.0f - p), seed, step, channelwise, accumulate);
  return cn_check_launch();
}

template <int D>
__device__ __forceinline__ void nab_load(const bf16_t* p, float* f) {
  if (D == 4) {
    const u32x2 v = *reinterpret_cast<const u32x2*>(p);
    f[0] = cn_bf16_lo(v[0]); f[1] = cn_bf16_hi(v[0]); f[2] = cn_bf16_lo(v[1]); f[3] = cn_bf16_hi(v[1]);
    return;
  }
#pragma unroll
  for (int i = 0; i < D / 8; ++i) cn_unpack8(*reinterpret_cast<const u32x4*>(p + i * 8), f + i * 8);
}
template <int D>
__device__ __forceinline__ void nab_store(bf16_t* p, const float* f) {
  if (D == 4) {
    const u32x2 v = {cn_pack_bf16(f[0], f[1]), cn_pack_bf16(f[2], f[3])};
    *reinterpret_cast<u32x2*>(p) = v;
    return;
  }
#pragma unroll
  for (int i = 0; i < D / 8; ++i) *reinterpret_cast<u32x4*>(p + i * 8) = cn_pack8(f + i * 8);
}

template <int D>
__global__ __launch_bounds__(256) void cn_bna_fwd_kernel(const bf16_t* __restrict__ qkv, long ldq,
                                                        bf16_t* __restrict__ out, long ldo, float* __restrict__ attn,
                                                        int B, int C, int heads, int H, int W, int dil, float scale,
                                                        unsigned long long dthresh, float dscale,
                                                        unsigned long long dseed_, const unsigned long long* __restrict__ dstep) {
  const unsigned long long dseed = cn_step_seed(dseed_, dstep);
  const int HW = H * W;
  const long n = (long)B * HW * heads;
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= n) return;
  const int h = (int)(i % heads);
  const long bp = i / heads;
  const int p = (int)(bp % HW);
  const int b = (int)(bp / HW);
  const int y = p / W, x = p - y * W;
  const int sy = nab_window_start(y, H, dil), sx = nab_window_start(x, W, dil);
  const bf16_t* base = qkv + ((long)b * HW) * ldq + h * D;
  float q[D];
  nab_load<D>(base + (long)p * ldq, q);
  float lg[NAB_KK];
#pragma unroll
  for (int t = 0; t < NAB_KK; ++t) {
    const int kp = (sy + (t / 3) * dil) * W + sx + (t % 3) * dil;
    float kv[D];
    nab_load<D>(base + (long)kp * ldq + C, kv);
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) s += q[d] * kv[d];
    lg[t] = s * scale;
  }
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NAB_KK; ++t) mx = fmaxf(mx, lg[t]);
  float den = 0.f;
#pragma unroll
  for (int t = 0; t < NAB_KK; ++t) { lg[t] = expf(lg[t] - mx); den += lg[t]; }
  const float inv = 1.0f / den;
  float* ap = attn + ((long)(b * heads + h) * NAB_KK) * HW + p;
  float o[D];
#pragma unroll
  for (int d = 0; d < D; ++d) o[d] = 0.f;
#pragma unroll
  for (int t = 0; t < NAB_KK; ++t) {
    lg[t] *= inv;
    if (attn != nullptr) ap[(long)t * HW] = lg[t];
    lg[t] *= nab_keep(dthresh, dscale, dseed, (long)b * heads + h, t, HW, p);
    const int kp = (sy + (t / 3) * dil) * W + sx + (t % 3) * dil;
    float vv[D];
    nab_load<D>(base + (long)kp * ldq + 2 * C, vv);
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] += lg[t] * vv[d];
  }
  nab_store<D>(out + ((long)b * HW + p) * ldo + h * D, o);
}

// query side: dP = dOut.v ; dS = P*(dP - sum P dP) (saved to dattn) ; dq = scale * sum dS*k
template <int D>
__global__ __launch_bounds__(256) void cn_bna_bwd_q_kernel(const bf16_t* __restrict__ qkv, long ldq,
                                                          const bf16_t* __restrict__ dout, long ldo,
                                                          const float* __restrict__ attn, float* __restrict__ dattn,
                                                          bf16_t* __restrict__ dqkv, long lddq, int B, int C,
                                                          int heads, int H, int W, int dil, float scale,
                                                          unsigned long long dthresh, float dscale,
                                                          unsigned long long dseed_, const unsigned long long* __restrict__ dstep) {
  const unsigned long long dseed = cn_step_seed(dseed_, dstep);
  const int HW = H * W;
  const long n = (long)B * HW * heads;
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= n) return;
  const int h = (int)(i % heads);
  const long bp = i / heads;
  const int p = (int)(bp % HW);
  const int b = (int)(bp / HW);
  const int y = p / W, x = p - y * W;
  const int sy = nab_window_start(y, H, dil), sx = nab_window_start(x, W, dil);
  const bf16_t* base = qkv + ((long)b * HW) * ldq + h * D;
  float g[D];
  nab_load<D>(dout + ((long)b * HW + p) * ldo + h * D, g);
  const float* ap = attn + ((long)(b * heads + h) * NAB_KK) * HW + p;
  float dp[NAB_KK], pr[NAB_KK];
  float dot = 0.f;
#pragma unroll
  for (int t = 0; t < NAB_KK; ++t) {
    const int kp = (sy + (t / 3) * dil) * W + sx + (t % 3) * dil;
    float vv[D];
    nab_load<D>(base + (long)kp * ldq + 2 * C, vv);
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) s += g[d] * vv[d];
    s *= nab_keep(dthresh, dscale, dseed, (long)b * heads + h, t, HW, p);
    dp[t] = s;
    pr[t] = ap[(long)t * HW];
    dot += pr[t] * s;
  }
  float* dap = dattn + ((long)(b * heads + h) * NAB_KK) * HW + p;
  float dq[D];
#pragma unroll
  for (int d = 0; d < D; ++d) dq[d] = 0.f;
#pragma unroll
  for (int t = 0; t < NAB_KK; ++t) {
    const float ds = pr[t] * (dp[t] - dot);
    dap[(long)t * HW] = ds;
    const int kp = (sy + (t / 3) * dil) * W + sx + (t % 3) * dil;
    float kv[D];
    nab_load<D>(base + (long)kp * ldq + C, kv);
#pragma unroll
    for (int d = 0; d < D; ++d) dq[d] += ds * kv[d];
  }
#pragma unroll
  for (int d = 0; d < D; ++d) dq[d] *= scale;
  nab_store<D>(dqkv + ((long)b * HW + p) * lddq + h * D, dq);
}

// key side, gather form (deterministic): visit every query whose window contains this key pixel
template <int D>
__global__ __launch_bounds__(256) void cn_bna_bwd_kv_kernel(const bf16_t* __restrict__ qkv, long ldq,
                                                           const bf16_t* __restrict__ dout, long ldo,
                                                           const float* __restrict__ attn,
                                                           const float* __restrict__ dattn,
                                                           bf16_t* __restrict__ dqkv, long lddq, int B, int C,
                                                           int heads, int H, int W, int dil, float scale,
                                                           unsigned long long dthresh, float dscale,
                                                           unsigned long long dseed_, const unsigned long long* __restrict__ dstep) {
  const unsigned long long dseed = cn_step_seed(dseed_, dstep);
  const int HW = H * W;
  const long n = (long)B * HW * heads;
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= n) return;
  const int h = (int)(i % heads);
  const long bp = i / heads;
  const int p = (int)(bp % HW);
  const int b = (int)(bp / HW);
  const int y = p / W, x = p - y * W;
  const bf16_t* qb = qkv + ((long)b * HW) * ldq + h * D;
  const bf16_t* gb = dout + ((long)b * HW) * ldo + h * D;
  const float* ap = attn + ((long)(b * heads + h) * NAB_KK) * HW;
  const float* dap = dattn + ((long)(b * heads + h) * NAB_KK) * HW;
  float dk[D], dv[D];
#pragma unroll
  for (int d = 0; d < D; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
  for (int my = -2; my <= 2; ++my) {
    const int qy = y + my * dil;
    if (qy < 0 || qy >= H) continue;
    const int offy = y - nab_window_start(qy, H, dil);
    if (offy < 0 || offy > 2 * dil) continue;
    const int ti = offy / dil;
    for (int mx = -2; mx <= 2; ++mx) {
      const int qx = x + mx * dil;
      if (qx < 0 || qx >= W) continue;
      const int offx = x - nab_window_start(qx, W, dil);
      if (offx < 0 || offx > 2 * dil) continue;
      const int t = ti * NAB_K + offx / dil;
      const int qpix = qy * W + qx;
      const float ds = dap[(long)t * HW + qpix];
      const float pr = ap[(long)t * HW + qpix] * nab_keep(dthresh, dscale, dseed, (long)b * heads + h, t, HW, qpix);
      float qv[D], gv[D];
      nab_load<D>(qb + (long)qpix * ldq, qv);
      nab_load<D>(gb + (long)qpix * ldo, gv);
#pragma unroll
      for (int d = 0; d < D; ++d) { dk[d] += ds * qv[d]; dv[d] += pr * gv[d]; }
    }
  }
#pragma unroll
  for (int d = 0; d < D; ++d) dk[d] *= scale;
  bf16_t* o = dqkv + ((long)b * HW + p) * lddq + h * D;
  nab_store<D>(o + C, dk);
  nab_store<D>(o + 2 * C, dv);
}

#define NAB_DISPATCH(D_, KERNEL, ...)                                                          \
  switch (D_) {                                                                                \
    case 4: CN_LAUNCH((KERNEL<4>), grid, dim3(256), 0, stream, __VA_ARGS__); break;  \
    case 8: CN_LAUNCH((KERNEL<8>), grid, dim3(256), 0, stream, __VA_ARGS__); break;  \
    case 16: CN_LAUNCH((KERNEL<16>), grid, dim3(256), 0, stream, __VA_ARGS__); break; \
    case 32: CN_LAUNCH((KERNEL<32>), grid, dim3(256), 0, stream, __VA_ARGS__); break; \
    case 64: CN_LAUNCH((KERNEL<64>), grid, dim3(256), 0, stream, __VA_ARGS__); break; \
    default: return CN_ERR_ARG;                                                                \
  }

extern "C" int cn_na2d_fwd_bf16(const void* qkv, long ldq, void* out, long ldo, float* attn, int B, int C, int heads,
                                int H, int W, int kernel_size, int dilation, float attn_drop, unsigned long long seed,
                                const unsigned long long* step, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (kernel_size != NAB_K || heads <= 0 || C % heads != 0) return CN_ERR_ARG;
  if (kernel_size * dilation > H || kernel_size * dilation > W) return CN_ERR_ARG;
  const int D = C / heads;
  const float scale = 1.0f / sqrtf((float)D);
  const long n = (long)B * H * W * heads;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (!(attn_drop >= 0.f && attn_drop < 1.f)) return CN_ERR_ARG;
  NAB_DISPATCH(D, cn_bna_fwd_kernel, (const bf16_t*)qkv, ldq, (bf16_t*)out, ldo, attn, B, C, heads, H, W, dilation,
               scale, nab_thresh(attn_drop), 1.0f / (1.0f - attn_drop), seed, step);
  return cn_check_launch();
}

// dqkv bf16 [B][H][W][3C] fully overwritten; dattn: scratch of attn's size.
extern "C" int cn_na2d_bwd_bf16(const void* qkv, long ldq, const void* dout, long ldo, const float* attn, float* dattn,
                                void* dqkv, long lddq, int B, int C, int heads, int H, int W, int kernel_size,
                                int dilation, float attn_drop, unsigned long long seed, const unsigned long long* step,
                                void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (kernel_size != NAB_K || heads <= 0 || C % heads != 0) return CN_ERR_ARG;
  const int D = C / heads;
  const float scale = 1.0f / sqrtf((float)D);
  const long n = (long)B * H * W * heads;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (!(attn_drop >= 0.f && attn_drop < 1.f)) return CN_ERR_ARG;
  const unsigned long long th = nab_thresh(attn_drop);
  const float ds = 1.0f / (1.0f - attn_drop);
  NAB_DISPATCH(D, cn_bna_bwd_q_kernel, (const bf16_t*)qkv, ldq, (const bf16_t*)dout, ldo, attn, dattn, (bf16_t*)dqkv,
               lddq, B, C, heads, H, W, dilation, scale, th, ds, seed, step);
  NAB_DISPATCH(D, cn_bna_bwd_kv_kernel, (const bf16_t*)qkv, ldq, (const bf16_t*)dout, ldo, attn, dattn, (bf16_t*)dqkv,
               lddq, B, C, heads, H, W, dilation, scale, th, ds, seed, step);
  return cn_check_launch();
}
