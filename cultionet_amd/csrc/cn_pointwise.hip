// Streaming kernels: bilinear resize (align_corners=True), slice copy / add, the fused
// final-combine heads and label preparation. All HBM-bound (gfx950).
//
// Reference ops: F.interpolate(mode="bilinear", align_corners=True) in check_upsample
// (/root/reference/src/cultionet/nn/functional.py:72-81), torch.cat in TowerUNetBlock /
// TowerUNetFinal (nn/modules/unet_parts.py:281-309,700-760), TowerUNetFinalCombine + SigmoidCrisp
// (unet_parts.py:43-193).
#include "cn_common.h"

// ---------------------------------------------------------------------------
// Bilinear resize, align_corners=True. Planes = B*C, both tensors [B][C][H][W] with batch strides.
// src index math follows ATen's area_pixel_compute_source_index (fp32).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void bl_src(int o, float scale, int in_size, int& i0, int& i1, float& l1) {
#pragma clang fp contract(off)  // ATen rounds scale*o before subtracting floor(): an fma here shifts lambda by ~1e-6
  const float src = scale * (float)o;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = src - i0;
}

// The source may live on a larger stored grid Hp x Wp (row pitch Wp, plane Hp*Wp) of which [0,Hi) x [0,Wi) is the image:
// the output_padding grid the engine's ConvTranspose2d writes (cn_conv_transpose2d_fwd_f32).
__global__ __launch_bounds__(256) void cn_bilinear_fwd_kernel(const float* __restrict__ x, long xbs,
                                                             float* __restrict__ y, long ybs, int C, int Hi, int Wi,
                                                             int Ho, int Wo, float sh, float sw, int Hp, int Wp) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= Ho * Wo) return;
  const int c = blockIdx.y, b = blockIdx.z;
  const int oy = p / Wo, ox = p - oy * Wo;
  int y0, y1, x0, x1;
  float ly, lx;
  bl_src(oy, sh, Hi, y0, y1, ly);
  bl_src(ox, sw, Wi, x0, x1, lx);
  const float* xp = x + b * xbs + (long)c * Hp * Wp;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float v = hy * (hx * xp[y0 * Wp + x0] + lx * xp[y0 * Wp + x1]) +
                  ly * (hx * xp[y1 * Wp + x0] + lx * xp[y1 * Wp + x1]);
  y[b * ybs + (long)c * Ho * Wo + p] = v;
}

// Adjoint in gather form: every input pixel sums the output pixels that read it (deterministic).
__global__ __launch_bounds__(256) void cn_bilinear_bwd_kernel(const float* __restrict__ dy, long dybs,
                                                             float* __restrict__ dx, long dxbs, int C, int Hi,
                                                             int Wi, int Ho, int Wo, float sh, float sw,
                                                             float inv_sh, float inv_sw, int accumulate, int Hp, int Wp) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= Hp * Wp) return;
  const int c = blockIdx.y, b = blockIdx.z;
  const int iy = p / Wp, ix = p - iy * Wp;
  if (iy >= Hi || ix >= Wi) {  // padding of the stored grid: no output reads it
    if (!accumulate) dx[b * dxbs + (long)c * Hp * Wp + p] = 0.f;
    return;
  }
  // candidate outputs: src in (iy-1, iy+1)  ->  o in ((iy-1)/s, (iy+1)/s), widened by one for rounding
  int oy_lo = (int)floorf((iy - 1) * inv_sh) - 1, oy_hi = (int)ceilf((iy + 1) * inv_sh) + 1;
  int ox_lo = (int)floorf((ix - 1) * inv_sw) - 1, ox_hi = (int)ceilf((ix + 1) * inv_sw) + 1;
  if (sh == 0.f) { oy_lo = 0; oy_hi = Ho - 1; }
  if (sw == 0.f) { ox_lo = 0; ox_hi = Wo - 1; }
  oy_lo = max(oy_lo, 0); oy_hi = min(oy_hi, Ho - 1);
  ox_lo = max(ox_lo, 0); ox_hi = min(ox_hi, Wo - 1);
  const float* dp = dy + b * dybs + (long)c * Ho * Wo;
  float acc = 0.f;
  if (ox_hi - ox_lo < 8) {
    // near 1:1 resizes (every one in TowerUNet): the column weights of the <= 8 candidates are computed once
    float wxv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ox = ox_lo + j;
      float wx = 0.f;
      if (ox <= ox_hi) {
        int x0, x1; float lx;
        bl_src(ox, sw, Wi, x0, x1, lx);
        if (x0 == ix) wx += 1.f - lx;
        if (x1 == ix) wx += lx;
      }
      wxv[j] = wx;
    }
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1; float ly;
      bl_src(oy, sh, Hi, y0, y1, ly);
      float wy = 0.f;
      if (y0 == iy) wy += 1.f - ly;
      if (y1 == iy) wy += ly;
      if (wy == 0.f) continue;
      const float* row = dp + oy * Wo + ox_lo;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (wxv[j] != 0.f) acc += wy * wxv[j] * row[j];
    }
    float* o = dx + b * dxbs + (long)c * Hp * Wp + p;
    *o = accumulate ? *o + acc : acc;
    return;
  }
  for (int oy = oy_lo; oy <= oy_hi; ++oy) {
    int y0, y1; float ly;
    bl_src(oy, sh, Hi, y0, y1, ly);
    float wy = 0.f;
    if (y0 == iy) wy += 1.f - ly;
    if (y1 == iy) wy += ly;
    if (wy == 0.f) continue;
    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
      int x0, x1; float lx;
      bl_src(ox, sw, Wi, x0, x1, lx);
      float wx = 0.f;
      if (x0 == ix) wx += 1.f - lx;
      if (x1 == ix) wx += lx;
      if (wx != 0.f) acc += wy * wx * dp[oy * Wo + ox];
    }
  }
  float* o = dx + b * dxbs + (long)c * Hp * Wp + p;
  *o = accumulate ? *o + acc : acc;
}

// Same adjoint for resizes that shrink by less than 2x per axis (every resize of TowerUNet is within a few pixels
// of 1:1): an input pixel then has at most 4 contributing output rows / columns. Their indices and weights depend
// only on the pixel, so a lane computes them ONCE and reuses them for BL_CH channels (the per-channel work is the
// <= 16 weighted loads). Pixels with more candidates than that (never, for scale > 0.5) take the general path.
#define BL_CH 16
__device__ __forceinline__ int bl_candidates(int i, int in_size, int out_size, float scale, float inv_scale, int* idx,
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
    bl_src(o, scale, in_size, i0, i1, l1);
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
    bl_src(o, scale, in_size, i0, i1, l1);
    float w = 0.f;
    if (i0 == i) w += 1.f - l1;
    if (i1 == i) w += l1;
    idx[k] = o;
    wgt[k] = live ? w : 0.f;
  }
  return n;
}

__global__ __launch_bounds__(256) void cn_bilinear_bwd_near_kernel(const float* __restrict__ dy, long dybs,
                                                                  float* __restrict__ dx, long dxbs, int C, int Hi,
                                                                  int Wi, int Ho, int Wo, float sh, float sw,
                                                                  float inv_sh, float inv_sw, int accumulate, int Hp,
                                                                  int Wp) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= Hp * Wp) return;
  const int c_begin = blockIdx.y * BL_CH, b = blockIdx.z;
  const int iy = p / Wp, ix = p - iy * Wp;
  const bool padding = iy >= Hi || ix >= Wi;  // outside the image on the stored grid: written as zeros below (ny = 0)
  int oyv[4] = {0, 0, 0, 0}, oxv[4] = {0, 0, 0, 0};
  float wyv[4] = {0.f, 0.f, 0.f, 0.f}, wxv[4] = {0.f, 0.f, 0.f, 0.f};
  const int ny = padding ? 0 : bl_candidates(iy, Hi, Ho, sh, inv_sh, oyv, wyv);
  const int nx = padding ? 0 : bl_candidates(ix, Wi, Wo, sw, inv_sw, oxv, wxv);
  int c_end = c_begin + BL_CH;
  if (c_end > C) c_end = C;
  if (ny <= 4 && nx <= 4) {
    // The 3 x 3 first candidate slots are loaded UNCONDITIONALLY (nine independent loads in flight per channel; a
    // branch per slot serialised them: 150 us for a [8,128,99,99] -> 100x100 adjoint that moves 82 MB). Unused slots
    // alias slot 0 -- a live candidate -- with weight 0, which leaves the sum unchanged; a fourth candidate per axis
    // (resizes that shrink by almost 2x) goes through the predicated tail.
    if (ny == 0 || nx == 0) {  // no output reads this pixel
      for (int c = c_begin; c < c_end; ++c) {
        float* o = dx + b * dxbs + (long)c * Hp * Wp + p;
        if (!accumulate) *o = 0.f;
      }
      return;
    }
    float w[4][4];
    int off[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool live = k < ny && j < nx;
        w[k][j] = live ? wyv[k] * wxv[j] : 0.f;
        off[k][j] = live ? oyv[k] * Wo + oxv[j] : oyv[0] * Wo + oxv[0];
      }
    const bool tail = ny == 4 || nx == 4;
    for (int c = c_begin; c < c_end; ++c) {
      const float* dp = dy + b * dybs + (long)c * Ho * Wo;
      float v[3][3];
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < 3; ++j) v[k][j] = dp[off[k][j]];
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc += w[k][j] * v[k][j];
      if (tail) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if ((k == 3 || j == 3) && w[k][j] != 0.f) acc += w[k][j] * dp[off[k][j]];
      }
      float* o = dx + b * dxbs + (long)c * Hp * Wp + p;
      *o = accumulate ? *o + acc : acc;
    }
    return;
  }
  // general path for this pixel
  int oy_lo = (int)floorf((iy - 1) * inv_sh) - 1, oy_hi = (int)ceilf((iy + 1) * inv_sh) + 1;
  int ox_lo = (int)floorf((ix - 1) * inv_sw) - 1, ox_hi = (int)ceilf((ix + 1) * inv_sw) + 1;
  if (sh == 0.f) { oy_lo = 0; oy_hi = Ho - 1; }
  if (sw == 0.f) { ox_lo = 0; ox_hi = Wo - 1; }
  oy_lo = max(oy_lo, 0); oy_hi = min(oy_hi, Ho - 1);
  ox_lo = max(ox_lo, 0); ox_hi = min(ox_hi, Wo - 1);
  for (int c = c_begin; c < c_end; ++c) {
    const float* dp = dy + b * dybs + (long)c * Ho * Wo;
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1; float ly;
      bl_src(oy, sh, Hi, y0, y1, ly);
      float wy = 0.f;
      if (y0 == iy) wy += 1.f - ly;
      if (y1 == iy) wy += ly;
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1; float lx;
        bl_src(ox, sw, Wi, x0, x1, lx);
        float wx = 0.f;
        if (x0 == ix) wx += 1.f - lx;
        if (x1 == ix) wx += lx;
        if (wx != 0.f) acc += wy * wx * dp[oy * Wo + ox];
      }
    }
    float* o = dx + b * dxbs + (long)c * Hp * Wp + p;
    *o = accumulate ? *o + acc : acc;
  }
}

static inline float bl_scale(int in_size, int out_size) {
  return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
}

// Hp x Wp: stored grid of the SOURCE (>= Hi x Wi; the image is its top-left Hi x Wi; 0 = dense, Hp = Hi, Wp = Wi).
extern "C" int cn_bilinear_fwd_f32(const float* x, long xbs, float* y, long ybs, int B, int C, int Hi, int Wi, int Ho,
                                   int Wo, int Hp, int Wp, void* stream) {
  if (B <= 0 || C <= 0) return CN_OK;
  if (Hp <= 0) Hp = Hi;
  if (Wp <= 0) Wp = Wi;
  if (Hp < Hi || Wp < Wi) return CN_ERR_ARG;
  dim3 grid((Ho * Wo + 255) / 256, C, B);
  CN_LAUNCH(cn_bilinear_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, xbs, y, ybs, C, Hi, Wi, Ho,
                     Wo, bl_scale(Hi, Ho), bl_scale(Wi, Wo), Hp, Wp);
  return cn_check_launch();
}

// Adjoint; dx lives on the stored grid Hp x Wp (see above): its padding is written as zeros (left alone when accumulating).
extern "C" int cn_bilinear_bwd_f32(const float* dy, long dybs, float* dx, long dxbs, int B, int C, int Hi, int Wi,
                                   int Ho, int Wo, int Hp, int Wp, int accumulate, void* stream) {
  if (B <= 0 || C <= 0) return CN_OK;
  if (Hp <= 0) Hp = Hi;
  if (Wp <= 0) Wp = Wi;
  if (Hp < Hi || Wp < Wi) return CN_ERR_ARG;
  const float sh = bl_scale(Hi, Ho), sw = bl_scale(Wi, Wo);
  if (2 * Hi > Ho && 2 * Wi > Wo && (long)B * C * Hi * Wi >= 1 << 16) {  // near-1:1 resize of a large tensor
    dim3 gridn((Hp * Wp + 255) / 256, (C + BL_CH - 1) / BL_CH, B);
    CN_LAUNCH(cn_bilinear_bwd_near_kernel, gridn, dim3(256), 0, (hipStream_t)stream, dy, dybs, dx, dxbs, C,
                       Hi, Wi, Ho, Wo, sh, sw, sh > 0.f ? 1.f / sh : 0.f, sw > 0.f ? 1.f / sw : 0.f, accumulate, Hp, Wp);
    return cn_check_launch();
  }
  dim3 grid((Hp * Wp + 255) / 256, C, B);
  CN_LAUNCH(cn_bilinear_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, dybs, dx, dxbs, C, Hi, Wi,
                     Ho, Wo, sh, sw, sh > 0.f ? 1.f / sh : 0.f, sw > 0.f ? 1.f / sw : 0.f, accumulate, Hp, Wp);
  return cn_check_launch();
}

// ---------------------------------------------------------------------------
// ConvTranspose2d with stride >= kernel size (TowerUNetFinal's up_conv of final_c: k 3, stride 4, padding 1;
// /root/reference/src/cultionet/nn/modules/unet_parts.py:227-309 + convolution.py:45-68) as a DENSE 1x1 contraction plus a
// pointwise pass. With s >= k no two taps of a cell meet: every input pixel (a, b) owns the k x k block of outputs at
// (s*a - pad + ky, s*b - pad + kx) and every other output is bias only. The contraction
//     P[b][co*k*k + ky*k + kx][a][b'] = sum_ci w[ci][co][ky][kx] * x[b][ci][a][b']
// is a plain GEMM over the SMALL grid (cn_conv_transpose2d_*_f32 with a 1x1 kernel on the weight tensor viewed as
// [Cin][Cout*k*k]: the 128 -> 128 layer at 8 x 25 x 25 is 1.5 GFLOP). The parity-class launches it replaces ran the same
// arithmetic as 16 classes of one tap each (66 us), a 9-tap stride-4 gather (45 us) and a dword-staged weight gradient
// at 10 TFLOP/s (142 us), plus the resize pair. The two kernels here do what is left:
//   forward : z = resize(bias + scatter(P))      one pass, the (s*n - c)^2 intermediate never exists
//   backward: dP = gather(resize^T(dz))          the adjoint of both, written in P's layout
// (the bias gradient is the per-channel sum of dz: the resize weights of every output pixel sum to one).
// ---------------------------------------------------------------------------
#define CT_CH 16
// y(p, q) of channel c = bias + P[c*KK + ky*K + kx][a][b'] when p = s*a - pad + ky, q = s*b' - pad + kx with ky, kx < K;
// returns the offset of that element relative to channel c's block of P, or -1 (bias only).
__device__ __forceinline__ int ct_src(int p, int q, int K, int s, int pad, int Hc, int Wc) {
  const int pa = p + pad, qa = q + pad;
  const int a = pa / s, b = qa / s;
  const int ky = pa - a * s, kx = qa - b * s;
  if (ky >= K || kx >= K || a >= Hc || b >= Wc) return -1;
  return ((ky * K + kx) * Hc + a) * Wc + b;
}

// grid (pixels of z / 256, channel chunks, B)
__global__ __launch_bounds__(256) void cn_convt_taps_fwd_kernel(const float* __restrict__ P, long pbs,
                                                               const float* __restrict__ bias,
                                                               float* __restrict__ z, long zbs, int C, int Hc, int Wc,
                                                               int K, int s, int pad, int Hy, int Wy, int Ho, int Wo,
                                                               float sh, float sw) {
  const int px = blockIdx.x * 256 + threadIdx.x;
  if (px >= Ho * Wo) return;
  const int c0 = blockIdx.y * CT_CH, b = blockIdx.z;
  const int oy = px / Wo, ox = px - oy * Wo;
  int y0, y1, x0, x1;
  float ly, lx;
  if (Ho == Hy && Wo == Wy) {  // no resize behind the transposed convolution
    y0 = y1 = oy; x0 = x1 = ox; ly = lx = 0.f;
  } else {
    bl_src(oy, sh, Hy, y0, y1, ly);
    bl_src(ox, sw, Wy, x0, x1, lx);
  }
  const float hy = 1.f - ly, hx = 1.f - lx;
  const int o00 = ct_src(y0, x0, K, s, pad, Hc, Wc), o01 = ct_src(y0, x1, K, s, pad, Hc, Wc);
  const int o10 = ct_src(y1, x0, K, s, pad, Hc, Wc), o11 = ct_src(y1, x1, K, s, pad, Hc, Wc);
  const long cstride = (long)K * K * Hc * Wc;
  const int c1 = c0 + CT_CH < C ? c0 + CT_CH : C;
  for (int c = c0; c < c1; ++c) {
    const float* pc = P + b * pbs + c * cstride;
    // (unused slots re-read element 0 with weight 0: four independent loads per channel, no branch per slot)
    const float v00 = pc[o00 < 0 ? 0 : o00], v01 = pc[o01 < 0 ? 0 : o01];
    const float v10 = pc[o10 < 0 ? 0 : o10], v11 = pc[o11 < 0 ? 0 : o11];
    const float bv = bias != nullptr ? bias[c] : 0.f;
    const float a00 = (o00 < 0 ? 0.f : v00) + bv, a01 = (o01 < 0 ? 0.f : v01) + bv;
    const float a10 = (o10 < 0 ? 0.f : v10) + bv, a11 = (o11 < 0 ? 0.f : v11) + bv;
    z[b * zbs + (long)c * Ho * Wo + px] = hy * (hx * a00 + lx * a01) + ly * (hx * a10 + lx * a11);
  }
}

// grid (elements of one channel's block of dP / 256, channel chunks, B); a thread owns one (tap, cell) for CT_CH channels
__global__ __launch_bounds__(256) void cn_convt_taps_bwd_kernel(const float* __restrict__ dz, long dzbs,
                                                               float* __restrict__ dP, long dpbs, int C, int Hc, int Wc,
                                                               int K, int s, int pad, int Hy, int Wy, int Ho, int Wo,
                                                               float sh, float sw, float inv_sh, float inv_sw) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int per = K * K * Hc * Wc;
  if (e >= per) return;
  const int c0 = blockIdx.y * CT_CH, b = blockIdx.z;
  const int t = e / (Hc * Wc), cell = e - t * (Hc * Wc);
  const int ky = t / K, kx = t - ky * K;
  const int a = cell / Wc, bb = cell - a * Wc;
  const int p = s * a - pad + ky, q = s * bb - pad + kx;
  const int c1 = c0 + CT_CH < C ? c0 + CT_CH : C;
  const bool inside = p >= 0 && p < Hy && q >= 0 && q < Wy;
  int oyv[4] = {0, 0, 0, 0}, oxv[4] = {0, 0, 0, 0};
  float wyv[4] = {0.f, 0.f, 0.f, 0.f}, wxv[4] = {0.f, 0.f, 0.f, 0.f};
  int ny = 0, nx = 0;
  if (inside) {
    if (Ho == Hy && Wo == Wy) {
      ny = nx = 1; oyv[0] = p; oxv[0] = q; wyv[0] = wxv[0] = 1.f;
    } else {
      ny = bl_candidates(p, Hy, Ho, sh, inv_sh, oyv, wyv);
      nx = bl_candidates(q, Wy, Wo, sw, inv_sw, oxv, wxv);
    }
  }
  if (ny > 4) ny = 4;  // (a resize that shrinks by 2x or more would need the general adjoint: refused by the launcher)
  if (nx > 4) nx = 4;
  for (int c = c0; c < c1; ++c) {
    const float* dp = dz + b * dzbs + (long)c * Ho * Wo;
    float acc = 0.f;
    for (int k = 0; k < ny; ++k)
      for (int j = 0; j < nx; ++j) acc += wyv[k] * wxv[j] * dp[oyv[k] * Wo + oxv[j]];
    dP[b * dpbs + (long)c * per + e] = acc;
  }
}

// P / dP: [B][C*K*K][Hc][Wc] (batch strides pbs / dpbs); z / dz: [B][C][Ho][Wo]; (Hy, Wy) = (Hc-1)*s - 2*pad + K, the
// size of the transposed convolution's own output; (Ho, Wo) = the size check_upsample resizes it to (== (Hy, Wy): none).
extern "C" int cn_convt_taps_fwd_f32(const float* P, long pbs, const float* bias, float* z, long zbs, int B, int C,
                                     int Hc, int Wc, int K, int stride, int pad, int Ho, int Wo, void* stream) {
  if (B <= 0 || C <= 0) return CN_OK;
  if (K < 1 || stride < K || pad < 0 || pad >= stride) return CN_ERR_ARG;
  const int Hy = (Hc - 1) * stride - 2 * pad + K, Wy = (Wc - 1) * stride - 2 * pad + K;
  if (Hy < 1 || Wy < 1 || Ho < 1 || Wo < 1) return CN_ERR_ARG;
  dim3 grid((Ho * Wo + 255) / 256, (C + CT_CH - 1) / CT_CH, B);
  CN_LAUNCH(cn_convt_taps_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, P, pbs, bias, z, zbs, C, Hc, Wc, K, stride,
            pad, Hy, Wy, Ho, Wo, bl_scale(Hy, Ho), bl_scale(Wy, Wo));
  return cn_check_launch();
}

extern "C" int cn_convt_taps_bwd_f32(const float* dz, long dzbs, float* dP, long dpbs, int B, int C, int Hc, int Wc,
                                     int K, int stride, int pad, int Ho, int Wo, void* stream) {
  if (B <= 0 || C <= 0) return CN_OK;
  if (K < 1 || stride < K || pad < 0 || pad >= stride) return CN_ERR_ARG;
  const int Hy = (Hc - 1) * stride - 2 * pad + K, Wy = (Wc - 1) * stride - 2 * pad + K;
  if (Hy < 1 || Wy < 1 || Ho < 1 || Wo < 1) return CN_ERR_ARG;
  if (2 * Hy <= Ho || 2 * Wy <= Wo) return CN_ERR_ARG;  // <= 4 outputs read an input pixel only for resizes below 2x
  const float sh = bl_scale(Hy, Ho), sw = bl_scale(Wy, Wo);
  dim3 grid((K * K * Hc * Wc + 255) / 256, (C + CT_CH - 1) / CT_CH, B);
  CN_LAUNCH(cn_convt_taps_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, dz, dzbs, dP, dpbs, C, Hc, Wc, K, stride,
            pad, Hy, Wy, Ho, Wo, sh, sw, sh > 0.f ? 1.f / sh : 0.f, sw > 0.f ? 1.f / sw : 0.f);
  return cn_check_launch();
}

// ---------------------------------------------------------------------------
// Strided-batch copy / add:  dst[b][i] (+)= src[b][i], i < n  (channel slices of NCHW tensors)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cn_copy_kernel(const float* __restrict__ src, long sbs,
                                                     float* __restrict__ dst, long dbs, long n, int accumulate) {
  const int b = blockIdx.y;
  const float* s = src + b * sbs;
  float* d = dst + b * dbs;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    d[i] = accumulate ? d[i] + s[i] : s[i];
  }
}

extern "C" int cn_copy_f32(const float* src, long sbs, float* dst, long dbs, int B, long n, int accumulate,
                           void* stream) {
  if (B <= 0 || n <= 0) return CN_OK;
  long bx = (n + 1023) / 1024;
  if (bx > 2048) bx = 2048;
  CN_LAUNCH(cn_copy_kernel, dim3((unsigned)bx, B), dim3(256), 0, (hipStream_t)stream, src, sbs, dst, dbs, n,
                     accumulate);
  return cn_check_launch();
}

// dst[b][i] = a[b][i] + c[b][i]
__global__ __launch_bounds__(256) void cn_add_kernel(const float* __restrict__ a, long abs_, const float* __restrict__ c,
                                                    long cbs, float* __restrict__ dst, long dbs, long n) {
  const int b = blockIdx.y;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    dst[b * dbs + i] = a[b * abs_ + i] + c[b * cbs + i];
}

extern "C" int cn_add_f32(const float* a, long abs_, const float* c, long cbs, float* dst, long dbs, int B, long n,
                          void* stream) {
  if (B <= 0 || n <= 0) return CN_OK;
  long bx = (n + 1023) / 1024;
  if (bx > 2048) bx = 2048;
  CN_LAUNCH(cn_add_kernel, dim3((unsigned)bx, B), dim3(256), 0, (hipStream_t)stream, a, abs_, c, cbs, dst,
                     dbs, n);
  return cn_check_launch();
}

__global__ void cn_fill_kernel(float* __restrict__ p, long n, float v) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = v;
}

extern "C" int cn_fill_f32(float* p, long n, float v, void* stream) {
  if (n <= 0) return CN_OK;
  long bx = (n + 1023) / 1024;
  if (bx > 2048) bx = 2048;
  CN_LAUNCH(cn_fill_kernel, dim3((unsigned)bx), dim3(256), 0, (hipStream_t)stream, p, n, v);
  return cn_check_launch();
}

// ---------------------------------------------------------------------------
// TowerUNetFinalCombine (unet_parts.py:101-193), fused:
//   s_k   = sum_t (1/gamma[k][t]) * h_t[:, k]          k = 0 dist, 1 edge, 2 crop; t = towers a,b,c
//   z_k   = w[k] * s_k + bias[k]                        (the 1x1 Conv2d(1,1))
//   out_k = sigmoid(z_k)                                (dist, crop)
//   out_1 = sigmoid(z_1 / (smooth + sigmoid(crisp)))    (edge: SigmoidCrisp)
// h_t: [B][3][HW] tower outputs (channel = task). params: HOST array of 16 DEVICE pointers to the
// scalar parameters: [3k+t] gamma[k][t], [9+k] w[k], [12+k] bias[k], [15] crisp gamma.
// ---------------------------------------------------------------------------
struct CnPtr16 { const float* p[16]; };
struct CnMutPtr16 { float* p[16]; };

__global__ __launch_bounds__(256) void cn_final_combine_fwd_kernel(const float* __restrict__ ha,
                                                                  const float* __restrict__ hb,
                                                                  const float* __restrict__ hc,
                                                                  const CnPtr16 pp,
                                                                  float* __restrict__ dist, float* __restrict__ edge,
                                                                  float* __restrict__ crop, int B, int HW,
                                                                  float smooth) {
  float prm[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) prm[k] = *pp.p[k];
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= (long)B * HW) return;
  const long b = i / HW, p = i - b * HW;
  const float crisp = 1.0f / (smooth + cn_sigmoid(prm[15]));
  float* outs[3] = {dist, edge, crop};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const long off = (b * 3 + k) * HW + p;
    const float s = (1.0f / prm[3 * k + 0]) * ha[off] + (1.0f / prm[3 * k + 1]) * hb[off] + (1.0f / prm[3 * k + 2]) * hc[off];
    float z = prm[9 + k] * s + prm[12 + k];
    if (k == 1) z *= crisp;
    outs[k][i] = cn_sigmoid(z);
  }
}

// Backward: dh_t (written); dparams: 16 device pointers, atomically accumulated into. Grid-stride over the pixels
// with the 16 parameter-gradient sums in registers: ONE block reduction and 16 atomics per block (<= 256 blocks) --
// one block per 256 pixels meant 16 same-address float atomics from each of 1250 blocks at batch 32 (102 us for a
// kernel whose forward takes 6).
__global__ __launch_bounds__(256) void cn_final_combine_bwd_kernel(
    const float* __restrict__ ha, const float* __restrict__ hb, const float* __restrict__ hc,
    const CnPtr16 pp, const float* __restrict__ dist, const float* __restrict__ edge,
    const float* __restrict__ crop, const float* __restrict__ ddist, const float* __restrict__ dedge,
    const float* __restrict__ dcrop, float* __restrict__ dha, float* __restrict__ dhb, float* __restrict__ dhc,
    const CnMutPtr16 dpp, int B, int HW, float smooth) {
  __shared__ float scratch[4];
  float prm[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) prm[k] = *pp.p[k];
  const float sg = cn_sigmoid(prm[15]);
  const float crisp = 1.0f / (smooth + sg);
  const float* outs[3] = {dist, edge, crop};
  const float* douts[3] = {ddist, dedge, dcrop};
  float acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  const long n = (long)B * HW;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
    const long b = i / HW, p = i - b * HW;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const long off = (b * 3 + k) * HW + p;
      const float ia = 1.0f / prm[3 * k + 0], ib = 1.0f / prm[3 * k + 1], ic = 1.0f / prm[3 * k + 2];
      const float va = ha[off], vb = hb[off], vc = hc[off];
      const float s = va * ia + vb * ib + vc * ic;
      const float o = outs[k][i];
      float dz = douts[k][i] * o * (1.f - o);  // wrt sigmoid argument
      if (k == 1) {
        const float zlin = prm[9 + k] * s + prm[12 + k];
        acc[15] += dz * zlin;
        dz *= crisp;
      }
      const float ds = dz * prm[9 + k];
      acc[9 + k] += dz * s;
      acc[12 + k] += dz;
      dha[off] = ds * ia;
      dhb[off] = ds * ib;
      dhc[off] = ds * ic;
      acc[3 * k + 0] -= ds * va * ia * ia;
      acc[3 * k + 1] -= ds * vb * ib * ib;
      acc[3 * k + 2] -= ds * vc * ic * ic;
    }
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float t = cn_block_sum<float, 256>(acc[k], scratch);
    // d/dgamma [1/(smooth + sigmoid(gamma))] = -sg(1-sg) * crisp^2
    if (threadIdx.x == 0) atomicAdd(dpp.p[k], k == 15 ? t * (-sg * (1.f - sg) * crisp * crisp) : t);
  }
}

extern "C" int cn_final_combine_fwd_f32(const float* ha, const float* hb, const float* hc,
                                        const float* const* params, float* dist, float* edge, float* crop, int B,
                                        int HW, float smooth, void* stream) {
  const long n = (long)B * HW;
  if (n <= 0) return CN_OK;
  CnPtr16 pp;
  for (int k = 0; k < 16; ++k) pp.p[k] = params[k];
  CN_LAUNCH(cn_final_combine_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, ha, hb, hc, pp, dist, edge, crop, B, HW, smooth);
  return cn_check_launch();
}

extern "C" int cn_final_combine_bwd_f32(const float* ha, const float* hb, const float* hc,
                                        const float* const* params, const float* dist, const float* edge,
                                        const float* crop, const float* ddist, const float* dedge,
                                        const float* dcrop, float* dha, float* dhb, float* dhc,
                                        float* const* dparams, int B, int HW, float smooth, void* stream) {
  const long n = (long)B * HW;
  if (n <= 0) return CN_OK;
  CnPtr16 pp;
  CnMutPtr16 dpp;
  for (int k = 0; k < 16; ++k) { pp.p[k] = params[k]; dpp.p[k] = dparams[k]; }
  const long nb = (n + 255) / 256;
  CN_LAUNCH(cn_final_combine_bwd_kernel, dim3((unsigned)(nb < 256 ? nb : 256)), dim3(256), 0,
                     (hipStream_t)stream, ha, hb, hc, pp, dist, edge, crop, ddist, dedge, dcrop, dha, dhb, dhc,
                     dpp, B, HW, smooth);
  return cn_check_launch();
}

// ---------------------------------------------------------------------------
// Dropout (nn.Dropout2d after each encoder block: convolution.py:495,511; natten attn_drop / proj_drop).
// Counter-based mask: keep(i) = splitmix64(seed + i) >= p * 2^64, never stored -- backward recomputes it.
//   channelwise != 0: one decision per (b, c) plane (Dropout2d); else one per element (Dropout).
// y = x * keep / (1 - p). Same entry point serves backward (x := dy, accumulate into dx).
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long cn_splitmix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void cn_dropout_kernel(const float* __restrict__ x, long xbs, float* __restrict__ y,
                                                        long ybs, int C, int L, unsigned long long thresh,
                                                        float scale, unsigned long long seed_, const unsigned long long* __restrict__ step, int channelwise,
                                                        int accumulate) {
  const unsigned long long seed = cn_step_seed(seed_, step);
  const int c = blockIdx.y, b = blockIdx.z;
  const unsigned long long plane = (unsigned long long)b * C + c;
  const float* xp = x + b * xbs + (long)c * L;
  float* yp = y + b * ybs + (long)c * L;
  float pk = 0.f;
  if (channelwise) pk = cn_splitmix64(seed + plane) >= thresh ? scale : 0.f;
  for (int l = blockIdx.x * 256 + threadIdx.x; l < L; l += gridDim.x * 256) {
    const float k = channelwise ? pk : (cn_splitmix64(seed + plane * (unsigned long long)L + l) >= thresh ? scale : 0.f);
    const float v = xp[l] * k;
    yp[l] = accumulate ? yp[l] + v : v;
  }
}

extern "C" int cn_dropout_f32(const float* x, long xbs, float* y, long ybs, int B, int C, int L, float p,
                              unsigned long long seed, const unsigned long long* step, int channelwise, int accumulate,
                              void* stream) {
  if (B <= 0 || C <= 0 || L <= 0) return CN_OK;
  if (!(p >= 0.f && p < 1.f)) return CN_ERR_ARG;
  const double t = (double)p * 18446744073709551616.0;  // p * 2^64
  const unsigned long long thresh = t >= 18446744073709551615.0 ? ~0ull : (unsigned long long)t;
  int bx = (L + 1023) / 1024;
  if (bx < 1) bx = 1;
  CN_LAUNCH(cn_dropout_kernel, dim3(bx, C, B), dim3(256), 0, (hipStream_t)stream, x, xbs, y, ybs, C, L,
                     thresh, 1.0f / (1.0f - p), seed, step, channelwise, accumulate);
  return cn_check_launch();
}

__global__ void cn_rng_advance_kernel(unsigned long long* word, unsigned long long value, int set) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *word = set ? value : *word + value;
}

extern "C" int cn_rng_advance_u64(unsigned long long* word, unsigned long long value, int set, void* stream) {
  if (word == nullptr) return CN_ERR_ARG;
  CN_LAUNCH(cn_rng_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, word, value, set);
  return cn_check_launch();
}

// ---------------------------------------------------------------------------
// F.adaptive_max_pool2d (pool_by_max=True: convolution.py:499-503). Window of output o along one axis:
// [floor(o*In/Out), ceil((o+1)*In/Out)). idx: int32 flat argmax inside the input plane (first max, as ATen).
// Backward in gather form (windows may overlap when In % Out != 0): deterministic, no atomics.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int amp_start(int o, int in, int out) { return (int)(((long)o * in) / out); }
__device__ __forceinline__ int amp_end(int o, int in, int out) { return (int)((((long)(o + 1)) * in + out - 1) / out); }

__global__ __launch_bounds__(256) void cn_adaptive_maxpool_fwd_kernel(const float* __restrict__ x, long xbs,
                                                                     float* __restrict__ y, long ybs,
                                                                     int* __restrict__ idx, int C, int Hi, int Wi,
                                                                     int Ho, int Wo) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= Ho * Wo) return;
  const int c = blockIdx.y, b = blockIdx.z;
  const int oy = p / Wo, ox = p - oy * Wo;
  const int y0 = amp_start(oy, Hi, Ho), y1 = amp_end(oy, Hi, Ho);
  const int x0 = amp_start(ox, Wi, Wo), x1 = amp_end(ox, Wi, Wo);
  const float* xp = x + b * xbs + (long)c * Hi * Wi;
  float best = -INFINITY;
  int bi = y0 * Wi + x0;
  for (int iy = y0; iy < y1; ++iy)
    for (int ix = x0; ix < x1; ++ix) {
      const float v = xp[iy * Wi + ix];
      if (v > best || v != v) { best = v; bi = iy * Wi + ix; }
    }
  y[b * ybs + (long)c * Ho * Wo + p] = best;
  if (idx) idx[((long)b * C + c) * Ho * Wo + p] = bi;
}

__global__ __launch_bounds__(256) void cn_adaptive_maxpool_bwd_kernel(const float* __restrict__ dy, long dybs,
                                                                     const int* __restrict__ idx,
                                                                     float* __restrict__ dx, long dxbs, int C, int Hi,
                                                                     int Wi, int Ho, int Wo, int accumulate) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= Hi * Wi) return;
  const int c = blockIdx.y, b = blockIdx.z;
  const int iy = p / Wi, ix = p - iy * Wi;
  // outputs whose window can contain (iy, ix): start(o) <= i < end(o)
  int oy_lo = (int)(((long)iy * Ho) / Hi) - 1, oy_hi = (int)((((long)(iy + 1)) * Ho + Hi - 1) / Hi);
  int ox_lo = (int)(((long)ix * Wo) / Wi) - 1, ox_hi = (int)((((long)(ix + 1)) * Wo + Wi - 1) / Wi);
  oy_lo = max(oy_lo, 0); oy_hi = min(oy_hi, Ho - 1);
  ox_lo = max(ox_lo, 0); ox_hi = min(ox_hi, Wo - 1);
  const float* dp = dy + b * dybs + (long)c * Ho * Wo;
  const int* ip = idx + ((long)b * C + c) * Ho * Wo;
  float acc = 0.f;
  for (int oy = oy_lo; oy <= oy_hi; ++oy)
    for (int ox = ox_lo; ox <= ox_hi; ++ox)
      if (ip[oy * Wo + ox] == p) acc += dp[oy * Wo + ox];
  float* o = dx + b * dxbs + (long)c * Hi * Wi + p;
  *o = accumulate ? *o + acc : acc;
}

extern "C" int cn_adaptive_maxpool_fwd_f32(const float* x, long xbs, float* y, long ybs, int* idx, int B, int C,
                                           int Hi, int Wi, int Ho, int Wo, void* stream) {
  if (B <= 0 || C <= 0 || Ho <= 0 || Wo <= 0) return CN_OK;
  CN_LAUNCH(cn_adaptive_maxpool_fwd_kernel, dim3((Ho * Wo + 255) / 256, C, B), dim3(256), 0,
                     (hipStream_t)stream, x, xbs, y, ybs, idx, C, Hi, Wi, Ho, Wo);
  return cn_check_launch();
}

extern "C" int cn_adaptive_maxpool_bwd_f32(const float* dy, long dybs, const int* idx, float* dx, long dxbs, int B,
                                           int C, int Hi, int Wi, int Ho, int Wo, int accumulate, void* stream) {
  if (B <= 0 || C <= 0) return CN_OK;
  CN_LAUNCH(cn_adaptive_maxpool_bwd_kernel, dim3((Hi * Wi + 255) / 256, C, B), dim3(256), 0,
                     (hipStream_t)stream, dy, dybs, idx, dx, dxbs, C, Hi, Wi, Ho, Wo, accumulate);
  return cn_check_launch();
}
