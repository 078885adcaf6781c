// BatchNorm (+SiLU, +residual) and LayerNorm over channels on bf16 NHWC activations (gfx950, HBM-bound).
//
// Mixed-precision path (BASELINE configs[2]; the reference's precision="16-mixed", model.py:168-186): activations
// bf16 [pixels][C] with a pixel stride `ld` (channel slices of concat buffers in place), statistics, parameters and
// parameter gradients fp32 (reference ops: nn.BatchNorm2d + SiLU of ConvBlock2d, nn/modules/convolution.py:71-120;
// nn.LayerNorm around NeighborhoodAttention2D, convolution.py:338-353).
// A thread owns ONE 8-channel group (16 bytes) and walks pixels, so its per-channel constants and partial sums live
// in registers and every access is a full 16-byte lane load of a contiguous pixel row.
#include "cn_bf16.h"

#define BBN_MAX_BLOCKS 512

// ---- per-channel sums over pixels ---------------------------------------------------------------------------
// MODE 0: {sum x, sum x^2}                      (BatchNorm forward statistics)
// MODE 1: {sum dz, sum dz*xhat}, dz = dy*act'(z) (BatchNorm backward)
// part[(blk*2 + k) * C + c]
template <int MODE>
__global__ __launch_bounds__(256) void cn_bbn_partial_kernel(const bf16_t* __restrict__ x, long ldx,
                                                            const bf16_t* __restrict__ dy, long lddy,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, long P, int C, int act,
                                                            long rows_per_block, float* __restrict__ part) {
  __shared__ float red[256 * 16];
  const int C8 = C >> 3;
  const int R = 256 / C8;
  const int tid = threadIdx.x;
  const int row = tid / C8, cg = tid - row * C8;
  const bool live = row < R;
  float a1[8], a2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a1[j] = a2[j] = 0.f;
  float m[8], rs[8], ga[8], be[8];
  if (MODE == 1 && live) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      m[j] = mean[cg * 8 + j]; rs[j] = rstd[cg * 8 + j]; ga[j] = gamma[cg * 8 + j]; be[j] = beta[cg * 8 + j];
    }
  }
  const long p0 = blockIdx.x * rows_per_block;
  const long p1 = p0 + rows_per_block < P ? p0 + rows_per_block : P;
  if (live) {
    // two pixels per iteration: both 16-byte loads (four in MODE 1) are in flight before either is consumed
    auto accum = [&](const u32x4& xr, const u32x4& dr) {
      float xv[8];
      cn_unpack8(xr, xv);
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { a1[j] += xv[j]; a2[j] += xv[j] * xv[j]; }
      } else {
        float dv[8];
        cn_unpack8(dr, dv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = (xv[j] - m[j]) * rs[j];
          float dz = dv[j];
          if (act == 1) dz *= cn_silu_grad(ga[j] * xh + be[j]);
          a1[j] += dz;
          a2[j] += dz * xh;
        }
      }
    };
    const u32x4 z4 = {0u, 0u, 0u, 0u};
    long p = p0 + row;
    for (; p + R < p1; p += 2 * R) {
      const u32x4 xa = *reinterpret_cast<const u32x4*>(x + p * ldx + cg * 8);
      const u32x4 xb = *reinterpret_cast<const u32x4*>(x + (p + R) * ldx + cg * 8);
      u32x4 da = z4, db = z4;
      if (MODE == 1) {
        da = *reinterpret_cast<const u32x4*>(dy + p * lddy + cg * 8);
        db = *reinterpret_cast<const u32x4*>(dy + (p + R) * lddy + cg * 8);
      }
      accum(xa, da);
      accum(xb, db);
    }
    if (p < p1) {
      const u32x4 xa = *reinterpret_cast<const u32x4*>(x + p * ldx + cg * 8);
      u32x4 da = z4;
      if (MODE == 1) da = *reinterpret_cast<const u32x4*>(dy + p * lddy + cg * 8);
      accum(xa, da);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[tid * 16 + j] = a1[j]; red[tid * 16 + 8 + j] = a2[j]; }
  __syncthreads();
  for (int idx = tid; idx < C8 * 16; idx += 256) {
    const int g2 = idx >> 4, j = idx & 15;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += red[(r * C8 + g2) * 16 + j];
    part[((long)blockIdx.x * 2 + (j >> 3)) * C + g2 * 8 + (j & 7)] = s;
  }
}

// Per-channel fp64 combine of the partial-sum rows: one WAVE per channel (4 channels per 256-thread block), lane l
// walks rows l, l+64, ... in four independent chains, then a shuffle tree. Rows come from the statistics pass
// (<= 512) or from the convolution epilogue (one per pixel tile: 2560 at batch 32 x 100^2), so the serial depth is what
// matters: 16 lanes per channel took 17.5 us for 2560 rows, one thread per channel 80 us for 512.
#define BBN_FIN_CH 4  // channels per finalize block
__device__ __forceinline__ void bbn_combine(const float* __restrict__ part, int nblk, int C, int c, int sub, double& s,
                                            double& ss) {
  s = 0.0; ss = 0.0;
  if (c < C) {
    double s1 = 0.0, q1 = 0.0, s2 = 0.0, q2 = 0.0, s3 = 0.0, q3 = 0.0;
    int i = sub;
    for (; i + 192 < nblk; i += 256) {
      s += part[((long)i * 2) * C + c]; ss += part[((long)i * 2 + 1) * C + c];
      s1 += part[((long)(i + 64) * 2) * C + c]; q1 += part[((long)(i + 64) * 2 + 1) * C + c];
      s2 += part[((long)(i + 128) * 2) * C + c]; q2 += part[((long)(i + 128) * 2 + 1) * C + c];
      s3 += part[((long)(i + 192) * 2) * C + c]; q3 += part[((long)(i + 192) * 2 + 1) * C + c];
    }
    for (; i < nblk; i += 64) { s += part[((long)i * 2) * C + c]; ss += part[((long)i * 2 + 1) * C + c]; }
    s += s1 + s2 + s3; ss += q1 + q2 + q3;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off, 64); ss += __shfl_xor(ss, off, 64); }
}

// Forward finalize: batch mean / rstd (fp64 combine), running-statistics update (momentum, unbiased variance).
// `part` rows come from the statistics pass or straight from the convolution's epilogue (one row per pixel tile).
__global__ __launch_bounds__(256) void cn_bbn_finalize_fwd_kernel(const float* __restrict__ part, int nblk, int C,
                                                                 double count,
                                                                 float eps, float momentum,
                                                                 float* __restrict__ running_mean,
                                                                 float* __restrict__ running_var,
                                                                 float* __restrict__ mean, float* __restrict__ rstd,
                                                                 int training) {
  const int c = blockIdx.x * BBN_FIN_CH + (threadIdx.x >> 6), sub = threadIdx.x & 63;
  if (!training) {
    if (c < C && sub == 0) {
      mean[c] = running_mean[c];
      rstd[c] = 1.0f / sqrtf(running_var[c] + eps);
    }
    return;
  }
  double s = 0.0, ss = 0.0;
  bbn_combine(part, nblk, C, c, sub, s, ss);
  if (c >= C || sub != 0) return;
  const double md = s / count;
  double var = ss / count - md * md;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)md;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean != nullptr) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)md;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// Backward finalize: coef = {mean(dz), mean(dz*xhat)}; dgamma += sum dz*xhat, dbeta += sum dz.
__global__ __launch_bounds__(256) void cn_bbn_finalize_bwd_kernel(const float* __restrict__ part, int nblk, int C,
                                                                 double count, float* __restrict__ coef,
                                                                 float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, int training) {
  const int c = blockIdx.x * BBN_FIN_CH + (threadIdx.x >> 6), sub = threadIdx.x & 63;
  double s1, s2;
  bbn_combine(part, nblk, C, c, sub, s1, s2);
  if (c >= C || sub != 0) return;
  coef[c] = training ? (float)(s1 / count) : 0.f;
  coef[C + c] = training ? (float)(s2 / count) : 0.f;
  dgamma[c] += (float)s2;
  dbeta[c] += (float)s1;
}

// MODE 0: y = act(gamma*(x-mean)*rstd + beta) (+ res)
// MODE 1: dx (+)= gamma*rstd*(dz - coef0 - xhat*coef1), dz = dy*act'(z)
template <int MODE>
__global__ __launch_bounds__(256) void cn_bbn_apply_kernel(const bf16_t* __restrict__ x, long ldx,
                                                          const bf16_t* __restrict__ dy, long lddy,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ coef,
                                                          const bf16_t* __restrict__ res, long ldr,
                                                          bf16_t* __restrict__ y, long ldy, long P, int C, int act,
                                                          int accumulate, long rows_per_block) {
  const int C8 = C >> 3;
  const int R = 256 / C8;
  const int tid = threadIdx.x;
  const int row = tid / C8, cg = tid - row * C8;
  if (row >= R) return;
  float sc[8], sh[8], m[8], rs[8], c0[8], c1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cg * 8 + j;
    m[j] = mean[c]; rs[j] = rstd[c];
    sc[j] = gamma[c] * rs[j];
    sh[j] = beta[c] - m[j] * sc[j];
    if (MODE == 1) { c0[j] = coef[c]; c1[j] = coef[C + c]; }
  }
  const long p0 = blockIdx.x * rows_per_block;
  const long p1 = p0 + rows_per_block < P ? p0 + rows_per_block : P;
  const u32x4 z4 = {0u, 0u, 0u, 0u};
  auto body = [&](long p, const u32x4& xr, const u32x4& ar, const u32x4& br) {
    float xv[8], o[8];
    cn_unpack8(xr, xv);
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float z = xv[j] * sc[j] + sh[j];
        if (act == 1) z = cn_silu(z);
        o[j] = z;
      }
      if (res != nullptr) {
        float rv[8];
        cn_unpack8(ar, rv);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += rv[j];
      }
    } else {
      float dv[8];
      cn_unpack8(ar, dv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (xv[j] - m[j]) * rs[j];
        float dz = dv[j];
        if (act == 1) dz *= cn_silu_grad(xv[j] * sc[j] + sh[j]);
        o[j] = sc[j] * (dz - c0[j] - xh * c1[j]);
      }
      if (accumulate) {
        float ov[8];
        cn_unpack8(br, ov);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += ov[j];
      }
    }
    *reinterpret_cast<u32x4*>(y + p * ldy + cg * 8) = cn_pack8(o);
  };
  // second operand: residual (MODE 0) or dy (MODE 1); third: the dx being accumulated into
  auto ld2 = [&](long p) -> u32x4 {
    if (MODE == 0) return res != nullptr ? *reinterpret_cast<const u32x4*>(res + p * ldr + cg * 8) : z4;
    return *reinterpret_cast<const u32x4*>(dy + p * lddy + cg * 8);
  };
  auto ld3 = [&](long p) -> u32x4 {
    return (MODE == 1 && accumulate) ? *reinterpret_cast<const u32x4*>(y + p * ldy + cg * 8) : z4;
  };
  long p = p0 + row;
  for (; p + R < p1; p += 2 * R) {  // two pixels per iteration: all loads issued before the first is consumed
    const u32x4 xa = *reinterpret_cast<const u32x4*>(x + p * ldx + cg * 8);
    const u32x4 xb = *reinterpret_cast<const u32x4*>(x + (p + R) * ldx + cg * 8);
    const u32x4 aa = ld2(p), ab = ld2(p + R);
    const u32x4 ba = ld3(p), bb = ld3(p + R);
    body(p, xa, aa, ba);
    body(p + R, xb, ab, bb);
  }
  if (p < p1) {
    const u32x4 xa = *reinterpret_cast<const u32x4*>(x + p * ldx + cg * 8);
    const u32x4 aa = ld2(p);
    const u32x4 ba = ld3(p);
    body(p, xa, aa, ba);
  }
}

// grid of the apply passes: no per-block output rows, so many more (smaller) blocks than the statistics passes
static inline void bbn_apply_grid(long P, int C, int& nblk, long& rows) {
  const int R = 256 / (C >> 3);
  long want = (P + (long)R * 4 - 1) / ((long)R * 4);  // >= 4 row-iterations per block
  if (want > 4096) want = 4096;
  if (want < 1) want = 1;
  rows = (P + want - 1) / want;
  rows = (rows + R - 1) / R * R;
  nblk = (int)((P + rows - 1) / rows);
}

static inline void bbn_grid(long P, int C, int& nblk, long& rows) {
  const int R = 256 / (C >> 3);
  long want = (P + (long)R * 8 - 1) / ((long)R * 8);  // >= 8 row-iterations per block
  if (want > BBN_MAX_BLOCKS) want = BBN_MAX_BLOCKS;
  if (want < 1) want = 1;
  rows = (P + want - 1) / want;
  rows = (rows + R - 1) / R * R;
  nblk = (int)((P + rows - 1) / rows);
}

// Floats of scratch for the calls below: per-block partial sums + backward coefficients.
extern "C" long cn_bn_workspace_floats_bf16(int C) { return (long)BBN_MAX_BLOCKS * 2 * C + 2 * C; }

// y = act(bn(x)) (+ res). x, res, y: bf16 [P][C] rows with pixel strides; C % 8 == 0, C <= 2048.
// training: batch statistics (saved to mean / rstd, running stats updated); else running statistics.
// conv_sums (nullable, training only): conv_rows rows of {sum, sum of squares}[C] from cn_conv2d_fwd_bf16's epilogue
// -- skips the statistics pass over x.
extern "C" int cn_bn_act_fwd_bf16(const void* x, long ldx, const float* gamma, const float* beta, float* running_mean,
                                  float* running_var, const void* res, long ldr, void* y, long ldy, float* mean,
                                  float* rstd, float* ws, long P, int C, int training, float momentum, float eps,
                                  int act, const float* conv_sums, int conv_rows, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (P <= 0 || C <= 0) return CN_OK;
  if ((C & 7) || C > 2048) return CN_ERR_ARG;
  if (!training && (running_mean == nullptr || running_var == nullptr)) return CN_ERR_ARG;
  int nblk;
  long rows;
  bbn_grid(P, C, nblk, rows);
  if (training && !(conv_sums != nullptr && conv_rows > 0))
    CN_LAUNCH((cn_bbn_partial_kernel<0>), dim3(nblk), dim3(256), 0, stream, (const bf16_t*)x, ldx, nullptr,
                       0L, nullptr, nullptr, nullptr, nullptr, P, C, act, rows, ws);
  const bool fused = training && conv_sums != nullptr && conv_rows > 0;
  CN_LAUNCH(cn_bbn_finalize_fwd_kernel, dim3((C + BBN_FIN_CH - 1) / BBN_FIN_CH), dim3(256), 0, stream, fused ? conv_sums : ws,
                     fused ? conv_rows : nblk, C, (double)P, eps, momentum, running_mean, running_var, mean, rstd,
                     training);
  int ablk;
  long arows;
  bbn_apply_grid(P, C, ablk, arows);
  CN_LAUNCH((cn_bbn_apply_kernel<0>), dim3(ablk), dim3(256), 0, stream, (const bf16_t*)x, ldx, nullptr, 0L,
                     mean, rstd, gamma, beta, nullptr, (const bf16_t*)res, ldr, (bf16_t*)y, ldy, P, C, act, 0, arows);
  return cn_check_launch();
}

// dx (+)= d act(bn(x)) / dx . dy; dgamma / dbeta are ACCUMULATED. dx nullable (parameter gradients only).
extern "C" int cn_bn_act_bwd_bf16(const void* x, long ldx, const void* dy, long lddy, const float* mean,
                                  const float* rstd, const float* gamma, const float* beta, void* dx, long lddx,
                                  float* dgamma, float* dbeta, float* ws, long P, int C, int training, int act,
                                  int accumulate_dx, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (P <= 0 || C <= 0) return CN_OK;
  if ((C & 7) || C > 2048) return CN_ERR_ARG;
  int nblk;
  long rows;
  bbn_grid(P, C, nblk, rows);
  float* coef = ws + (long)BBN_MAX_BLOCKS * 2 * C;
  CN_LAUNCH((cn_bbn_partial_kernel<1>), dim3(nblk), dim3(256), 0, stream, (const bf16_t*)x, ldx,
                     (const bf16_t*)dy, lddy, mean, rstd, gamma, beta, P, C, act, rows, ws);
  CN_LAUNCH(cn_bbn_finalize_bwd_kernel, dim3((C + BBN_FIN_CH - 1) / BBN_FIN_CH), dim3(256), 0, stream, ws, nblk, C, (double)P,
                     coef, dgamma, dbeta, training);
  if (dx != nullptr) {
    int ablk;
    long arows;
    bbn_apply_grid(P, C, ablk, arows);
    CN_LAUNCH((cn_bbn_apply_kernel<1>), dim3(ablk), dim3(256), 0, stream, (const bf16_t*)x, ldx,
                       (const bf16_t*)dy, lddy, mean, rstd, gamma, beta, coef, nullptr, 0L, (bf16_t*)dx, lddx, P, C,
                       act, accumulate_dx, arows);
  }
  return cn_check_launch();
}

// ---- LayerNorm over the channel axis (rows of the NHWC image) -------------------------------------------------
// C8 = C/8 lanes share a pixel (C8 a power of two <= 64); statistics by lane shuffles, two-pass in registers.
template <int C8>
__device__ __forceinline__ float ln_group_sum(float v) {
#pragma unroll
  for (int off = C8 >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <int C8>
__global__ __launch_bounds__(256) void cn_bln_fwd_kernel(const bf16_t* __restrict__ x, long ldx,
                                                        const float* __restrict__ w, const float* __restrict__ bi,
                                                        const bf16_t* __restrict__ res, long ldr,
                                                        bf16_t* __restrict__ y, long ldy, long P, float eps) {
  constexpr int R = 256 / C8;
  const int tid = threadIdx.x;
  const int row = tid / C8, cg = tid % C8;
  float wv[8], bv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { wv[j] = w[cg * 8 + j]; bv[j] = bi[cg * 8 + j]; }
  const float invC = 1.0f / (float)(C8 * 8);
  for (long p = (long)blockIdx.x * R + row; p < P; p += (long)gridDim.x * R) {
    float xv[8], o[8];
    cn_unpack8(*reinterpret_cast<const u32x4*>(x + p * ldx + cg * 8), xv);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += xv[j];
    const float mu = ln_group_sum<C8>(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = xv[j] - mu; q += d * d; }
    const float rs = rsqrtf(ln_group_sum<C8>(q) * invC + eps);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (xv[j] - mu) * rs * wv[j] + bv[j];
    if (res != nullptr) {
      float rv[8];
      cn_unpack8(*reinterpret_cast<const u32x4*>(res + p * ldr + cg * 8), rv);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += rv[j];
    }
    *reinterpret_cast<u32x4*>(y + p * ldy + cg * 8) = cn_pack8(o);
  }
}

template <int C8>
__global__ __launch_bounds__(256) void cn_bln_bwd_kernel(const bf16_t* __restrict__ x, long ldx,
                                                        const bf16_t* __restrict__ dy, long lddy,
                                                        const float* __restrict__ w, bf16_t* __restrict__ dx,
                                                        long lddx, float* __restrict__ dw, float* __restrict__ db,
                                                        long P, float eps, int accumulate) {
  __shared__ float red[256 * 16];
  constexpr int R = 256 / C8;
  const int tid = threadIdx.x;
  const int row = tid / C8, cg = tid % C8;
  float wv[8], gw[8], gb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { wv[j] = w[cg * 8 + j]; gw[j] = 0.f; gb[j] = 0.f; }
  const float invC = 1.0f / (float)(C8 * 8);
  for (long p = (long)blockIdx.x * R + row; p < P; p += (long)gridDim.x * R) {
    float xv[8], dv[8], o[8];
    cn_unpack8(*reinterpret_cast<const u32x4*>(x + p * ldx + cg * 8), xv);
    cn_unpack8(*reinterpret_cast<const u32x4*>(dy + p * lddy + cg * 8), dv);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += xv[j];
    const float mu = ln_group_sum<C8>(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = xv[j] - mu; q += d * d; }
    const float rs = rsqrtf(ln_group_sum<C8>(q) * invC + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = (xv[j] - mu) * rs;
      const float gj = dv[j] * wv[j];
      s1 += gj;
      s2 += gj * xh;
      gw[j] += dv[j] * xh;
      gb[j] += dv[j];
      xv[j] = xh;
      dv[j] = gj;
    }
    const float m1 = ln_group_sum<C8>(s1) * invC, m2 = ln_group_sum<C8>(s2) * invC;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = rs * (dv[j] - m1 - xv[j] * m2);
    if (accumulate) {
      float ov[8];
      cn_unpack8(*reinterpret_cast<const u32x4*>(dx + p * lddx + cg * 8), ov);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += ov[j];
    }
    *reinterpret_cast<u32x4*>(dx + p * lddx + cg * 8) = cn_pack8(o);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[tid * 16 + j] = gw[j]; red[tid * 16 + 8 + j] = gb[j]; }
  __syncthreads();
  for (int idx = tid; idx < C8 * 16; idx += 256) {
    const int g2 = idx >> 4, j = idx & 15;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += red[(r * C8 + g2) * 16 + j];
    atomicAdd((j >> 3 ? db : dw) + g2 * 8 + (j & 7), s);
  }
}

#define BLN_DISPATCH(KERNEL, ...)                                                                       \
  switch (C >> 3) {                                                                                     \
    case 1: CN_LAUNCH((KERNEL<1>), grid, dim3(256), 0, stream, __VA_ARGS__); break;           \
    case 2: CN_LAUNCH((KERNEL<2>), grid, dim3(256), 0, stream, __VA_ARGS__); break;           \
    case 4: CN_LAUNCH((KERNEL<4>), grid, dim3(256), 0, stream, __VA_ARGS__); break;           \
    case 8: CN_LAUNCH((KERNEL<8>), grid, dim3(256), 0, stream, __VA_ARGS__); break;           \
    case 16: CN_LAUNCH((KERNEL<16>), grid, dim3(256), 0, stream, __VA_ARGS__); break;         \
    case 32: CN_LAUNCH((KERNEL<32>), grid, dim3(256), 0, stream, __VA_ARGS__); break;         \
    case 64: CN_LAUNCH((KERNEL<64>), grid, dim3(256), 0, stream, __VA_ARGS__); break;         \
    default: return CN_ERR_ARG;                                                                         \
  }

// y = LayerNorm_C(x) * w + b (+ res); C in {8, 16, 32, 64, 128, 256, 512}.
extern "C" int cn_layernorm_c_fwd_bf16(const void* x, long ldx, const float* w, const float* b, const void* res,
                                       long ldr, void* y, long ldy, long P, int C, float eps, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (P <= 0) return CN_OK;
  if (C & 7) return CN_ERR_ARG;
  const int R = 256 / (C >> 3);
  long nb = (P + R - 1) / R;
  if (nb > 4096) nb = 4096;
  const dim3 grid((unsigned)nb);
  BLN_DISPATCH(cn_bln_fwd_kernel, (const bf16_t*)x, ldx, w, b, (const bf16_t*)res, ldr, (bf16_t*)y, ldy, P, eps);
  return cn_check_launch();
}

// dx (+)= ...; dw / db ACCUMULATED with fp32 atomics (one add per block and channel).
extern "C" int cn_layernorm_c_bwd_bf16(const void* x, long ldx, const void* dy, long lddy, const float* w, void* dx,
                                       long lddx, float* dw, float* db, long P, int C, float eps, int accumulate_dx,
                                       void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (P <= 0) return CN_OK;
  if (C & 7) return CN_ERR_ARG;
  const int R = 256 / (C >> 3);
  long nb = (P + (long)R * 8 - 1) / ((long)R * 8);
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  const dim3 grid((unsigned)nb);
  BLN_DISPATCH(cn_bln_bwd_kernel, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, w, (bf16_t*)dx, lddx, dw, db, P, eps,
               accumulate_dx);
  return cn_check_launch();
}

// bias gradients on the bf16 path: out[c] (+)= sum_p x[p][c]. ws: cn_bn_workspace_floats_bf16(C) floats.
__global__ __launch_bounds__(256) void cn_bsum_finalize_kernel(const float* __restrict__ part, int nblk, int C,
                                                              float* __restrict__ out, int accumulate) {
  const int c = blockIdx.x * BBN_FIN_CH + (threadIdx.x >> 6), sub = threadIdx.x & 63;
  double s, ss;
  bbn_combine(part, nblk, C, c, sub, s, ss);
  if (c >= C || sub != 0) return;
  out[c] = accumulate ? out[c] + (float)s : (float)s;
}

extern "C" int cn_channel_sum_bf16(const void* x, long ldx, long P, int C, float* out, int accumulate, float* ws,
                                   void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (P <= 0 || C <= 0) return CN_OK;
  if ((C & 7) || C > 2048) return CN_ERR_ARG;
  int nblk;
  long rows;
  bbn_grid(P, C, nblk, rows);
  CN_LAUNCH((cn_bbn_partial_kernel<0>), dim3(nblk), dim3(256), 0, stream, (const bf16_t*)x, ldx, nullptr, 0L,
                     nullptr, nullptr, nullptr, nullptr, P, C, 0, rows, ws);
  CN_LAUNCH(cn_bsum_finalize_kernel, dim3((C + BBN_FIN_CH - 1) / BBN_FIN_CH), dim3(256), 0, stream, ws, nblk, C, out, accumulate);
  return cn_check_launch();
}
