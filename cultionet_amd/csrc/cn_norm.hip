// BatchNorm (train / eval) with fused SiLU + residual add, LayerNorm over the channel axis
// of NCHW tensors, and per-channel sums. HBM-bound streaming kernels (gfx950).
//
// Reference ops: nn.BatchNorm2d / BatchNorm3d + SiLU inside ConvBlock2d
// (/root/reference/src/cultionet/nn/modules/convolution.py:71-120, models/nunet.py:18-57),
// nn.LayerNorm over channels in NHWC (nunet.py:93-97, convolution.py:338-353).
// Here tensors stay NCHW: a LayerNorm "row" is the C values of one pixel (stride L),
// lanes walk consecutive pixels so every access is coalesced.
#include "cn_common.h"

#define BN_SPLIT_MAX 256  // workspace rows per channel
#define BN_SPLIT_GROUP 64 // cap of the grouped kernels and the channel sums (every thread walks the partials serially there)

// ---------------------------------------------------------------------------
// BatchNorm statistics: x viewed as [B][C][L] (batch stride xbs). Partial sums in fp64.
// part[c][split][2] = {sum, sumsq} over this block's share.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cn_bn_partial_kernel(const float* __restrict__ x, long xbs, int B, int C,
                                                           int L, int splits, double* __restrict__ part, int vec4) {
  __shared__ double scratch[4];
  const int c = blockIdx.x, sp = blockIdx.y;
  double s = 0.0, ss = 0.0;
  if (vec4) {  // L % 4 == 0, 16-byte aligned planes: one float4 per lane and iteration
    const int L4 = L >> 2;
    const int per = (L4 + splits - 1) / splits;
    const int beg = sp * per;
    const int end = (beg + per < L4) ? beg + per : L4;
    for (int b = 0; b < B; ++b) {
      const float4* xp = reinterpret_cast<const float4*>(x + b * xbs + (long)c * L);
      for (int l = beg + threadIdx.x; l < end; l += 256) {
        const float4 v = xp[l];
        s += (double)((v.x + v.y) + (v.z + v.w));
        ss += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
      }
    }
  } else {
    const int per = (L + splits - 1) / splits;
    const int beg = sp * per;
    const int end = (beg + per < L) ? beg + per : L;
    for (int b = 0; b < B; ++b) {
      const float* xp = x + b * xbs + (long)c * L;
      for (int l = beg + threadIdx.x; l < end; l += 256) {
        const float v = xp[l];
        s += v;
        ss += (double)v * v;
      }
    }
  }
  s = cn_block_sum<double, 256>(s, scratch);
  ss = cn_block_sum<double, 256>(ss, scratch);
  if (threadIdx.x == 0) {
    part[((long)c * splits + sp) * 2 + 0] = s;
    part[((long)c * splits + sp) * 2 + 1] = ss;
  }
}

// Sum of a channel's `splits` partial pairs by the whole block (thread t takes partials t, t + 256, ...): every thread
// returns the totals.
__device__ __forceinline__ void cn_bn_sum_parts(const double* __restrict__ part, int c, int splits, double& s,
                                                double& ss) {
  __shared__ double scratch[4];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < splits; i += 256) {
    a += part[((long)c * splits + i) * 2 + 0];
    b += part[((long)c * splits + i) * 2 + 1];
  }
  s = cn_block_sum<double, 256>(a, scratch);
  ss = cn_block_sum<double, 256>(b, scratch);
}

// Eval mode: mean/rstd from running statistics.
__global__ void cn_bn_eval_stats_kernel(const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                        int C, float eps, float* __restrict__ mean, float* __restrict__ rstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean[c] = running_mean[c];
  rstd[c] = 1.0f / sqrtf(running_var[c] + eps);
}

// y = act(gamma * (x - mean) * rstd + beta) (+ residual). act: 0 none, 1 SiLU.
// Training (part != nullptr): every block reduces its channel's `splits` partial sums itself (a few dozen
// doubles) instead of waiting for a separate finalize launch; block (0, c, 0) publishes mean / rstd for the
// backward pass and updates the running statistics.
__global__ __launch_bounds__(256) void cn_bn_apply_kernel(const float* __restrict__ x, long xbs,
                                                         float* __restrict__ mean, float* __restrict__ rstd,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         const float* __restrict__ res, long rbs,
                                                         float* __restrict__ y, long ybs, int B, int C, int L, int act,
                                                         const double* __restrict__ part, int splits, double count,
                                                         float eps, float momentum, float* __restrict__ running_mean,
                                                         float* __restrict__ running_var) {
  const int c = blockIdx.y, b = blockIdx.z;
  float m, rs;
  if (part != nullptr) {
    double s, ss;
    cn_bn_sum_parts(part, c, splits, s, ss);
    const double md = s / count;
    double var = ss / count - md * md;
    if (var < 0.0) var = 0.0;
    m = (float)md;
    rs = (float)(1.0 / sqrt(var + (double)eps));
    if (blockIdx.x == 0 && b == 0 && threadIdx.x == 0) {
      mean[c] = m;
      rstd[c] = rs;
      if (running_mean != nullptr) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
      }
    }
  } else {
    m = mean[c];
    rs = rstd[c];
  }
  const float sc = gamma[c] * rs, be = beta[c];
  const long r0 = (long)c * L;
  const float* xp = x + b * xbs + r0;
  const float* rp = res ? res + b * rbs + r0 : nullptr;
  float* yp = y + b * ybs + r0;
  for (int l = blockIdx.x * 256 + threadIdx.x; l < L; l += gridDim.x * 256) {
    float z = (xp[l] - m) * sc + be;
    if (act == 1) z = cn_silu(z);
    if (rp) z += rp[l];
    yp[l] = z;
  }
}

// Backward pass 1: per-channel partial sums of dz and dz*xhat, dz = dy * act'(z).
__global__ __launch_bounds__(256) void cn_bn_bwd_partial_kernel(const float* __restrict__ x, long xbs,
                                                               const float* __restrict__ dy, long dybs,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int B, int C, int L,
                                                               int act, int splits, double* __restrict__ part,
                                                               int vec4) {
  __shared__ double scratch[4];
  const int c = blockIdx.x, sp = blockIdx.y;
  const float m = mean[c], rs = rstd[c], ga = gamma[c], be = beta[c];
  double s1 = 0.0, s2 = 0.0;
  auto term = [&](float xv, float dv) {
    const float xh = (xv - m) * rs;
    float dz = dv;
    if (act == 1) dz *= cn_silu_grad(ga * xh + be);
    s1 += dz;
    s2 += (double)dz * xh;
  };
  if (vec4) {
    const int L4 = L >> 2;
    const int per = (L4 + splits - 1) / splits;
    const int beg = sp * per;
    const int end = (beg + per < L4) ? beg + per : L4;
    for (int b = 0; b < B; ++b) {
      const float4* xp = reinterpret_cast<const float4*>(x + b * xbs + (long)c * L);
      const float4* dp = reinterpret_cast<const float4*>(dy + b * dybs + (long)c * L);
      for (int l = beg + threadIdx.x; l < end; l += 256) {
        const float4 xv = xp[l], dv = dp[l];
        term(xv.x, dv.x); term(xv.y, dv.y); term(xv.z, dv.z); term(xv.w, dv.w);
      }
    }
  } else {
    const int per = (L + splits - 1) / splits;
    const int beg = sp * per;
    const int end = (beg + per < L) ? beg + per : L;
    for (int b = 0; b < B; ++b) {
      const float* xp = x + b * xbs + (long)c * L;
      const float* dp = dy + b * dybs + (long)c * L;
      for (int l = beg + threadIdx.x; l < end; l += 256) term(xp[l], dp[l]);
    }
  }
  s1 = cn_block_sum<double, 256>(s1, scratch);
  s2 = cn_block_sum<double, 256>(s2, scratch);
  if (threadIdx.x == 0) {
    part[((long)c * splits + sp) * 2 + 0] = s1;
    part[((long)c * splits + sp) * 2 + 1] = s2;
  }
}

// Backward pass 2: dx (+)= gamma*rstd*(dz - mean(dz) - xhat*mean(dz*xhat))   [train]
//                  dx (+)= gamma*rstd*dz                                        [eval]
// Every block reduces its channel's partial sums itself; block (0, c, 0) accumulates dgamma / dbeta.
__global__ __launch_bounds__(256) void cn_bn_bwd_apply_kernel(const float* __restrict__ x, long xbs,
                                                             const float* __restrict__ dy, long dybs,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const double* __restrict__ part, int splits,
                                                             double count, int training,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             int accumulate_params, float* __restrict__ dx, long dxbs,
                                                             int B, int C, int L, int act, int accumulate) {
  const int c = blockIdx.y, b = blockIdx.z;
  double s1, s2;
  cn_bn_sum_parts(part, c, splits, s1, s2);
  if (blockIdx.x == 0 && b == 0 && threadIdx.x == 0) {
    if (accumulate_params) {
      dgamma[c] += (float)s2;
      dbeta[c] += (float)s1;
    } else {
      dgamma[c] = (float)s2;
      dbeta[c] = (float)s1;
    }
  }
  if (dx == nullptr) return;
  const float m = mean[c], rs = rstd[c], ga = gamma[c], be = beta[c];
  const float c1 = training ? (float)(s1 / count) : 0.f, c2 = training ? (float)(s2 / count) : 0.f;
  const long r0 = (long)c * L;
  const float* xp = x + b * xbs + r0;
  const float* dp = dy + b * dybs + r0;
  float* dxp = dx + b * dxbs + r0;
  for (int l = blockIdx.x * 256 + threadIdx.x; l < L; l += gridDim.x * 256) {
    const float xh = (xp[l] - m) * rs;
    float dz = dp[l];
    if (act == 1) dz *= cn_silu_grad(ga * xh + be);
    float g = (dz - c1 - xh * c2) * (ga * rs);
    if (accumulate) g += dxp[l];
    dxp[l] = g;
  }
}

// ---- small tensors (B*L <= 2048 values per channel: the 13x13 level): ONE launch, one block per
// channel, the channel's values held in registers between the statistics and the normalisation -- x (and dy) are
// read once and the partial-sums round trip + second launch disappear. Same fp64 statistics as the two-pass path.
template <int NV>
__global__ __launch_bounds__(256) void cn_bn_fused_fwd_kernel(const float* __restrict__ x, long xbs,
                                                             float* __restrict__ mean, float* __restrict__ rstd,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ res, long rbs,
                                                             float* __restrict__ y, long ybs, int B, int C, int L,
                                                             int act, float eps, float momentum,
                                                             float* __restrict__ running_mean,
                                                             float* __restrict__ running_var) {
  __shared__ double scratch[4];
  const int c = blockIdx.x, n = B * L;
  float v[NV];
  double s = 0.0, ss = 0.0;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = threadIdx.x + k * 256;
    v[k] = 0.f;
    if (i < n) {
      const int b = i / L, l = i - b * L;
      v[k] = x[b * xbs + (long)c * L + l];
      s += v[k];
      ss += (double)v[k] * v[k];
    }
  }
  s = cn_block_sum<double, 256>(s, scratch);
  ss = cn_block_sum<double, 256>(ss, scratch);
  const double count = (double)n;
  const double md = s / count;
  double var = ss / count - md * md;
  if (var < 0.0) var = 0.0;
  const float m = (float)md, rs = (float)(1.0 / sqrt(var + (double)eps));
  if (threadIdx.x == 0) {
    mean[c] = m;
    rstd[c] = rs;
    if (running_mean != nullptr) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  }
  const float sc = gamma[c] * rs, be = beta[c];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = threadIdx.x + k * 256;
    if (i < n) {
      const int b = i / L, l = i - b * L;
      float z = (v[k] - m) * sc + be;
      if (act == 1) z = cn_silu(z);
      if (res) z += res[b * rbs + (long)c * L + l];
      y[b * ybs + (long)c * L + l] = z;
    }
  }
}

template <int NV>
__global__ __launch_bounds__(256) void cn_bn_fused_bwd_kernel(const float* __restrict__ x, long xbs,
                                                             const float* __restrict__ dy, long dybs,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int training,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             int accumulate_params, float* __restrict__ dx, long dxbs,
                                                             int B, int C, int L, int act, int accumulate) {
  __shared__ double scratch[4];
  const int c = blockIdx.x, n = B * L;
  const float m = mean[c], rs = rstd[c], ga = gamma[c], be = beta[c];
  float xh[NV], dz[NV];
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = threadIdx.x + k * 256;
    xh[k] = 0.f;
    dz[k] = 0.f;
    if (i < n) {
      const int b = i / L, l = i - b * L;
      xh[k] = (x[b * xbs + (long)c * L + l] - m) * rs;
      float d = dy[b * dybs + (long)c * L + l];
      if (act == 1) d *= cn_silu_grad(ga * xh[k] + be);
      dz[k] = d;
      s1 += d;
      s2 += (double)d * xh[k];
    }
  }
  s1 = cn_block_sum<double, 256>(s1, scratch);
  s2 = cn_block_sum<double, 256>(s2, scratch);
  if (threadIdx.x == 0) {
    if (accumulate_params) {
      dgamma[c] += (float)s2;
      dbeta[c] += (float)s1;
    } else {
      dgamma[c] = (float)s2;
      dbeta[c] = (float)s1;
    }
  }
  if (dx == nullptr) return;
  const double count = (double)n;
  const float c1 = training ? (float)(s1 / count) : 0.f, c2 = training ? (float)(s2 / count) : 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = threadIdx.x + k * 256;
    if (i < n) {
      const int b = i / L, l = i - b * L;
      float g = (dz[k] - c1 - xh[k] * c2) * (ga * rs);
      float* o = dx + b * dxbs + (long)c * L + l;
      if (accumulate) g += *o;
      *o = g;
    }
  }
}

#define BN_FUSED_MAX (8 * 256)  // larger channels are faster as two wide launches than as one block per channel

// Single BatchNorm with few channels (BatchNorm3d of the time reduction: C = 3 channels of 3.2M elements at batch 32
// were 192 blocks on 256 CUs, ~2 TB/s): up to 256 splits per channel, reduced block-parallel by the apply kernels.
static int bn_splits_wide(int C, long L) {
  int s = (2048 + C - 1) / C;
  const long maxs = (L + 511) / 512;
  if (s > maxs) s = (int)maxs;
  if (s > BN_SPLIT_MAX) s = BN_SPLIT_MAX;
  if (s < 1) s = 1;
  return s;
}

static int bn_splits(int C, long L) {
  int s = (1024 + C - 1) / C;
  const long maxs = (L + 511) / 512;
  if (s > maxs) s = (int)maxs;
  if (s > BN_SPLIT_GROUP) s = BN_SPLIT_GROUP;
  if (s < 1) s = 1;
  return s;
}

static dim3 plane_grid(int B, int C, int L) {
  int bx = (L + 1023) / 1024;  // ~4 elements per thread
  if (bx < 1) bx = 1;
  return dim3(bx, C, B);
}

extern "C" int cn_bn_workspace_doubles(int C) { return C * BN_SPLIT_MAX * 2; }

// Forward. x,y viewed as [B][C][L] (L = H*W for BatchNorm2d, T*H*W for BatchNorm3d).
// training != 0: batch statistics (saved to mean/rstd [C]) and running stats updated in place.
// ws: workspace of cn_bn_workspace_doubles(C) doubles.
extern "C" int cn_bn_act_fwd_f32(const float* x, long xbs, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, const float* res, long rbs, float* y,
                                 long ybs, float* mean, float* rstd, double* ws, int B, int C, int L, int training,
                                 float momentum, float eps, int act, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || C <= 0 || L <= 0) return CN_OK;
  if (training && (long)B * L <= BN_FUSED_MAX && C >= 32) {
#define CN_BN_FWD(NV_)                                                                                               \
  CN_LAUNCH((cn_bn_fused_fwd_kernel<NV_>), dim3(C), dim3(256), 0, stream, x, xbs, mean, rstd, gamma, beta, res, \
                     rbs, y, ybs, B, C, L, act, eps, momentum, running_mean, running_var)
    CN_BN_FWD(8);
#undef CN_BN_FWD
    return cn_check_launch();
  }
  if (training) {
    const int splits = bn_splits_wide(C, L);
    CN_LAUNCH(cn_bn_partial_kernel, dim3(C, splits), dim3(256), 0, stream, x, xbs, B, C, L, splits, ws,
                       (int)(L % 4 == 0 && xbs % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0));
    CN_LAUNCH(cn_bn_apply_kernel, plane_grid(B, C, L), dim3(256), 0, stream, x, xbs, mean, rstd, gamma, beta,
                       res, rbs, y, ybs, B, C, L, act, (const double*)ws, splits, (double)B * L, eps, momentum,
                       running_mean, running_var);
  } else {
    CN_LAUNCH(cn_bn_eval_stats_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, running_mean,
                       running_var, C, eps, mean, rstd);
    CN_LAUNCH(cn_bn_apply_kernel, plane_grid(B, C, L), dim3(256), 0, stream, x, xbs, mean, rstd, gamma, beta,
                       res, rbs, y, ybs, B, C, L, act, (const double*)nullptr, 0, 1.0, eps, momentum,
                       (float*)nullptr, (float*)nullptr);
  }
  return cn_check_launch();
}

// Backward. dgamma/dbeta (+)= per accumulate_params; dx (+)= per accumulate_dx.
// coef: scratch [2*C] floats; ws as in forward.
extern "C" int cn_bn_act_bwd_f32(const float* x, long xbs, const float* dy, long dybs, const float* mean,
                                 const float* rstd, const float* gamma, const float* beta, float* dx, long dxbs,
                                 float* dgamma, float* dbeta, float* coef, double* ws, int B, int C, int L,
                                 int training, int act, int accumulate_dx, int accumulate_params, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || C <= 0 || L <= 0) return CN_OK;
  if ((long)B * L <= BN_FUSED_MAX && C >= 32) {
#define CN_BN_BWD(NV_)                                                                                              \
  CN_LAUNCH((cn_bn_fused_bwd_kernel<NV_>), dim3(C), dim3(256), 0, stream, x, xbs, dy, dybs, mean, rstd, gamma, \
                     beta, training, dgamma, dbeta, accumulate_params, dx, dxbs, B, C, L, act, accumulate_dx)
    CN_BN_BWD(8);
#undef CN_BN_BWD
    return cn_check_launch();
  }
  const int splits = bn_splits_wide(C, L);
  CN_LAUNCH(cn_bn_bwd_partial_kernel, dim3(C, splits), dim3(256), 0, stream, x, xbs, dy, dybs, mean, rstd,
                     gamma, beta, B, C, L, act, splits, ws,
                     (int)(L % 4 == 0 && xbs % 4 == 0 && dybs % 4 == 0 &&
                           ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0));
  (void)coef;
  CN_LAUNCH(cn_bn_bwd_apply_kernel, dx != nullptr ? plane_grid(B, C, L) : dim3(1, C, 1), dim3(256), 0,
                     stream, x, xbs, dy, dybs, mean, rstd, gamma, beta, (const double*)ws, splits, (double)B * L,
                     training, dgamma, dbeta, accumulate_params, dx, dxbs, B, C, L, act, accumulate_dx);
  return cn_check_launch();
}

// ---------------------------------------------------------------------------
// Grouped BatchNorm (+ SiLU): G (<= 4) BatchNorm layers over G same-shaped tensors in one launch pair -- the
// dilation branches of ResidualAConv (convolution.py:376-395). sum_outputs != 0 fuses the ResUNet-a sum
//   y = res + sum_g act(bn_g(x_g))
// into the one normalisation pass (the sequential form re-reads and re-writes the running sum once per branch).
// ---------------------------------------------------------------------------
#define BN_MAX_GROUPS 4
struct CnBnGroupArgs {
  const float* x[BN_MAX_GROUPS];
  const float* gamma[BN_MAX_GROUPS];
  const float* beta[BN_MAX_GROUPS];
  float* mean[BN_MAX_GROUPS];
  float* rstd[BN_MAX_GROUPS];
  float* running_mean[BN_MAX_GROUPS];
  float* running_var[BN_MAX_GROUPS];
  float* y[BN_MAX_GROUPS];         // forward outputs (sum mode: y[0] only)
  const float* dy[BN_MAX_GROUPS];  // backward: output gradients (sum mode: all the same pointer)
  float* dx[BN_MAX_GROUPS];
  float* dgamma[BN_MAX_GROUPS];
  float* dbeta[BN_MAX_GROUPS];
  int accumulate_dx[BN_MAX_GROUPS];
  long xbs, ybs, dybs, dxbs, rbs;
  const float* res;
  int G, B, C, L, act, splits, training, sum_outputs, accumulate_params;
  int vec4;  // backward: L % 4 == 0 and every x / dy / dx plane 16-byte aligned => float4 accesses
  float eps, momentum;
  double* part;  // [G][C][splits][2]
};

__global__ __launch_bounds__(256) void cn_bn_group_partial_kernel(const CnBnGroupArgs a) {
  __shared__ double scratch[4];
  const int c = blockIdx.x, sp = blockIdx.y, g = blockIdx.z;
  double s = 0.0, ss = 0.0;
  if (a.vec4) {
    const int L4 = a.L >> 2;
    const int per = (L4 + a.splits - 1) / a.splits;
    const int beg = sp * per;
    const int end = (beg + per < L4) ? beg + per : L4;
    for (int b = 0; b < a.B; ++b) {
      const float4* xp = reinterpret_cast<const float4*>(a.x[g] + b * a.xbs + (long)c * a.L);
      for (int l = beg + threadIdx.x; l < end; l += 256) {
        const float4 v = xp[l];
        s += (double)((v.x + v.y) + (v.z + v.w));
        ss += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
      }
    }
  } else {
    const int per = (a.L + a.splits - 1) / a.splits;
    const int beg = sp * per;
    const int end = (beg + per < a.L) ? beg + per : a.L;
    for (int b = 0; b < a.B; ++b) {
      const float* xp = a.x[g] + b * a.xbs + (long)c * a.L;
      for (int l = beg + threadIdx.x; l < end; l += 256) {
        const float v = xp[l];
        s += v;
        ss += (double)v * v;
      }
    }
  }
  s = cn_block_sum<double, 256>(s, scratch);
  ss = cn_block_sum<double, 256>(ss, scratch);
  if (threadIdx.x == 0) {
    double* p = a.part + (((long)g * a.C + c) * a.splits + sp) * 2;
    p[0] = s;
    p[1] = ss;
  }
}

// grid = (blocks over L, C, B)
__global__ __launch_bounds__(256) void cn_bn_group_apply_kernel(const CnBnGroupArgs a) {
  const int c = blockIdx.y, b = blockIdx.z;
  float m[BN_MAX_GROUPS], sc[BN_MAX_GROUPS], be[BN_MAX_GROUPS];
  const double count = (double)a.B * a.L;
#pragma unroll
  for (int g = 0; g < BN_MAX_GROUPS; ++g) {
    m[g] = sc[g] = be[g] = 0.f;
    if (g < a.G) {
      float rs;
      if (a.training) {
        double s = 0.0, ss = 0.0;
        const double* p = a.part + ((long)g * a.C + c) * a.splits * 2;
        for (int i = 0; i < a.splits; ++i) {
          s += p[2 * i];
          ss += p[2 * i + 1];
        }
        const double md = s / count;
        double var = ss / count - md * md;
        if (var < 0.0) var = 0.0;
        m[g] = (float)md;
        rs = (float)(1.0 / sqrt(var + (double)a.eps));
        if (blockIdx.x == 0 && b == 0 && threadIdx.x == 0) {
          a.mean[g][c] = m[g];
          a.rstd[g][c] = rs;
          if (a.running_mean[g] != nullptr) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            a.running_mean[g][c] = (1.f - a.momentum) * a.running_mean[g][c] + a.momentum * m[g];
            a.running_var[g][c] = (1.f - a.momentum) * a.running_var[g][c] + a.momentum * (float)unbiased;
          }
        }
      } else {
        m[g] = a.running_mean[g][c];
        rs = 1.0f / sqrtf(a.running_var[g][c] + a.eps);
        if (blockIdx.x == 0 && b == 0 && threadIdx.x == 0) {
          a.mean[g][c] = m[g];
          a.rstd[g][c] = rs;
        }
      }
      sc[g] = a.gamma[g][c] * rs;
      be[g] = a.beta[g][c];
    }
  }
  const long r0 = (long)c * a.L;
  const float* rp = a.res ? a.res + b * a.rbs + r0 : nullptr;
  if (a.vec4) {  // 16-byte accesses: L % 4 == 0 and every plane 16-byte aligned
    const float4 z4 = {0.f, 0.f, 0.f, 0.f};
    for (int l = blockIdx.x * 256 + threadIdx.x; l < (a.L >> 2); l += gridDim.x * 256) {
      float4 acc = rp ? reinterpret_cast<const float4*>(rp)[l] : z4;
#pragma unroll
      for (int g = 0; g < BN_MAX_GROUPS; ++g) {
        if (g < a.G) {
          const float4 xv = reinterpret_cast<const float4*>(a.x[g] + b * a.xbs + r0)[l];
          float4 z;
          z.x = (xv.x - m[g]) * sc[g] + be[g];
          z.y = (xv.y - m[g]) * sc[g] + be[g];
          z.z = (xv.z - m[g]) * sc[g] + be[g];
          z.w = (xv.w - m[g]) * sc[g] + be[g];
          if (a.act == 1) { z.x = cn_silu(z.x); z.y = cn_silu(z.y); z.z = cn_silu(z.z); z.w = cn_silu(z.w); }
          if (a.sum_outputs) { acc.x += z.x; acc.y += z.y; acc.z += z.z; acc.w += z.w; }
          else reinterpret_cast<float4*>(a.y[g] + b * a.ybs + r0)[l] = z;
        }
      }
      if (a.sum_outputs) reinterpret_cast<float4*>(a.y[0] + b * a.ybs + r0)[l] = acc;
    }
    return;
  }
  for (int l = blockIdx.x * 256 + threadIdx.x; l < a.L; l += gridDim.x * 256) {
    float acc = rp ? rp[l] : 0.f;
#pragma unroll
    for (int g = 0; g < BN_MAX_GROUPS; ++g) {
      if (g < a.G) {
        float z = (a.x[g][b * a.xbs + r0 + l] - m[g]) * sc[g] + be[g];
        if (a.act == 1) z = cn_silu(z);
        if (a.sum_outputs) acc += z;  // res + f_0 + f_1 + ...: the order of the sequential form
        else a.y[g][b * a.ybs + r0 + l] = z;
      }
    }
    if (a.sum_outputs) a.y[0][b * a.ybs + r0 + l] = acc;
  }
}

__global__ __launch_bounds__(256) void cn_bn_group_bwd_partial_kernel(const CnBnGroupArgs a) {
  __shared__ double scratch[4];
  const int c = blockIdx.x, sp = blockIdx.y, g = blockIdx.z;
  const float m = a.mean[g][c], rs = a.rstd[g][c], ga = a.gamma[g][c], be = a.beta[g][c];
  double s1 = 0.0, s2 = 0.0;
  auto term = [&](float xv, float dv) {
    const float xh = (xv - m) * rs;
    float dz = dv;
    if (a.act == 1) dz *= cn_silu_grad(ga * xh + be);
    s1 += dz;
    s2 += (double)dz * xh;
  };
  if (a.vec4) {  // 16-byte lane loads (scalar loads ran this pass at 1.9 TB/s)
    const int L4 = a.L >> 2;
    const int per = (L4 + a.splits - 1) / a.splits;
    const int beg = sp * per;
    const int end = (beg + per < L4) ? beg + per : L4;
    for (int b = 0; b < a.B; ++b) {
      const float4* xp = reinterpret_cast<const float4*>(a.x[g] + b * a.xbs + (long)c * a.L);
      const float4* dp = reinterpret_cast<const float4*>(a.dy[g] + b * a.dybs + (long)c * a.L);
      for (int l = beg + threadIdx.x; l < end; l += 256) {
        const float4 xv = xp[l], dv = dp[l];
        term(xv.x, dv.x); term(xv.y, dv.y); term(xv.z, dv.z); term(xv.w, dv.w);
      }
    }
  } else {
    const int per = (a.L + a.splits - 1) / a.splits;
    const int beg = sp * per;
    const int end = (beg + per < a.L) ? beg + per : a.L;
    for (int b = 0; b < a.B; ++b) {
      const float* xp = a.x[g] + b * a.xbs + (long)c * a.L;
      const float* dp = a.dy[g] + b * a.dybs + (long)c * a.L;
      for (int l = beg + threadIdx.x; l < end; l += 256) term(xp[l], dp[l]);
    }
  }
  s1 = cn_block_sum<double, 256>(s1, scratch);
  s2 = cn_block_sum<double, 256>(s2, scratch);
  if (threadIdx.x == 0) {
    double* p = a.part + (((long)g * a.C + c) * a.splits + sp) * 2;
    p[0] = s1;
    p[1] = s2;
  }
}

// grid = (blocks over L x G, C, B)
__global__ __launch_bounds__(256) void cn_bn_group_bwd_apply_kernel(const CnBnGroupArgs a, int bx_per_group) {
  const int g = blockIdx.x / bx_per_group, bx = blockIdx.x - g * bx_per_group;
  const int c = blockIdx.y, b = blockIdx.z;
  double s1 = 0.0, s2 = 0.0;
  const double* p = a.part + ((long)g * a.C + c) * a.splits * 2;
  for (int i = 0; i < a.splits; ++i) {
    s1 += p[2 * i];
    s2 += p[2 * i + 1];
  }
  if (bx == 0 && b == 0 && threadIdx.x == 0) {
    if (a.accumulate_params) {
      a.dgamma[g][c] += (float)s2;
      a.dbeta[g][c] += (float)s1;
    } else {
      a.dgamma[g][c] = (float)s2;
      a.dbeta[g][c] = (float)s1;
    }
  }
  if (a.dx[g] == nullptr) return;
  const double count = (double)a.B * a.L;
  const float m = a.mean[g][c], rs = a.rstd[g][c], ga = a.gamma[g][c], be = a.beta[g][c];
  const float c1 = a.training ? (float)(s1 / count) : 0.f, c2 = a.training ? (float)(s2 / count) : 0.f;
  const long r0 = (long)c * a.L;
  const float* xp = a.x[g] + b * a.xbs + r0;
  const float* dp = a.dy[g] + b * a.dybs + r0;
  float* dxp = a.dx[g] + b * a.dxbs + r0;
  const int acc = a.accumulate_dx[g];
  const float gs = ga * rs;
  auto grad = [&](float xv, float dv, float prev) {
    const float xh = (xv - m) * rs;
    float dz = dv;
    if (a.act == 1) dz *= cn_silu_grad(ga * xh + be);
    float gr = (dz - c1 - xh * c2) * gs;
    if (acc) gr += prev;
    return gr;
  };
  if (a.vec4) {
    const float4* x4 = reinterpret_cast<const float4*>(xp);
    const float4* d4 = reinterpret_cast<const float4*>(dp);
    float4* o4 = reinterpret_cast<float4*>(dxp);
    const float4 z = {0.f, 0.f, 0.f, 0.f};
    for (int l = bx * 256 + threadIdx.x; l < (a.L >> 2); l += bx_per_group * 256) {
      const float4 xv = x4[l], dv = d4[l];
      const float4 pv = acc ? o4[l] : z;
      float4 o;
      o.x = grad(xv.x, dv.x, pv.x); o.y = grad(xv.y, dv.y, pv.y);
      o.z = grad(xv.z, dv.z, pv.z); o.w = grad(xv.w, dv.w, pv.w);
      o4[l] = o;
    }
    return;
  }
  for (int l = bx * 256 + threadIdx.x; l < a.L; l += bx_per_group * 256) dxp[l] = grad(xp[l], dp[l], acc ? dxp[l] : 0.f);
}

// Host arrays of G device pointers; ws: G * cn_bn_workspace_doubles(C) doubles.
// sum_outputs == 0: ys[g] = act(bn_g(xs[g])) (res must be NULL); != 0: ys[0] = res + sum_g act(bn_g(xs[g])).
extern "C" int cn_bn_act_group_fwd_f32(int G, const float* const* xs, long xbs, const float* const* gammas,
                                       const float* const* betas, float* const* running_means,
                                       float* const* running_vars, const float* res, long rbs, float* const* ys,
                                       long ybs, float* const* means, float* const* rstds, double* ws, int B, int C,
                                       int L, int training, float momentum, float eps, int act, int sum_outputs,
                                       void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (G < 1 || G > BN_MAX_GROUPS || (!sum_outputs && res != nullptr)) return CN_ERR_ARG;
  if (B <= 0 || C <= 0 || L <= 0) return CN_OK;
  CnBnGroupArgs a = {};
  for (int g = 0; g < G; ++g) {
    a.x[g] = xs[g]; a.gamma[g] = gammas[g]; a.beta[g] = betas[g];
    a.running_mean[g] = running_means ? running_means[g] : nullptr;
    a.running_var[g] = running_vars ? running_vars[g] : nullptr;
    if (!training && (a.running_mean[g] == nullptr || a.running_var[g] == nullptr)) return CN_ERR_ARG;
    a.mean[g] = means[g]; a.rstd[g] = rstds[g];
    a.y[g] = sum_outputs ? ys[0] : ys[g];
  }
  a.xbs = xbs; a.ybs = ybs; a.rbs = rbs; a.res = res;
  a.G = G; a.B = B; a.C = C; a.L = L; a.act = act; a.training = training; a.sum_outputs = sum_outputs;
  a.eps = eps; a.momentum = momentum; a.part = ws;
  a.splits = bn_splits(C * G, L);
  a.vec4 = (L % 4 == 0) && xbs % 4 == 0 && ybs % 4 == 0 && (res == nullptr || rbs % 4 == 0) &&
           (reinterpret_cast<uintptr_t>(res) & 15) == 0;
  for (int g = 0; g < G; ++g)
    if ((reinterpret_cast<uintptr_t>(a.x[g]) | reinterpret_cast<uintptr_t>(a.y[g])) & 15) a.vec4 = 0;
  if (training)
    CN_LAUNCH(cn_bn_group_partial_kernel, dim3(C, a.splits, G), dim3(256), 0, stream, a);
  CN_LAUNCH(cn_bn_group_apply_kernel, plane_grid(B, C, L), dim3(256), 0, stream, a);
  return cn_check_launch();
}

// dys: per-group output gradients (the same pointer G times after a summed forward). dxs[g] may be NULL.
extern "C" int cn_bn_act_group_bwd_f32(int G, const float* const* xs, long xbs, const float* const* dys, long dybs,
                                       const float* const* means, const float* const* rstds,
                                       const float* const* gammas, const float* const* betas, float* const* dxs,
                                       long dxbs, const int* accumulate_dx, float* const* dgammas,
                                       float* const* dbetas, double* ws, int B, int C, int L, int training, int act,
                                       int accumulate_params, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (G < 1 || G > BN_MAX_GROUPS) return CN_ERR_ARG;
  if (B <= 0 || C <= 0 || L <= 0) return CN_OK;
  CnBnGroupArgs a = {};
  for (int g = 0; g < G; ++g) {
    a.x[g] = xs[g]; a.dy[g] = dys[g]; a.gamma[g] = gammas[g]; a.beta[g] = betas[g];
    a.mean[g] = const_cast<float*>(means[g]); a.rstd[g] = const_cast<float*>(rstds[g]);
    a.dx[g] = dxs ? dxs[g] : nullptr; a.dgamma[g] = dgammas[g]; a.dbeta[g] = dbetas[g];
    a.accumulate_dx[g] = accumulate_dx ? accumulate_dx[g] : 0;
  }
  a.xbs = xbs; a.dybs = dybs; a.dxbs = dxbs;
  a.G = G; a.B = B; a.C = C; a.L = L; a.act = act; a.training = training; a.accumulate_params = accumulate_params;
  a.part = ws;
  a.splits = bn_splits(C * G, L);
  a.vec4 = (L % 4 == 0) && xbs % 4 == 0 && dybs % 4 == 0 && dxbs % 4 == 0;
  for (int g = 0; g < G; ++g)
    if ((reinterpret_cast<uintptr_t>(a.x[g]) | reinterpret_cast<uintptr_t>(a.dy[g]) |
         reinterpret_cast<uintptr_t>(a.dx[g])) & 15)
      a.vec4 = 0;
  CN_LAUNCH(cn_bn_group_bwd_partial_kernel, dim3(C, a.splits, G), dim3(256), 0, stream, a);
  const dim3 pg = plane_grid(B, C, L);
  CN_LAUNCH(cn_bn_group_bwd_apply_kernel, dim3(pg.x * G, pg.y, pg.z), dim3(256), 0, stream, a, (int)pg.x);
  return cn_check_launch();
}

// ---------------------------------------------------------------------------
// Per-channel sum over [B][C][L] (bias gradients): out[c] (+)= sum x[b,c,l]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cn_channel_sum_kernel(const float* __restrict__ x, long xbs, int B, int C,
                                                            int L, int splits, float* __restrict__ out) {
  __shared__ double scratch[4];
  const int c = blockIdx.x, sp = blockIdx.y;
  const int per = (L + splits - 1) / splits;
  const int beg = sp * per;
  const int end = (beg + per < L) ? beg + per : L;
  double s = 0.0;
  for (int b = 0; b < B; ++b) {
    const float* xp = x + b * xbs + (long)c * L;
    for (int l = beg + threadIdx.x; l < end; l += 256) s += xp[l];
  }
  s = cn_block_sum<double, 256>(s, scratch);
  if (threadIdx.x == 0) atomicAdd(out + c, (float)s);
}

extern "C" int cn_channel_sum_f32(const float* x, long xbs, int B, int C, int L, float* out, int accumulate,
                                  void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (C <= 0) return CN_OK;
  if (!accumulate && hipMemsetAsync(out, 0, sizeof(float) * C, stream) != hipSuccess) return CN_ERR_LAUNCH;
  const int splits = bn_splits(C, L);
  CN_LAUNCH(cn_channel_sum_kernel, dim3(C, splits), dim3(256), 0, stream, x, xbs, B, C, L, splits, out);
  return cn_check_launch();
}

// ---------------------------------------------------------------------------
// LayerNorm over C for NCHW tensors: one lane per pixel, values strided by L.
//   y = (x - mu) * rstd * w[c] + b[c] (+ residual);  mu/rstd saved per pixel [B][L].
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cn_ln_fwd_kernel(const float* __restrict__ x, long xbs,
                                                       const float* __restrict__ w, const float* __restrict__ bvec,
                                                       const float* __restrict__ res, long rbs,
                                                       float* __restrict__ y, long ybs, float* __restrict__ mu,
                                                       float* __restrict__ rstd, int B, int C, int L, float eps) {
  const long p = blockIdx.x * 256L + threadIdx.x;
  if (p >= (long)B * L) return;
  const long b = p / L, l = p - b * L;
  const float* xp = x + b * xbs + l;
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += xp[(long)c * L];
  const float m = s / C;
  float v = 0.f;
  for (int c = 0; c < C; ++c) {
    const float d = xp[(long)c * L] - m;
    v += d * d;
  }
  const float rs = 1.0f / sqrtf(v / C + eps);
  mu[p] = m;
  rstd[p] = rs;
  float* yp = y + b * ybs + l;
  const float* rp = res ? res + b * rbs + l : nullptr;
  for (int c = 0; c < C; ++c) {
    float o = (xp[(long)c * L] - m) * rs * w[c] + bvec[c];
    if (rp) o += rp[(long)c * L];
    yp[(long)c * L] = o;
  }
}

// dx (+)= rstd * (g - mean_c(g) - xhat * mean_c(g*xhat)), g = dy*w; dw/db via block partials + atomics.
template <int MAXC>
__global__ __launch_bounds__(256) void cn_ln_bwd_kernel(const float* __restrict__ x, long xbs,
                                                       const float* __restrict__ dy, long dybs,
                                                       const float* __restrict__ w, const float* __restrict__ mu,
                                                       const float* __restrict__ rstd, float* __restrict__ dx,
                                                       long dxbs, float* __restrict__ dw, float* __restrict__ db,
                                                       int B, int C, int L, int accumulate_dx) {
  __shared__ float red[2][4][MAXC];
  const long p = blockIdx.x * 256L + threadIdx.x;
  const bool ok = p < (long)B * L;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  long b = 0, l = 0;
  float m = 0.f, rs = 0.f;
  if (ok) {
    b = p / L;
    l = p - b * L;
    m = mu[p];
    rs = rstd[p];
  }
  const float* xp = x + b * xbs + l;
  const float* dp = dy + b * dybs + l;
  float s1 = 0.f, s2 = 0.f;
  for (int c = 0; c < C; ++c) {
    float g = 0.f, xh = 0.f, d = 0.f;
    if (ok) {
      d = dp[(long)c * L];
      xh = (xp[(long)c * L] - m) * rs;
      g = d * w[c];
    }
    s1 += g;
    s2 += g * xh;
    // per-channel parameter gradients: wave reduce, then one LDS slot per wave
    const float dwv = cn_wave_sum(d * xh);
    const float dbv = cn_wave_sum(d);
    if (lane == 0) {
      red[0][wid][c] = dwv;
      red[1][wid][c] = dbv;
    }
  }
  if (ok) {
    const float a1 = s1 / C, a2 = s2 / C;
    float* dxp = dx + b * dxbs + l;
    for (int c = 0; c < C; ++c) {
      const float xh = (xp[(long)c * L] - m) * rs;
      float g = rs * (dp[(long)c * L] * w[c] - a1 - xh * a2);
      if (accumulate_dx) g += dxp[(long)c * L];
      dxp[(long)c * L] = g;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    atomicAdd(dw + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
    atomicAdd(db + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
  }
}

// ---- C <= 128: 64 pixels per block, the channels split over the 4 waves (CPW each) and held in registers, so x
// (and dy) are read ONCE and a 100x100 chip batch gives ~5 waves per SIMD instead of ~1. The per-pixel statistics
// are combined through LDS (two passes over the registers: mean, then centred variance, like the reference).
template <int CPW>
__global__ __launch_bounds__(256) void cn_ln_fwd_reg_kernel(const float* __restrict__ x, long xbs,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ bvec,
                                                           const float* __restrict__ res, long rbs,
                                                           float* __restrict__ y, long ybs, float* __restrict__ mu,
                                                           float* __restrict__ rstd, int B, int C, int L, float eps) {
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long p = blockIdx.x * 64L + lane;
  const bool ok = p < (long)B * L;
  const long b = ok ? p / L : 0, l = ok ? p - b * L : 0;
  const int c0 = wid * CPW;
  const float* xp = x + b * xbs + l + (long)c0 * L;
  float xv[CPW];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    xv[i] = (ok && c0 + i < C) ? xp[(long)i * L] : 0.f;
    s += xv[i];
  }
  red[0][wid][lane] = s;
  __syncthreads();
  const float m = (red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane]) / C;
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const float d = xv[i] - m;
    v += (c0 + i < C) ? d * d : 0.f;
  }
  red[1][wid][lane] = v;
  __syncthreads();
  const float rs = 1.0f / sqrtf((red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane]) / C + eps);
  if (!ok) return;
  if (wid == 0) {
    mu[p] = m;
    rstd[p] = rs;
  }
  float* yp = y + b * ybs + l + (long)c0 * L;
  const float* rp = res ? res + b * rbs + l + (long)c0 * L : nullptr;
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    if (c0 + i < C) {
      float o = (xv[i] - m) * rs * w[c0 + i] + bvec[c0 + i];
      if (rp) o += rp[(long)i * L];
      yp[(long)i * L] = o;
    }
  }
}

// Backward: tiles of 64 pixels (grid-stride), channels split over the 4 waves and held in registers. The parameter
// gradients are kept per lane across the block's tiles, reduced once per block and written to the workspace
// part[block][2][C] (same-address float atomics serialise at ~200 ns each: one per tile and channel was the
// whole cost of this kernel); cn_ln_param_finalize_kernel sums the partials.
template <int CPW>
__global__ __launch_bounds__(256) void cn_ln_bwd_reg_kernel(const float* __restrict__ x, long xbs,
                                                           const float* __restrict__ dy, long dybs,
                                                           const float* __restrict__ w, const float* __restrict__ mu,
                                                           const float* __restrict__ rstd, float* __restrict__ dx,
                                                           long dxbs, float* __restrict__ part, int B, int C, int L,
                                                           int accumulate_dx, int ntiles) {
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int c0 = wid * CPW;
  float dwacc[CPW], dbacc[CPW];
#pragma unroll
  for (int i = 0; i < CPW; ++i) dwacc[i] = dbacc[i] = 0.f;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long p = tile * 64L + lane;
    const bool ok = p < (long)B * L;
    const long b = ok ? p / L : 0, l = ok ? p - b * L : 0;
    const float m = ok ? mu[p] : 0.f, rs = ok ? rstd[p] : 0.f;
    const float* xp = x + b * xbs + l + (long)c0 * L;
    const float* dp = dy + b * dybs + l + (long)c0 * L;
    float xh[CPW], g[CPW];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
      const bool cok = ok && c0 + i < C;
      const float d = cok ? dp[(long)i * L] : 0.f;
      xh[i] = cok ? (xp[(long)i * L] - m) * rs : 0.f;
      g[i] = cok ? d * w[c0 + i] : 0.f;
      s1 += g[i];
      s2 += g[i] * xh[i];
      dwacc[i] = fmaf(d, xh[i], dwacc[i]);
      dbacc[i] += d;
    }
    __syncthreads();  // the previous tile's readers of red[] are done
    red[0][wid][lane] = s1;
    red[1][wid][lane] = s2;
    __syncthreads();
    if (ok) {
      const float a1 = (red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane]) / C;
      const float a2 = (red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane]) / C;
      float* dxp = dx + b * dxbs + l + (long)c0 * L;
#pragma unroll
      for (int i = 0; i < CPW; ++i) {
        if (c0 + i < C) {
          float o = rs * (g[i] - a1 - xh[i] * a2);
          if (accumulate_dx) o += dxp[(long)i * L];
          dxp[(long)i * L] = o;
        }
      }
    }
  }
  float* pw = part + (long)blockIdx.x * 2 * C;
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const float dwv = cn_wave_sum_to_lane63(dwacc[i]);
    const float dbv = cn_wave_sum_to_lane63(dbacc[i]);
    if (lane == 63 && c0 + i < C) {
      pw[c0 + i] = dwv;
      pw[C + c0 + i] = dbv;
    }
  }
}

// dw[c] += sum_blocks part[blk][0][c], db[c] += sum_blocks part[blk][1][c]; grid = (ceil(2C / 32), 16): a block
// sums 1/16 of the block list for 32 values (8 sub-slices, LDS reduce) and issues one atomic per value.
__global__ __launch_bounds__(256) void cn_ln_param_finalize_kernel(const float* __restrict__ part, int nblk, int C,
                                                                  float* __restrict__ dw, float* __restrict__ db) {
  __shared__ float red[8][32];
  const int v = blockIdx.x * 32 + (threadIdx.x & 31), slice = blockIdx.y * 8 + (threadIdx.x >> 5);
  float s = 0.f;
  if (v < 2 * C)
    for (int k = slice; k < nblk; k += 8 * gridDim.y) s += part[(long)k * 2 * C + v];
  red[threadIdx.x >> 5][threadIdx.x & 31] = s;
  __syncthreads();
  if ((threadIdx.x >> 5) == 0 && v < 2 * C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x & 31];
    atomicAdd(v < C ? dw + v : db + (v - C), t);
  }
}

extern "C" int cn_layernorm_c_fwd_f32(const float* x, long xbs, const float* w, const float* b, const float* res,
                                      long rbs, float* y, long ybs, float* mu, float* rstd, int B, int C, int L,
                                      float eps, void* stream) {
  const long P = (long)B * L;
  if (P <= 0) return CN_OK;
  if (C <= 128) {
    const dim3 grid((unsigned)((P + 63) / 64));
#define CN_LN_FWD(CPW_)                                                                                             \
  CN_LAUNCH((cn_ln_fwd_reg_kernel<CPW_>), grid, dim3(256), 0, (hipStream_t)stream, x, xbs, w, b, res, rbs, y, \
                     ybs, mu, rstd, B, C, L, eps)
    if (C <= 32) CN_LN_FWD(8);
    else if (C <= 64) CN_LN_FWD(16);
    else CN_LN_FWD(32);
#undef CN_LN_FWD
    return cn_check_launch();
  }
  CN_LAUNCH(cn_ln_fwd_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, xbs,
                     w, b, res, rbs, y, ybs, mu, rstd, B, C, L, eps);
  return cn_check_launch();
}

// Workspace of cn_layernorm_c_bwd_f32 in floats (per-block partial parameter gradients).
extern "C" int cn_layernorm_c_workspace_floats(int B, int C, int L) {
  const long ntiles = ((long)B * L + 63) / 64;
  const long nblk = ntiles < 2048 ? ntiles : 2048;
  return C <= 128 ? (int)(nblk * 2 * C) : 0;
}

// dw/db are ACCUMULATED: zero them first for a fresh gradient. ws: cn_layernorm_c_workspace_floats(B, C, L)
// floats of scratch (C <= 128; wider layers use the atomic kernel and ignore it).
extern "C" int cn_layernorm_c_bwd_f32(const float* x, long xbs, const float* dy, long dybs, const float* w,
                                      const float* mu, const float* rstd, float* dx, long dxbs, float* dw,
                                      float* db, int B, int C, int L, int accumulate_dx, float* ws, long ws_floats,
                                      void* stream) {
  const long P = (long)B * L;
  if (P <= 0) return CN_OK;
  if (C > 512) return CN_ERR_ARG;
  if (C <= 128) {
    const long ntiles = (P + 63) / 64;
    const int nblk = (int)(ntiles < 2048 ? ntiles : 2048);
    if (ws == nullptr || ws_floats < (long)nblk * 2 * C) return CN_ERR_ARG;
    const dim3 grid((unsigned)nblk);
#define CN_LN_BWD(CPW_)                                                                                           \
  CN_LAUNCH((cn_ln_bwd_reg_kernel<CPW_>), grid, dim3(256), 0, (hipStream_t)stream, x, xbs, dy, dybs, w, mu, \
                     rstd, dx, dxbs, ws, B, C, L, accumulate_dx, (int)ntiles)
    if (C <= 32) CN_LN_BWD(8);
    else if (C <= 64) CN_LN_BWD(16);
    else CN_LN_BWD(32);
#undef CN_LN_BWD
    CN_LAUNCH(cn_ln_param_finalize_kernel, dim3((2 * C + 31) / 32, 16), dim3(256), 0, (hipStream_t)stream, ws,
                       nblk, C, dw, db);
    return cn_check_launch();
  }
  CN_LAUNCH((cn_ln_bwd_kernel<512>), dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     x, xbs, dy, dybs, w, mu, rstd, dx, dxbs, dw, db, B, C, L, accumulate_dx);
  return cn_check_launch();
}


// ---------------------------------------------------------------------------------------------------------------
// Eval-mode BatchNorm folded into the convolution in front of it (ConvBlock2d, convolution.py:71-120, in inference
// mode): scale[c] = gamma[c] / sqrt(running_var[c] + eps), shift[c] = beta[c] - running_mean[c] * scale[c]
// (+ conv_bias[c] * scale[c] when the convolution has a bias). The packed weights are multiplied by `scale`
// (cn_pack_weights_scaled_bf16) and `shift` becomes the fused launch's bias.
// ---------------------------------------------------------------------------------------------------------------
__global__ void cn_bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                  const float* __restrict__ mean, const float* __restrict__ var,
                                  const float* __restrict__ conv_bias, float eps, int C, float* __restrict__ scale,
                                  float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float s = gamma[c] / sqrtf(var[c] + eps);
  scale[c] = s;
  shift[c] = beta[c] - mean[c] * s + (conv_bias != nullptr ? conv_bias[c] * s : 0.f);
}

extern "C" int cn_bn_fold_f32(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                              const float* conv_bias, float eps, int C, float* scale, float* shift, void* stream) {
  if (C <= 0) return CN_OK;
  if (!gamma || !beta || !running_mean || !running_var || !scale || !shift) return CN_ERR_ARG;
  CN_LAUNCH(cn_bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean,
            running_var, conv_bias, eps, C, scale, shift);
  return cn_check_launch();
}
