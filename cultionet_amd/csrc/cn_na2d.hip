// Neighborhood attention core (q.k over a clamped/dilated KxK window -> softmax -> .v) on NCHW
// planes, forward and backward. HBM/L2-bound; one lane per (pixel, head).
//
// Restates natten 0.17.1's unfused NA2D (na2d_qk -> softmax -> na2d_av, no rel-pos bias,
// non-causal) used at /root/reference/src/cultionet/nn/modules/convolution.py:341-350.
// The surrounding qkv / proj Linear layers run as 1x1 convolutions on the MFMA kernel
// (cn_conv.hip), so q,k,v arrive as channel blocks of one [B, 3C, H, W] tensor:
// channel of (which, head, d) = which*C + head*D + d  (== natten's reshape(B,H,W,3,heads,D)).
#include "cn_common.h"

#define NA_K 3
#define NA_KK 9

// natten get_window_start (K = 3, n = 1)
__device__ __forceinline__ int na_window_start(int i, int len, int dil) {
  if (dil <= 1) return max(i - 1, 0) + ((i + 1 >= len) ? (len - i - 2) : 0);
  const int ni = i - dil;
  if (ni < 0) return i % dil;
  if (i + dil >= len) {
    const int imodd = i % dil;
    const int a = (len / dil) * dil;
    const int b = len - a;
    if (imodd < b) return len - b + imodd - 2 * dil;
    return a + imodd - NA_K * dil;
  }
  return ni;
}

// attn_drop (nn.Dropout on the soft-maxed logits): keep/(1-p) factor of tap t, recomputed from a counter hash
// (same splitmix64 stream as cn_dropout_f32) in forward and backward; 1.0 when drop is off.
__device__ __forceinline__ unsigned long long na_splitmix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ float na_keep(unsigned long long thresh, float scale, unsigned long long seed, long bh,
                                         int t, int HW, int p) {
  if (thresh == 0ull) return 1.0f;
  const unsigned long long i = ((unsigned long long)bh * NA_KK + t) * (unsigned long long)HW + p;
  return na_splitmix64(seed + i) >= thresh ? scale : 0.f;
}

// qkv [B][3C][H][W] (batch stride qbs); out [B][C][H][W]; attn [B][heads][9][H][W] (saved probs).
template <int D>
__global__ __launch_bounds__(256) void cn_na2d_fwd_kernel(const float* __restrict__ qkv, long qbs,
                                                         float* __restrict__ out, long obs,
                                                         float* __restrict__ attn, int B, int C, int heads, int H,
                                                         int W, int dil, float scale, unsigned long long dthresh,
                                                         float dscale, unsigned long long dseed_, const unsigned long long* __restrict__ dstep, int Dr) {
  const int DD = D > 0 ? D : Dr;  // D == 0: head dimension known at run time only (any width)
  const unsigned long long dseed = cn_step_seed(dseed_, dstep);
  const int HW = H * W;
  int bx, h, b;
  if (!cn_xcd_block((HW + 255) / 256, heads, ((HW + 255) / 256) * heads * B, bx, h, b)) return;
  const int p = bx * 256 + threadIdx.x;
  if (p >= HW) return;
  const int y = p / W, x = p - y * W;
  const int sy = na_window_start(y, H, dil), sx = na_window_start(x, W, dil);
  const float* qp = qkv + b * qbs + (long)(h * DD) * HW;
  const float* kp = qp + (long)C * HW;
  const float* vp = kp + (long)C * HW;
  // Loop order: channel plane outermost, the 9 taps inside. A tap-outer order revisits each of the D planes once
  // per tap with the other D-1 planes in between (reuse distance D * ~1.3 KB > the 32 KB L1): every tap then
  // misses to L2 and the kernel moves 9x the k / v bytes. Plane-outer touches a plane's few lines 9 times in a row.
  int kpix[NA_KK];
#pragma unroll
  for (int i = 0; i < NA_K; ++i)
#pragma unroll
    for (int j = 0; j < NA_K; ++j) kpix[i * NA_K + j] = (sy + i * dil) * W + sx + j * dil;
  float lg[NA_KK];
#pragma unroll
  for (int t = 0; t < NA_KK; ++t) lg[t] = 0.f;
#pragma unroll 4
  for (int d = 0; d < DD; ++d) {
    const float qd = qp[(long)d * HW + p] * scale;
    const float* kd = kp + (long)d * HW;
#pragma unroll
    for (int t = 0; t < NA_KK; ++t) lg[t] += qd * kd[kpix[t]];
  }
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NA_KK; ++t) mx = fmaxf(mx, lg[t]);
  float den = 0.f;
#pragma unroll
  for (int t = 0; t < NA_KK; ++t) {
    lg[t] = expf(lg[t] - mx);
    den += lg[t];
  }
  const float inv = 1.0f / den;
  float* ap = attn + ((long)(b * heads + h) * NA_KK) * HW + p;
#pragma unroll
  for (int t = 0; t < NA_KK; ++t) {
    lg[t] *= inv;
    if (attn) ap[(long)t * HW] = lg[t];
    lg[t] *= na_keep(dthresh, dscale, dseed, (long)b * heads + h, t, HW, p);
  }
  float* op = out + b * obs + (long)(h * DD) * HW + p;
#pragma unroll 4
  for (int d = 0; d < DD; ++d) {
    const float* vd = vp + (long)d * HW;
    float o = 0.f;
#pragma unroll
    for (int t = 0; t < NA_KK; ++t) o += lg[t] * vd[kpix[t]];
    op[(long)d * HW] = o;
  }
}

// Backward, query side: dP = dOut.v ; dS = P*(dP - sum P dP) ; dq = scale * sum dS*k ; saves dS.
template <int D>
__global__ __launch_bounds__(256) void cn_na2d_bwd_q_kernel(const float* __restrict__ qkv, long qbs,
                                                           const float* __restrict__ dout, long dobs,
                                                           const float* __restrict__ attn,
                                                           float* __restrict__ dattn, float* __restrict__ dqkv,
                                                           long dqbs, int B, int C, int heads, int H, int W, int dil,
                                                           float scale, unsigned long long dthresh, float dscale,
                                                           unsigned long long dseed_, const unsigned long long* __restrict__ dstep, int Dr) {
  const int DD = D > 0 ? D : Dr;  // D == 0: head dimension known at run time only (any width)
  const unsigned long long dseed = cn_step_seed(dseed_, dstep);
  const int HW = H * W;
  int bx, h, b;
  if (!cn_xcd_block((HW + 255) / 256, heads, ((HW + 255) / 256) * heads * B, bx, h, b)) return;
  const int p = bx * 256 + threadIdx.x;
  if (p >= HW) return;
  const int y = p / W, x = p - y * W;
  const int sy = na_window_start(y, H, dil), sx = na_window_start(x, W, dil);
  const float* kp = qkv + b * qbs + (long)(C + h * DD) * HW;
  const float* vp = kp + (long)C * HW;
  const float* dop = dout + b * dobs + (long)(h * DD) * HW + p;
  int kpix[NA_KK];
#pragma unroll
  for (int i = 0; i < NA_K; ++i)
#pragma unroll
    for (int j = 0; j < NA_K; ++j) kpix[i * NA_K + j] = (sy + i * dil) * W + sx + j * dil;
  const float* ap = attn + ((long)(b * heads + h) * NA_KK) * HW + p;
  float pr[NA_KK], dp[NA_KK];
#pragma unroll
  for (int t = 0; t < NA_KK; ++t) dp[t] = 0.f;
  // plane-outer (see the forward kernel): dP[t] = sum_d dOut[d] * v[d][tap t]
#pragma unroll 4
  for (int d = 0; d < DD; ++d) {
    const float gd = dop[(long)d * HW];
    const float* vd = vp + (long)d * HW;
#pragma unroll
    for (int t = 0; t < NA_KK; ++t) dp[t] += gd * vd[kpix[t]];
  }
  float dot = 0.f;
#pragma unroll
  for (int t = 0; t < NA_KK; ++t) {
    pr[t] = ap[(long)t * HW];
    dp[t] *= na_keep(dthresh, dscale, dseed, (long)b * heads + h, t, HW, p);
    dot += pr[t] * dp[t];
  }
  float* dap = dattn + ((long)(b * heads + h) * NA_KK) * HW + p;
#pragma unroll
  for (int t = 0; t < NA_KK; ++t) {
    dp[t] = pr[t] * (dp[t] - dot);  // dS
    dap[(long)t * HW] = dp[t];
  }
  float* dqp = dqkv + b * dqbs + (long)(h * DD) * HW + p;
#pragma unroll 4
  for (int d = 0; d < DD; ++d) {
    const float* kd = kp + (long)d * HW;
    float dq = 0.f;
#pragma unroll
    for (int t = 0; t < NA_KK; ++t) dq += dp[t] * kd[kpix[t]];
    dqp[(long)d * HW] = dq * scale;
  }
}

// Backward, key side (gather form, deterministic): for key pixel (y,x) visit every query whose
// window contains it: dk = scale * sum dS[q,t]*q[q] ; dv = sum P[q,t]*dOut[q].
template <int D>
__global__ __launch_bounds__(256) void cn_na2d_bwd_kv_kernel(const float* __restrict__ qkv, long qbs,
                                                            const float* __restrict__ dout, long dobs,
                                                            const float* __restrict__ attn,
                                                            const float* __restrict__ dattn,
                                                            float* __restrict__ dqkv, long dqbs, int B, int C,
                                                            int heads, int H, int W, int dil, float scale,
                                                            unsigned long long dthresh, float dscale,
                                                            unsigned long long dseed_, const unsigned long long* __restrict__ dstep, int Dr) {
  const int DD = D > 0 ? D : Dr;  // D == 0: head dimension known at run time only (any width)
  const unsigned long long dseed = cn_step_seed(dseed_, dstep);
  const int HW = H * W;
  int bx, h, b;
  if (!cn_xcd_block((HW + 255) / 256, heads, ((HW + 255) / 256) * heads * B, bx, h, b)) return;
  const int p = bx * 256 + threadIdx.x;
  if (p >= HW) return;
  const int y = p / W, x = p - y * W;
  const float* qp = qkv + b * qbs + (long)(h * DD) * HW;
  const float* dop = dout + b * dobs + (long)(h * DD) * HW;
  const float* ap = attn + ((long)(b * heads + h) * NA_KK) * HW;
  const float* dap = dattn + ((long)(b * heads + h) * NA_KK) * HW;
  // CH channels of the head at a time: all D of them when D is a compile-time constant, eight per sweep of the window
  // for a run-time head dimension (D == 0)
  constexpr int CH = D > 0 ? D : 8;
  float* dkp = dqkv + b * dqbs + (long)(C + h * DD) * HW + p;
  float* dvp = dkp + (long)C * HW;
  for (int d0 = 0; d0 < DD; d0 += CH) {
    float dk[CH], dv[CH];
#pragma unroll
    for (int d = 0; d < CH; ++d) {
      dk[d] = 0.f;
      dv[d] = 0.f;
    }
    for (int my = -2; my <= 2; ++my) {
      const int qy = y + my * dil;
      if (qy < 0 || qy >= H) continue;
      const int offy = y - na_window_start(qy, H, dil);
      if (offy < 0 || offy > 2 * dil) continue;  // same residue class => divisible by dil
      const int i = offy / dil;
      for (int mx = -2; mx <= 2; ++mx) {
        const int qx = x + mx * dil;
        if (qx < 0 || qx >= W) continue;
        const int offx = x - na_window_start(qx, W, dil);
        if (offx < 0 || offx > 2 * dil) continue;
        const int t = i * NA_K + offx / dil;
        const int qpix = qy * W + qx;
        const float ds = dap[(long)t * HW + qpix];
        const float pr = ap[(long)t * HW + qpix] * na_keep(dthresh, dscale, dseed, (long)b * heads + h, t, HW, qpix);
#pragma unroll
        for (int d = 0; d < CH; ++d) {
          if (D > 0 || d0 + d < DD) {
            dk[d] += ds * qp[(long)(d0 + d) * HW + qpix];
            dv[d] += pr * dop[(long)(d0 + d) * HW + qpix];
          }
        }
      }
    }
#pragma unroll
    for (int d = 0; d < CH; ++d) {
      if (D > 0 || d0 + d < DD) {
        dkp[(long)(d0 + d) * HW] = dk[d] * scale;
        dvp[(long)(d0 + d) * HW] = dv[d];
      }
    }
  }
}

// Head dimensions of the reference's widths (hidden 8 .. 128: h/4, h/2, h for powers of two) are compiled in; any other
// width (hidden 24, 40, 48, 96 ...) runs the same kernels with the head dimension as a run-time loop bound (D = 0).
#define NA_DISPATCH(D_, KERNEL, ...)                                                                          \
  switch (D_) {                                                                                               \
    case 2: CN_LAUNCH((KERNEL<2>), grid, dim3(256), 0, stream, __VA_ARGS__, D_); break;             \
    case 4: CN_LAUNCH((KERNEL<4>), grid, dim3(256), 0, stream, __VA_ARGS__, D_); break;             \
    case 8: CN_LAUNCH((KERNEL<8>), grid, dim3(256), 0, stream, __VA_ARGS__, D_); break;             \
    case 16: CN_LAUNCH((KERNEL<16>), grid, dim3(256), 0, stream, __VA_ARGS__, D_); break;           \
    case 32: CN_LAUNCH((KERNEL<32>), grid, dim3(256), 0, stream, __VA_ARGS__, D_); break;           \
    case 64: CN_LAUNCH((KERNEL<64>), grid, dim3(256), 0, stream, __VA_ARGS__, D_); break;           \
    default: CN_LAUNCH((KERNEL<0>), grid, dim3(256), 0, stream, __VA_ARGS__, D_); break;            \
  }

// kernel_size must be 3 (every NATTEN_PARAMS entry used by TowerUNet: unet_parts.py:19-40).
static unsigned long long na_thresh(float p) {
  if (!(p > 0.f)) return 0ull;
  const double t = (double)p * 18446744073709551616.0;
  return t >= 18446744073709551615.0 ? ~0ull : (unsigned long long)t;
}

extern "C" int cn_na2d_fwd_f32(const float* qkv, long qbs, float* out, long obs, float* attn, int B, int C,
                               int heads, int H, int W, int kernel_size, int dilation, float attn_drop,
                               unsigned long long seed, const unsigned long long* step, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (kernel_size != NA_K || heads <= 0 || C % heads != 0) return CN_ERR_ARG;
  if (kernel_size * dilation > H || kernel_size * dilation > W) return CN_ERR_ARG;
  const int D = C / heads;
  const float scale = 1.0f / sqrtf((float)D);
  dim3 grid(cn_xcd_grid((long)((H * W + 255) / 256) * heads * B));  // XCD-aware order, decoded in the kernels
  if (!(attn_drop >= 0.f && attn_drop < 1.f)) return CN_ERR_ARG;
  NA_DISPATCH(D, cn_na2d_fwd_kernel, qkv, qbs, out, obs, attn, B, C, heads, H, W, dilation, scale,
              na_thresh(attn_drop), 1.0f / (1.0f - attn_drop), seed, step);
  return cn_check_launch();
}

// dqkv [B][3C][H][W] is fully overwritten (dq, dk, dv); dattn is scratch of attn's size.
extern "C" int cn_na2d_bwd_f32(const float* qkv, long qbs, const float* dout, long dobs, const float* attn,
                               float* dattn, float* dqkv, long dqbs, int B, int C, int heads, int H, int W,
                               int kernel_size, int dilation, float attn_drop, unsigned long long seed,
                               const unsigned long long* step, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (kernel_size != NA_K || heads <= 0 || C % heads != 0) return CN_ERR_ARG;
  const int D = C / heads;
  const float scale = 1.0f / sqrtf((float)D);
  dim3 grid(cn_xcd_grid((long)((H * W + 255) / 256) * heads * B));  // XCD-aware order, decoded in the kernels
  if (!(attn_drop >= 0.f && attn_drop < 1.f)) return CN_ERR_ARG;
  const unsigned long long th = na_thresh(attn_drop);
  const float ds = 1.0f / (1.0f - attn_drop);
  NA_DISPATCH(D, cn_na2d_bwd_q_kernel, qkv, qbs, dout, dobs, attn, dattn, dqkv, dqbs, B, C, heads, H, W, dilation,
              scale, th, ds, seed, step);
  NA_DISPATCH(D, cn_na2d_bwd_kv_kernel, qkv, qbs, dout, dobs, attn, dattn, dqkv, dqbs, B, C, heads, H, W, dilation,
              scale, th, ds, seed, step);
  return cn_check_launch();
}
