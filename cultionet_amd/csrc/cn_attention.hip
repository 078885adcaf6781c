// Spatial-channel attention (CBAM-style) of ResidualAConv with attention_weights="spatial_channel"
// (reference: nn/modules/attention.py:12-126, applied at nn/modules/convolution.py:388-393):
//   channel:  ca[b,c]  = sigmoid( fc1(mean_hw x) + fc2(max_hw x) ),  fc = 1x1 conv C->C/2 -> SiLU -> 1x1 conv C/2->C
//   spatial:  sa[b,hw] = sigmoid( conv3x3_{2->1}( [mean_c x, max_c x] ) )        (the 3x3 conv runs on cn_thin_*)
//   out      *= 1 + gamma * 0.5 * (ca + sa)
// All HBM-bound streaming / reduction kernels over NCHW planes; the C x C/2 matrices are tiny (one block per sample).
#include "cn_common.h"

// ---------------------------------------------------------------------------
// Pools. (1) per (b, c): mean and max over the L = H*W pixels (+ index of the first maximum);
//        (2) per (b, pixel): mean and max over the C channels (+ channel of the first maximum) -> pooled [B][2][L].
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cn_sca_hw_pool_kernel(const float* __restrict__ x, long xbs, int C, int L,
                                                            float* __restrict__ avg, float* __restrict__ mx,
                                                            int* __restrict__ idx) {
  __shared__ float sv[256];
  __shared__ int si[256];
  __shared__ float scratch[4];
  const int c = blockIdx.x, b = blockIdx.y;
  const float* xp = x + b * xbs + (long)c * L;
  float s = 0.f, m = -INFINITY;
  int mi = 0;
  for (int l = threadIdx.x; l < L; l += 256) {
    const float v = xp[l];
    s += v;
    if (v > m) { m = v; mi = l; }
  }
  s = cn_block_sum<float, 256>(s, scratch);
  sv[threadIdx.x] = m;
  si[threadIdx.x] = mi;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      const float o = sv[threadIdx.x + off];
      const int oi = si[threadIdx.x + off];
      if (o > sv[threadIdx.x] || (o == sv[threadIdx.x] && oi < si[threadIdx.x])) {
        sv[threadIdx.x] = o;
        si[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    avg[b * C + c] = s / L;
    mx[b * C + c] = sv[0];
    idx[b * C + c] = si[0];
  }
}

__global__ __launch_bounds__(256) void cn_sca_c_pool_kernel(const float* __restrict__ x, long xbs, int C, int L,
                                                           float* __restrict__ pooled, int* __restrict__ cidx) {
  const int l = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (l >= L) return;
  const float* xp = x + b * xbs + l;
  float s = 0.f, m = -INFINITY;
  int mi = 0;
  for (int c = 0; c < C; ++c) {
    const float v = xp[(long)c * L];
    s += v;
    if (v > m) { m = v; mi = c; }
  }
  pooled[((long)b * 2 + 0) * L + l] = s / C;
  pooled[((long)b * 2 + 1) * L + l] = m;
  cidx[(long)b * L + l] = mi;
}

extern "C" int cn_sca_pool_fwd_f32(const float* x, long xbs, int B, int C, int L, float* avg, float* mx, int* idx,
                                   float* pooled, int* cidx, void* stream) {
  if (B <= 0 || C <= 0 || L <= 0) return CN_OK;
  CN_LAUNCH(cn_sca_hw_pool_kernel, dim3(C, B), dim3(256), 0, (hipStream_t)stream, x, xbs, C, L, avg, mx, idx);
  CN_LAUNCH(cn_sca_c_pool_kernel, dim3((L + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, x, xbs, C, L,
                     pooled, cidx);
  return cn_check_launch();
}

// d x[b,c,l] (+)= davg[b,c]/L + [l == idx[b,c]] dmx[b,c] + dpooled[b,0,l]/C + [c == cidx[b,l]] dpooled[b,1,l]
__global__ __launch_bounds__(256) void cn_sca_pool_bwd_kernel(const float* __restrict__ davg,
                                                             const float* __restrict__ dmx,
                                                             const int* __restrict__ idx,
                                                             const float* __restrict__ dpooled,
                                                             const int* __restrict__ cidx, float* __restrict__ dx,
                                                             long dxbs, int C, int L, int accumulate) {
  const int c = blockIdx.y, b = blockIdx.z;
  const float da = davg[b * C + c] / L, dm = dmx[b * C + c];
  const int mi = idx[b * C + c];
  float* dp = dx + b * dxbs + (long)c * L;
  for (int l = blockIdx.x * 256 + threadIdx.x; l < L; l += gridDim.x * 256) {
    float g = da + (l == mi ? dm : 0.f) + dpooled[((long)b * 2) * L + l] / C +
              (cidx[(long)b * L + l] == c ? dpooled[((long)b * 2 + 1) * L + l] : 0.f);
    if (accumulate) g += dp[l];
    dp[l] = g;
  }
}

extern "C" int cn_sca_pool_bwd_f32(const float* davg, const float* dmx, const int* idx, const float* dpooled,
                                   const int* cidx, float* dx, long dxbs, int B, int C, int L, int accumulate,
                                   void* stream) {
  if (B <= 0 || C <= 0 || L <= 0) return CN_OK;
  int bx = (L + 1023) / 1024;
  if (bx < 1) bx = 1;
  CN_LAUNCH(cn_sca_pool_bwd_kernel, dim3(bx, C, B), dim3(256), 0, (hipStream_t)stream, davg, dmx, idx, dpooled,
                     cidx, dx, dxbs, C, L, accumulate);
  return cn_check_launch();
}

// ---------------------------------------------------------------------------
// Channel MLPs (one block per sample; C <= 1024). w1 [Ch][C], w2 [C][Ch] = the 1x1 conv weights, no biases.
//   hpre_*[b][j] = sum_c w1[j][c] v[c];  ca[b][c] = sigmoid( sum_j w2a[c][j] silu(hpre_a[j]) + w2m[c][j] silu(hpre_m[j]) )
// ---------------------------------------------------------------------------
#define SCA_MAXC 1024
__global__ __launch_bounds__(256) void cn_sca_mlp_fwd_kernel(const float* __restrict__ avg, const float* __restrict__ mx,
                                                            const float* __restrict__ w1a,
                                                            const float* __restrict__ w2a,
                                                            const float* __restrict__ w1m,
                                                            const float* __restrict__ w2m, float* __restrict__ hpre_a,
                                                            float* __restrict__ hpre_m, float* __restrict__ ca, int C,
                                                            int Ch) {
  __shared__ float va[SCA_MAXC], vm[SCA_MAXC], ha[SCA_MAXC / 2], hm[SCA_MAXC / 2];
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    va[c] = avg[b * C + c];
    vm[c] = mx[b * C + c];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < Ch; j += 256) {
    float sa = 0.f, sm = 0.f;
    for (int c = 0; c < C; ++c) {
      sa = fmaf(w1a[(long)j * C + c], va[c], sa);
      sm = fmaf(w1m[(long)j * C + c], vm[c], sm);
    }
    hpre_a[b * Ch + j] = sa;
    hpre_m[b * Ch + j] = sm;
    ha[j] = cn_silu(sa);
    hm[j] = cn_silu(sm);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f, t = 0.f;
    for (int j = 0; j < Ch; ++j) {
      s = fmaf(w2a[(long)c * Ch + j], ha[j], s);
      t = fmaf(w2m[(long)c * Ch + j], hm[j], t);
    }
    ca[b * C + c] = cn_sigmoid(s + t);
  }
}

extern "C" int cn_sca_mlp_fwd_f32(const float* avg, const float* mx, const float* w1a, const float* w2a,
                                  const float* w1m, const float* w2m, float* hpre_a, float* hpre_m, float* ca, int B,
                                  int C, int Ch, void* stream) {
  if (C > SCA_MAXC || Ch > SCA_MAXC / 2 || Ch < 1) return CN_ERR_ARG;
  if (B <= 0) return CN_OK;
  CN_LAUNCH(cn_sca_mlp_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, avg, mx, w1a, w2a, w1m, w2m,
                     hpre_a, hpre_m, ca, C, Ch);
  return cn_check_launch();
}

// Backward of the two MLPs + the sigmoid. Weight gradients are ACCUMULATED (atomics over the samples).
__global__ __launch_bounds__(256) void cn_sca_mlp_bwd_kernel(
    const float* __restrict__ avg, const float* __restrict__ mx, const float* __restrict__ w1a,
    const float* __restrict__ w2a, const float* __restrict__ w1m, const float* __restrict__ w2m,
    const float* __restrict__ hpre_a, const float* __restrict__ hpre_m, const float* __restrict__ ca,
    const float* __restrict__ dca, float* __restrict__ dw1a, float* __restrict__ dw2a, float* __restrict__ dw1m,
    float* __restrict__ dw2m, float* __restrict__ davg, float* __restrict__ dmx, int C, int Ch) {
  __shared__ float va[SCA_MAXC], vm[SCA_MAXC], dpre[SCA_MAXC];
  __shared__ float ha[SCA_MAXC / 2], hm[SCA_MAXC / 2], dha[SCA_MAXC / 2], dhm[SCA_MAXC / 2];
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    va[c] = avg[b * C + c];
    vm[c] = mx[b * C + c];
    const float a = ca[b * C + c];
    dpre[c] = dca[b * C + c] * a * (1.f - a);
  }
  for (int j = threadIdx.x; j < Ch; j += 256) {
    ha[j] = cn_silu(hpre_a[b * Ch + j]);
    hm[j] = cn_silu(hpre_m[b * Ch + j]);
  }
  __syncthreads();
  // second layers: dW2[c][j] += dpre[c] h[j];  dh[j] = sum_c W2[c][j] dpre[c]
  for (int i = threadIdx.x; i < C * Ch; i += 256) {
    const int c = i / Ch, j = i - c * Ch;
    atomicAdd(dw2a + i, dpre[c] * ha[j]);
    atomicAdd(dw2m + i, dpre[c] * hm[j]);
  }
  for (int j = threadIdx.x; j < Ch; j += 256) {
    float sa = 0.f, sm = 0.f;
    for (int c = 0; c < C; ++c) {
      sa = fmaf(w2a[(long)c * Ch + j], dpre[c], sa);
      sm = fmaf(w2m[(long)c * Ch + j], dpre[c], sm);
    }
    dha[j] = sa * cn_silu_grad(hpre_a[b * Ch + j]);
    dhm[j] = sm * cn_silu_grad(hpre_m[b * Ch + j]);
  }
  __syncthreads();
  // first layers: dW1[j][c] += dh[j] v[c];  dv[c] = sum_j W1[j][c] dh[j]
  for (int i = threadIdx.x; i < C * Ch; i += 256) {
    const int j = i / C, c = i - j * C;
    atomicAdd(dw1a + i, dha[j] * va[c]);
    atomicAdd(dw1m + i, dhm[j] * vm[c]);
  }
  for (int c = threadIdx.x; c < C; c += 256) {
    float sa = 0.f, sm = 0.f;
    for (int j = 0; j < Ch; ++j) {
      sa = fmaf(w1a[(long)j * C + c], dha[j], sa);
      sm = fmaf(w1m[(long)j * C + c], dhm[j], sm);
    }
    davg[b * C + c] = sa;
    dmx[b * C + c] = sm;
  }
}

extern "C" int cn_sca_mlp_bwd_f32(const float* avg, const float* mx, const float* w1a, const float* w2a,
                                  const float* w1m, const float* w2m, const float* hpre_a, const float* hpre_m,
                                  const float* ca, const float* dca, float* dw1a, float* dw2a, float* dw1m,
                                  float* dw2m, float* davg, float* dmx, int B, int C, int Ch, void* stream) {
  if (C > SCA_MAXC || Ch > SCA_MAXC / 2 || Ch < 1) return CN_ERR_ARG;
  if (B <= 0) return CN_OK;
  CN_LAUNCH(cn_sca_mlp_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, avg, mx, w1a, w2a, w1m, w2m,
                     hpre_a, hpre_m, ca, dca, dw1a, dw2a, dw1m, dw2m, davg, dmx, C, Ch);
  return cn_check_launch();
}

// ---------------------------------------------------------------------------
// y = out * (1 + gamma * 0.5 * (ca[b,c] + sigmoid(sconv[b,l])))
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cn_sca_apply_fwd_kernel(const float* __restrict__ out, long obs,
                                                              const float* __restrict__ ca,
                                                              const float* __restrict__ sconv,
                                                              const float* __restrict__ gamma, float* __restrict__ y,
                                                              long ybs, int C, int L) {
  const int c = blockIdx.y, b = blockIdx.z;
  const float g = 0.5f * gamma[0], a = ca[b * C + c];
  const float* op = out + b * obs + (long)c * L;
  const float* sp = sconv + (long)b * L;
  float* yp = y + b * ybs + (long)c * L;
  for (int l = blockIdx.x * 256 + threadIdx.x; l < L; l += gridDim.x * 256)
    yp[l] = op[l] * (1.f + g * (a + cn_sigmoid(sp[l])));
}

extern "C" int cn_sca_apply_fwd_f32(const float* out, long obs, const float* ca, const float* sconv,
                                    const float* gamma, float* y, long ybs, int B, int C, int L, void* stream) {
  if (B <= 0 || C <= 0 || L <= 0) return CN_OK;
  int bx = (L + 1023) / 1024;
  if (bx < 1) bx = 1;
  CN_LAUNCH(cn_sca_apply_fwd_kernel, dim3(bx, C, B), dim3(256), 0, (hipStream_t)stream, out, obs, ca, sconv,
                     gamma, y, ybs, C, L);
  return cn_check_launch();
}

// Backward, pass A (block per (c, b)): dout (+)= dy * att;  dca[b,c] = 0.5 gamma sum_l dy out;
//                                      dgamma += 0.5 sum_l dy out (ca + sa)
__global__ __launch_bounds__(256) void cn_sca_apply_bwd_a_kernel(const float* __restrict__ dy, long dybs,
                                                                const float* __restrict__ out, long obs,
                                                                const float* __restrict__ ca,
                                                                const float* __restrict__ sconv,
                                                                const float* __restrict__ gamma,
                                                                float* __restrict__ dout, long dobs, int accumulate,
                                                                float* __restrict__ dca, float* __restrict__ dgpart,
                                                                int C, int L) {
  __shared__ float scratch[4];
  const int c = blockIdx.x, b = blockIdx.y;
  const float g = 0.5f * gamma[0], a = ca[b * C + c];
  const float* dp = dy + b * dybs + (long)c * L;
  const float* op = out + b * obs + (long)c * L;
  const float* sp = sconv + (long)b * L;
  float* dop = dout ? dout + b * dobs + (long)c * L : nullptr;
  float s1 = 0.f, s2 = 0.f;
  for (int l = threadIdx.x; l < L; l += 256) {
    const float sa = cn_sigmoid(sp[l]);
    const float d = dp[l], t = d * op[l];
    s1 += t;
    s2 += t * (a + sa);
    if (dop) {
      float v = d * (1.f + g * (a + sa));
      if (accumulate) v += dop[l];
      dop[l] = v;
    }
  }
  s1 = cn_block_sum<float, 256>(s1, scratch);
  s2 = cn_block_sum<float, 256>(s2, scratch);
  if (threadIdx.x == 0) {
    dca[b * C + c] = g * s1;
    dgpart[b * C + c] = 0.5f * s2;  // summed by cn_sca_sum_kernel (same-address atomics would serialise)
  }
}

// pass B (lane per (b, pixel)): dsconv[b,l] = 0.5 gamma sa (1 - sa) sum_c dy out
__global__ __launch_bounds__(256) void cn_sca_apply_bwd_b_kernel(const float* __restrict__ dy, long dybs,
                                                                const float* __restrict__ out, long obs,
                                                                const float* __restrict__ sconv,
                                                                const float* __restrict__ gamma,
                                                                float* __restrict__ dsconv, int C, int L) {
  const int l = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (l >= L) return;
  const float* dp = dy + b * dybs + l;
  const float* op = out + b * obs + l;
  float s = 0.f;
  for (int c = 0; c < C; ++c) s = fmaf(dp[(long)c * L], op[(long)c * L], s);
  const float sa = cn_sigmoid(sconv[(long)b * L + l]);
  dsconv[(long)b * L + l] = 0.5f * gamma[0] * sa * (1.f - sa) * s;
}

__global__ __launch_bounds__(256) void cn_sca_sum_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
  __shared__ float scratch[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += v[i];
  s = cn_block_sum<float, 256>(s, scratch);
  if (threadIdx.x == 0) out[0] += s;
}

// dout nullable. dgamma is ACCUMULATED; dca / dsconv are overwritten. scratch: B*C floats.
extern "C" int cn_sca_apply_bwd_f32(const float* dy, long dybs, const float* out, long obs, const float* ca,
                                    const float* sconv, const float* gamma, float* dout, long dobs,
                                    int accumulate_dout, float* dca, float* dsconv, float* dgamma, float* scratch,
                                    int B, int C, int L, void* stream) {
  if (B <= 0 || C <= 0 || L <= 0) return CN_OK;
  CN_LAUNCH(cn_sca_apply_bwd_a_kernel, dim3(C, B), dim3(256), 0, (hipStream_t)stream, dy, dybs, out, obs, ca,
                     sconv, gamma, dout, dobs, accumulate_dout, dca, scratch, C, L);
  CN_LAUNCH(cn_sca_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, B * C, dgamma);
  CN_LAUNCH(cn_sca_apply_bwd_b_kernel, dim3((L + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, dy, dybs,
                     out, obs, sconv, gamma, dsconv, C, L);
  return cn_check_launch();
}
