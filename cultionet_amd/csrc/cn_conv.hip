// Implicit-GEMM convolution family for gfx950 (MI355X), fp32 on the f32 MFMA pipe.
//
// One kernel serves every contraction on the TowerUNet path that gathers a
// K x (taps) neighbourhood of an NCHW tensor and reduces over input channels:
//   * Conv2d forward            (reference: torch.nn.Conv2d inside ConvBlock2d,
//                                /root/reference/src/cultionet/nn/modules/convolution.py:71-120)
//   * Conv2d backward-data      (autograd of the above; stride>1 handled as s*s parity classes)
//   * ConvTranspose2d forward   (convolution.py:45-68; s*s parity classes, no zero-stuffing)
//   * ConvTranspose2d backward-data
//   * 1x1 convs / the NA qkv+proj Linear layers (convolution.py:341-350) as T=1 taps.
//
// Mapping (im2col-free): a block owns MT consecutive flattened grid pixels of one
// image and NT output channels. Per K-chunk of 8 input channels it stages the
// halo rows [8][rows][pitch] and the packed weights [taps*8][NT] in LDS; MFMA
// v_mfma_f32_32x32x2_f32 has A = weights (rows = cout), B = pixels (cols), so
// the accumulator's lane index is the pixel and stores are 128-B coalesced
// along W of NCHW. fp32 MFMA is bit-for-bit an fp32 fma chain (no TF32 on gfx950).
#include "cn_common.h"
#include "cn_conv_geom.h"
#include "cn_profile.h"

#define KC 8
#define NI 12  // max staged plane = NI*256 floats per channel

template <int WAVES_N, int TN>
__global__ __launch_bounds__(256) void cn_conv_igemm_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ wp,
                                                           const float* __restrict__ bias,
                                                           float* __restrict__ y, const CnConvGeom g) {
  constexpr int WAVES_M = 4 / WAVES_N;
  constexpr int MT = WAVES_M * 64;
  constexpr int NT = WAVES_N * TN * 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* in_lds = smem;                 // [KC][plane]
  float* w_lds = smem + g.w_lds_off;    // [ntaps*KC][NT]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int b = blockIdx.x / g.tiles_per_img;
  const int m0 = (blockIdx.x - b * g.tiles_per_img) * MT;
  const int n0 = blockIdx.y * NT;
  const int Mimg = g.Hg * g.Wg;
  const int gy0 = m0 / g.Wg;
  const int iy_base = gy0 * g.is + g.min_dy;
  const int HWin = g.Hin * g.Win;

  // per-thread decode of the staged halo plane (same for every K-chunk)
  int goff[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int e = tid + i * 256;
    if (e < g.plane) {
      const int r = e / g.pitch, c = e - r * g.pitch;
      const int iy = iy_base + r, ix = c + g.min_dx;
      goff[i] = (iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Win) ? iy * g.Win + ix : -1;
    } else {
      goff[i] = -2;
    }
  }

  int pix_lds[2], out_off[2];
  bool pix_ok[2];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const int p = m0 + wm * 64 + tm * 32 + l31;
    pix_ok[tm] = p < Mimg;
    const int pc = pix_ok[tm] ? p : Mimg - 1;
    const int gy = pc / g.Wg, gx = pc - gy * g.Wg;
    pix_lds[tm] = ((gy - gy0) * g.is) * g.pitch + gx * g.is + half * g.plane;
    out_off[tm] = (gy * g.os + g.oy0) * g.Wout + gx * g.os + g.ox0;
  }

  f32x16 acc[TN][2];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tn][tm][r] = 0.f;

  const float* xb = x + (long)b * g.xbs;
  const int nw4 = g.ntaps * KC * (NT / 4);

  if (g.ntaps > 0) {
    for (int c0 = 0; c0 < g.Cin; c0 += KC) {
      __syncthreads();
      // ---- stage halo rows of KC input channels (zero outside the image / past Cin)
#pragma unroll
      for (int ci = 0; ci < KC; ++ci) {
        const bool cok = (c0 + ci) < g.Cin;
        const float* xc = xb + (long)(c0 + ci) * HWin;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          if (goff[i] != -2) {
            float v = 0.f;
            if (cok && goff[i] >= 0) v = xc[goff[i]];
            in_lds[ci * g.plane + tid + i * 256] = v;
          }
        }
      }
      // ---- stage packed weights [tap][c0..c0+KC)[n0..n0+NT)
      for (int f = tid; f < nw4; f += 256) {
        const int row = f / (NT / 4), c4 = f - row * (NT / 4);
        const int t = row / KC, ci = row - t * KC;
        const float4 v = *reinterpret_cast<const float4*>(
            wp + ((long)(g.wt[t] * g.Kpad + c0 + ci) * g.Npad + n0 + c4 * 4));
        *reinterpret_cast<float4*>(w_lds + row * NT + c4 * 4) = v;
      }
      __syncthreads();
      // ---- MFMA over (tap, channel pair)
      for (int t = 0; t < g.ntaps; ++t) {
        const int tapoff = (g.dy[t] - g.min_dy) * g.pitch + (g.dx[t] - g.min_dx);
        const float* wrow = w_lds + (t * KC + half) * NT + wn * (TN * 32) + l31;
#pragma unroll
        for (int cp = 0; cp < KC / 2; ++cp) {
          float a[TN], bb[2];
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) a[tn] = wrow[(2 * cp) * NT + tn * 32];
#pragma unroll
          for (int tm = 0; tm < 2; ++tm) bb[tm] = in_lds[pix_lds[tm] + (2 * cp) * g.plane + tapoff];
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
              acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tn], bb[tm], acc[tn][tm], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: D[i = cout][j = pixel]; lane = pixel -> coalesced along W
  float* yb = y + (long)b * g.ybs;
  const int HWout = g.Hout * g.Wout;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = n0 + wn * (TN * 32) + tn * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (co < g.Cout) {
        const float bv = g.has_bias ? bias[co] : 0.f;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          if (pix_ok[tm]) {
            float* dst = yb + (long)co * HWout + out_off[tm];
            float v = acc[tn][tm][r] + bv;
            if (g.accumulate) v += *dst;
            *dst = v;
          }
        }
      }
    }
  }
}

// Wp[t][k][n] = (k < K && n < N) ? w[k*sk + n*sn + t*st] : 0
__global__ void cn_pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int T, int K,
                                       int N, int Kpad, int Npad, long sk, long sn, long st) {
  const long total = (long)T * Kpad * Npad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i % Npad);
    const long r = i / Npad;
    const int k = (int)(r % Kpad);
    const int t = (int)(r / Kpad);
    wp[i] = (k < K && n < N) ? w[k * sk + n * sn + t * st] : 0.f;
  }
}

// --------------------------------------------------------------------------
// host side: geometry + launch
// --------------------------------------------------------------------------
static int cn_pick_nt(int cout) { return cout <= 32 ? 32 : (cout <= 64 ? 64 : 128); }

extern "C" int cn_conv_npad(int n_out) {
  const int nt = cn_pick_nt(n_out);
  return (n_out + nt - 1) / nt * nt;
}
extern "C" int cn_conv_kpad(int k_in) { return (k_in + KC - 1) / KC * KC; }

extern "C" int cn_pack_weights_f32(const float* w, float* wp, int T, int K, int N, long sk, long sn, long st,
                                   void* stream) {
  const int Kpad = cn_conv_kpad(K), Npad = cn_conv_npad(N);
  const long total = (long)T * Kpad * Npad;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(cn_pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wp, T, K, N,
                     Kpad, Npad, sk, sn, st);
  return cn_check_launch();
}

template <int WAVES_N, int TN>
static int cn_launch_igemm(const float* x, const float* wp, const float* bias, float* y, CnConvGeom g,
                           hipStream_t stream) {
  constexpr int MT = (4 / WAVES_N) * 64;
  constexpr int NT = WAVES_N * TN * 32;
  const int Mimg = g.Hg * g.Wg;
  if (Mimg <= 0 || g.B <= 0) return CN_OK;
  int rows_g = (MT + g.Wg - 2) / g.Wg + 1;
  if (rows_g > g.Hg) rows_g = g.Hg;
  int max_dy = g.min_dy, max_dx = g.min_dx;
  for (int t = 0; t < g.ntaps; ++t) {
    if (g.dy[t] > max_dy) max_dy = g.dy[t];
    if (g.dx[t] > max_dx) max_dx = g.dx[t];
  }
  g.rows_cap = (rows_g - 1) * g.is + (max_dy - g.min_dy) + 1;
  g.pitch = (g.Wg - 1) * g.is + (max_dx - g.min_dx) + 1;
  g.plane = g.rows_cap * g.pitch;
  if (g.plane > NI * 256) return CN_ERR_LDS;
  g.w_lds_off = (KC * g.plane + 3) / 4 * 4;
  g.tiles_per_img = (Mimg + MT - 1) / MT;
  const size_t lds = (size_t)(g.w_lds_off + (g.ntaps > 0 ? g.ntaps : 1) * KC * NT) * sizeof(float);
  if (lds > 160 * 1024) return CN_ERR_LDS;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)cn_conv_igemm_kernel<WAVES_N, TN>,
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  dim3 grid(g.B * g.tiles_per_img, (g.Cout + NT - 1) / NT);
  cn_prof_before(stream);
  hipLaunchKernelGGL((cn_conv_igemm_kernel<WAVES_N, TN>), grid, dim3(256), lds, stream, x, wp, bias, y, g);
  cn_prof_after(stream, NT == 128 ? 0 : 1, 2.0 * g.B * Mimg * (double)g.Cout * g.Cin * g.ntaps);
  return cn_check_launch();
}

int cn_conv_igemm_launch(const float* x, const float* wp, const float* bias, float* y, CnConvGeom g,
                         hipStream_t stream) {
  if (g.ntaps < 0 || g.ntaps > CN_MAX_TAPS) return CN_ERR_ARG;
  g.min_dy = 0;
  g.min_dx = 0;
  for (int t = 0; t < g.ntaps; ++t) {
    if (t == 0 || g.dy[t] < g.min_dy) g.min_dy = g.dy[t];
    if (t == 0 || g.dx[t] < g.min_dx) g.min_dx = g.dx[t];
  }
  g.Kpad = cn_conv_kpad(g.Cin);
  g.Npad = cn_conv_npad(g.Cout);
  const int nt = cn_pick_nt(g.Cout);
  if (nt == 32) return cn_launch_igemm<1, 1>(x, wp, bias, y, g, stream);
  if (nt == 64) return cn_launch_igemm<1, 2>(x, wp, bias, y, g, stream);
  return cn_launch_igemm<2, 2>(x, wp, bias, y, g, stream);
}

static inline int floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

// Conv2d forward. x [B,Cin,Hin,Win] (batch stride xbs), wp packed [KH*KW][Kpad(Cin)][Npad(Cout)],
// y [B,Cout,Hout,Wout] (batch stride ybs).
extern "C" int cn_conv2d_fwd_f32(const float* x, long xbs, const float* wp, const float* bias, float* y,
                                 long ybs, int B, int Cin, int Hin, int Win, int Cout, int KH, int KW,
                                 int stride, int pad, int dil, int accumulate, void* stream) {
  if (KH * KW > CN_MAX_TAPS || stride < 1 || dil < 1) return CN_ERR_ARG;
  CnConvGeom g = {};
  g.B = B; g.Cin = Cin; g.Hin = Hin; g.Win = Win; g.Cout = Cout;
  g.Hout = (Hin + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  g.Wout = (Win + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  g.xbs = xbs; g.ybs = ybs;
  g.Hg = g.Hout; g.Wg = g.Wout; g.is = stride; g.os = 1; g.oy0 = 0; g.ox0 = 0;
  g.ntaps = KH * KW;
  for (int ky = 0; ky < KH; ++ky)
    for (int kx = 0; kx < KW; ++kx) {
      const int t = ky * KW + kx;
      g.dy[t] = ky * dil - pad; g.dx[t] = kx * dil - pad; g.wt[t] = t;
    }
  g.accumulate = accumulate; g.has_bias = bias != nullptr;
  return cn_conv_igemm_launch(x, wp, bias, y, g, (hipStream_t)stream);
}

// Shared by Conv2d backward-data and ConvTranspose2d forward:
//   out[o] (+)= bias + sum_k src[(o + pad - k*dil)/s] * W[k]   where divisible,
// decomposed into s*s parity classes of the output grid so that no MAC is wasted.
// src [B,Csrc,Hs,Ws], out [B,Cdst,Ho,Wo]; wp packed [KH*KW][Kpad(Csrc)][Npad(Cdst)].
static int cn_scatter_conv(const float* src, long sbs, const float* wp, const float* bias, float* out, long obs,
                           int B, int Csrc, int Hs, int Ws, int Cdst, int Ho, int Wo, int KH, int KW,
                           int stride, int pad, int dil, int accumulate, hipStream_t stream) {
  if (KH * KW > CN_MAX_TAPS || stride < 1 || dil < 1) return CN_ERR_ARG;
  for (int py = 0; py < stride; ++py)
    for (int px = 0; px < stride; ++px) {
      CnConvGeom g = {};
      g.B = B; g.Cin = Csrc; g.Hin = Hs; g.Win = Ws; g.Cout = Cdst; g.Hout = Ho; g.Wout = Wo;
      g.xbs = sbs; g.ybs = obs;
      g.Hg = (Ho - py + stride - 1) / stride;
      g.Wg = (Wo - px + stride - 1) / stride;
      if (g.Hg <= 0 || g.Wg <= 0) continue;
      g.is = 1; g.os = stride; g.oy0 = py; g.ox0 = px;
      int nt = 0;
      for (int ky = 0; ky < KH; ++ky) {
        const int ny = py + pad - ky * dil;
        if (((ny % stride) + stride) % stride != 0) continue;
        for (int kx = 0; kx < KW; ++kx) {
          const int nx = px + pad - kx * dil;
          if (((nx % stride) + stride) % stride != 0) continue;
          g.dy[nt] = floordiv(ny, stride); g.dx[nt] = floordiv(nx, stride); g.wt[nt] = ky * KW + kx;
          ++nt;
        }
      }
      g.ntaps = nt;
      g.accumulate = accumulate; g.has_bias = bias != nullptr;
      const int rc = cn_conv_igemm_launch(src, wp, bias, out, g, stream);
      if (rc != CN_OK) return rc;
    }
  return CN_OK;
}

// Conv2d backward-data: dx [B,Cin,Hin,Win] (+)= conv^T(dy [B,Cout,Hout,Wout]); wp packed with K=Cout, N=Cin.
extern "C" int cn_conv2d_bwd_data_f32(const float* dy, long dybs, const float* wp_t, float* dx, long dxbs,
                                      int B, int Cin, int Hin, int Win, int Cout, int KH, int KW, int stride,
                                      int pad, int dil, int accumulate, void* stream) {
  const int Hout = (Hin + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  const int Wout = (Win + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  return cn_scatter_conv(dy, dybs, wp_t, nullptr, dx, dxbs, B, Cout, Hout, Wout, Cin, Hin, Win, KH, KW, stride,
                         pad, dil, accumulate, (hipStream_t)stream);
}

// ConvTranspose2d forward: y [B,Cout,Hout,Wout], Hout = (Hin-1)*s - 2*pad + K; wp packed with K=Cin, N=Cout.
extern "C" int cn_conv_transpose2d_fwd_f32(const float* x, long xbs, const float* wp, const float* bias,
                                           float* y, long ybs, int B, int Cin, int Hin, int Win, int Cout,
                                           int KH, int KW, int stride, int pad, int accumulate, void* stream) {
  const int Hout = (Hin - 1) * stride - 2 * pad + KH;
  const int Wout = (Win - 1) * stride - 2 * pad + KW;
  return cn_scatter_conv(x, xbs, wp, bias, y, ybs, B, Cin, Hin, Win, Cout, Hout, Wout, KH, KW, stride, pad, 1,
                         accumulate, (hipStream_t)stream);
}

// ConvTranspose2d backward-data: dx [B,Cin,Hin,Win] (+)= conv_stride_s(dy); wp_t packed with K=Cout, N=Cin.
extern "C" int cn_conv_transpose2d_bwd_data_f32(const float* dy, long dybs, const float* wp_t, float* dx,
                                                long dxbs, int B, int Cin, int Hin, int Win, int Cout, int KH,
                                                int KW, int stride, int pad, int accumulate, void* stream) {
  CnConvGeom g = {};
  const int Hout = (Hin - 1) * stride - 2 * pad + KH;
  const int Wout = (Win - 1) * stride - 2 * pad + KW;
  g.B = B; g.Cin = Cout; g.Hin = Hout; g.Win = Wout; g.Cout = Cin; g.Hout = Hin; g.Wout = Win;
  g.xbs = dybs; g.ybs = dxbs;
  g.Hg = Hin; g.Wg = Win; g.is = stride; g.os = 1;
  g.ntaps = KH * KW;
  if (g.ntaps > CN_MAX_TAPS) return CN_ERR_ARG;
  for (int ky = 0; ky < KH; ++ky)
    for (int kx = 0; kx < KW; ++kx) {
      const int t = ky * KW + kx;
      g.dy[t] = ky - pad; g.dx[t] = kx - pad; g.wt[t] = t;
    }
  g.accumulate = accumulate; g.has_bias = 0;
  return cn_conv_igemm_launch(dy, wp_t, nullptr, dx, g, (hipStream_t)stream);
}

// --------------------------------------------------------------------------
// nn.Conv3d(kernel (k,1,1)) of PreTimeReduction (models/nunet.py:18-57) as a 1x1 contraction over
// the [B, C*T, H, W] view: the (k,1,1) kernel becomes a banded [Cout*Tout] x [Cin*Tin] matrix.
//   w [Cout][Cin][k];  Wexp[(co,t')][(ci,t)] = w[co][ci][t-t'] for 0 <= t-t' < k, else 0.
// transposed == 0: packed for forward  (K = (ci,t),  N = (co,t'))
// transposed != 0: packed for bwd-data (K = (co,t'), N = (ci,t))
// --------------------------------------------------------------------------
__global__ void cn_pack_timeconv_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin,
                                        int Tin, int k, int transposed, int Kpad, int Npad) {
  const int Tout = Tin - k + 1;
  const int Kd = transposed ? Cout * Tout : Cin * Tin;
  const int Nd = transposed ? Cin * Tin : Cout * Tout;
  const long total = (long)Kpad * Npad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i % Npad), kk = (int)(i / Npad);
    float v = 0.f;
    if (kk < Kd && n < Nd) {
      const int in_idx = transposed ? n : kk, out_idx = transposed ? kk : n;
      const int ci = in_idx / Tin, t = in_idx - ci * Tin;
      const int co = out_idx / Tout, tp = out_idx - co * Tout;
      const int dt = t - tp;
      if (dt >= 0 && dt < k) v = w[((long)co * Cin + ci) * k + dt];
    }
    wp[i] = v;
  }
}

extern "C" int cn_pack_timeconv_f32(const float* w, float* wp, int Cout, int Cin, int Tin, int k, int transposed,
                                    void* stream) {
  const int Tout = Tin - k + 1;
  if (Tout < 1) return CN_ERR_ARG;
  const int Kd = transposed ? Cout * Tout : Cin * Tin, Nd = transposed ? Cin * Tin : Cout * Tout;
  const int Kpad = cn_conv_kpad(Kd), Npad = cn_conv_npad(Nd);
  const long total = (long)Kpad * Npad;
  hipLaunchKernelGGL(cn_pack_timeconv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, w, wp, Cout, Cin, Tin, k, transposed, Kpad, Npad);
  return cn_check_launch();
}

// dw[co][ci][dt] += sum_{t'} dWexp[(co,t')][(ci,t'+dt)]   (dWexp dense [Cout*Tout][Cin*Tin])
__global__ void cn_fold_timeconv_kernel(const float* __restrict__ dwexp, float* __restrict__ dw, int Cout, int Cin,
                                        int Tin, int k) {
  const int Tout = Tin - k + 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Cout * Cin * k) return;
  const int dt = i % k, ci = (i / k) % Cin, co = i / (k * Cin);
  float s = 0.f;
  for (int tp = 0; tp < Tout; ++tp) s += dwexp[(long)(co * Tout + tp) * (Cin * Tin) + ci * Tin + tp + dt];
  dw[i] += s;
}

extern "C" int cn_fold_timeconv_grad_f32(const float* dwexp, float* dw, int Cout, int Cin, int Tin, int k,
                                         void* stream) {
  const int n = Cout * Cin * k;
  hipLaunchKernelGGL(cn_fold_timeconv_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, dwexp, dw, Cout,
                     Cin, Tin, k);
  return cn_check_launch();
}
