// Direct (VALU) 3x3 "same" convolutions for thin layers: the head streams of TowerUNetFinal
// (reference: nn/modules/unet_parts.py:196-224 StreamConv2d, :227-309 TowerUNetFinal):
//   128 -> 3 (x3 streams, same input)   => one pass producing 9 channels
//   3 -> 1   (x3 streams, own inputs)   => one grouped pass over a 9-channel tensor
//   3 -> 3   (fuse conv)
// These are HBM-bound (<= 81 MACs per input element): padding them to 32-wide MFMA tiles wastes > 90% of the
// matrix pipe and re-reads the 128-channel input once per stream. Here a lane owns one pixel, the weights are
// wave-uniform (scalar loads, SGPR operands of v_fmac) and the input is read once for all streams.
//
// Weight sets: up to 3 tensors [CP][Cin][3][3] (the separate nn.Conv2d parameters of the streams).
//   GROUPED == false: every set sees all Cin input channels; output channel = set*CP + c.
//   GROUPED == true : set g sees input channels [g*Cin, (g+1)*Cin) of an NG*Cin-channel tensor.
#include "cn_common.h"

namespace {

struct Taps {
  int off[9];
  bool ok[9];
};

// neighbour offsets of pixel (oy, ox) for tap t = (ky, kx): (oy + sgn*(ky-1)*dil, ox + sgn*(kx-1)*dil)
__device__ __forceinline__ Taps cn_thin_taps(int oy, int ox, int H, int W, int dil, int sgn) {
  Taps tp;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int iy = oy + sgn * (t / 3 - 1) * dil, ix = ox + sgn * (t % 3 - 1) * dil;
    tp.ok[t] = iy >= 0 && iy < H && ix >= 0 && ix < W;
    tp.off[t] = tp.ok[t] ? iy * W + ix : 0;
  }
  return tp;
}

#define CN_THIN_W(g) ((g) == 0 ? w0 : ((g) == 1 ? w1 : w2))
#define CN_THIN_B(g) ((g) == 0 ? b0 : ((g) == 1 ? b1 : b2))
#define CN_THIN_DW(g) ((g) == 0 ? dw0 : ((g) == 1 ? dw1 : dw2))

template <int NG, int CP, bool GROUPED>
__global__ __launch_bounds__(256) void cn_thin_fwd_kernel(const float* __restrict__ x, long xbs,
                                                         const float* __restrict__ w0, const float* __restrict__ w1,
                                                         const float* __restrict__ w2, const float* __restrict__ b0,
                                                         const float* __restrict__ b1, const float* __restrict__ b2,
                                                         float* __restrict__ y, long ybs, int Cin, int H, int W,
                                                         int dil) {
  const int HW = H * W;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const bool live = pix < HW;
  const int pc = live ? pix : 0;
  const int oy = pc / W, ox = pc - oy * W;
  const Taps tp = cn_thin_taps(oy, ox, H, W, dil, 1);
  const float* xb = x + (long)blockIdx.y * xbs;
  float acc[NG * CP];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int c = 0; c < CP; ++c) acc[g * CP + c] = CN_THIN_B(g) != nullptr ? CN_THIN_B(g)[c] : 0.f;

  if (!GROUPED) {
    for (int ci = 0; ci < Cin; ++ci) {
      const float* xc = xb + (long)ci * HW;
      float xv[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) xv[t] = tp.ok[t] ? xc[tp.off[t]] : 0.f;
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int c = 0; c < CP; ++c) {
          const float* wr = CN_THIN_W(g) + ((long)c * Cin + ci) * 9;
#pragma unroll
          for (int t = 0; t < 9; ++t) acc[g * CP + c] = fmaf(xv[t], wr[t], acc[g * CP + c]);
        }
    }
  } else {
#pragma unroll
    for (int g = 0; g < NG; ++g)
      for (int ci = 0; ci < Cin; ++ci) {
        const float* xc = xb + (long)(g * Cin + ci) * HW;
        float xv[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) xv[t] = tp.ok[t] ? xc[tp.off[t]] : 0.f;
#pragma unroll
        for (int c = 0; c < CP; ++c) {
          const float* wr = CN_THIN_W(g) + ((long)c * Cin + ci) * 9;
#pragma unroll
          for (int t = 0; t < 9; ++t) acc[g * CP + c] = fmaf(xv[t], wr[t], acc[g * CP + c]);
        }
      }
  }
  if (live) {
    float* yb = y + (long)blockIdx.y * ybs + pix;
#pragma unroll
    for (int o = 0; o < NG * CP; ++o) yb[(long)o * HW] = acc[o];
  }
}

// dx[b][ci][p] (+)= sum_{co,t} dy[b][co][p - (t-1)*dil] * w[co][ci][t]
template <int NG, int CP, bool GROUPED>
__global__ __launch_bounds__(256) void cn_thin_bwd_data_kernel(const float* __restrict__ dy, long dybs,
                                                              const float* __restrict__ w0,
                                                              const float* __restrict__ w1,
                                                              const float* __restrict__ w2, float* __restrict__ dx,
                                                              long dxbs, int Cin, int H, int W, int dil,
                                                              int accumulate) {
  const int HW = H * W;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const bool live = pix < HW;
  const int pc = live ? pix : 0;
  const int oy = pc / W, ox = pc - oy * W;
  const Taps tp = cn_thin_taps(oy, ox, H, W, dil, -1);
  const float* dyb = dy + (long)blockIdx.y * dybs;
  float* dxb = dx + (long)blockIdx.y * dxbs + pix;

  if (!GROUPED) {
    float dv[NG * CP][9];
#pragma unroll
    for (int o = 0; o < NG * CP; ++o)
#pragma unroll
      for (int t = 0; t < 9; ++t) dv[o][t] = tp.ok[t] ? dyb[(long)o * HW + tp.off[t]] : 0.f;
    for (int ci = 0; ci < Cin; ++ci) {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int c = 0; c < CP; ++c) {
          const float* wr = CN_THIN_W(g) + ((long)c * Cin + ci) * 9;
#pragma unroll
          for (int t = 0; t < 9; ++t) s = fmaf(dv[g * CP + c][t], wr[t], s);
        }
      if (live) {
        float* d = dxb + (long)ci * HW;
        *d = accumulate ? *d + s : s;
      }
    }
  } else {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      float dv[CP][9];
#pragma unroll
      for (int c = 0; c < CP; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) dv[c][t] = tp.ok[t] ? dyb[(long)(g * CP + c) * HW + tp.off[t]] : 0.f;
      for (int ci = 0; ci < Cin; ++ci) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CP; ++c) {
          const float* wr = CN_THIN_W(g) + ((long)c * Cin + ci) * 9;
#pragma unroll
          for (int t = 0; t < 9; ++t) s = fmaf(dv[c][t], wr[t], s);
        }
        if (live) {
          float* d = dxb + (long)(g * Cin + ci) * HW;
          *d = accumulate ? *d + s : s;
        }
      }
    }
  }
}

// dw[co][ci][t] += sum_{b,p} dy[b][co][p] * x[b][ci][p + (t-1)*dil]
// grid = (pixel chunks, input channels); a lane keeps the (output channel x tap) partial sums of its pixels in
// registers, the block reduces them through wave shuffles + LDS and issues one atomic per weight.
template <int NG, int CP, bool GROUPED>
__global__ __launch_bounds__(256) void cn_thin_bwd_weight_kernel(const float* __restrict__ x, long xbs,
                                                                const float* __restrict__ dy, long dybs,
                                                                float* __restrict__ dw0, float* __restrict__ dw1,
                                                                float* __restrict__ dw2, int B, int Cin, int H,
                                                                int W, int dil, int per_chunk) {
  constexpr int NO = GROUPED ? CP : NG * CP;  // output channels paired with this input channel
  __shared__ float red[4][NO * 9];
  const int HW = H * W;
  const int cit = blockIdx.y;                  // input channel of the whole tensor
  const int grp = GROUPED ? cit / Cin : 0;
  const int ci = GROUPED ? cit - grp * Cin : cit;
  const long P = (long)B * HW;
  long q = (long)blockIdx.x * per_chunk + threadIdx.x;
  long q_end = (long)(blockIdx.x + 1) * per_chunk;
  if (q_end > P) q_end = P;
  float acc[NO * 9];
#pragma unroll
  for (int i = 0; i < NO * 9; ++i) acc[i] = 0.f;
  for (; q < q_end; q += 256) {
    const int b = (int)(q / HW);
    const int pix = (int)(q - (long)b * HW);
    const int oy = pix / W, ox = pix - oy * W;
    const Taps tp = cn_thin_taps(oy, ox, H, W, dil, 1);
    const float* xc = x + (long)b * xbs + (long)cit * HW;
    float xv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) xv[t] = tp.ok[t] ? xc[tp.off[t]] : 0.f;
    const float* dyb = dy + (long)b * dybs + (long)(GROUPED ? grp * CP : 0) * HW + pix;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      const float d = dyb[(long)o * HW];
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[o * 9 + t] = fmaf(d, xv[t], acc[o * 9 + t]);
    }
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NO * 9; ++i) {
    const float v = cn_wave_sum(acc[i]);
    if (lane == 0) red[wid][i] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NO * 9; i += 256) {
    const float v = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    const int o = i / 9, t = i - o * 9;
    float* dst;
    if (GROUPED) {
      float* dwg = grp == 0 ? dw0 : (grp == 1 ? dw1 : dw2);
      dst = dwg + ((long)o * Cin + ci) * 9 + t;
    } else {
      const int g = o / CP, c = o - g * CP;
      float* dwg = g == 0 ? dw0 : (g == 1 ? dw1 : dw2);
      dst = dwg + ((long)c * Cin + ci) * 9 + t;
    }
    atomicAdd(dst, v);
  }
}

int cn_thin_cfg(int nsets, int cout_per_set, int grouped) {
  if (nsets == 3 && cout_per_set == 3 && !grouped) return 0;  // 128 -> 3 x3 streams on one input
  if (nsets == 3 && cout_per_set == 1 && grouped) return 1;   // 3 -> 1 x3 streams on their own inputs
  if (nsets == 1 && cout_per_set == 3 && !grouped) return 2;  // fuse conv 3 -> 3
  if (nsets == 1 && cout_per_set == 1 && !grouped) return 3;  // a single C -> 1 stream
  return -1;
}

}  // namespace

#define CN_THIN_DISPATCH(cfg, CALL)        \
  switch (cfg) {                           \
    case 0: { CALL(3, 3, false); } break;  \
    case 1: { CALL(3, 1, true); } break;   \
    case 2: { CALL(1, 3, false); } break;  \
    default: { CALL(1, 1, false); } break; \
  }

// y[b][set*CP + c][p] = bias_set[c] + sum_{ci,t} x[b][ci (+ set*Cin if grouped)][p + (t-1)*dil] * w_set[c][ci][t]
// ws / biases: HOST arrays of nsets device pointers (biases, or single entries, may be NULL).
// Returns CN_ERR_ARG for (nsets, cout_per_set, grouped) outside {(3,3,0), (3,1,1), (1,3,0), (1,1,0)}.
extern "C" int cn_thin_conv3x3_fwd_f32(const float* x, long xbs, const float* const* ws, const float* const* biases,
                                       float* y, long ybs, int B, int Cin, int H, int W, int nsets, int cout_per_set,
                                       int grouped, int dil, void* stream) {
  const int cfg = cn_thin_cfg(nsets, cout_per_set, grouped);
  if (cfg < 0 || dil < 1 || Cin < 1) return CN_ERR_ARG;
  if (B <= 0 || H <= 0 || W <= 0) return CN_OK;
  const float* w[3] = {ws[0], nsets > 1 ? ws[1] : nullptr, nsets > 2 ? ws[2] : nullptr};
  const float* b[3] = {nullptr, nullptr, nullptr};
  if (biases)
    for (int i = 0; i < nsets; ++i) b[i] = biases[i];
  const dim3 grid(cn_cdiv((long)H * W, 256), B);
#define CN_CALL(NG_, CP_, GR_)                                                                                      \
  hipLaunchKernelGGL((cn_thin_fwd_kernel<NG_, CP_, GR_>), grid, dim3(256), 0, (hipStream_t)stream, x, xbs, w[0], w[1], \
                     w[2], b[0], b[1], b[2], y, ybs, Cin, H, W, dil)
  CN_THIN_DISPATCH(cfg, CN_CALL)
#undef CN_CALL
  return cn_check_launch();
}

extern "C" int cn_thin_conv3x3_bwd_data_f32(const float* dy, long dybs, const float* const* ws, float* dx, long dxbs,
                                            int B, int Cin, int H, int W, int nsets, int cout_per_set, int grouped,
                                            int dil, int accumulate, void* stream) {
  const int cfg = cn_thin_cfg(nsets, cout_per_set, grouped);
  if (cfg < 0 || dil < 1 || Cin < 1) return CN_ERR_ARG;
  if (B <= 0 || H <= 0 || W <= 0) return CN_OK;
  const float* w[3] = {ws[0], nsets > 1 ? ws[1] : nullptr, nsets > 2 ? ws[2] : nullptr};
  const dim3 grid(cn_cdiv((long)H * W, 256), B);
#define CN_CALL(NG_, CP_, GR_)                                                                                   \
  hipLaunchKernelGGL((cn_thin_bwd_data_kernel<NG_, CP_, GR_>), grid, dim3(256), 0, (hipStream_t)stream, dy, dybs, \
                     w[0], w[1], w[2], dx, dxbs, Cin, H, W, dil, accumulate)
  CN_THIN_DISPATCH(cfg, CN_CALL)
#undef CN_CALL
  return cn_check_launch();
}

// NOTE accumulates into dws (like cn_conv2d_bwd_weight_f32).
extern "C" int cn_thin_conv3x3_bwd_weight_f32(const float* x, long xbs, const float* dy, long dybs, float* const* dws,
                                              int B, int Cin, int H, int W, int nsets, int cout_per_set, int grouped,
                                              int dil, void* stream) {
  const int cfg = cn_thin_cfg(nsets, cout_per_set, grouped);
  if (cfg < 0 || dil < 1 || Cin < 1) return CN_ERR_ARG;
  if (B <= 0 || H <= 0 || W <= 0) return CN_OK;
  float* dw[3] = {dws[0], nsets > 1 ? dws[1] : nullptr, nsets > 2 ? dws[2] : nullptr};
  const int cin_total = grouped ? nsets * Cin : Cin;
  const long P = (long)B * H * W;
  // ~2048 blocks on the chip, but at least 16 pixels per lane so the block reduction stays a small fraction
  long chunks = 2048 / cin_total;
  if (chunks < 1) chunks = 1;
  long per = (P + chunks - 1) / chunks;
  if (per < 16 * 256) per = 16 * 256;
  per = (per + 255) / 256 * 256;
  chunks = (P + per - 1) / per;
  const dim3 grid((unsigned)chunks, cin_total);
#define CN_CALL(NG_, CP_, GR_)                                                                                      \
  hipLaunchKernelGGL((cn_thin_bwd_weight_kernel<NG_, CP_, GR_>), grid, dim3(256), 0, (hipStream_t)stream, x, xbs, dy, \
                     dybs, dw[0], dw[1], dw[2], B, Cin, H, W, dil, (int)per)
  CN_THIN_DISPATCH(cfg, CN_CALL)
#undef CN_CALL
  return cn_check_launch();
}
