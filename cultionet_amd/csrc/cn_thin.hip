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
// grid = (row chunks x column tiles, input channels). A lane owns one image column and walks down the rows of its
// chunk (two rows in flight per block: threads 0-127 / 128-255), so the row index and the vertical tap validity are
// wave-uniform and the loop carries no divisions; the (output channel x tap) partial sums stay in registers,
// the block reduces them (DPP wave sums + LDS) and issues one atomic per weight.
template <int NG, int CP, bool GROUPED>
__global__ __launch_bounds__(256) void cn_thin_bwd_weight_kernel(const float* __restrict__ x, long xbs,
                                                                const float* __restrict__ dy, long dybs,
                                                                float* __restrict__ dw0, float* __restrict__ dw1,
                                                                float* __restrict__ dw2, int B, int Cin, int H,
                                                                int W, int dil, int rows_per_chunk, int col_tiles) {
  constexpr int NO = GROUPED ? CP : NG * CP;  // output channels paired with this input channel
  __shared__ float red[4][NO * 9];
  const int HW = H * W;
  const int cit = blockIdx.y;                  // input channel of the whole tensor
  const int grp = GROUPED ? cit / Cin : 0;
  const int ci = GROUPED ? cit - grp * Cin : cit;
  const int chunk = blockIdx.x / col_tiles, ct = blockIdx.x - chunk * col_tiles;
  const int half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 7);
  const int ox = ct * 128 + (threadIdx.x & 127);
  const bool col_ok = ox < W;
  bool okx[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int ix = ox + (kx - 1) * dil;
    okx[kx] = col_ok && ix >= 0 && ix < W;
  }
  const int R = B * H;
  int r_end = (chunk + 1) * rows_per_chunk;
  if (r_end > R) r_end = R;
  float acc[NO * 9];
#pragma unroll
  for (int i = 0; i < NO * 9; ++i) acc[i] = 0.f;
  for (int r = chunk * rows_per_chunk + half; r < r_end; r += 2) {
    const int b = r / H, oy = r - b * H;
    const float* xc = x + (long)b * xbs + (long)cit * HW + ox;
    float xv[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy + (ky - 1) * dil;
      const bool oky = iy >= 0 && iy < H;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
        xv[ky * 3 + kx] = (oky && okx[kx]) ? xc[iy * W + (kx - 1) * dil] : 0.f;
    }
    const float* dyb = dy + (long)b * dybs + (long)(GROUPED ? grp * CP : 0) * HW + oy * W + ox;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      const float d = col_ok ? dyb[(long)o * HW] : 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[o * 9 + t] = fmaf(d, xv[t], acc[o * 9 + t]);
    }
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NO * 9; ++i) {
    const float v = cn_wave_sum_to_lane63(acc[i]);
    if (lane == 63) red[wid][i] = v;
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

// ---- wide-input variants (Cin >= 16, shared input): one lane per pixel leaves ~1 wave per SIMD on a 100x100
// chip batch, so the input channels are split over the 4 waves of a block (64 pixels per block) and each wave
// keeps ITS quarter of the weights in LDS as [ci][(o,t) = NO*9, padded to a multiple of 4] (broadcast
// ds_read_b128; the scalar cache cannot hold the 41 KB of a 128 -> 9 layer).
template <int NG, int CP>
__device__ __forceinline__ void cn_thin_stage_weights(float* wl, int Cq, int Cin, const float* __restrict__ w0,
                                                      const float* __restrict__ w1, const float* __restrict__ w2) {
  constexpr int NO = NG * CP, RS = (NO * 9 + 3) / 4 * 4;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int idx = lane; idx < Cq * RS; idx += 64) {  // pad entries are zeroed (they are multiplied, by zeros)
    const int cil = idx / RS, r = idx - cil * RS;
    const int o = r / 9, t = r - o * 9;
    const int g = o / CP, c = o - g * CP;
    const int ci = wid * Cq + cil;
    const float* wg = g == 0 ? w0 : (g == 1 ? w1 : w2);
    wl[(wid * Cq + cil) * RS + r] = (ci < Cin && r < NO * 9) ? wg[((long)c * Cin + ci) * 9 + t] : 0.f;
  }
}

// out[ci][RS] = the NO*9 weights of input channel ci in (o, t) order, zero padded: the layout the ks kernels
// consume. One tiny launch per call; without it every block gathers the 41 KB itself through a chain of dependent
// global loads, which was most of the kernels' time.
template <int NG, int CP>
__global__ __launch_bounds__(256) void cn_thin_pack_kernel(const float* __restrict__ w0, const float* __restrict__ w1,
                                                          const float* __restrict__ w2, float* __restrict__ out,
                                                          int Cin) {
  constexpr int NO = NG * CP, RS = (NO * 9 + 3) / 4 * 4;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Cin * RS) return;
  const int ci = idx / RS, r = idx - ci * RS;
  const int o = r / 9, t = r - o * 9;
  const int g = o / CP, c = o - g * CP;
  const float* wg = g == 0 ? w0 : (g == 1 ? w1 : w2);
  out[idx] = r < NO * 9 ? wg[((long)c * Cin + ci) * 9 + t] : 0.f;
}

template <int NG, int CP>
__global__ __launch_bounds__(256) void cn_thin_fwd_ks_kernel(const float* __restrict__ x, long xbs,
                                                            const float* __restrict__ w0,
                                                            const float* __restrict__ w1,
                                                            const float* __restrict__ w2,
                                                            const float* __restrict__ b0,
                                                            const float* __restrict__ b1,
                                                            const float* __restrict__ b2, float* __restrict__ y,
                                                            long ybs, int Cin, int H, int W, int dil, int Cq, int B,
                                                            const float* __restrict__ wpk) {
  constexpr int NO = NG * CP, RS = (NO * 9 + 3) / 4 * 4;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  // wpk != NULL: weights pre-packed [Cin][RS] in global memory (read one dword per lane, L1/L2 resident);
  // else staged by this block into LDS [4][Cq][RS]
  float* wl = sm;
  float* red = wpk ? sm : sm + 4 * Cq * RS;  // [4][NO][64]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int HW = H * W;
  int bx, by, bz;
  const int gx = (HW + 63) / 64;
  if (!cn_xcd_block(gx, B, gx * B, bx, by, bz)) return;
  const int pix = bx * 64 + lane;
  const bool live = pix < HW;
  const int pc = live ? pix : 0;
  const int oy = pc / W, ox = pc - oy * W;
  const Taps tp = cn_thin_taps(oy, ox, H, W, dil, 1);
  if (!wpk) {
    cn_thin_stage_weights<NG, CP>(wl, Cq, Cin, w0, w1, w2);
    __syncthreads();
  }
  const float* xb = x + (long)by * xbs;
  float acc[NO];
#pragma unroll
  for (int o = 0; o < NO; ++o) acc[o] = 0.f;
  // channels in groups of U: all U*9 input loads are issued before the first FMA (memory-level parallelism)
  constexpr int U = 2;
  for (int c0 = 0; c0 < Cq; c0 += U) {
    float xv[U][9];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int ci = wid * Cq + c0 + u;
      if (ci >= Cin) ci = Cin - 1;  // staged weights of channels past the end are zero
      const float* xc = xb + (long)ci * HW;
#pragma unroll
      for (int t = 0; t < 9; ++t) xv[u][t] = tp.ok[t] ? xc[tp.off[t]] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (c0 + u < Cq) {
        // wave-uniform weights: one dword per lane out of LDS + v_readlane (an LDS broadcast of all RS weights to
        // all 64 lanes made this kernel LDS-bandwidth-bound)
        const int cw = wid * Cq + c0 + u;
        const float* wrow = wpk ? wpk + (long)(cw < Cin ? cw : 0) * RS : wl + cw * RS;
        const bool wok = !wpk || cw < Cin;  // packed rows exist for real channels only
        // (round 5: scalar loads of the packed row instead -- s_load_dwordx16 into SGPRs, no v_readlane, v_pk_fma_f32 --
        // cut the vector instructions per channel from ~165 to ~45 and made THIS kernel slower, 64 -> 89 us: the
        // compiler waits for every 16-weight batch right where it is used (lgkmcnt(0)), five exposed scalar-cache
        // misses per channel; the same change pays in cn_thin_bwd_data_ks_kernel, whose operands are all registers)
        const int wa = wok ? __float_as_int(wrow[lane]) : 0;
        const int wb = wok ? __float_as_int(wrow[64 + (lane < RS - 64 ? lane : 0)]) : 0;
#pragma unroll
        for (int o = 0; o < NO; ++o)
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const int i = o * 9 + t;
            acc[o] = fmaf(xv[u][t], __int_as_float(__builtin_amdgcn_readlane(i < 64 ? wa : wb, i & 63)), acc[o]);
          }
      }
    }
  }
#pragma unroll
  for (int o = 0; o < NO; ++o) red[(wid * NO + o) * 64 + lane] = acc[o];
  __syncthreads();
  for (int idx = threadIdx.x; idx < NO * 64; idx += 256) {
    const int o = idx >> 6, l = idx & 63;
    const int g = o / CP, c = o - g * CP;
    const float* bg = g == 0 ? b0 : (g == 1 ? b1 : b2);
    float v = bg != nullptr ? bg[c] : 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += red[(w * NO + o) * 64 + l];
    const int p = bx * 64 + l;
    if (p < HW) y[(long)by * ybs + (long)o * HW + p] = v;
  }
}

template <int NG, int CP>
__global__ __launch_bounds__(256) void cn_thin_bwd_data_ks_kernel(const float* __restrict__ dy, long dybs,
                                                                 const float* __restrict__ w0,
                                                                 const float* __restrict__ w1,
                                                                 const float* __restrict__ w2,
                                                                 float* __restrict__ dx, long dxbs, int Cin, int H,
                                                                 int W, int dil, int accumulate, int Cq, int B,
                                                                 const float* __restrict__ wpk) {
  constexpr int NO = NG * CP, RS = (NO * 9 + 3) / 4 * 4;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* wl = sm;  // [4][Cq][RS] (only without pre-packed weights)
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int HW = H * W;
  int bx, by, bz;
  const int gx = (HW + 63) / 64;
  if (!cn_xcd_block(gx, B, gx * B, bx, by, bz)) return;
  const int pix = bx * 64 + lane;
  const bool live = pix < HW;
  const int pc = live ? pix : 0;
  const int oy = pc / W, ox = pc - oy * W;
  const Taps tp = cn_thin_taps(oy, ox, H, W, dil, -1);
  if (!wpk) cn_thin_stage_weights<NG, CP>(wl, Cq, Cin, w0, w1, w2);
  const float* dyb = dy + (long)by * dybs;
  f32x2 dv[RS / 2];  // (o, t) pairs in the order of the staged weights; the pad entries are zero
#pragma unroll
  for (int i = 0; i < RS; ++i) {
    const int o = i / 9, t = i - o * 9;
    const float v = (i < NO * 9 && tp.ok[t]) ? dyb[(long)o * HW + tp.off[t]] : 0.f;
    dv[i >> 1][i & 1] = v;
  }
  __syncthreads();
  float* dxb = dx + (long)by * dxbs + pix;
  for (int cil = 0; cil < Cq; ++cil) {
    const int ci = wid * Cq + cil;
    if (ci >= Cin) break;
    // The RS weights of this channel are wave-uniform. Broadcasting them out of LDS (ds_read_b128, same address in
    // every lane) costs the full 64-lane LDS bandwidth and made the kernel LDS-bound; instead every lane reads ONE
    // weight (two conflict-free dword reads per channel) and v_readlane moves them to scalar registers.
    f32x2 s2[2] = {{0.f, 0.f}, {0.f, 0.f}};  // packed fma (v_pk_fma_f32), two independent chains
    if (wpk) {
      // (round 5) packed weights: the row is wave-uniform -> scalar loads (s_load_dwordx16), and the packed FMA reads
      // the SGPR pair directly: no v_readlane per weight (two of the three vector instructions per pair went into them)
      const float* wr = wpk + (long)__builtin_amdgcn_readfirstlane(ci) * RS;
#pragma unroll
      for (int j = 0; j < RS / 2; ++j) {
        const f32x2 w2 = {wr[2 * j], wr[2 * j + 1]};
        s2[j & 1] = __builtin_elementwise_fma(dv[j], w2, s2[j & 1]);
      }
    } else {
    const float* wrow = wl + (wid * Cq + cil) * RS;
    const int wa = __float_as_int(wrow[lane]);
    const int wb = __float_as_int(wrow[64 + (lane < RS - 64 ? lane : 0)]);
#pragma unroll
    for (int j = 0; j < RS / 2; ++j) {
      const int i0 = 2 * j, i1 = 2 * j + 1;
      const f32x2 w2 = {__int_as_float(__builtin_amdgcn_readlane(i0 < 64 ? wa : wb, i0 & 63)),
                        __int_as_float(__builtin_amdgcn_readlane(i1 < 64 ? wa : wb, i1 & 63))};
      s2[j & 1] = __builtin_elementwise_fma(dv[j], w2, s2[j & 1]);
    }
    }
    const float r = (s2[0][0] + s2[0][1]) + (s2[1][0] + s2[1][1]);
    if (live) {
      float* d = dxb + (long)ci * HW;
      *d = accumulate ? *d + r : r;
    }
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
                                       int grouped, int dil, float* wpack, void* stream) {
  const int cfg = cn_thin_cfg(nsets, cout_per_set, grouped);
  if (cfg < 0 || dil < 1 || Cin < 1) return CN_ERR_ARG;
  if (B <= 0 || H <= 0 || W <= 0) return CN_OK;
  const float* w[3] = {ws[0], nsets > 1 ? ws[1] : nullptr, nsets > 2 ? ws[2] : nullptr};
  const float* b[3] = {nullptr, nullptr, nullptr};
  if (biases)
    for (int i = 0; i < nsets; ++i) b[i] = biases[i];
  if (cfg == 0 && Cin >= 16) {
    const int Cq = (Cin + 3) / 4;
    // wpack (Cin * 84 floats): the weights are packed once per call and read from global memory; without it every
    // block stages them into LDS itself
    const size_t lds = sizeof(float) * (size_t)((wpack ? 0 : 4 * Cq * 84) + 4 * 9 * 64);
    if (lds <= 64 * 1024) {
      if (wpack)
        CN_LAUNCH((cn_thin_pack_kernel<3, 3>), dim3(cn_cdiv((long)Cin * 84, 256)), dim3(256), 0,
                           (hipStream_t)stream, w[0], w[1], w[2], wpack, Cin);
      CN_LAUNCH((cn_thin_fwd_ks_kernel<3, 3>), dim3(cn_xcd_grid((long)cn_cdiv((long)H * W, 64) * B)), dim3(256), lds,
                         (hipStream_t)stream, x, xbs, w[0], w[1], w[2], b[0], b[1], b[2], y, ybs, Cin, H, W, dil, Cq, B,
                         (const float*)wpack);
      return cn_check_launch();
    }
  }
  const dim3 grid(cn_cdiv((long)H * W, 256), B);
#define CN_CALL(NG_, CP_, GR_)                                                                                      \
  CN_LAUNCH((cn_thin_fwd_kernel<NG_, CP_, GR_>), grid, dim3(256), 0, (hipStream_t)stream, x, xbs, w[0], w[1], \
                     w[2], b[0], b[1], b[2], y, ybs, Cin, H, W, dil)
  CN_THIN_DISPATCH(cfg, CN_CALL)
#undef CN_CALL
  return cn_check_launch();
}

extern "C" int cn_thin_conv3x3_bwd_data_f32(const float* dy, long dybs, const float* const* ws, float* dx, long dxbs,
                                            int B, int Cin, int H, int W, int nsets, int cout_per_set, int grouped,
                                            int dil, int accumulate, float* wpack, void* stream) {
  const int cfg = cn_thin_cfg(nsets, cout_per_set, grouped);
  if (cfg < 0 || dil < 1 || Cin < 1) return CN_ERR_ARG;
  if (B <= 0 || H <= 0 || W <= 0) return CN_OK;
  const float* w[3] = {ws[0], nsets > 1 ? ws[1] : nullptr, nsets > 2 ? ws[2] : nullptr};
  if (cfg == 0 && Cin >= 16) {
    const int Cq = (Cin + 3) / 4;
    const size_t lds = sizeof(float) * (size_t)(wpack ? 0 : 4 * Cq * 84);
    if (lds <= 64 * 1024) {
      if (wpack)
        CN_LAUNCH((cn_thin_pack_kernel<3, 3>), dim3(cn_cdiv((long)Cin * 84, 256)), dim3(256), 0,
                           (hipStream_t)stream, w[0], w[1], w[2], wpack, Cin);
      CN_LAUNCH((cn_thin_bwd_data_ks_kernel<3, 3>), dim3(cn_xcd_grid((long)cn_cdiv((long)H * W, 64) * B)), dim3(256), lds,
                         (hipStream_t)stream, dy, dybs, w[0], w[1], w[2], dx, dxbs, Cin, H, W, dil, accumulate, Cq, B,
                         (const float*)wpack);
      return cn_check_launch();
    }
  }
  const dim3 grid(cn_cdiv((long)H * W, 256), B);
#define CN_CALL(NG_, CP_, GR_)                                                                                   \
  CN_LAUNCH((cn_thin_bwd_data_kernel<NG_, CP_, GR_>), grid, dim3(256), 0, (hipStream_t)stream, dy, dybs, \
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
  const int col_tiles = (W + 127) / 128;
  const int R = B * H;
  // ~1024 blocks on the chip, but at least 16 rows per lane so the block reduction stays a small fraction
  int chunks = 1024 / (cin_total * col_tiles);
  if (chunks < 1) chunks = 1;
  int per = (R + chunks - 1) / chunks;
  if (per < 32) per = 32;
  per = (per + 1) & ~1;
  chunks = (R + per - 1) / per;
  const dim3 grid((unsigned)(chunks * col_tiles), cin_total);
#define CN_CALL(NG_, CP_, GR_)                                                                                      \
  CN_LAUNCH((cn_thin_bwd_weight_kernel<NG_, CP_, GR_>), grid, dim3(256), 0, (hipStream_t)stream, x, xbs, dy, \
                     dybs, dw[0], dw[1], dw[2], B, Cin, H, W, dil, per, col_tiles)
  CN_THIN_DISPATCH(cfg, CN_CALL)
#undef CN_CALL
  return cn_check_launch();
}
