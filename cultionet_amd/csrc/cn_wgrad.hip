// Weight-gradient contraction for Conv2d / ConvTranspose2d / 1x1 / Linear, fp32 MFMA (gfx950).
//
//   dW[a][b][t] += sum_{n, g} S[n, a, g] * Bg[n, b, g*s + off(t)]
//
// S is the tensor that lives on the "small" pixel grid (dy for Conv2d, x for
// ConvTranspose2d), Bg the gathered one. This is a GEMM with M = A channels,
// N = (b, tap) flattened exactly as the weight tensor is laid out
// ([A][Bc][KH][KW] for both torch Conv2d and ConvTranspose2d), K = all pixels of
// the batch. K is split over blocks (grid.z) and over the waves of a block when
// A is small; partial results are combined with no-return f32 atomics whose
// wave-instruction covers 2 x 128 contiguous bytes of dW.
// Reference op: autograd of torch.nn.Conv2d / ConvTranspose2d used by
// /root/reference/src/cultionet/nn/modules/convolution.py:45-120.
#include "cn_common.h"
#include "cn_profile.h"

#define WG_MAX_TAPS 9
#define WG_BC 32   // b-channels per block
#define WG_NS 4    // max S elements per lane per row of the chunk  (chunk pixels <= 256)
#define WG_NB 13   // max Bg plane elements per lane               (plane_b <= 832)

struct CnWgradGeom {
  int N;
  int A, Hs, Ws; long sbs;
  int Bc, Hb, Wb; long bbs;
  int s;
  int T;
  int offy[WG_MAX_TAPS], offx[WG_MAX_TAPS];
  int min_oy, min_ox;
  long sa;           // dW offset = a*sa + b*T + t
  int PR, Wsp, pitch_s;
  int rows_b, pitch_b, plane_b;
  int a_tiles;       // 32-row tiles of A per block (1, 2 or 4)
  int chunks_per_img, total_chunks, chunks_per_split;
  int b_lds_off;
};

template <int T>
__global__ __launch_bounds__(256, 2) void cn_wgrad_kernel(const float* __restrict__ S, const float* __restrict__ Bg,
                                                      float* __restrict__ dW, const CnWgradGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_lds = smem;                // [a_tiles*32][pitch_s]
  float* b_lds = smem + g.b_lds_off;  // [WG_BC][plane_b]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int at = wid % g.a_tiles, kp = wid / g.a_tiles, kparts = 4 / g.a_tiles;
  const int a0 = blockIdx.y * g.a_tiles * 32;
  const int b0 = blockIdx.x * WG_BC;
  const int npix = g.PR * g.Wsp;  // staged (padded) grid pixels per chunk, even

  // ---- per-lane staging decode (chunk independent)
  int so[WG_NS];  // (row << 16 | col) of the element inside the chunk; col 0xFFFF = zero pad; -2 = skip
#pragma unroll
  for (int i = 0; i < WG_NS; ++i) {
    const int e = lane + i * 64;
    if (e < npix) {
      const int r = e / g.Wsp, c = e - r * g.Wsp;
      so[i] = (r << 16) | ((c < g.Ws) ? c : 0xFFFF);
    } else {
      so[i] = -2;
    }
  }
  int bo[WG_NB];  // (row << 16 | col) inside the staged Bg plane; -1 -> skip
#pragma unroll
  for (int i = 0; i < WG_NB; ++i) {
    const int e = lane + i * 64;
    if (e < g.plane_b) {
      const int r = e / g.pitch_b;
      bo[i] = (r << 16) | (e - r * g.pitch_b);
    } else {
      bo[i] = -1;
    }
  }

  // ---- per-lane operand addresses
  int boff[T];
#pragma unroll
  for (int j = 0; j < T; ++j) {
    const int n = j * 32 + l31;
    const int bl = n / T, t = n - bl * T;
    boff[j] = bl * g.plane_b + (g.offy[t] - g.min_oy) * g.pitch_b + (g.offx[t] - g.min_ox) + half * g.s;
  }
  const int aoff = (at * 32 + l31) * g.pitch_s + half;

  f32x16 acc[T];
#pragma unroll
  for (int j = 0; j < T; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int HWs = g.Hs * g.Ws, HWb = g.Hb * g.Wb;
  const int arows = g.a_tiles * 32;
  int chunk = blockIdx.z * g.chunks_per_split;
  int chunk_end = chunk + g.chunks_per_split;
  if (chunk_end > g.total_chunks) chunk_end = g.total_chunks;

  // this wave's share of the k-steps of a chunk
  const int steps = npix >> 1;
  const int q_begin = (steps * kp) / kparts, q_end = (steps * (kp + 1)) / kparts;
  const int e_begin = q_begin * 2;
  const int r_begin = e_begin / g.Wsp, c_begin = e_begin - r_begin * g.Wsp;

  for (; chunk < chunk_end; ++chunk) {
    const int n = chunk / g.chunks_per_img;
    const int gy0 = (chunk - n * g.chunks_per_img) * g.PR;
    const int pr = (g.Hs - gy0 < g.PR) ? g.Hs - gy0 : g.PR;
    __syncthreads();
    // ---- stage S rows [arows][PR*Wsp]; zero for a >= A, padded col, rows past the image
    {
      const float* Sn = S + (long)n * g.sbs + (long)gy0 * g.Ws;
      for (int a = wid; a < arows; a += 4) {
        const bool aok = (a0 + a) < g.A;
        const float* Sa = Sn + (long)(a0 + a) * HWs;
#pragma unroll
        for (int i = 0; i < WG_NS; ++i) {
          if (so[i] != -2) {
            const int r = so[i] >> 16, c = so[i] & 0xFFFF;
            float v = 0.f;
            if (aok && c != 0xFFFF && r < pr) v = Sa[r * g.Ws + c];
            s_lds[a * g.pitch_s + lane + i * 64] = v;
          }
        }
      }
    }
    // ---- stage Bg halo planes [WG_BC][rows_b][pitch_b]; zero outside the image / past Bc
    {
      const float* Bn = Bg + (long)n * g.bbs;
      const int iy0 = gy0 * g.s + g.min_oy;
      for (int bl = wid; bl < WG_BC; bl += 4) {
        const bool bok = (b0 + bl) < g.Bc;
        const float* Bb = Bn + (long)(b0 + bl) * HWb;
#pragma unroll
        for (int i = 0; i < WG_NB; ++i) {
          if (bo[i] >= 0) {
            const int iy = iy0 + (bo[i] >> 16), ix = g.min_ox + (bo[i] & 0xFFFF);
            float v = 0.f;
            if (bok && iy >= 0 && iy < g.Hb && ix >= 0 && ix < g.Wb) v = Bb[iy * g.Wb + ix];
            b_lds[bl * g.plane_b + lane + i * 64] = v;
          }
        }
      }
    }
    __syncthreads();
    // ---- MFMA: k = pixel pairs (col, col+1) of the chunk
    int r = r_begin, c = c_begin;
    for (int q = q_begin; q < q_end; ++q) {
      const float av = s_lds[aoff + r * g.Wsp + c];
      const int bbase = (r * g.s) * g.pitch_b + c * g.s;
#pragma unroll
      for (int j = 0; j < T; ++j) {
        const float bv = b_lds[boff[j] + bbase];
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
      }
      c += 2;
      if (c >= g.Wsp) { c = 0; ++r; }
    }
  }

  // ---- epilogue: D[i = a][j = n]; lanes walk n = b*T + t -> contiguous in dW
#pragma unroll
  for (int j = 0; j < T; ++j) {
    const int n = j * 32 + l31;
    const int bl = n / T;
    const bool bok = (b0 + bl) < g.Bc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int a = a0 + at * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (bok && a < g.A) atomicAdd(dW + (long)a * g.sa + (long)b0 * T + n, acc[j][r]);
    }
  }
}

template <int T>
static int cn_wgrad_launch_t(const float* S, const float* Bg, float* dW, CnWgradGeom g, hipStream_t stream) {
  // chunk: PR grid rows so that a chunk holds ~<=128 pixels (<= 256 hard limit)
  g.Wsp = (g.Ws + 1) & ~1;
  if (g.Wsp > 64 * WG_NS) return CN_ERR_LDS;
  g.PR = 128 / g.Wsp;
  if (g.PR < 1) g.PR = 1;
  if (g.PR > g.Hs) g.PR = g.Hs;
  int max_oy = g.min_oy, max_ox = g.min_ox;
  for (int t = 0; t < g.T; ++t) {
    if (g.offy[t] > max_oy) max_oy = g.offy[t];
    if (g.offx[t] > max_ox) max_ox = g.offx[t];
  }
  for (;;) {
    g.rows_b = (g.PR - 1) * g.s + (max_oy - g.min_oy) + 1;
    g.pitch_b = (g.Wsp - 1) * g.s + (max_ox - g.min_ox) + 1;
    g.plane_b = g.rows_b * g.pitch_b;
    if (g.plane_b <= 64 * WG_NB || g.PR == 1) break;
    --g.PR;
  }
  if (g.plane_b > 64 * WG_NB) return CN_ERR_LDS;
  g.pitch_s = (g.PR * g.Wsp) | 1;
  g.a_tiles = g.A > 64 ? 4 : (g.A > 32 ? 2 : 1);
  // keep two blocks per CU when possible: A tile of 64 rows for large A
  if (g.a_tiles == 4) g.a_tiles = 2;
  size_t lds;
  for (;;) {
    g.b_lds_off = (g.a_tiles * 32 * g.pitch_s + 3) / 4 * 4;
    lds = (size_t)(g.b_lds_off + WG_BC * g.plane_b) * sizeof(float);
    if (lds <= 160 * 1024 || g.a_tiles == 1) break;
    g.a_tiles >>= 1;
  }
  if (lds > 160 * 1024) return CN_ERR_LDS;
  g.chunks_per_img = (g.Hs + g.PR - 1) / g.PR;
  g.total_chunks = g.N * g.chunks_per_img;
  const int gx = (g.Bc + WG_BC - 1) / WG_BC, gy = (g.A + g.a_tiles * 32 - 1) / (g.a_tiles * 32);
  // aim for ~3 blocks per CU overall
  int splits = (768 + gx * gy - 1) / (gx * gy);
  if (splits > g.total_chunks) splits = g.total_chunks;
  if (splits < 1) splits = 1;
  g.chunks_per_split = (g.total_chunks + splits - 1) / splits;
  splits = (g.total_chunks + g.chunks_per_split - 1) / g.chunks_per_split;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)cn_wgrad_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr_set = true;
  }
  cn_prof_before(stream);
  hipLaunchKernelGGL((cn_wgrad_kernel<T>), dim3(gx, gy, splits), dim3(256), lds, stream, S, Bg, dW, g);
  cn_prof_after(stream, T == 9 ? 2 : 3, 2.0 * g.N * g.Hs * g.Ws * (double)g.A * g.Bc * g.T);
  return cn_check_launch();
}

// Generic entry: dW[a][b][t] += sum S[n,a,gy,gx] * Bg[n,b,gy*s+offy[t],gx*s+offx[t]]
static int cn_wgrad_generic(const float* S, long sbs, int A, int Hs, int Ws, const float* Bg, long bbs, int Bc,
                            int Hb, int Wb, int s, int KH, int KW, int dil, int pad, float* dW, int N,
                            hipStream_t stream) {
  CnWgradGeom g = {};
  g.N = N; g.A = A; g.Hs = Hs; g.Ws = Ws; g.sbs = sbs; g.Bc = Bc; g.Hb = Hb; g.Wb = Wb; g.bbs = bbs; g.s = s;
  g.T = KH * KW;
  if (N <= 0 || A <= 0 || Bc <= 0 || Hs <= 0 || Ws <= 0) return CN_OK;
  for (int ky = 0; ky < KH; ++ky)
    for (int kx = 0; kx < KW; ++kx) {
      g.offy[ky * KW + kx] = ky * dil - pad;
      g.offx[ky * KW + kx] = kx * dil - pad;
    }
  g.min_oy = -pad; g.min_ox = -pad;
  g.sa = (long)Bc * g.T;
  if (g.T == 1) return cn_wgrad_launch_t<1>(S, Bg, dW, g, stream);
  if (g.T == 9) return cn_wgrad_launch_t<9>(S, Bg, dW, g, stream);
  return CN_ERR_ARG;
}

// Conv2d: dw [Cout][Cin][KH][KW] += x (*) dy. NOTE accumulates: zero dw first for a fresh gradient.
extern "C" int cn_conv2d_bwd_weight_f32(const float* x, long xbs, const float* dy, long dybs, float* dw, int B,
                                        int Cin, int Hin, int Win, int Cout, int KH, int KW, int stride, int pad,
                                        int dil, void* stream) {
  const int Hout = (Hin + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  const int Wout = (Win + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  return cn_wgrad_generic(dy, dybs, Cout, Hout, Wout, x, xbs, Cin, Hin, Win, stride, KH, KW, dil, pad, dw, B,
                          (hipStream_t)stream);
}

// ConvTranspose2d: dw [Cin][Cout][KH][KW] += x (small grid) (*) dy (gathered at stride s).
extern "C" int cn_conv_transpose2d_bwd_weight_f32(const float* x, long xbs, const float* dy, long dybs, float* dw,
                                                  int B, int Cin, int Hin, int Win, int Cout, int KH, int KW,
                                                  int stride, int pad, void* stream) {
  const int Hout = (Hin - 1) * stride - 2 * pad + KH;
  const int Wout = (Win - 1) * stride - 2 * pad + KW;
  return cn_wgrad_generic(x, xbs, Cin, Hin, Win, dy, dybs, Cout, Hout, Wout, stride, KH, KW, 1, pad, dw, B,
                          (hipStream_t)stream);
}
