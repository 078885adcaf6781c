// Input / output edges of the hot path, HBM-bound streaming kernels.
//   input : EdgeDataset.get scaling + clip + z-score (/root/reference/src/cultionet/data/datasets.py:443-446,
//           utils/normalize.py:63-82) fused into one pass from the stored integer reflectances to fp32
//   output: LightningGTiffWriter slice-off-padding + x10000 + clip to uint16
//           (/root/reference/src/cultionet/callbacks.py:176-227)
#include "cn_common.h"

template <typename TIn>
__global__ __launch_bounds__(256) void cn_prepare_chips_kernel(const TIn* __restrict__ x, float* __restrict__ y,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ stdv, int C, long L,
                                                              float scale, float lo, float hi) {
  const int c = blockIdx.y, b = blockIdx.z;
  const long base = ((long)b * C + c) * L;
  const float m = mean ? mean[c] : 0.f, inv = stdv ? 1.0f / stdv[c] : 1.f;
  for (long l = blockIdx.x * 256L + threadIdx.x; l < L; l += (long)gridDim.x * 256) {
    float v = (float)x[base + l] * scale;
    v = fminf(fmaxf(v, lo), hi);
    y[base + l] = (v - m) * inv;
  }
}

// x: [B][C][L] raw values (dtype: 0 f32, 1 i32, 2 i16, 3 u16); y: fp32 [B][C][L];
// y = (clip(x * scale, lo, hi) - mean[c]) / std[c]   (mean/std nullable: no z-score)
extern "C" int cn_prepare_chips_f32(const void* x, int dtype, float* y, const float* mean, const float* stdv, int B,
                                    int C, long L, float scale, float lo, float hi, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || C <= 0 || L <= 0) return CN_OK;
  long bx = (L + 1023) / 1024;
  if (bx > 1024) bx = 1024;
  dim3 grid((unsigned)bx, C, B);
  switch (dtype) {
    case 0: CN_LAUNCH(cn_prepare_chips_kernel<float>, grid, dim3(256), 0, stream, (const float*)x, y, mean, stdv, C, L, scale, lo, hi); break;
    case 1: CN_LAUNCH(cn_prepare_chips_kernel<int>, grid, dim3(256), 0, stream, (const int*)x, y, mean, stdv, C, L, scale, lo, hi); break;
    case 2: CN_LAUNCH(cn_prepare_chips_kernel<short>, grid, dim3(256), 0, stream, (const short*)x, y, mean, stdv, C, L, scale, lo, hi); break;
    case 3: CN_LAUNCH(cn_prepare_chips_kernel<unsigned short>, grid, dim3(256), 0, stream, (const unsigned short*)x, y, mean, stdv, C, L, scale, lo, hi); break;
    default: return CN_ERR_ARG;
  }
  return cn_check_launch();
}

// out[b][k][i][j] = (uint16) clip(p_k[b][pad_top+i][pad_left+j] * scale, 0, scale), k = distance, edge, crop
__global__ __launch_bounds__(256) void cn_predictions_u16_kernel(const float* __restrict__ dist,
                                                                const float* __restrict__ edge,
                                                                const float* __restrict__ crop,
                                                                unsigned short* __restrict__ out, int H, int W,
                                                                int pad_top, int pad_left, int h, int w, float scale) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= h * w) return;
  const int k = blockIdx.y, b = blockIdx.z;
  const float* src = k == 0 ? dist : (k == 1 ? edge : crop);
  const int i = p / w, j = p - i * w;
  float v = src[(long)b * H * W + (long)(pad_top + i) * W + pad_left + j] * scale;
  v = fminf(fmaxf(v, 0.f), scale);
  out[((long)b * 3 + k) * h * w + p] = (unsigned short)v;
}

// dist/edge/crop: [B][1][H][W] probabilities; out: uint16 [B][3][h][w] (window without padding)
extern "C" int cn_predictions_to_u16(const float* dist, const float* edge, const float* crop, unsigned short* out,
                                     int B, int H, int W, int pad_top, int pad_left, int h, int w, float scale,
                                     void* stream) {
  if (B <= 0 || h <= 0 || w <= 0) return CN_OK;
  if (pad_top < 0 || pad_left < 0 || pad_top + h > H || pad_left + w > W) return CN_ERR_ARG;
  CN_LAUNCH(cn_predictions_u16_kernel, dim3((h * w + 255) / 256, 3, B), dim3(256), 0, (hipStream_t)stream, dist,
                     edge, crop, out, H, W, pad_top, pad_left, h, w, scale);
  return cn_check_launch();
}

// ---- sliding-window predict (BASELINE configs[4]): window + padding tiling of a scene and the stitch back --------
// Tiling semantics of the reference's predict data path (data/create.py:176-212: chunks of window_size, each grown by
// `padding` with map_overlap(boundary=0), data/store.py:69-100: zero-filled to (window_size + 2*padding)^2): window n
// with origin (r0, c0) is the crop [r0-pad, r0-pad+S) x [c0-pad, c0-pad+S) of the ZERO-extended scene, S = ws + 2*pad.
// Fused with EdgeDataset.get's scaling / clip and the z-score (as cn_prepare_chips_f32), so the scene stays in its
// stored integer type in HBM and every window batch is produced by one launch.
template <typename TIn>
__global__ __launch_bounds__(256) void cn_window_chips_kernel(const TIn* __restrict__ scene, float* __restrict__ out,
                                                             const int* __restrict__ win_rc,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ stdv, int T, int H, int W,
                                                             int S, int pad, float scale, float lo, float hi) {
  const int plane = blockIdx.y, n = blockIdx.z;
  const int r0 = win_rc[2 * n] - pad, c0 = win_rc[2 * n + 1] - pad;
  const int c = plane / T;
  const float m = mean ? mean[c] : 0.f, inv = stdv ? 1.0f / stdv[c] : 1.f;
  const TIn* src = scene + (long)plane * H * W;
  float* dst = out + ((long)n * gridDim.y + plane) * S * S;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < S * S; i += gridDim.x * 256) {
    const int y = i / S, x = i - y * S;
    const int sy = r0 + y, sx = c0 + x;
    float v = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? (float)src[(long)sy * W + sx] : 0.f;
    v = fminf(fmaxf(v * scale, lo), hi);
    dst[i] = (v - m) * inv;
  }
}

// scene: [C*T][H][W] raw values (dtype as cn_prepare_chips_f32); out: fp32 [nwin][C*T][S][S];
// win_rc: DEVICE int [nwin][2] window origins (row, col) in scene coordinates.
extern "C" int cn_window_chips_f32(const void* scene, int dtype, float* out, const int* win_rc, int nwin, int C, int T,
                                   int H, int W, int S, int pad, const float* mean, const float* stdv, float scale,
                                   float lo, float hi, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (nwin <= 0 || C <= 0 || T <= 0 || S <= 0) return CN_OK;
  if (pad < 0 || 2 * pad >= S) return CN_ERR_ARG;
  int bx = (S * S + 1023) / 1024;
  dim3 grid((unsigned)bx, C * T, nwin);
#define CN_WC(TY) CN_LAUNCH(cn_window_chips_kernel<TY>, grid, dim3(256), 0, stream, (const TY*)scene, out, \
                                     win_rc, mean, stdv, T, H, W, S, pad, scale, lo, hi)
  switch (dtype) {
    case 0: CN_WC(float); break;
    case 1: CN_WC(int); break;
    case 2: CN_WC(short); break;
    case 3: CN_WC(unsigned short); break;
    default: return CN_ERR_ARG;
  }
#undef CN_WC
  return cn_check_launch();
}

// LightningGTiffWriter.write_on_batch_end (callbacks.py:176-227) for a batch of windows, on the device: drop the
// padding, x scale, clip to [0, scale], cast to uint16 and write each window into its place of the [3][H][W] mosaic
// (windows are clipped at the scene's bottom / right edge as upstream clips window_height / window_width).
__global__ __launch_bounds__(256) void cn_stitch_u16_kernel(const float* __restrict__ dist, const float* __restrict__ edge,
                                                           const float* __restrict__ crop,
                                                           unsigned short* __restrict__ out,
                                                           const int* __restrict__ win_rc, int S, int pad, int ws, int H,
                                                           int W, float scale) {
  const int k = blockIdx.y, n = blockIdx.z;
  const int r0 = win_rc[2 * n], c0 = win_rc[2 * n + 1];
  const int h = min(ws, H - r0), w = min(ws, W - c0);
  const float* src = (k == 0 ? dist : (k == 1 ? edge : crop)) + (long)n * S * S;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < h * w; i += gridDim.x * 256) {
    const int y = i / w, x = i - y * w;
    float v = src[(long)(pad + y) * S + pad + x] * scale;
    v = fminf(fmaxf(v, 0.f), scale);
    out[((long)k * H + r0 + y) * W + c0 + x] = (unsigned short)v;
  }
}

extern "C" int cn_stitch_predictions_u16(const float* dist, const float* edge, const float* crop, unsigned short* out,
                                         const int* win_rc, int nwin, int S, int pad, int ws, int H, int W, float scale,
                                         void* stream) {
  if (nwin <= 0) return CN_OK;
  if (ws <= 0 || pad < 0 || ws + 2 * pad > S) return CN_ERR_ARG;
  CN_LAUNCH(cn_stitch_u16_kernel, dim3((ws * ws + 1023) / 1024, 3, nwin), dim3(256), 0, (hipStream_t)stream,
                     dist, edge, crop, out, win_rc, S, pad, ws, H, W, scale);
  return cn_check_launch();
}
