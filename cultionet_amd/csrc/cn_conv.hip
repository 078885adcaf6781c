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

#include <cstdlib>

#define KC 8

static int cn_reduce_slices(const CnConvGeom& g, hipStream_t stream);

// ---- epilogue shared by the two implicit-GEMM kernels: D[i = cout][j = pixel]; lane = pixel -> coalesced along W.
// Split-K / summed-group launches store each block's partial tile into its own workspace slice and
// cn_conv_reduce_kernel sums the slices. (Round 4 measured the alternative -- a per-tile ticket, the block that draws a
// tile's last ticket summing its slices in index order: 67 launches fewer per fp32 step and 9 % SLOWER end to end,
// 369 -> 335 chips/s. A tile's slices are 64 KiB x 6..16 read by ONE block with dword loads at memory latency,
// ~45 us per launch, where the reduce kernel spreads the same bytes over the chip in 8 us. Tickets pay for rows of a few
// hundred floats -- cn_ticket.h -- not for slabs.)
template <int TN, int TM>
__device__ __forceinline__ void cn_igemm_epilogue(const CnConvGeom& g, const f32x16 (&acc)[TN][TM],
                                                  const bool (&pix_ok)[TM], const int (&out_off)[TM], int co_base,
                                                  int half, int grp, int split, int b,
                                                  const float* __restrict__ bias, float* __restrict__ y) {
  const int HWout = g.Hout * g.Wout;
  const bool sliced = g.slice_stride != 0;  // split-K partials go to private workspace slices (no atomics)
  float* yb = sliced ? g.part + (long)(grp * g.splits + split) * g.slice_stride + (long)b * g.Cout * HWout
                     : y + (long)b * g.ybs;
  const bool first = split == 0;  // shared_y: every group adds its bias (sum semantics)
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co_base + tn * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (co < g.Cout) {
        const float bv = (bias != nullptr && first) ? bias[co] : 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          if (pix_ok[tm]) {
            float* dst = yb + (long)co * HWout + out_off[tm];
            const float v = acc[tn][tm][r] + bv;
            if (sliced) {
              *dst = acc[tn][tm][r];
            } else if (g.atomic_out) {
              atomicAdd(dst, v);
            } else if (g.accumulate) {
              *dst += v;
            } else {
              *dst = v;
            }
          }
        }
      }
    }
  }
}

// Software-pipelined implicit GEMM.
//   NI_T: halo plane capacity = NI_T*256 floats per channel (register prefetch uses NI_T*KC VGPRs)
// grid = (tiles over all classes and images, N tiles, K splits)
template <int WAVES_N, int TN, int NI_T>
__global__ __launch_bounds__(256) void cn_conv_igemm_kernel(const CnConvGeom g) {
  constexpr int WAVES_M = 4 / WAVES_N;
  constexpr int MT = WAVES_M * 64;
  constexpr int NT = WAVES_N * TN * 32;
  constexpr int WI = (CN_MAX_TAPS * KC * (NT / 4) + 255) / 256;  // float4 weight loads per thread per chunk
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* in_lds = smem;               // [KC][plane]
  float* w_lds = smem + g.w_lds_off;  // [ntaps*KC][NT]
  int* tap_lds = reinterpret_cast<int*>(smem + g.tap_lds_off);  // [0..8] LDS tap offsets, [9..17] packed tap ids

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;

  // ---- which class / image / tile
  int bx, by, split;
  if (!cn_xcd_block(g.grid_x, g.grid_y, g.grid_x * g.grid_y * g.splits, bx, by, split)) return;
  int ci_ = 0;
  if (g.interleave) {
    ci_ = bx % g.ncls;
  } else {
#pragma unroll 1
    for (int c = 1; c < g.ncls; ++c)
      if (bx >= g.cls[c].block_begin) ci_ = c;
  }
  const int grp = g.cls[ci_].grp;
  const float* __restrict__ x = g.gx[grp];
  const float* __restrict__ wp = g.gwp[grp];
  const float* __restrict__ bias = g.gbias[grp];
  float* __restrict__ y = g.gy[grp];
  const int Hg = g.cls[ci_].Hg, Wg = g.cls[ci_].Wg, ntaps = g.cls[ci_].ntaps;
  const int pitch = g.cls[ci_].pitch, plane = g.cls[ci_].plane;
  const int min_dy = g.cls[ci_].min_dy, min_dx = g.cls[ci_].min_dx;
  const int oy0 = g.cls[ci_].oy0, ox0 = g.cls[ci_].ox0;
  const int tiles_per_img = g.cls[ci_].tiles_per_img;
  const int tile = g.interleave ? bx / g.ncls : bx - g.cls[ci_].block_begin;
  const int b = tile / tiles_per_img;
  const int m0 = (tile - b * tiles_per_img) * MT;
  if (tid < ntaps) {
    tap_lds[tid] = (g.cls[ci_].dy[tid] - min_dy) * pitch + (g.cls[ci_].dx[tid] - min_dx);
    tap_lds[CN_MAX_TAPS + tid] = g.cls[ci_].wt[tid];
  }
  __syncthreads();
  const int n0 = by * NT;
  const int Mimg = Hg * Wg;
  const int gy0 = m0 / Wg;
  const int iy_base = gy0 * g.is + min_dy;
  const int HWin = g.Hin * g.Win;

  // per-thread decode of the staged halo plane (same for every K-chunk)
  int goff[NI_T];
#pragma unroll
  for (int i = 0; i < NI_T; ++i) {
    const int e = tid + i * 256;
    if (e < plane) {
      const int r = e / pitch, c = e - r * pitch;
      const int iy = iy_base + r, ix = c + min_dx;
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
    const int gy = pc / Wg, gx = pc - gy * Wg;
    pix_lds[tm] = ((gy - gy0) * g.is) * pitch + gx * g.is + half * plane;
    out_off[tm] = (gy * g.os + oy0) * g.Wout + gx * g.os + ox0;
  }

  f32x16 acc[TN][2];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tn][tm][r] = 0.f;

  const float* xb = x + (long)b * g.xbs;
  const int nw4 = ntaps * KC * (NT / 4);
  const int nchunks = (g.Cin + KC - 1) / KC;
  int ch = split * g.chunks_per_split;
  int ch_end = ch + g.chunks_per_split;
  if (ch_end > nchunks) ch_end = nchunks;

  if (ntaps > 0 && ch < ch_end) {
    float xin[KC][NI_T];
    f32x4 win[WI];
    // chunk-independent lane offsets + bounds-checked buffer loads (see the vec kernel below): the chunk only moves the
    // scalar offset; zero-filled / idle elements read as zeros through an offset beyond num_records
    constexpr unsigned CN_OOBD = 0x80000000u;
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, 0x7fffffff, 0x00020000);
    unsigned xo[NI_T], wo[WI];
#pragma unroll
    for (int i = 0; i < NI_T; ++i) xo[i] = goff[i] >= 0 ? (unsigned)goff[i] * 4u : CN_OOBD;
#pragma unroll
    for (int j = 0; j < WI; ++j) {
      const int f = tid + j * 256;
      wo[j] = CN_OOBD;
      if (f < nw4) {
        const int row = f / (NT / 4), c4 = f - row * (NT / 4);
        const int t = row / KC, ci = row - t * KC;
        wo[j] = (unsigned)(((tap_lds[CN_MAX_TAPS + t] * g.Kpad + ci) * g.Npad + n0 + c4 * 4) * 4);
      }
    }
#define CN_PREFETCH(c0_)                                                                                   \
  {                                                                                                        \
    const int c0 = (c0_);                                                                                  \
    _Pragma("unroll") for (int ci = 0; ci < KC; ++ci) {                                                    \
      if ((c0 + ci) < g.Cin) { /* wave-uniform */                                                          \
        const int so = (c0 + ci) * HWin * 4;                                                               \
        _Pragma("unroll") for (int i = 0; i < NI_T; ++i)                                                   \
            xin[ci][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xo[i], so, 0)); \
      } else {                                                                                             \
        _Pragma("unroll") for (int i = 0; i < NI_T; ++i) xin[ci][i] = 0.f;                                 \
      }                                                                                                    \
    }                                                                                                      \
    const int wso = c0 * g.Npad * 4;                                                                       \
    _Pragma("unroll") for (int j = 0; j < WI; ++j)                                                         \
        win[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wo[j], wso, 0));     \
  }
    CN_PREFETCH(ch * KC);
    for (; ch < ch_end; ++ch) {
      __syncthreads();  // previous chunk's MFMAs are done reading LDS
#pragma unroll
      for (int ci = 0; ci < KC; ++ci)
#pragma unroll
        for (int i = 0; i < NI_T; ++i)
          if (goff[i] != -2) in_lds[ci * plane + tid + i * 256] = xin[ci][i];
#pragma unroll
      for (int j = 0; j < WI; ++j) {
        const int f = tid + j * 256;
        if (f < nw4) *reinterpret_cast<f32x4*>(w_lds + f * 4) = win[j];
      }
      __syncthreads();
      if (ch + 1 < ch_end) CN_PREFETCH((ch + 1) * KC);  // in flight while the MFMAs below run
      float opa[2][TN], opb[2][2];
#define CN_LOAD_OPS(bf_, t_, cp_, toff_)                                                                   \
  {                                                                                                        \
    const float* wrow = w_lds + ((t_) * KC + half) * NT + wn * (TN * 32) + l31;                            \
    _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) opa[bf_][tn] = wrow[(2 * (cp_)) * NT + tn * 32];     \
    opb[bf_][0] = in_lds[pix_lds[0] + (2 * (cp_)) * plane + (toff_)];                                      \
    opb[bf_][1] = in_lds[pix_lds[1] + (2 * (cp_)) * plane + (toff_)];                                      \
  }
#define CN_MFMA(bf_)                                                                                       \
  _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) _Pragma("unroll") for (int tm = 0; tm < 2; ++tm)       \
      acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa[bf_][tn], opb[bf_][tm], acc[tn][tm], 0, 0, 0);
      int toff_cur = tap_lds[0];
      CN_LOAD_OPS(0, 0, 0, toff_cur);
#pragma unroll 1
      for (int t = 0; t < ntaps; ++t) {
        const int toff_nxt = (t + 1 < ntaps) ? tap_lds[t + 1] : 0;
        CN_LOAD_OPS(1, t, 1, toff_cur);
        CN_MFMA(0);
        CN_LOAD_OPS(0, t, 2, toff_cur);
        CN_MFMA(1);
        CN_LOAD_OPS(1, t, 3, toff_cur);
        CN_MFMA(0);
        if (t + 1 < ntaps) CN_LOAD_OPS(0, t + 1, 0, toff_nxt);
        CN_MFMA(1);
        toff_cur = toff_nxt;
      }
#undef CN_LOAD_OPS
#undef CN_MFMA
    }
#undef CN_PREFETCH
  }

  cn_igemm_epilogue<TN, 2>(g, acc, pix_ok, out_off, n0 + wn * (TN * 32), half, grp, split, b, bias, y);
}

// Same GEMM with 16-byte staging ("flattened rows"): when channel planes are 16-byte aligned (H*W % 4 == 0) the
// halo rows of a tile are ONE contiguous flat range of the plane, copied with aligned float4 loads into an
// unpadded LDS image (pitch = Win). Row overruns fall outside [0, H*W) and are zero-filled per chunk; column
// overruns wrap into the neighbouring row and are masked per lane and tap at operand-read time instead.
// 4x fewer staging instructions than the dword path (the limiter of the f32 MFMA loop: one VMEM/LDS
// instruction per wave moves at most 16 B per lane whatever its width).
//   NV: float4 chunks per thread per channel (vplane <= NV*1024 floats)
template <int WAVES_N, int TN, int TM, int NV, int RP>
__global__ __launch_bounds__(256, (NV <= 1 ? 2 : 1)) void cn_conv_igemm_vec_kernel(const CnConvGeom g) {
  constexpr int WAVES_M = 4 / WAVES_N;
  constexpr int MT = WAVES_M * TM * 32;
  constexpr int NT = WAVES_N * TN * 32;
  constexpr int WI = (CN_MAX_TAPS * KC * (NT / 4) + 255) / 256;
  constexpr int VS = NV * 1024;  // LDS channel stride (floats): compile-time so channel-pair offsets are DS immediates
  // RP == 3: unpadded rows (as RP == 0) over ODD planes (H*W % 4 == 1: 25x25, 13x13, 99x99 ...): plane c starts
  // (c % 4) floats past a 16-byte boundary, so channel ci of a chunk is fetched from its plane base rounded DOWN
  // (delta = ci % 4 floats early; the chunk's first channel is a multiple of 4) and its LDS image is laid at channel
  // stride VS + 1: 16-byte piece e4 of channel ci lands at ci*VS + (ci & ~3) + e4 (aligned), i.e. plane element q at
  // ci*(VS+1) + q - f0 -- the same image as the aligned kernel's at a channel stride that is still a DS immediate.
  // The two pieces that straddle a plane's ends are masked element-wise when they are written to LDS.
  constexpr bool PADR = RP == 1 || RP == 2;
  constexpr int ODD = RP == 3 ? 1 : 0;
  constexpr int VSS = VS + ODD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* in_lds = smem;               // [KC][VS]
  float* w_lds = smem + g.w_lds_off;  // [ntaps*KC][NT]
  int* tap_lds = reinterpret_cast<int*>(smem + g.tap_lds_off);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;

  int bx, by, split;
  if (!cn_xcd_block(g.grid_x, g.grid_y, g.grid_x * g.grid_y * g.splits, bx, by, split)) return;
  int ci_ = 0;
  if (g.interleave) {
    ci_ = bx % g.ncls;
  } else {
#pragma unroll 1
    for (int c = 1; c < g.ncls; ++c)
      if (bx >= g.cls[c].block_begin) ci_ = c;
  }
  const int grp = g.cls[ci_].grp;
  const float* __restrict__ x = g.gx[grp];
  const float* __restrict__ wp = g.gwp[grp];
  const float* __restrict__ bias = g.gbias[grp];
  float* __restrict__ y = g.gy[grp];
  const int Hg = g.cls[ci_].Hg, Wg = g.cls[ci_].Wg, ntaps = g.cls[ci_].ntaps;
  const int vplane = g.cls[ci_].vplane;
  const int min_dy = g.cls[ci_].min_dy, min_dx = g.cls[ci_].min_dx;
  const int oy0 = g.cls[ci_].oy0, ox0 = g.cls[ci_].ox0;
  const int tiles_per_img = g.cls[ci_].tiles_per_img;
  const int tile = g.interleave ? bx / g.ncls : bx - g.cls[ci_].block_begin;
  const int b = tile / tiles_per_img;
  const int m0 = (tile - b * tiles_per_img) * MT;
  const int Win = g.Win;
  // RP: rows padded in LDS by a zero gap of >= 4 floats (serves both neighbours' halos); pitch % 4 == 0
  const int pitch = PADR ? ((Win + 3) / 4 * 4 + 4) : Win;
  if (tid < ntaps) {
    tap_lds[tid] = PADR ? (g.cls[ci_].dy[tid] - min_dy) * pitch + g.cls[ci_].dx[tid]
                      : (g.cls[ci_].dy[tid] - min_dy) * Win + (g.cls[ci_].dx[tid] - min_dx);
    tap_lds[CN_MAX_TAPS + tid] = g.cls[ci_].wt[tid];
    tap_lds[2 * CN_MAX_TAPS + tid] = g.cls[ci_].dx[tid];
  }
  __syncthreads();
  const int n0 = by * NT;
  const int Mimg = Hg * Wg;
  const int gy0 = m0 / Wg;
  const int HWin = g.Hin * Win;
  // flattened-row image: RP == false: unpadded, origin `start` aligned down to 16 bytes (shift sh), column
  // overruns masked at read time. RP == true (Win % 4 == 0): whole rows [iy_base, iy_base+rows) x [0, Win), each
  // followed by a 4-float zero gap in LDS, so taps that step over a row end read zeros and need no mask.
  const int start = PADR ? (gy0 * g.is + min_dy) * Win : (gy0 * g.is + min_dy) * Win + min_dx;
  const int f0 = (start >> 2) << 2;  // aligned down (arithmetic shift: floor)
  const int sh = start - f0;         // 0 when RP == 1; 0 or 2 when RP == 2

  int goff[NV];   // flat index of this thread's float4 chunk, -1 zero-fill, -2 idle
  int ldst[NV];   // LDS float offset of the chunk (RP == 2: of its first half) inside a channel image
  int ldst2[NV];  // RP == 2 (Win % 4 == 2): offset of the second half -- a chunk may straddle two rows
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int e4 = 4 * (tid + i * 256) - (PADR ? sh : 0);  // RP: element index relative to the first staged row
    const int lim = PADR ? g.cls[ci_].rows * Win : vplane;
    ldst[i] = 0;
    ldst2[i] = 0;
    if (e4 + 3 >= 0 && e4 < lim) {
      const int fq = start + e4 - (PADR ? 0 : sh);
      goff[i] = (fq >= 0 && (ODD ? fq < HWin : fq + 3 < HWin)) ? fq : -1;  // ODD: fq % 4 == 0, the last piece is partial
      if (RP == 1) {
        const int r = e4 / Win;
        ldst[i] = r * pitch + 4 + (e4 - r * Win);
      } else if (RP == 2) {
        // halves (2 floats) never straddle a row because Win is even; a half before the first row is dropped
        const int ea = e4 < 0 ? e4 + 2 : e4, eb = e4 + 2;
        const int ra = ea / Win, rb = eb / Win;
        ldst[i] = e4 < 0 ? -1 : ra * pitch + 4 + (ea - ra * Win);
        ldst2[i] = eb < lim ? rb * pitch + 4 + (eb - rb * Win) : -1;
      } else {
        ldst[i] = e4;
      }
    } else {
      goff[i] = -2;
    }
  }
  if (PADR) {  // gaps are never written by the staging: zero the channel images once
    for (int e = tid; e < KC * VS; e += 256) in_lds[e] = 0.f;
  }

  int pix_lds[TM], out_off[TM];
  unsigned colmask[TM];  // bit t: tap t's column gx*is + dx[t] lies inside [0, Win)
  bool pix_ok[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int p = m0 + wm * (TM * 32) + tm * 32 + l31;
    pix_ok[tm] = p < Mimg;
    const int pc = pix_ok[tm] ? p : Mimg - 1;
    const int gy = pc / Wg, gx = pc - gy * Wg;
    pix_lds[tm] = PADR ? ((gy - gy0) * g.is) * pitch + 4 + gx * g.is + half * VS
                     : ((gy - gy0) * g.is) * Win + gx * g.is + sh + half * VSS;
    out_off[tm] = (gy * g.os + oy0) * g.Wout + gx * g.os + ox0;
    unsigned m = 0;
    for (int t = 0; t < ntaps; ++t) {
      const int ix = gx * g.is + tap_lds[2 * CN_MAX_TAPS + t];
      if (ix >= 0 && ix < Win) m |= 1u << t;
    }
    colmask[tm] = m;
  }

  f32x16 acc[TN][TM];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tn][tm][r] = 0.f;
  const float* xb = x + (long)b * g.xbs;
  const int nw4 = ntaps * KC * (NT / 4);
  const int nchunks = (g.Cin + KC - 1) / KC;
  int ch = split * g.chunks_per_split;
  int ch_end = ch + g.chunks_per_split;
  if (ch_end > nchunks) ch_end = nchunks;

  if (ntaps > 0 && ch < ch_end) {
    f32x4 xin[KC][NV];
    f32x4 win[WI];
    // Per-chunk prefetch = bounds-checked buffer loads with everything that does not depend on the chunk folded into
    // per-lane byte offsets once: the chunk only moves the SCALAR offset (channel plane / weight row). Zero-filled and
    // idle pieces carry an offset beyond num_records (the hardware returns zeros): no EXEC masking, no zero
    // initialisation, no tap-table reads or 64-bit address arithmetic per piece (was ~290 instructions per chunk, issued
    // with the matrix pipe of a one / two waves-per-SIMD kernel waiting).
    constexpr unsigned CN_OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, 0x7fffffff, 0x00020000);
    unsigned xo[NV], wo[WI];
#pragma unroll
    for (int i = 0; i < NV; ++i) xo[i] = goff[i] >= 0 ? (unsigned)goff[i] * 4u : CN_OOB;
#pragma unroll
    for (int j = 0; j < WI; ++j) {
      const int f = tid + j * 256;
      wo[j] = CN_OOB;
      if (f < nw4) {
        const int row = f / (NT / 4), c4 = f - row * (NT / 4);
        const int t = row / KC, ci = row - t * KC;
        wo[j] = (unsigned)(((tap_lds[CN_MAX_TAPS + t] * g.Kpad + ci) * g.Npad + n0 + c4 * 4) * 4);
      }
    }
#define CN_PREFETCH_V(c0_)                                                                                 \
  {                                                                                                        \
    const int c0 = (c0_);                                                                                  \
    _Pragma("unroll") for (int ci = 0; ci < KC; ++ci) {                                                    \
      if ((c0 + ci) < g.Cin) { /* wave-uniform */                                                          \
        const int so = ODD ? (((c0 + ci) * HWin) & ~3) * 4 : (c0 + ci) * HWin * 4;                                                               \
        _Pragma("unroll") for (int i = 0; i < NV; ++i)                                                     \
            xin[ci][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xo[i], so, 0)); \
      } else {                                                                                             \
        _Pragma("unroll") for (int i = 0; i < NV; ++i) xin[ci][i] = f32x4{0.f, 0.f, 0.f, 0.f};             \
      }                                                                                                    \
    }                                                                                                      \
    const int wso = c0 * g.Npad * 4;                                                                       \
    _Pragma("unroll") for (int j = 0; j < WI; ++j)                                                         \
        win[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wo[j], wso, 0));     \
  }
    CN_PREFETCH_V(ch * KC);
    for (; ch < ch_end; ++ch) {
      __syncthreads();
#pragma unroll
      for (int ci = 0; ci < KC; ++ci)
#pragma unroll
        for (int i = 0; i < NV; ++i)
          if (goff[i] != -2) {
            if (RP == 2) {
              if (ldst[i] >= 0) *reinterpret_cast<float2*>(in_lds + ci * VS + ldst[i]) = make_float2(xin[ci][i][0], xin[ci][i][1]);
              if (ldst2[i] >= 0) *reinterpret_cast<float2*>(in_lds + ci * VS + ldst2[i]) = make_float2(xin[ci][i][2], xin[ci][i][3]);
            } else if (ODD) {
              f32x4 v = xin[ci][i];
              const int dl = ci & 3;  // floats this channel's pieces start before the plane-relative index (unrolled: a constant)
              const bool first = dl > 0 && goff[i] == 0;         // first piece: its first dl floats are the previous plane's
              const bool last = dl < 3 && goff[i] == HWin - 1;   // last piece: only elements 0..dl are inside the plane
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if ((first && j < dl) || (last && j > dl)) v[j] = 0.f;
              *reinterpret_cast<f32x4*>(in_lds + ci * VS + (ci & ~3) + ldst[i]) = v;
            } else {
              *reinterpret_cast<f32x4*>(in_lds + ci * VS + ldst[i]) = xin[ci][i];
            }
          }
#pragma unroll
      for (int j = 0; j < WI; ++j) {
        const int f = tid + j * 256;
        if (f < nw4) *reinterpret_cast<f32x4*>(w_lds + f * 4) = win[j];
      }
      __syncthreads();
      if (ch + 1 < ch_end) CN_PREFETCH_V((ch + 1) * KC);
      // Per tap: one LDS base address and one 0/-1 column mask per pixel tile; the four channel-pair steps then
      // need no address VALU (compile-time immediates) and one v_and per operand. The f32 MFMA shares the SIMD's
      // vector issue port with VALU work, so instructions-per-MFMA is what bounds this loop.
#pragma unroll 1
      for (int t = 0; t < ntaps; ++t) {
        const int toff = tap_lds[t];
        const float* wrow = w_lds + (t * KC + half) * NT + wn * (TN * 32) + l31;
        const float* brow[TM];
        int msk[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          brow[tm] = in_lds + pix_lds[tm] + toff;
          msk[tm] = PADR ? -1 : __builtin_amdgcn_sbfe((int)colmask[tm], t, 1);
        }
#pragma unroll
        for (int cp = 0; cp < KC / 2; ++cp) {
          float a[TN], bb[TM];
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) a[tn] = wrow[(2 * cp) * NT + tn * 32];
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
            bb[tm] = PADR ? brow[tm][(2 * cp) * VS] : __int_as_float(__float_as_int(brow[tm][(2 * cp) * VSS]) & msk[tm]);
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
              acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tn], bb[tm], acc[tn][tm], 0, 0, 0);
        }
      }
    }
#undef CN_PREFETCH_V
  }

  cn_igemm_epilogue<TN, TM>(g, acc, pix_ok, out_off, n0 + wn * (TN * 32), half, grp, split, b, bias, y);
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

// Batched repack of every weight tensor of a model in ONE launch (descriptor table in device memory):
// after an optimizer step all ~180 packed copies are stale at once.
struct CnPackDesc {
  const float* w;
  float* wp;
  int T, K, N, Kpad, Npad;
  long sk, sn, st;
};

// Tile-transposed: the source is contiguous along (mid, tap) for a fixed index of its slowest dimension, the
// packed copy along n. A block stages 32 (slow) x 32 (mid) x T floats through LDS so that BOTH the reads (runs of
// 32*T floats) and the writes (runs of 32 n) are coalesced; the strided gather this replaces read 588 MB to write
// 85 MB per step (r01 PMC pass).
#define CN_PK_TS 32
#define CN_PK_PITCH (32 * CN_MAX_TAPS + 1)
__global__ __launch_bounds__(256) void cn_pack_weights_batched_kernel(const CnPackDesc* __restrict__ descs) {
  __shared__ float tile[CN_PK_TS * CN_PK_PITCH];
  const CnPackDesc d = descs[blockIdx.y];
  const bool n_slow = d.sn > d.sk;             // which of (k, n) is the slowest source dimension
  const long ss = n_slow ? d.sn : d.sk, sm = n_slow ? d.sk : d.sn;
  const int S = n_slow ? d.N : d.K, M = n_slow ? d.K : d.N;
  const int Spad = n_slow ? d.Npad : d.Kpad, Mpad = n_slow ? d.Kpad : d.Npad;
  if (d.T > CN_MAX_TAPS || d.st != 1 || sm != d.T) {  // generic strides: plain gather
    const long total = (long)d.T * d.Kpad * d.Npad;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
      const int n = (int)(i % d.Npad);
      const long r = i / d.Npad;
      const int k = (int)(r % d.Kpad);
      const int t = (int)(r / d.Kpad);
      d.wp[i] = (k < d.K && n < d.N) ? d.w[k * d.sk + n * d.sn + t * d.st] : 0.f;
    }
    return;
  }
  const int T = d.T;
  const int ts = (Spad + CN_PK_TS - 1) / CN_PK_TS, tm = (Mpad + 31) / 32;
  const int run = 32 * T;
  for (int tl = blockIdx.x; tl < ts * tm; tl += gridDim.x) {
    const int s0 = (tl / tm) * CN_PK_TS, m0 = (tl % tm) * 32;
    __syncthreads();
    for (int s = threadIdx.x >> 6; s < CN_PK_TS; s += 4)  // one wave per source row: contiguous run of 32*T floats
      for (int j = threadIdx.x & 63; j < run; j += 64) {
        const long mi = (long)m0 * T + j;
        tile[s * CN_PK_PITCH + j] = (s0 + s < S && mi < (long)M * T) ? d.w[(s0 + s) * ss + mi] : 0.f;
      }
    __syncthreads();
    // output rows: (t, k) with 32 consecutive n each
    const int ln = threadIdx.x & 31;
    for (int r = threadIdx.x >> 5; r < 32 * T; r += 8) {
      const int t = r / 32, o = r % 32;  // o: the non-n index inside the tile
      int k, n;
      float v;
      if (n_slow) { k = m0 + o; n = s0 + ln; v = tile[ln * CN_PK_PITCH + o * T + t]; }
      else        { k = s0 + o; n = m0 + ln; v = tile[o * CN_PK_PITCH + ln * T + t]; }
      if (k < d.Kpad && n < d.Npad) d.wp[((long)t * d.Kpad + k) * d.Npad + n] = v;
    }
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
  CN_LAUNCH(cn_pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wp, T, K, N,
                     Kpad, Npad, sk, sn, st);
  return cn_check_launch();
}

// descs: DEVICE array of n descriptors {w, wp, T, K, N, Kpad, Npad, sk, sn, st} (layout of CnPackDesc: two
// pointers, five ints + 4 bytes padding, three longs = 64 bytes).
extern "C" int cn_pack_weights_batched_f32(const void* descs, int n, void* stream) {
  if (n <= 0) return CN_OK;
  CN_LAUNCH(cn_pack_weights_batched_kernel, dim3(64, n), dim3(256), 0, (hipStream_t)stream,
                     (const CnPackDesc*)descs);
  return cn_check_launch();
}

template <int WAVES_N, int TN, int TM, int NV, int RP>
static int cn_launch_igemm_v(const float* x, const float* wp, const float* bias, float* y, CnConvGeom& g,
                             int total_tiles, int max_taps, int splits, double flops, hipStream_t stream) {
  constexpr int NT = WAVES_N * TN * 32;
  g.tap_lds_off = g.w_lds_off + (max_taps > 0 ? max_taps : 1) * KC * NT;
  // the staging prefetch addresses an image through a buffer resource with 32-bit byte offsets (num_records 2 GiB):
  // refuse images (or packed weight sets) beyond that instead of wrapping silently
  if ((long)g.Cin * g.Hin * g.Win * 4 >= (1L << 31) || (long)(max_taps > 0 ? max_taps : 1) * g.Kpad * g.Npad * 4 >= (1L << 31))
    return CN_ERR_ARG;
  const size_t lds = (size_t)(g.tap_lds_off + 3 * CN_MAX_TAPS + 1) * sizeof(float);
  if (lds > 160 * 1024) return CN_ERR_LDS;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)cn_conv_igemm_vec_kernel<WAVES_N, TN, TM, NV, RP>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  g.grid_x = total_tiles;
  g.grid_y = (g.Cout + NT - 1) / NT;
  g.splits = splits < 1 ? 1 : splits;
  dim3 grid(cn_xcd_grid((long)g.grid_x * g.grid_y * g.splits));
  cn_prof_name("cn_conv_igemm_vec_kernel<%d, %d, %d, %d, %d>", WAVES_N, TN, TM, NV, RP);
  cn_prof_desc("igemm_vec<%d,%d,%d,%d,%d> G%d B%d %d->%d %dx%d->%dx%d taps%d cls%d is%d os%d grid%dx%dx%d", WAVES_N, TN,
               TM, NV, RP, g.G, g.B, g.Cin, g.Cout, g.Hin, g.Win, g.Hout, g.Wout, max_taps, g.ncls, g.is, g.os,
               total_tiles, (g.Cout + NT - 1) / NT, splits);
  cn_prof_bytes(4.0 * g.G * ((double)g.B * g.Cin * g.Hin * g.Win + (double)g.B * g.Cout * g.Hout * g.Wout / (g.shared_y ? g.G : 1) +
                             (double)(max_taps > 0 ? max_taps : 1) * g.Cin * g.Cout));
  cn_prof_before(stream);
  CN_LAUNCH((cn_conv_igemm_vec_kernel<WAVES_N, TN, TM, NV, RP>), grid, dim3(256), lds, stream, g);
  cn_prof_after(stream, NT == 128 ? 0 : 1, flops);  // the contraction kernel alone (matches rocprof's per-kernel rows)
  const int rrc = cn_reduce_slices(g, stream);
  return rrc != CN_OK ? rrc : cn_check_launch();
}

template <int WAVES_N, int TN, int NI_T>
static int cn_launch_igemm_t(const float* x, const float* wp, const float* bias, float* y, CnConvGeom& g,
                             int total_tiles, int max_taps, int splits, double flops, hipStream_t stream) {
  constexpr int NT = WAVES_N * TN * 32;
  g.tap_lds_off = g.w_lds_off + (max_taps > 0 ? max_taps : 1) * KC * NT;
  // the staging prefetch addresses an image through a buffer resource with 32-bit byte offsets (num_records 2 GiB):
  // refuse images (or packed weight sets) beyond that instead of wrapping silently
  if ((long)g.Cin * g.Hin * g.Win * 4 >= (1L << 31) || (long)(max_taps > 0 ? max_taps : 1) * g.Kpad * g.Npad * 4 >= (1L << 31))
    return CN_ERR_ARG;
  const size_t lds = (size_t)(g.tap_lds_off + 3 * CN_MAX_TAPS + 1) * sizeof(float);
  if (lds > 160 * 1024) return CN_ERR_LDS;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)cn_conv_igemm_kernel<WAVES_N, TN, NI_T>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  g.grid_x = total_tiles;
  g.grid_y = (g.Cout + NT - 1) / NT;
  g.splits = splits < 1 ? 1 : splits;
  dim3 grid(cn_xcd_grid((long)g.grid_x * g.grid_y * g.splits));
  cn_prof_name("cn_conv_igemm_kernel<%d, %d, %d>", WAVES_N, TN, NI_T);
  cn_prof_desc("igemm_dw<%d,%d,%d> G%d B%d %d->%d %dx%d->%dx%d taps%d cls%d is%d os%d grid%dx%dx%d", WAVES_N, TN, NI_T,
               g.G, g.B, g.Cin, g.Cout, g.Hin, g.Win, g.Hout, g.Wout, max_taps, g.ncls, g.is, g.os, total_tiles,
               (g.Cout + NT - 1) / NT, splits);
  cn_prof_bytes(4.0 * g.G * ((double)g.B * g.Cin * g.Hin * g.Win + (double)g.B * g.Cout * g.Hout * g.Wout / (g.shared_y ? g.G : 1) +
                             (double)(max_taps > 0 ? max_taps : 1) * g.Cin * g.Cout));
  cn_prof_before(stream);
  CN_LAUNCH((cn_conv_igemm_kernel<WAVES_N, TN, NI_T>), grid, dim3(256), lds, stream, g);
  cn_prof_after(stream, NT == 128 ? 0 : 1, flops);  // the contraction kernel alone (matches rocprof's per-kernel rows)
  const int rrc = cn_reduce_slices(g, stream);
  return rrc != CN_OK ? rrc : cn_check_launch();
}

// Per-class LDS / tiling geometry for pixel tiles of MT pixels.
struct CnPlan {
  int total_tiles, max_plane, max_vplane, max_taps;
  int max_rplane, max_absdx;  // row-padded image size (rows*(Win+4)+4) and largest |dx|
  double flops;
};

static CnPlan cn_plan(CnConvGeom& g, int MT) {
  CnPlan p = {0, 0, 0, 0, 0, 0, 0.0};
  for (int c = 0; c < g.ncls; ++c) {
    CnConvClass& k = g.cls[c];
    const int Mimg = k.Hg * k.Wg;
    int rows_g = (MT + k.Wg - 2) / k.Wg + 1;
    if (rows_g > k.Hg) rows_g = k.Hg;
    k.min_dy = 0; k.min_dx = 0;
    int max_dy = 0, max_dx = 0;
    for (int t = 0; t < k.ntaps; ++t) {
      if (t == 0 || k.dy[t] < k.min_dy) k.min_dy = k.dy[t];
      if (t == 0 || k.dx[t] < k.min_dx) k.min_dx = k.dx[t];
      if (t == 0 || k.dy[t] > max_dy) max_dy = k.dy[t];
      if (t == 0 || k.dx[t] > max_dx) max_dx = k.dx[t];
    }
    const int rows = (rows_g - 1) * g.is + (max_dy - k.min_dy) + 1;
    k.pitch = (k.Wg - 1) * g.is + (max_dx - k.min_dx) + 1;
    k.plane = rows * k.pitch;
    k.vplane = (rows * g.Win + (max_dx - k.min_dx) + 4 + 3) / 4 * 4;
    k.rows = rows;
    {
      const int rpitch = (g.Win + 3) / 4 * 4 + 4;  // row-padded image: + one chunk of slack for the aligned-down start
      if ((rows + 1) * rpitch + 4 > p.max_rplane) p.max_rplane = (rows + 1) * rpitch + 4;
    }
    if (-k.min_dx > p.max_absdx) p.max_absdx = -k.min_dx;
    if (max_dx > p.max_absdx) p.max_absdx = max_dx;
    k.tiles_per_img = (Mimg + MT - 1) / MT;
    k.block_begin = p.total_tiles;
    p.total_tiles += g.B * k.tiles_per_img;
    if (k.plane > p.max_plane) p.max_plane = k.plane;
    if (k.vplane > p.max_vplane) p.max_vplane = k.vplane;
    if (k.ntaps > p.max_taps) p.max_taps = k.ntaps;
    p.flops += 2.0 * g.B * Mimg * (double)g.Cout * g.Cin * k.ntaps;
  }
  // parity classes of a strided scatter: 1 / 2 / 2 / 4 taps per class at stride 2. Laid out class after class, the
  // XCD-aware block order hands each XCD ONE class -- two XCDs run the 4-tap class for the whole launch while two others
  // finish the 1-tap class in a quarter of the time (0.56 of the machine at best). With equal tile counts the classes are
  // interleaved instead (CnConvGeom::interleave; cn_scatter_conv_g sorts them by descending taps).
  g.interleave = 0;
  if (g.ncls > 1 && g.want_interleave) {
    bool same = true;
    for (int c = 1; c < g.ncls; ++c) same = same && g.cls[c].tiles_per_img == g.cls[0].tiles_per_img;
    g.interleave = same ? 1 : 0;
  }
  return p;
}

// Tile / split-K choice. The f32 MFMA loop keeps ~2 blocks per CU resident (512 slots on the chip), so a launch
// costs ceil(blocks / 512) rounds of (MT pixels x chunks-per-split); pick the pixel tile and the K split that
// minimise it (e.g. 8 chips of 100x100: MT=160 gives 504 blocks = one round, MT=128 gives 632 = two).
struct CnChoice { int cfg, splits, cps; double cost; };

// Cost model, in units of (one 9-tap K-chunk of one pixel row of a 128-cout tile) = 7.5 us / 128, calibrated on the
// K-split sweeps of tools/ksplit.py (128->128 at 25^2 / 50^2, 256->256 at 13^2; within ~10 % for 1..8 splits):
//   a block alone on its CU runs a chunk in ~MT units, two co-resident blocks in ~1.56 MT each (they share the
//   SIMDs' MFMA pipes); staging + barriers are ~1/3 of a 9-tap chunk and do not shrink with fewer taps; every
//   round of blocks pays a fixed prologue/epilogue of ~240 units (14 us); a K-split adds the reduce launch (~140
//   units) plus writing and re-reading the partial slices (mostly L2 / Infinity-Cache resident: ~8 TB/s). 512 blocks fit the chip at once (2 per CU).
static double cn_launch_cost(long blocks, int cps, double mt_units, int splits, double out_elems) {
  // per-chunk cost: alone on the CU 36.7 + 0.614 MT', two co-resident blocks 30 + 1.19 MT' each (MT' = pixel tile
  // scaled by NT/128 and taps/9); fitted to the (tile, split) sweeps of tools/kcfg.py on 128->128 at 50^2
  // (Round 5 re-checked R INSIDE the step, where a split conv also competes with the weight-gradient stream for the CUs it
  // fills: R = 0 / 80 / 170 / 340 / 680 / 1400 -> 386.0 / 386.0 / 387.3 / 385.3 / 377.1 / 359.8 chips/s, no split at all
  // 347.8: the isolated calibration holds.)
  const double F = 273.0, R = 170.0;
  const double c1 = 36.7 + 0.614 * mt_units, c2 = 30.0 + 1.19 * mt_units;
  const long full = blocks / 512, rem = blocks % 512;
  double c = full * (F + cps * c2);
  if (rem > 0) c += F + cps * (rem > 256 ? c2 : c1);
  if (splits > 1) c += R + out_elems * 4.0 * (2.0 * splits + 1.0) / 8.0e6 * (128.0 / 7.5);
  return c;
}

// Split-K scratch is registered PER STREAM (cn_conv_set_workspace): launches on different streams never share a
// buffer. Each entry point binds its stream's buffer into these thread-locals before planning the launch.
static thread_local float* g_conv_ws;
static thread_local long g_conv_ws_floats;

static CnChoice cn_choose(const CnConvGeom& g0, const int* mts, int ncfg, int NT, bool allow_split) {
  const int nchunks = (g0.Cin + KC - 1) / KC;
  const int ny = (g0.Cout + NT - 1) / NT;
  const int G = g0.G > 0 ? g0.G : 1;
  const double out_elems = (double)g0.B * g0.Cout * g0.Hout * g0.Wout * (g0.shared_y ? 1 : G);
  CnChoice best = {0, 1, nchunks > 0 ? nchunks : 1, 1e300};
  int forced = 0;
  {
    static const char* dbg = getenv("CN_DBG_SPLITS");  // tuning aid: force the K split
    if (dbg) forced = atoi(dbg) < 1 ? 1 : atoi(dbg);
  }
  // with a workspace the splits must fit it as slices (the atomic fallback on a large output is far slower)
  long ws_cap = 32;
  if (g_conv_ws != nullptr) {
    const long stride = ((long)g0.B * g0.Cout * g0.Hout * g0.Wout + 3) / 4 * 4;
    ws_cap = g_conv_ws_floats / (stride * G);
  }
  int forced_cfg = -1;
  {
    static const char* dbgc = getenv("CN_DBG_CFG");  // tuning aid: force the pixel-tile config index
    if (dbgc) forced_cfg = atoi(dbgc);
  }
  for (int i = 0; i < ncfg; ++i) {
    if (forced_cfg >= 0 && i != (forced_cfg < ncfg ? forced_cfg : ncfg - 1)) continue;
    CnConvGeom gi = g0;
    const CnPlan p = cn_plan(gi, mts[i]);
    if (p.total_tiles <= 0) continue;
    const long base = (long)p.total_tiles * ny;
    int max_splits = (allow_split && p.max_taps > 0 && nchunks >= 4) ? nchunks / 2 : 1;
    if (max_splits > 32) max_splits = 32;
    if (max_splits > ws_cap) max_splits = ws_cap < 1 ? 1 : (int)ws_cap;
    // average taps per tile over the classes (parity classes of a strided scatter have 0..4 of the 9 taps)
    const double taps = p.total_tiles > 0 ? p.flops / (2.0 * g0.Cout * g0.Cin * (double)mts[i] * p.total_tiles) : 9.0;
    const double chunk_units = (double)mts[i] * NT / 128.0 * (taps < 9.0 ? taps : 9.0) / 9.0;
    for (int sp = 1; sp <= max_splits; ++sp) {
      if (forced && sp != (forced > max_splits ? max_splits : forced)) continue;
      const int cps = (nchunks + sp - 1) / sp;
      const int splits = (nchunks + cps - 1) / cps;  // the split count that cps really gives
      if (splits != sp && !forced) continue;
      const double cost = cn_launch_cost(base * splits, cps, chunk_units, splits, out_elems);
      if (cost < best.cost * (i == best.cfg ? 1.0 : 0.98)) best = {i, splits, cps, cost};  // switch tile only for >= 2 %
    }
  }
  return best;
}

// Optional scratch for split-K partial slices (cn_conv_set_workspace): with it, a K-split launch stores each
// split's partial tile into its own slice with plain stores and cn_conv_reduce_kernel sums the slices (+ bias)
// into y -- no memset of y, no float atomics. Without it (or if it is too small) the atomic path is used.

#include <mutex>
#include <unordered_map>
namespace {
struct CnWs { float* p; long n; };
std::mutex g_ws_mu;
std::unordered_map<void*, CnWs> g_ws_by_stream;
}  // namespace

extern "C" int cn_conv_set_workspace(void* stream, float* ws, long ws_floats) {
  if (ws != nullptr && ((reinterpret_cast<uintptr_t>(ws) & 15) || ws_floats < 0)) return CN_ERR_ARG;
  std::lock_guard<std::mutex> lk(g_ws_mu);
  if (ws == nullptr) g_ws_by_stream.erase(stream);
  else g_ws_by_stream[stream] = {ws, ws_floats};
  return CN_OK;
}

// Bind the scratch registered for `stream` (none => the atomic split-K path) for the launch being planned.
static void cn_bind_ws(void* stream) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  auto it = g_ws_by_stream.find(stream);
  if (it == g_ws_by_stream.end()) { g_conv_ws = nullptr; g_conv_ws_floats = 0; }
  else { g_conv_ws = it->second.p; g_conv_ws_floats = it->second.n; }
}

struct CnReduceArgs {
  const float* part;
  long slice_stride;
  int nsum;  // slices summed per output (splits, or G * splits when the groups share y)
  int ngrp;  // distinct outputs
  float* y[CN_MAX_GROUPS];
  const float* bias[CN_MAX_GROUPS * CN_MAX_GROUPS];  // [output][contributing group]
  int nbias;                                          // biases per output (1, or G when shared)
  long ybs;
  int B, Cout, HW, accumulate;
};

// y_g[b][c][p] (+)= sum_bias + sum_s part[g * nsum + s][b][c][p]
__global__ __launch_bounds__(256) void cn_conv_reduce_kernel(const CnReduceArgs a) {
  const long n = (long)a.Cout * a.HW;
  const int b = blockIdx.y, gi = blockIdx.z;
  const float* p0 = a.part + (long)gi * a.nsum * a.slice_stride + (long)b * n;
  float* yb = a.y[gi] + (long)b * a.ybs;
  for (long i = (blockIdx.x * 256L + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
    if (i + 3 < n && (n & 3) == 0 && (a.ybs & 3) == 0 && (reinterpret_cast<uintptr_t>(yb) & 15) == 0) {
      f32x4 s = *reinterpret_cast<const f32x4*>(p0 + i);
      for (int k = 1; k < a.nsum; ++k) s += *reinterpret_cast<const f32x4*>(p0 + (long)k * a.slice_stride + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = (int)((i + e) / a.HW);
        for (int j = 0; j < a.nbias; ++j)
          if (a.bias[gi * CN_MAX_GROUPS + j] != nullptr) s[e] += a.bias[gi * CN_MAX_GROUPS + j][c];
      }
      f32x4* d = reinterpret_cast<f32x4*>(yb + i);
      *d = a.accumulate ? *d + s : s;
    } else {
      for (long e = i; e < n && e < i + 4; ++e) {
        float s = p0[e];
        for (int k = 1; k < a.nsum; ++k) s += p0[(long)k * a.slice_stride + e];
        const int c = (int)(e / a.HW);
        for (int j = 0; j < a.nbias; ++j)
          if (a.bias[gi * CN_MAX_GROUPS + j] != nullptr) s += a.bias[gi * CN_MAX_GROUPS + j][c];
        yb[e] = a.accumulate ? yb[e] + s : s;
      }
    }
  }
}

static int cn_finish_split(CnConvGeom& g, const CnPlan& p, int splits, int cps, float* y, hipStream_t stream) {
  (void)y;
  const int nchunks = (g.Cin + KC - 1) / KC;
  g.chunks_per_split = cps > 0 ? cps : (nchunks > 0 ? nchunks : 1);
  if (p.max_taps == 0) g.chunks_per_split = nchunks > 0 ? nchunks : 1;
  g.splits = splits < 1 ? 1 : splits;
  g.atomic_out = g.splits > 1 || (g.G > 1 && g.shared_y);
  g.part = nullptr;
  g.slice_stride = 0;
  if (g.atomic_out) {
    const long stride = ((long)g.B * g.Cout * g.Hout * g.Wout + 3) / 4 * 4;
    if (g_conv_ws != nullptr && stride * g.G * g.splits <= g_conv_ws_floats) {
      g.part = g_conv_ws;
      g.slice_stride = stride;
      return CN_OK;
    }
  }
  if (g.atomic_out && !g.accumulate) {
    const int ny = g.shared_y ? 1 : g.G;
    for (int i = 0; i < ny; ++i)
      if (hipMemsetAsync(g.gy[i], 0, sizeof(float) * (size_t)g.B * g.Cout * g.Hout * g.Wout, stream) != hipSuccess)
        return CN_ERR_LAUNCH;
  }
  return CN_OK;
}

// After a sliced launch: sum the slices into the outputs.
static int cn_reduce_slices(const CnConvGeom& g, hipStream_t stream) {
  if (g.slice_stride == 0) return CN_OK;
  CnReduceArgs a = {};
  a.part = g.part;
  a.slice_stride = g.slice_stride;
  a.ybs = g.ybs; a.B = g.B; a.Cout = g.Cout; a.HW = g.Hout * g.Wout; a.accumulate = g.accumulate;
  if (g.shared_y) {
    a.nsum = g.G * g.splits; a.ngrp = 1; a.y[0] = g.gy[0]; a.nbias = g.G;
    for (int j = 0; j < g.G; ++j) a.bias[j] = g.gbias[j];
  } else {
    a.nsum = g.splits; a.ngrp = g.G; a.nbias = 1;
    for (int i = 0; i < g.G; ++i) { a.y[i] = g.gy[i]; a.bias[i * CN_MAX_GROUPS] = g.gbias[i]; }
  }
  const long n = (long)g.Cout * a.HW;
  long bx = (n + 1023) / 1024;
  if (bx > 1024) bx = 1024;
  CN_LAUNCH(cn_conv_reduce_kernel, dim3((unsigned)bx, g.B, a.ngrp), dim3(256), 0, stream, a);
  return cn_check_launch();
}

template <int WAVES_N, int TN, int TM>
static int cn_launch_vec_cfg(const float* x, const float* wp, const float* bias, float* y, CnConvGeom& g, int splits,
                             int cps, hipStream_t stream) {
  constexpr int MT = (4 / WAVES_N) * TM * 32;
  const CnPlan p = cn_plan(g, MT);
  if (p.total_tiles <= 0) return CN_OK;
  // row-padded LDS image (no read-time masks) when rows are whole 16-byte pieces and the padded image fits
  const int rp = g.odd_planes ? 3
                 : (p.max_rplane <= 4096 && p.max_absdx <= 4) ? (g.Win % 4 == 0 ? 1 : (g.Win % 2 == 0 ? 2 : 0)) : 0;
  const int img = (rp == 1 || rp == 2) ? p.max_rplane : p.max_vplane;
  // (odd planes: the channel images sit at stride VS + 1, the last one ends up to 8 floats later)
  g.w_lds_off = KC * 1024 * (img <= 1024 ? 1 : (img <= 2048 ? 2 : 4)) + (rp == 3 ? 16 : 0);
  const int rc = cn_finish_split(g, p, splits, cps, y, stream);
  if (rc != CN_OK) return rc;
#define CN_GO(NV_, RP_) \
  return cn_launch_igemm_v<WAVES_N, TN, TM, NV_, RP_>(x, wp, bias, y, g, p.total_tiles, p.max_taps, splits, p.flops, stream)
  if (rp == 1) {
    if (img <= 1024) CN_GO(1, 1);
    if (img <= 2048) CN_GO(2, 1);
    CN_GO(4, 1);
  }
  if (rp == 2) {
    if (img <= 1024) CN_GO(1, 2);
    if (img <= 2048) CN_GO(2, 2);
    CN_GO(4, 2);
  }
  if (rp == 3) {
    if (img <= 1024) CN_GO(1, 3);
    if (img <= 2048) CN_GO(2, 3);
    CN_GO(4, 3);
  }
  if (img <= 1024) CN_GO(1, 0);
  if (img <= 2048) CN_GO(2, 0);
  CN_GO(4, 0);
#undef CN_GO
}

template <int WAVES_N, int TN>
static int cn_launch_dword_cfg(const float* x, const float* wp, const float* bias, float* y, CnConvGeom& g,
                               int splits, int cps, hipStream_t stream) {
  constexpr int MT = (4 / WAVES_N) * 64;
  const CnPlan p = cn_plan(g, MT);
  if (p.total_tiles <= 0) return CN_OK;
  if (p.max_plane > 12 * 256) return CN_ERR_LDS;
  g.w_lds_off = (KC * p.max_plane + 3) / 4 * 4;
  const int rc = cn_finish_split(g, p, splits, cps, y, stream);
  if (rc != CN_OK) return rc;
  if (p.max_plane <= 3 * 256)
    return cn_launch_igemm_t<WAVES_N, TN, 3>(x, wp, bias, y, g, p.total_tiles, p.max_taps, splits, p.flops, stream);
  if (p.max_plane <= 6 * 256)
    return cn_launch_igemm_t<WAVES_N, TN, 6>(x, wp, bias, y, g, p.total_tiles, p.max_taps, splits, p.flops, stream);
  return cn_launch_igemm_t<WAVES_N, TN, 12>(x, wp, bias, y, g, p.total_tiles, p.max_taps, splits, p.flops, stream);
}

// --------------------------------------------------------------------------
// 1x1 convolutions of large pixel counts (skip convs of the tower blocks, qkv / proj of the 100x100 attention) as a
// plain GEMM  y[co][p] = sum_ci wp[ci][co] x[ci][p]: no halo, so a K-chunk of 32 channels is staged densely
// ([32][NT] weights + [32][MT] pixels = 36 KB) and there are 4x fewer barriers per MFMA than the 8-channel chunks
// of the tap kernels give a single-tap layer.
// --------------------------------------------------------------------------
struct CnGemm1x1 {
  int B, Cin, Cout, HW, Kpad, Npad, tiles_per_img, grid_x, grid_y, accumulate;
  long xbs, ybs;
};
#define KC1 32

template <int TM>
__global__ __launch_bounds__(256) void cn_conv1x1_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        const CnGemm1x1 g) {
  constexpr int MT = TM * 32, NT = 128;
  constexpr int NA = KC1 * NT / 4 / 256;             // weight float4 per thread per chunk (4)
  constexpr int NB = (KC1 * MT / 4 + 255) / 256;     // pixel float4 per thread per chunk
  __shared__ __attribute__((aligned(16))) float a_lds[KC1 * NT];
  __shared__ __attribute__((aligned(16))) float b_lds[KC1 * MT];
  int bx, by, bz;
  if (!cn_xcd_block(g.grid_x, g.grid_y, g.grid_x * g.grid_y, bx, by, bz)) return;
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int b = bx / g.tiles_per_img;
  const int p0 = (bx - b * g.tiles_per_img) * MT;
  const int n0 = by * NT;
  const float* xb = x + (long)b * g.xbs;

  f32x16 acc[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f;

  f32x4 ra[NA], rb[NB];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#define CN_G1_PREFETCH(c0_)                                                                          \
  {                                                                                                  \
    const int c0 = (c0_);                                                                            \
    _Pragma("unroll") for (int k = 0; k < NA; ++k) {                                                 \
      const int i = tid + k * 256, row = i / (NT / 4), c4 = i - row * (NT / 4);                      \
      ra[k] = (c0 + row) < g.Kpad                                                                    \
                  ? *reinterpret_cast<const f32x4*>(wp + (long)(c0 + row) * g.Npad + n0 + c4 * 4)    \
                  : zero4;                                                                           \
    }                                                                                                \
    _Pragma("unroll") for (int k = 0; k < NB; ++k) {                                                 \
      const int i = tid + k * 256, row = i / (MT / 4), q = i - row * (MT / 4);                       \
      const int p = p0 + q * 4;                                                                      \
      rb[k] = (i < KC1 * MT / 4 && (c0 + row) < g.Cin && p < g.HW)                                   \
                  ? *reinterpret_cast<const f32x4*>(xb + (long)(c0 + row) * g.HW + p)                \
                  : zero4;                                                                           \
    }                                                                                                \
  }
  const int nchunks = (g.Cin + KC1 - 1) / KC1;
  CN_G1_PREFETCH(0);
  for (int ch = 0; ch < nchunks; ++ch) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NA; ++k) *reinterpret_cast<f32x4*>(a_lds + (tid + k * 256) * 4) = ra[k];
#pragma unroll
    for (int k = 0; k < NB; ++k)
      if (tid + k * 256 < KC1 * MT / 4) *reinterpret_cast<f32x4*>(b_lds + (tid + k * 256) * 4) = rb[k];
    __syncthreads();
    if (ch + 1 < nchunks) CN_G1_PREFETCH((ch + 1) * KC1);
    const float* ap = a_lds + half * NT + wn * 32 + l31;
    const float* bp = b_lds + half * MT + l31;
#pragma unroll
    for (int k2 = 0; k2 < KC1 / 2; ++k2) {
      const float av = ap[k2 * 2 * NT];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
        acc[tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[k2 * 2 * MT + tm * 32], acc[tm], 0, 0, 0);
    }
  }
#undef CN_G1_PREFETCH
  float* yb = y + (long)b * g.ybs;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (co < g.Cout) {
      const float bv = bias != nullptr ? bias[co] : 0.f;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int p = p0 + tm * 32 + l31;
        if (p < g.HW) {
          float* dst = yb + (long)co * g.HW + p;
          const float v = acc[tm][r] + bv;
          *dst = g.accumulate ? *dst + v : v;
        }
      }
    }
  }
}

// Returns CN_ERR_ARG when the shape is not for this kernel (the caller then takes the tap kernels).
static int cn_conv1x1_launch(const float* x, long xbs, const float* wp, const float* bias, float* y, long ybs, int B,
                             int Cin, int HW, int Cout, int accumulate, hipStream_t stream) {
  if ((HW & 3) || (xbs & 3) || (reinterpret_cast<uintptr_t>(x) & 15)) return CN_ERR_ARG;
  CnGemm1x1 g = {};
  g.B = B; g.Cin = Cin; g.Cout = Cout; g.HW = HW; g.xbs = xbs; g.ybs = ybs; g.accumulate = accumulate;
  g.Kpad = cn_conv_kpad(Cin);
  g.Npad = cn_conv_npad(Cout);
  if (cn_pick_nt(Cout) != 128) return CN_ERR_ARG;
  g.grid_y = (Cout + 127) / 128;
  // pixel tile: 160 or 128, whichever needs fewer (rounds x tile) at 512 resident blocks
  const long t160 = (long)B * ((HW + 159) / 160) * g.grid_y, t128 = (long)B * ((HW + 127) / 128) * g.grid_y;
  const bool use160 = ((t160 + 511) / 512) * 160 <= ((t128 + 511) / 512) * 128;
  const long blocks = use160 ? t160 : t128;
  if (blocks < 256) return CN_ERR_ARG;  // small launches need the K split of the tap kernels
  g.tiles_per_img = use160 ? (HW + 159) / 160 : (HW + 127) / 128;
  g.grid_x = B * g.tiles_per_img;
  const double flops = 2.0 * B * HW * (double)Cout * Cin;
  cn_prof_name("cn_conv1x1_kernel<%d>", use160 ? 5 : 4);
  cn_prof_desc("gemm1x1<%d> B%d %d->%d HW%d grid%dx%d", use160 ? 5 : 4, B, Cin, Cout, HW, g.grid_x, g.grid_y);
  cn_prof_before(stream);
  if (use160)
    CN_LAUNCH((cn_conv1x1_kernel<5>), dim3(cn_xcd_grid(blocks)), dim3(256), 0, stream, x, wp, bias, y, g);
  else
    CN_LAUNCH((cn_conv1x1_kernel<4>), dim3(cn_xcd_grid(blocks)), dim3(256), 0, stream, x, wp, bias, y, g);
  cn_prof_after(stream, 0, flops);
  return cn_check_launch();
}

// ---- opt-in autotuning (cn_conv_set_autotune): the first overwriting launch of each distinct shape times the
// (tile config x K split) candidates with HIP events on the launch stream and the winner is cached for the
// process (accumulating launches of the same shape reuse it). Off by default: launches then never synchronise.
#include <functional>
static int g_autotune = 0;
static std::unordered_map<uint64_t, CnChoice> g_tuned;  // guarded by g_tune_mu
static std::mutex g_tune_mu;

extern "C" int cn_conv_set_autotune(int on) {
  std::lock_guard<std::mutex> lk(g_tune_mu);
  g_autotune = on ? 1 : 0;
  if (!on) g_tuned.clear();
  return CN_OK;
}

static uint64_t cn_tune_key(const CnConvGeom& g, bool vec, int nt, bool allow_split) {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](long v) { h = (h ^ (uint64_t)v) * 1099511628211ull; };
  mix(g.B); mix(g.Cin); mix(g.Hin); mix(g.Win); mix(g.Cout); mix(g.Hout); mix(g.Wout); mix(g.is); mix(g.os);
  mix(g.ncls); mix(g.G); mix(g.shared_y); mix(vec); mix(nt); mix(allow_split); mix(g_conv_ws != nullptr);
  for (int c = 0; c < g.ncls; ++c) {
    const CnConvClass& k = g.cls[c];
    mix(k.Hg); mix(k.Wg); mix(k.ntaps); mix(k.grp);
    for (int t = 0; t < k.ntaps; ++t) { mix(k.dy[t]); mix(k.dx[t]); }
  }
  return h;
}

static CnChoice cn_autotune(const CnConvGeom& g, const int* mts, int ncfg, int nt, bool allow_split,
                            const CnChoice& model, const std::function<int(const CnChoice&, CnConvGeom&)>& run,
                            hipStream_t stream) {
  static const int split_set[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32};
  const int nchunks = (g.Cin + KC - 1) / KC;
  const int ny = (g.Cout + nt - 1) / nt;
  long ws_cap = 32;
  if (g_conv_ws != nullptr) {
    const long stride = ((long)g.B * g.Cout * g.Hout * g.Wout + 3) / 4 * 4;
    ws_cap = g_conv_ws_floats / (stride * (g.G > 0 ? g.G : 1));
  }
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return model;
  CnChoice best = model;
  float best_ms = 1e30f;
  for (int i = 0; i < ncfg; ++i) {
    CnConvGeom gi = g;
    const CnPlan p = cn_plan(gi, mts[i]);
    if (p.total_tiles <= 0) continue;
    int max_splits = (allow_split && p.max_taps > 0 && nchunks >= 4) ? nchunks / 2 : 1;
    if (max_splits > 32) max_splits = 32;
    if (max_splits > ws_cap) max_splits = ws_cap < 1 ? 1 : (int)ws_cap;
    for (int sp : split_set) {
      if (sp > max_splits) break;
      const int cps = (nchunks + sp - 1) / sp;
      const int splits = (nchunks + cps - 1) / cps;
      if (splits != sp) continue;
      if ((long)p.total_tiles * ny * splits > 8192) continue;  // far past any useful split
      const CnChoice c = {i, splits, cps, 0.0};
      float ms = 0.f;
      bool ok = true;
      for (int rep = 0; rep < 2 && ok; ++rep) {  // first run warms caches / attributes, second is timed
        CnConvGeom gg = g;
        ok = hipEventRecord(e0, stream) == hipSuccess && run(c, gg) == CN_OK &&
             hipEventRecord(e1, stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
             hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
      }
      if (ok && ms < best_ms) { best_ms = ms; best = c; }
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return best;
}

int cn_conv_igemm_launch(CnConvGeom& g, hipStream_t stream) {
  if (g.ncls < 1 || g.ncls > CN_MAX_CLASSES || g.B <= 0 || g.G < 1) return g.B <= 0 ? CN_OK : CN_ERR_ARG;
  g.splits = 1;
  const float* x = g.gx[0]; const float* wp = g.gwp[0]; const float* bias = g.gbias[0]; float* y = g.gy[0];
  g.Kpad = cn_conv_kpad(g.Cin);
  g.Npad = cn_conv_npad(g.Cout);
  const int nt = cn_pick_nt(g.Cout);
  const bool dense_out = g.ybs == (long)g.Cout * g.Hout * g.Wout;
  // strided outputs cannot be memset for the atomic path: split them only if the slices surely fit the workspace
  const bool ws_fits = g_conv_ws != nullptr &&
                       (((long)g.B * g.Cout * g.Hout * g.Wout + 3) / 4 * 4) * g.G * 32 <= g_conv_ws_floats;
  const bool allow_split = dense_out || g.accumulate || ws_fits;
  if (g.G > 1 && g.shared_y && !allow_split) return CN_ERR_ARG;
  // 16-byte staging needs every image 16-byte aligned and either aligned planes (H*W % 4 == 0) or the odd-plane variant
  // (H*W % 4 == 1: every odd square -- 25x25, 13x13, 99x99, 49x49, 97x97)
  static const bool odd_vec = getenv("CN_ODD_VEC") == nullptr || atoi(getenv("CN_ODD_VEC")) != 0;  // A/B switch
  const int hw4 = (int)(((long)g.Hin * g.Win) % 4);
  // (measured per layer at batch 8, same box: 25x25 and 49x49 planes 3-9 % faster than the dword kernel, the stride-4
  // scatter 25x25 -> 97x97 1.5x; strided gathers from 99x99 -- two float4 pieces per thread and channel -- 20 % slower:
  // those stay on the dword kernel)
  g.odd_planes = (hw4 == 1 && odd_vec && !(g.is > 1 && (long)g.Hin * g.Win > 4096)) ? 1 : 0;
  bool vec = (hw4 == 0 || g.odd_planes) && (g.xbs % 4 == 0);
  // ... and the same holds for ALIGNED large planes (round 6: with ConvTranspose2d on the 100 x 100 output_padding grid its
  // backward-data gather became eligible for the 16-byte kernel: 83.7 vs 78.3 us alone, and 336 vs 177 us inside the step
  // next to the weight-gradient stream -- one block per CU of a two-chunk image; step 390.0 -> 392.2 chips/s, same box)
  if (g.is > 1 && (long)g.Hin * g.Win > 4096) vec = false;
  // the kernels' buffer loads carry 31-bit byte offsets inside one image / one packed weight tensor
  if ((long)g.Cin * g.Hin * g.Win * 4 >= (1L << 31) || (long)CN_MAX_TAPS * cn_conv_kpad(g.Cin) * g.Npad * 4 >= (1L << 31))
    return CN_ERR_ARG;
  for (int i = 0; i < g.G; ++i) vec = vec && ((reinterpret_cast<uintptr_t>(g.gx[i]) & 15) == 0);
  if (vec) {  // the flattened-row image of the largest tile must fit 4096 floats
    CnConvGeom t = g;
    vec = cn_plan(t, 256).max_vplane <= 4 * 1024;
  }
  // ---- configuration: tile config x K split, from the cost model or (opt-in) measured once per shape
  static const int mts128[3] = {128, 96, 160};  // MT=192 (TM=6) needs > 256 registers: one wave per SIMD
  static const int mts128d[1] = {128};
  static const int mts256[1] = {256};
  const int* mts = nt == 128 ? (vec ? mts128 : mts128d) : mts256;
  const int ncfg = (nt == 128 && vec) ? 3 : 1;
  auto run = [&](const CnChoice& c, CnConvGeom& gg) -> int {
    if (nt == 128) {
      if (vec) {
        switch (c.cfg) {
          case 1: return cn_launch_vec_cfg<4, 1, 3>(x, wp, bias, y, gg, c.splits, c.cps, stream);
          case 2: return cn_launch_vec_cfg<4, 1, 5>(x, wp, bias, y, gg, c.splits, c.cps, stream);
          default: return cn_launch_vec_cfg<2, 2, 2>(x, wp, bias, y, gg, c.splits, c.cps, stream);
        }
      }
      return cn_launch_dword_cfg<2, 2>(x, wp, bias, y, gg, c.splits, c.cps, stream);
    }
    // The narrow-cout configurations own 256-pixel tiles. A strided gather of few output channels from a large plane
    // (ConvTranspose2d backward-data at stride 4 with 16-32 channels: 11 output rows = 44 input rows of 100) overflows
    // the dword kernel's halo plane: such a launch falls back to the 128-pixel tile of the 128-cout configuration --
    // wasteful on the cout side, but a non-default width must run, not fail with CN_ERR_LDS (found in round 6).
    int rc;
    if (nt == 64) {
      if (vec) return cn_launch_vec_cfg<1, 2, 2>(x, wp, bias, y, gg, c.splits, c.cps, stream);
      rc = cn_launch_dword_cfg<1, 2>(x, wp, bias, y, gg, c.splits, c.cps, stream);
    } else {
      if (vec) return cn_launch_vec_cfg<1, 1, 2>(x, wp, bias, y, gg, c.splits, c.cps, stream);
      rc = cn_launch_dword_cfg<1, 1>(x, wp, bias, y, gg, c.splits, c.cps, stream);
    }
    if (rc == CN_ERR_LDS) {
      CnConvGeom g2 = gg;
      rc = cn_launch_dword_cfg<2, 2>(x, wp, bias, y, g2, c.splits, c.cps, stream);
    }
    return rc;
  };
  CnChoice c = cn_choose(g, mts, ncfg, nt, allow_split);
  // (Round 6, measured for the interleaved parity-class launches at 8 x 50^2 -> 100^2, 128 -> 128: smaller pixel tiles so
  // that the classes balance dynamically -- 864 blocks of 96 pixels, 640 of 128 -- run 141.9 / 168.1 us against 105.7 at
  // 512 blocks of 160: these launches are bound by the per-chunk staging of few-tap classes, not by class imbalance.)
  if (g_autotune) {
    std::lock_guard<std::mutex> lk(g_tune_mu);  // tuning runs are serialised across threads
    const uint64_t key = cn_tune_key(g, vec, nt, allow_split);
    auto it = g_tuned.find(key);
    if (it != g_tuned.end()) {
      c = it->second;
    } else if (!g.accumulate && !cn_prof_on()) {
      // first sight of this shape on an overwriting launch: time the candidates (the output is simply rewritten)
      c = cn_autotune(g, mts, ncfg, nt, allow_split, c, run, stream);
      g_tuned.emplace(key, c);
    }
  }
  return run(c, g);
}

static inline int floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

// Fills the group tables; shared_y when every group names the same output (their results are summed).
static int cn_set_groups(CnConvGeom& g, int G, const float* const* xs, const float* const* wps,
                         const float* const* biases, float* const* ys) {
  if (G < 1 || G > CN_MAX_GROUPS) return CN_ERR_ARG;
  g.G = G;
  int same = 0;
  for (int i = 0; i < G; ++i) {
    g.gx[i] = xs[i]; g.gwp[i] = wps[i]; g.gbias[i] = biases ? biases[i] : nullptr; g.gy[i] = ys[i];
    if (ys[i] == ys[0]) ++same;
  }
  if (same != G && same != 1) return CN_ERR_ARG;  // all outputs distinct, or all the same
  for (int i = 1; i < G && same == 1; ++i)
    for (int j = 1; j < i; ++j)
      if (ys[i] == ys[j]) return CN_ERR_ARG;
  g.shared_y = (G > 1 && same == G) ? 1 : 0;
  g.has_bias = 0;
  for (int i = 0; i < G; ++i) g.has_bias |= g.gbias[i] != nullptr;
  return CN_OK;
}

// Gather form (Conv2d forward, ConvTranspose2d backward-data): one class per group,
// input coord = o*stride + k*dil - pad.
static int cn_gather_conv_g(int G, const float* const* xs, long xbs, const float* const* wps,
                            const float* const* biases, float* const* ys, long ybs, int B, int Cin, int Hin,
                            int Win, int Cout, int Hout, int Wout, int KH, int KW, int stride, const int* pads,
                            const int* dils, int accumulate, hipStream_t stream) {
  if (KH * KW > CN_MAX_TAPS || stride < 1) return CN_ERR_ARG;
  if (Hout <= 0 || Wout <= 0) return CN_OK;
  if (G == 1 && KH == 1 && KW == 1 && stride == 1 && pads[0] == 0 && Hout == Hin && Wout == Win) {
    const int r1 = cn_conv1x1_launch(xs[0], xbs, wps[0], biases ? biases[0] : nullptr, ys[0], ybs, B, Cin, Hin * Win,
                                     Cout, accumulate, stream);
    if (r1 != CN_ERR_ARG) return r1;
  }
  CnConvGeom g = {};
  const int rc = cn_set_groups(g, G, xs, wps, biases, ys);
  if (rc != CN_OK) return rc;
  g.B = B; g.Cin = Cin; g.Hin = Hin; g.Win = Win; g.Cout = Cout; g.Hout = Hout; g.Wout = Wout;
  g.xbs = xbs; g.ybs = ybs; g.is = stride; g.os = 1;
  g.ncls = G;
  for (int i = 0; i < G; ++i) {
    if (dils[i] < 1) return CN_ERR_ARG;
    CnConvClass& k = g.cls[i];
    k.grp = i;
    k.Hg = Hout; k.Wg = Wout; k.oy0 = 0; k.ox0 = 0;
    k.ntaps = KH * KW;
    for (int ky = 0; ky < KH; ++ky)
      for (int kx = 0; kx < KW; ++kx) {
        const int t = ky * KW + kx;
        k.dy[t] = ky * dils[i] - pads[i]; k.dx[t] = kx * dils[i] - pads[i]; k.wt[t] = t;
      }
  }
  g.accumulate = accumulate;
  return cn_conv_igemm_launch(g, stream);
}

static int cn_gather_conv(const float* x, long xbs, const float* wp, const float* bias, float* y, long ybs, int B,
                          int Cin, int Hin, int Win, int Cout, int Hout, int Wout, int KH, int KW, int stride,
                          int pad, int dil, int accumulate, hipStream_t stream) {
  return cn_gather_conv_g(1, &x, xbs, &wp, &bias, &y, ybs, B, Cin, Hin, Win, Cout, Hout, Wout, KH, KW, stride, &pad,
                          &dil, accumulate, stream);
}

// Conv2d forward. x [B,Cin,Hin,Win] (batch stride xbs), wp packed [KH*KW][Kpad(Cin)][Npad(Cout)],
// y [B,Cout,Hout,Wout] (batch stride ybs).
extern "C" int cn_conv2d_fwd_f32(const float* x, long xbs, const float* wp, const float* bias, float* y,
                                 long ybs, int B, int Cin, int Hin, int Win, int Cout, int KH, int KW,
                                 int stride, int pad, int dil, int accumulate, void* stream) {
  cn_bind_ws(stream);
  if (stride < 1) return CN_ERR_ARG;  // before it is divided by
  const int Hout = (Hin + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  const int Wout = (Win + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  return cn_gather_conv(x, xbs, wp, bias, y, ybs, B, Cin, Hin, Win, Cout, Hout, Wout, KH, KW, stride, pad, dil,
                        accumulate, (hipStream_t)stream);
}

// Scatter form shared by Conv2d backward-data and ConvTranspose2d forward:
//   out[o] (+)= bias + sum_k src[(o + pad - k*dil)/s] * W[k]   where divisible,
// as s*s parity classes of the output grid in ONE launch (no zero-stuffing, no wasted MACs).
// src [B,Csrc,Hs,Ws], out [B,Cdst,Ho,Wo]; wp packed [KH*KW][Kpad(Csrc)][Npad(Cdst)].
static int cn_scatter_conv_g(int G, const float* const* srcs, long sbs, const float* const* wps,
                             const float* const* biases, float* const* outs, long obs, int B, int Csrc, int Hs,
                             int Ws, int Cdst, int Ho, int Wo, const int* khs, const int* kws, int stride,
                             const int* pads, const int* dils, int accumulate, hipStream_t stream) {
  if (stride < 1 || G * stride * stride > CN_MAX_CLASSES) return CN_ERR_ARG;
  for (int i = 0; i < G; ++i)
    if (khs[i] * kws[i] > CN_MAX_TAPS || khs[i] < 1 || kws[i] < 1) return CN_ERR_ARG;
  if (G == 1 && khs[0] == 1 && kws[0] == 1 && stride == 1 && pads[0] == 0 && Ho == Hs && Wo == Ws) {
    const int r1 = cn_conv1x1_launch(srcs[0], sbs, wps[0], biases ? biases[0] : nullptr, outs[0], obs, B, Csrc,
                                     Hs * Ws, Cdst, accumulate, stream);
    if (r1 != CN_ERR_ARG) return r1;
  }
  CnConvGeom g = {};
  const int rc = cn_set_groups(g, G, srcs, wps, biases, outs);
  if (rc != CN_OK) return rc;
  g.B = B; g.Cin = Csrc; g.Hin = Hs; g.Win = Ws; g.Cout = Cdst; g.Hout = Ho; g.Wout = Wo;
  g.xbs = sbs; g.ybs = obs; g.is = 1; g.os = stride;
  g.accumulate = accumulate;
  int nc = 0;
  for (int gi = 0; gi < G; ++gi) {
    const int pad = pads[gi], dil = dils[gi], KH = khs[gi], KW = kws[gi];
    if (dil < 1) return CN_ERR_ARG;
    for (int py = 0; py < stride; ++py)
      for (int px = 0; px < stride; ++px) {
        CnConvClass& k = g.cls[nc];
        k.grp = gi;
        k.Hg = (Ho - py + stride - 1) / stride;
        k.Wg = (Wo - px + stride - 1) / stride;
        if (k.Hg <= 0 || k.Wg <= 0) continue;
        k.oy0 = py; k.ox0 = px;
        int nt = 0;
        for (int ky = 0; ky < KH; ++ky) {
          const int ny = py + pad - ky * dil;
          if (((ny % stride) + stride) % stride != 0) continue;
          for (int kx = 0; kx < KW; ++kx) {
            const int nx = px + pad - kx * dil;
            if (((nx % stride) + stride) % stride != 0) continue;
            k.dy[nt] = floordiv(ny, stride); k.dx[nt] = floordiv(nx, stride); k.wt[nt] = ky * KW + kx;
            ++nt;
          }
        }
        k.ntaps = nt;
        ++nc;
      }
  }
  g.ncls = nc;
  if (nc == 0) return CN_OK;
  // heavy classes first (stable: insertion sort over <= 16 classes), interleaved by cn_plan when their tile counts agree
  for (int i = 1; i < nc; ++i) {
    const CnConvClass key = g.cls[i];
    int j = i - 1;
    while (j >= 0 && g.cls[j].ntaps < key.ntaps) { g.cls[j + 1] = g.cls[j]; --j; }
    g.cls[j + 1] = key;
  }
  g.want_interleave = stride > 1 ? 1 : 0;  // (8 x 50^2 -> 100^2, 128 -> 128: 114.1 -> 105.7 us alone; the step does not move)
  return cn_conv_igemm_launch(g, stream);
}

static int cn_scatter_conv(const float* src, long sbs, const float* wp, const float* bias, float* out, long obs,
                           int B, int Csrc, int Hs, int Ws, int Cdst, int Ho, int Wo, int KH, int KW,
                           int stride, int pad, int dil, int accumulate, hipStream_t stream) {
  return cn_scatter_conv_g(1, &src, sbs, &wp, &bias, &out, obs, B, Csrc, Hs, Ws, Cdst, Ho, Wo, &KH, &KW, stride, &pad,
                           &dil, accumulate, stream);
}

// Conv2d backward-data: dx [B,Cin,Hin,Win] (+)= conv^T(dy [B,Cout,Hout,Wout]); wp packed with K=Cout, N=Cin.
extern "C" int cn_conv2d_bwd_data_f32(const float* dy, long dybs, const float* wp_t, float* dx, long dxbs,
                                      int B, int Cin, int Hin, int Win, int Cout, int KH, int KW, int stride,
                                      int pad, int dil, int accumulate, void* stream) {
  cn_bind_ws(stream);
  if (stride < 1) return CN_ERR_ARG;  // before it is divided by
  const int Hout = (Hin + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  const int Wout = (Win + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  return cn_scatter_conv(dy, dybs, wp_t, nullptr, dx, dxbs, B, Cout, Hout, Wout, Cin, Hin, Win, KH, KW, stride,
                         pad, dil, accumulate, (hipStream_t)stream);
}

// Grouped Conv2d forward: G (<= 4) convolutions of identical shape (they may differ in padding / dilation as long
// as the output size agrees) in ONE launch -- the dilation branches of ResidualAConv (convolution.py:376-395) and
// the three head streams of TowerUNetFinal (unet_parts.py:196-224). xs/wps/biases/ys are HOST arrays of G device
// pointers; inputs may alias (same x for every branch); outputs are all distinct or all the same (summed).
extern "C" int cn_conv2d_fwd_grouped_f32(int G, const float* const* xs, long xbs, const float* const* wps,
                                         const float* const* biases, float* const* ys, long ybs, int B, int Cin,
                                         int Hin, int Win, int Cout, int KH, int KW, int stride, const int* pads,
                                         const int* dils, int accumulate, void* stream) {
  cn_bind_ws(stream);
  if (G < 1 || G > CN_MAX_GROUPS) return CN_ERR_ARG;
  if (stride < 1) return CN_ERR_ARG;  // before it is divided by
  const int Hout = (Hin + 2 * pads[0] - dils[0] * (KH - 1) - 1) / stride + 1;
  const int Wout = (Win + 2 * pads[0] - dils[0] * (KW - 1) - 1) / stride + 1;
  for (int i = 1; i < G; ++i)
    if ((Hin + 2 * pads[i] - dils[i] * (KH - 1) - 1) / stride + 1 != Hout ||
        (Win + 2 * pads[i] - dils[i] * (KW - 1) - 1) / stride + 1 != Wout)
      return CN_ERR_ARG;
  return cn_gather_conv_g(G, xs, xbs, wps, biases, ys, ybs, B, Cin, Hin, Win, Cout, Hout, Wout, KH, KW, stride,
                          pads, dils, accumulate, (hipStream_t)stream);
}

// Grouped Conv2d backward-data; dxs all distinct, or all the same buffer (branches that share their input: the
// G contributions are summed into it). Kernel sizes are per group (khs / kws), so the 1x1 skip convolution of a
// ResidualAConv joins its 3x3 branches in the same launch.
extern "C" int cn_conv2d_bwd_data_grouped_f32(int G, const float* const* dys, long dybs, const float* const* wps_t,
                                              float* const* dxs, long dxbs, int B, int Cin, int Hin, int Win,
                                              int Cout, const int* khs, const int* kws, int stride, const int* pads,
                                              const int* dils, int accumulate, void* stream) {
  cn_bind_ws(stream);
  if (G < 1 || G > CN_MAX_GROUPS) return CN_ERR_ARG;
  if (stride < 1) return CN_ERR_ARG;  // before it is divided by
  const int Hout = (Hin + 2 * pads[0] - dils[0] * (khs[0] - 1) - 1) / stride + 1;
  const int Wout = (Win + 2 * pads[0] - dils[0] * (kws[0] - 1) - 1) / stride + 1;
  for (int i = 1; i < G; ++i)
    if ((Hin + 2 * pads[i] - dils[i] * (khs[i] - 1) - 1) / stride + 1 != Hout ||
        (Win + 2 * pads[i] - dils[i] * (kws[i] - 1) - 1) / stride + 1 != Wout)
      return CN_ERR_ARG;
  return cn_scatter_conv_g(G, dys, dybs, wps_t, nullptr, dxs, dxbs, B, Cout, Hout, Wout, Cin, Hin, Win, khs, kws,
                           stride, pads, dils, accumulate, (hipStream_t)stream);
}

// ConvTranspose2d forward: y [B,Cout,Hout,Wout], Hout = (Hin-1)*s - 2*pad + K + out_pad; wp packed with K=Cin, N=Cout.
// out_pad = nn.ConvTranspose2d's output_padding (0 <= out_pad < stride): the extra rows / columns at the bottom / right
// are ordinary transposed-convolution outputs (taps that fall outside x read zeros). The engine uses it to put the
// (2n-1)^2 result of the reference's ConvTranspose2d(3, 2, 1) on the 2n x 2n grid of the resize that follows
// (convolution.py:45-68): rows / columns [0, 2n-1) are exactly the reference's tensor, the planes are 16-byte aligned
// (99 x 99 is not), and nothing downstream needs an aligned copy or the dword-staging kernels.
extern "C" int cn_conv_transpose2d_fwd_f32(const float* x, long xbs, const float* wp, const float* bias,
                                           float* y, long ybs, int B, int Cin, int Hin, int Win, int Cout,
                                           int KH, int KW, int stride, int pad, int out_pad, int accumulate,
                                           void* stream) {
  cn_bind_ws(stream);
  if (out_pad < 0 || (out_pad > 0 && out_pad >= stride)) return CN_ERR_ARG;
  const int Hout = (Hin - 1) * stride - 2 * pad + KH + out_pad;
  const int Wout = (Win - 1) * stride - 2 * pad + KW + out_pad;
  return cn_scatter_conv(x, xbs, wp, bias, y, ybs, B, Cin, Hin, Win, Cout, Hout, Wout, KH, KW, stride, pad, 1,
                         accumulate, (hipStream_t)stream);
}

// ConvTranspose2d backward-data: dx [B,Cin,Hin,Win] (+)= conv_stride_s(dy); wp_t packed with K=Cout, N=Cin.
extern "C" int cn_conv_transpose2d_bwd_data_f32(const float* dy, long dybs, const float* wp_t, float* dx,
                                                long dxbs, int B, int Cin, int Hin, int Win, int Cout, int KH,
                                                int KW, int stride, int pad, int out_pad, int accumulate,
                                                void* stream) {
  cn_bind_ws(stream);
  if (out_pad < 0 || (out_pad > 0 && out_pad >= stride)) return CN_ERR_ARG;
  const int Hout = (Hin - 1) * stride - 2 * pad + KH + out_pad;
  const int Wout = (Win - 1) * stride - 2 * pad + KW + out_pad;
  return cn_gather_conv(dy, dybs, wp_t, nullptr, dx, dxbs, B, Cout, Hout, Wout, Cin, Hin, Win, KH, KW, stride, pad,
                        1, accumulate, (hipStream_t)stream);
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
  CN_LAUNCH(cn_pack_timeconv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
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
  CN_LAUNCH(cn_fold_timeconv_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, dwexp, dw, Cout,
                     Cin, Tin, k);
  return cn_check_launch();
}
