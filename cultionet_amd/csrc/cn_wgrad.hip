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
#include <cstdlib>
#include "cn_common.h"
#include "cn_profile.h"
#include "cn_slicesum.h"

#define WG_MAX_TAPS 9
#define WG_BC 32   // b-channels per block
#define WG_NS 4    // max S elements per lane per row of the chunk  (chunk pixels <= 256)
#define WG_NB 13   // max Bg plane elements per lane               (plane_b <= 832)
#define WG_KS 8    // 16-byte variant: DMA instructions per wave for the S image  (arows*npix <= 8192 floats)
#define WG_U 4     // 16-byte variant: k-steps per unrolled group
#define WG_KB 16   // 16-byte variant: DMA instructions per wave for the Bg image (32*plane_b <= 16384 floats)

struct CnWgradGeom {
  int N;
  int A, Hs, Ws; long sbs; long scs;   // S: channels, grid, batch stride, channel stride
  int Bc, Hb, Wb; long bbs; long bcs;  // Bg likewise
  double flops;                        // algorithmic FLOP of the launch (true, unpadded dims)
  int s;
  int T;
  int offy[WG_MAX_TAPS], offx[WG_MAX_TAPS];
  int min_oy, min_ox;
  long sa;           // dW offset = a*sa + b*T + t
  int PR, Wsp, pitch_s;
  int rows_b, pitch_b, plane_b;
  int a_tiles;       // 32-row tiles of A per block (1, 2 or 4)
  int chunks_per_img, total_chunks, chunks_per_split;
  int b_lds_off;   // float offset of the Bg planes inside one LDS buffer
  int buf_stride;  // floats per LDS buffer
  int nbuf;        // 2: double-buffered LDS-DMA pipeline, 1: single buffer
  int colsplit;    // unused
  int G, spg;                    // groups (same shapes, own tensors): logical z = group * spg + split
  const float* gS[4];
  const float* gB[4];
  float* gdW[4];
  long slice_stride;  // != 0: every (split, k-part) stores its partial dW into its own slice of a workspace
  int grid_x, grid_y, grid_z;  // logical grid (Bc tiles, A tiles, splits); launched 1-D in XCD-aware order
  int mq_lo, mq_hi;  // 16-byte variant: pixel pairs [0,mq_lo) and [mq_hi, Ws/2) of a row need column masks
};

typedef __attribute__((address_space(3))) void* cn_lds_ptr;
typedef const __attribute__((address_space(1))) void* cn_gbl_ptr;

__device__ float cn_zero_line[64];  // zero source for out-of-image lanes of an LDS-DMA load

// One LDS-DMA wave-instruction: lane l copies 4 bytes from its own `src` to lds_base[l] (no VGPR round trip).
__device__ __forceinline__ void cn_glds4(const float* src, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((cn_gbl_ptr)src, (cn_lds_ptr)lds_wave_base, 4, 0, 0);
}

// Staging is done entirely by LDS-DMA (global_load_lds): every lane always loads (out-of-image lanes read a
// zero line), so a chunk's ~70 wave-instructions are all in flight at once instead of round-tripping through
// VGPRs; with two LDS buffers the next chunk lands while the MFMAs of the current one run.
template <int T>
__global__ __launch_bounds__(256) void cn_wgrad_kernel(const float* __restrict__ S, const float* __restrict__ Bg,
                                                      float* __restrict__ dW, const CnWgradGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int at = wid % g.a_tiles, kp = wid / g.a_tiles, kparts = 4 / g.a_tiles;
  int bx, by, bz;
  if (!cn_xcd_block(g.grid_x, g.grid_y, g.grid_x * g.grid_y * g.grid_z, bx, by, bz)) return;
  const int a0 = by * g.a_tiles * 32;
  const int b0 = bx * WG_BC;
  const int npix = g.PR * g.Wsp;  // staged (padded) grid pixels per chunk, even
  const float* zero = cn_zero_line + lane;

  // ---- per-lane staging decode (chunk independent), packed (row << 16 | col)
  int so[WG_NS];  // col 0xFFFF = zero pad; -2 = lane idle for this piece
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
  int bo[WG_NB];  // -1 -> lane idle
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
  int chunk = bz * g.chunks_per_split;
  int chunk_end = chunk + g.chunks_per_split;
  if (chunk_end > g.total_chunks) chunk_end = g.total_chunks;

  // this wave's share of the k-steps of a chunk
  const int steps = npix >> 1;
  const int q_begin = (steps * kp) / kparts, q_end = (steps * (kp + 1)) / kparts;
  const int e_begin = q_begin * 2;
  const int r_begin = e_begin / g.Wsp, c_begin = e_begin - r_begin * g.Wsp;

  // ---- LDS-DMA staging of one chunk into buffer `buf`
  auto stage = [&](int ck, int buf) {
    float* s_lds = smem + buf * g.buf_stride;
    float* b_lds = s_lds + g.b_lds_off;
    const int n = ck / g.chunks_per_img;
    const int gy0 = (ck - n * g.chunks_per_img) * g.PR;
    const int pr = (g.Hs - gy0 < g.PR) ? g.Hs - gy0 : g.PR;
    const float* Sn = S + (long)n * g.sbs + (long)gy0 * g.Ws;
    for (int a = wid; a < arows; a += 4) {
      const bool aok = (a0 + a) < g.A;
      const float* Sa = Sn + (long)(a0 + a) * g.scs;
#pragma unroll
      for (int i = 0; i < WG_NS; ++i) {
        if (i * 64 < npix) {  // wave-uniform
          if (so[i] != -2) {
            const int r = so[i] >> 16, c = so[i] & 0xFFFF;
            const float* src = (aok && c != 0xFFFF && r < pr) ? Sa + r * g.Ws + c : zero;
            cn_glds4(src, s_lds + a * g.pitch_s + i * 64);
          }
        }
      }
    }
    const float* Bn = Bg + (long)n * g.bbs;
    const int iy0 = gy0 * g.s + g.min_oy;
    for (int bl = wid; bl < WG_BC; bl += 4) {
      const bool bok = (b0 + bl) < g.Bc;
      const float* Bb = Bn + (long)(b0 + bl) * g.bcs;
#pragma unroll
      for (int i = 0; i < WG_NB; ++i) {
        if (i * 64 < g.plane_b) {  // wave-uniform
          if (bo[i] >= 0) {
            const int iy = iy0 + (bo[i] >> 16), ix = g.min_ox + (bo[i] & 0xFFFF);
            const float* src = (bok && iy >= 0 && iy < g.Hb && ix >= 0 && ix < g.Wb) ? Bb + iy * g.Wb + ix : zero;
            cn_glds4(src, b_lds + bl * g.plane_b + i * 64);
          }
        }
      }
    }
  };

  int cur = 0;
  if (chunk < chunk_end && g.nbuf == 2) stage(chunk, 0);
  for (; chunk < chunk_end; ++chunk) {
    if (g.nbuf == 1) {
      __syncthreads();  // everyone is done reading the single buffer
      stage(chunk, 0);
    }
    __syncthreads();  // (vmcnt(0) first) this chunk's DMA has landed; previous compute finished
    if (g.nbuf == 2 && chunk + 1 < chunk_end) stage(chunk + 1, cur ^ 1);
    const float* s_lds = smem + cur * g.buf_stride;
    const float* b_lds = s_lds + g.b_lds_off;
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
    if (g.nbuf == 2) cur ^= 1;
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

__device__ float cn_zero_line16[256];  // 1 KiB of zeros: per-lane 16-byte zero source

// Diagnostic build (-DCNW_STAMP): s_memtime stamps of wave 0 of block 0 over its first chunks (tools/wgrad_stamps.py).
#ifdef CNW_STAMP
__device__ unsigned long long cnw_stamps[256];
#define CNW_ST() do { if (do_stamp && stamp_i < 256) cnw_stamps[stamp_i++] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int cn_wgrad_read_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(cnw_stamps), sizeof(unsigned long long) * 256) == hipSuccess ? 0 : -2;
}
#else
#define CNW_ST() do { } while (0)
#endif

__device__ __forceinline__ void cn_glds16(const float* src, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((cn_gbl_ptr)src, (cn_lds_ptr)lds_wave_base, 16, 0, 0);
}

// 16-byte LDS-DMA variant ("flattened rows"): needs 16-byte aligned channel planes (Hs*Ws % 4 == 0,
// Hb*Wb % 4 == 0, strides % 4 == 0) and (PR*Ws) % 4 == 0. The S rows of a chunk are one contiguous flat range;
// the Bg halo rows too (unpadded, pitch = Wb), so row overruns are zero-filled per 16-byte piece by a flat
// bound check and column overruns (which wrap into the neighbouring row) are masked per lane at operand read.
// One DMA wave-instruction moves 1 KiB instead of 256 B: the per-CU DMA issue rate (~1 per 100 cycles)
// is what bounded the dword version.
template <int T, int S_>
__global__ __launch_bounds__(256) void cn_wgrad_vec_kernel(const float* __restrict__ S0, const float* __restrict__ Bg0,
                                                          float* __restrict__ dW0, const CnWgradGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int at = wid % g.a_tiles, kp = wid / g.a_tiles, kparts = 4 / g.a_tiles;
  int bx, by, bz;
  if (!cn_xcd_block(g.grid_x, g.grid_y, g.grid_x * g.grid_y * g.grid_z, bx, by, bz)) return;
  const int grp = g.G > 1 ? bz / g.spg : 0;
  const int split = bz - grp * g.spg;
  const float* __restrict__ S = g.G > 1 ? g.gS[grp] : S0;
  const float* __restrict__ Bg = g.G > 1 ? g.gB[grp] : Bg0;
  float* __restrict__ dW = (g.G > 1 && g.slice_stride == 0) ? g.gdW[grp] : dW0;
  const int a0 = by * g.a_tiles * 32;
  const int b0 = bx * WG_BC;
  const int npix = g.PR * g.Ws;   // unpadded, even, multiple of 4
  const int n4s = npix >> 2;      // float4 pieces per S row (<= 64)
  const int n4b = g.plane_b >> 2; // float4 pieces per Bg channel image
  const float* zero = cn_zero_line16 + 4 * lane;
  const int HWs = g.Hs * g.Ws, HWb = g.Hb * g.Wb;

  int boff[T], ox[T];
#pragma unroll
  for (int j = 0; j < T; ++j) {
    const int n = j * 32 + l31;
    const int bl = n / T, t = n - bl * T;
    boff[j] = bl * g.plane_b + (g.offy[t] - g.min_oy) * g.Wb + (g.offx[t] - g.min_ox) + half * g.s;
    ox[j] = g.offx[t] + half * g.s;
  }
  const int aoff = (at * 32 + l31) * g.pitch_s + half;

  f32x16 acc[T];
#pragma unroll
  for (int j = 0; j < T; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int arows = g.a_tiles * 32;
  int chunk = split * g.chunks_per_split;
  int chunk_end = chunk + g.chunks_per_split;
  if (chunk_end > g.total_chunks) chunk_end = g.total_chunks;

  const int steps = npix >> 1;
  const int q_begin = (steps * kp) / kparts, q_end = (steps * (kp + 1)) / kparts;
  const int e_begin = q_begin * 2;
  const int r_begin = e_begin / g.Ws, c_begin = e_begin - r_begin * g.Ws;

  // dense LDS images: S [arows][npix], Bg [WG_BC][plane_b]; piece p of an image = its p-th 16 bytes, so one
  // DMA wave-instruction (64 consecutive pieces = 1 KiB) may span several rows / channels.
  // Per-lane BYTE offset of this lane's piece in DMA instruction k, relative to the chunk's base address (S: first
  // pixel of the chunk in channel 0 of image n; Bg: the 16-byte aligned start of the halo in channel 0), or WG_IDLE
  // for lanes past the image / past the channel count (their LDS rows only feed dW entries that are never stored).
  // (Measured and not kept: issuing an interior chunk's DMA instructions singly between the MFMA groups instead of
  // back to back -- each then stalls the wave ~330 instead of ~126 cycles: 252 -> 311 us at 128->128, 100^2.)
  // Interior chunks -- all but the first / last rows of an image -- then stage with ONE instruction per KiB: scalar
  // base + per-lane offset, no per-chunk address arithmetic or bound checks (those took ~3000 cycles per chunk with
  // the matrix pipe idle: one wave per SIMD). Boundary chunks re-derive the piece index and zero-fill per piece.
  constexpr unsigned WG_IDLE = 0xFFFFFFFFu;
  unsigned so[WG_KS], bo[WG_KB];
  {
    const int nsp = arows * n4s, nbp = WG_BC * n4b;
#pragma unroll
    for (int k = 0; k < WG_KS; ++k) {
      const int p = (wid + 4 * k) * 64 + lane;
      const int a = p / n4s, pc = p - a * n4s;
      so[k] = (p < nsp && (a0 + a) < g.A) ? (unsigned)(((long)(a0 + a) * g.scs + 4 * pc) * 4) : WG_IDLE;
    }
#pragma unroll
    for (int k = 0; k < WG_KB; ++k) {
      const int p = (wid + 4 * k) * 64 + lane;
      const int bl = p / n4b, pi = p - bl * n4b;
      bo[k] = (p < nbp && (b0 + bl) < g.Bc) ? (unsigned)(((long)(b0 + bl) * g.bcs + 4 * pi) * 4) : WG_IDLE;
    }
  }
  auto stage = [&](int ck, int buf) {
    float* s_lds = smem + buf * g.buf_stride;
    float* b_lds = s_lds + g.b_lds_off;
    const int n = ck / g.chunks_per_img;
    const int gy0 = (ck - n * g.chunks_per_img) * g.PR;
    const int nsp = arows * n4s, nbp = WG_BC * n4b;
    {
      const int fs0 = gy0 * g.Ws;
      const char* sbase = reinterpret_cast<const char*>(S + (long)n * g.sbs + fs0);
      if (fs0 + npix <= (int)g.scs) {  // wave-uniform: the whole chunk lies inside the plane
#pragma unroll
        for (int k = 0; k < WG_KS; ++k)
          if ((wid + 4 * k) * 64 < nsp && so[k] != WG_IDLE)
            cn_glds16(reinterpret_cast<const float*>(sbase + so[k]), s_lds + (wid + 4 * k) * 256);
      } else {
#pragma unroll
        for (int k = 0; k < WG_KS; ++k) {
          if ((wid + 4 * k) * 64 < nsp && so[k] != WG_IDLE) {
            const int p = (wid + 4 * k) * 64 + lane;
            const int pc = p % n4s;
            const bool ok = (fs0 + 4 * pc + 3) < (int)g.scs;
            const float* src = ok ? reinterpret_cast<const float*>(sbase + so[k]) : zero;
            cn_glds16(src, s_lds + (wid + 4 * k) * 256);
          }
        }
      }
    }
    {
      const int start = (gy0 * g.s + g.min_oy) * g.Wb + g.min_ox;
      const int f0 = (start >> 2) << 2;
      const char* bbase = reinterpret_cast<const char*>(Bg + (long)n * g.bbs + f0);
      if (f0 >= 0 && f0 + g.plane_b <= (int)g.bcs) {  // wave-uniform: the halo lies inside the plane
#pragma unroll
        for (int k = 0; k < WG_KB; ++k)
          if ((wid + 4 * k) * 64 < nbp && bo[k] != WG_IDLE)
            cn_glds16(reinterpret_cast<const float*>(bbase + bo[k]), b_lds + (wid + 4 * k) * 256);
      } else {
#pragma unroll
        for (int k = 0; k < WG_KB; ++k) {
          if ((wid + 4 * k) * 64 < nbp && bo[k] != WG_IDLE) {
            const int p = (wid + 4 * k) * 64 + lane;
            const int fq = f0 + 4 * (p % n4b);
            const bool ok = fq >= 0 && (fq + 3) < (int)g.bcs;
            const float* src = ok ? reinterpret_cast<const float*>(bbase + bo[k]) : zero;
            cn_glds16(src, b_lds + (wid + 4 * k) * 256);
          }
        }
      }
    }
  };

  int cur = 0;
#ifdef CNW_STAMP
  const bool do_stamp = T == 9 && S_ == 1 && blockIdx.x == 8 && tid == 0;
  int stamp_i = 0;
#endif
  CNW_ST();  // 0: prologue done
  if (chunk < chunk_end && g.nbuf == 2) stage(chunk, 0);
  for (; chunk < chunk_end; ++chunk) {
    CNW_ST();  // chunk top
    if (g.nbuf == 1) {
      __syncthreads();
      stage(chunk, 0);
    }
    __syncthreads();
    CNW_ST();  // after barrier
    if (g.nbuf == 2 && chunk + 1 < chunk_end) stage(chunk + 1, cur ^ 1);
    CNW_ST();  // after DMA issue
    const float* s_lds = smem + cur * g.buf_stride;
    const float* b_lds = s_lds + g.b_lds_off;
    const int gy0c = (chunk % g.chunks_per_img) * g.PR;
    const int start = (gy0c * g.s + g.min_oy) * g.Wb + g.min_ox;
    const int sh = start - ((start >> 2) << 2);
    // One wave per SIMD issues everything in order, so instructions-per-MFMA bounds this loop: the pixel pairs
    // of a row are walked in unrolled groups whose LDS offsets are compile-time immediates (stride S_), bases are
    // bumped once per group, and the column masks are applied only on the first / last pair of an image row.
    {
      // the pixel pairs of every row are split over the k-part waves
      const int nq_row = g.Ws >> 1;
      const int q_lo = (nq_row * kp) / kparts, q_hi = (nq_row * (kp + 1)) / kparts;
      // pairs [0, mq_lo) and [mq_hi, nq_row) of a row can reach outside [0, Wb) for some tap -> masked steps
      const int seg0 = min(max(g.mq_lo, q_lo), q_hi), seg1 = max(min(g.mq_hi, q_hi), seg0);
      for (int r = 0; r < g.PR; ++r) {
        const float* ap = s_lds + aoff + r * g.Ws;          // indexed by absolute column
        const float* bp[T];
#pragma unroll
        for (int j = 0; j < T; ++j) bp[j] = b_lds + boff[j] + sh + (r * S_) * g.Wb;
        int q = q_lo;
        // masked and left-over single pairs: all T + 1 operands are read before the first MFMA (read / wait / MFMA per
        // tap exposed one LDS round trip in front of every MFMA of the pair)
        for (; q < seg0; ++q) {  // leading masked pairs
          const float av = ap[2 * q];
          const int cs = 2 * q * S_;
          float bv[T];
#pragma unroll
          for (int j = 0; j < T; ++j) bv[j] = bp[j][cs];
#pragma unroll
          for (int j = 0; j < T; ++j) {
            const float b_ = ((unsigned)(cs + ox[j]) < (unsigned)g.Wb) ? bv[j] : 0.f;
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b_, acc[j], 0, 0, 0);
          }
        }
        // Interior pairs (no masks, immediate offsets), in groups of two k-steps with the operands double-buffered
        // in registers: the LDS reads of the NEXT group are issued before the 2*T MFMAs of the current one. One wave
        // per SIMD issues in order, so without this every group waited out the full LDS latency with the MFMA pipe
        // idle (the compiler emitted read-all / s_waitcnt 0 / MFMA-all). (Folding the edge pairs into this pipeline
        // as well was measured slower: 252 -> 273 us at 128->128, 100^2 -- the extra live operands cost more moves
        // than the two exposed LDS round trips per row.)
        CNW_ST();  // after leading masked pairs
        if constexpr (T == 1) {
          // one MFMA per k-step: the register pipeline below was measured slower (172 vs 161 us at 480->128, 100^2)
          for (; q + WG_U <= seg1; q += WG_U) {
            const float* apq = ap + 2 * q;
#pragma unroll
            for (int u = 0; u < WG_U; ++u)
              acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(apq[2 * u], (bp[0] + 2 * q * S_)[2 * u * S_], acc[0], 0, 0, 0);
          }
        } else {
          constexpr int GS = 2;  // k-steps per group
          const int ng = (seg1 - q) / GS;
          if (ng > 0) {
            float a0[GS], a1[GS], b0[GS][T], b1[GS][T];
#define WG_LOADG(qq_, a_, b_)                                                    \
  {                                                                              \
    const float* apq_ = ap + 2 * (qq_);                                          \
    _Pragma("unroll") for (int u = 0; u < GS; ++u) a_[u] = apq_[2 * u];          \
    _Pragma("unroll") for (int j = 0; j < T; ++j) {                              \
      const float* bq_ = bp[j] + 2 * (qq_) * S_;                                 \
      _Pragma("unroll") for (int u = 0; u < GS; ++u) b_[u][j] = bq_[2 * u * S_]; \
    }                                                                            \
  }
#define WG_MFMAG(a_, b_)                                                                                   \
  {                                                                                                        \
    _Pragma("unroll") for (int u = 0; u < GS; ++u)                                                         \
        _Pragma("unroll") for (int j = 0; j < T; ++j)                                                      \
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[u], b_[u][j], acc[j], 0, 0, 0);               \
  }
            WG_LOADG(q, a0, b0);
            int i = 0;
            for (; i + 2 <= ng; i += 2) {
              WG_LOADG(q + GS * (i + 1), a1, b1);
              WG_MFMAG(a0, b0);
              if (i + 2 < ng) WG_LOADG(q + GS * (i + 2), a0, b0);
              WG_MFMAG(a1, b1);
            }
            if (i < ng) WG_MFMAG(a0, b0);
#undef WG_LOADG
#undef WG_MFMAG
            q += GS * ng;
          }
        }
        CNW_ST();  // after pipelined groups
        for (; q < seg1; ++q) {
          const float av = ap[2 * q];
          float bv[T];
#pragma unroll
          for (int j = 0; j < T; ++j) bv[j] = bp[j][2 * q * S_];
#pragma unroll
          for (int j = 0; j < T; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[j], acc[j], 0, 0, 0);
        }
        for (; q < q_hi; ++q) {  // trailing masked pairs
          const float av = ap[2 * q];
          const int cs = 2 * q * S_;
          float bv[T];
#pragma unroll
          for (int j = 0; j < T; ++j) bv[j] = bp[j][cs];
#pragma unroll
          for (int j = 0; j < T; ++j) {
            const float b_ = ((unsigned)(cs + ox[j]) < (unsigned)g.Wb) ? bv[j] : 0.f;
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b_, acc[j], 0, 0, 0);
          }
        }
      }
    }
    CNW_ST();  // chunk end
    if (g.nbuf == 2) cur ^= 1;
  }
  CNW_ST();

  // the k-part waves of a block hold partial sums of the SAME dW tile: add them up through LDS (free after the
  // main loop) so the block writes its tile once
  if (kparts > 1) {
    __syncthreads();
    float* red = smem;
    if (kp > 0) {
      float* rp = red + (long)((kp - 1) * g.a_tiles + at) * (T * 16 * 64) + lane;
#pragma unroll
      for (int j = 0; j < T; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) rp[(j * 16 + r) * 64] = acc[j][r];
    }
    __syncthreads();
    if (kp > 0) return;
    for (int k = 1; k < kparts; ++k) {
      const float* rp = red + (long)((k - 1) * g.a_tiles + at) * (T * 16 * 64) + lane;
#pragma unroll
      for (int j = 0; j < T; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] += rp[(j * 16 + r) * 64];
    }
  }
  // many splits on a small dW: same-address float atomics serialise (~200 ns each), so each split stores its
  // partial into a private workspace slice instead and cn_wgrad_reduce_kernel sums the slices
  float* dWs = dW + (long)bz * g.slice_stride;
#pragma unroll
  for (int j = 0; j < T; ++j) {
    const int n = j * 32 + l31;
    const int bl = n / T;
    const bool bok = (b0 + bl) < g.Bc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int a = a0 + at * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (bok && a < g.A) {
        if (g.slice_stride != 0) dWs[(long)a * g.sa + (long)b0 * T + n] = acc[j][r];
        else atomicAdd(dW + (long)a * g.sa + (long)b0 * T + n, acc[j][r]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 3 x 3 weight gradient, TWELVE waves per block (round 5): the same tile, LDS images and LDS-DMA staging as
// cn_wgrad_vec_kernel<9, S_>, but a wave owns ONE KERNEL ROW of the taps -- 32 couts x (32 channels x 3 taps) = three
// accumulator tiles, 48 registers instead of 144 -- so three waves share a SIMD (wave = cout tile x kernel row x k-part).
// cn_wgrad_vec_kernel<9, *> runs one wave per SIMD (393 registers) and issues everything in order: while it issues a
// chunk's ~17 LDS-DMA instructions (~126 cycles each at the CU's issue rate), waits at the chunk barrier or for the first
// operands, the matrix pipe idles -- 14.4k of a chunk's ~20k cycles are MFMAs (0.58 of the f32 peak alone, four rounds
// running). Here a wave's DMA issue, barrier skew and LDS latency run under the other two waves' MFMAs.
// Columns of a wave's tile: n = j * 32 + lane % 32 (j < 3), channel n / 3, tap ky * 3 + n % 3 -- dW runs of three floats.
#define WG3_WAVES 12
#define WG3_KS 3   // DMA instructions per wave for the S image  (arows * npix <= 8192 floats = 32 KiB = 32 instructions)
#define WG3_KB 6   // DMA instructions per wave for the Bg image (32 * plane_b <= 16384 floats = 64 instructions)
template <int S_>
__global__ __launch_bounds__(WG3_WAVES * 64) void cn_wgrad_vec3_kernel(const float* __restrict__ S0,
                                                                       const float* __restrict__ Bg0,
                                                                       float* __restrict__ dW0, const CnWgradGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int T = 3;  // taps per wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  // wave = (k-part, kernel row, cout tile): the waves of one (cout tile, kernel row) are a_tiles * 3 apart
  const int at = wid % g.a_tiles;
  const int ky = (wid / g.a_tiles) % 3;
  const int kp = wid / (g.a_tiles * 3), kparts = WG3_WAVES / (g.a_tiles * 3);
  int bx, by, bz;
  if (!cn_xcd_block(g.grid_x, g.grid_y, g.grid_x * g.grid_y * g.grid_z, bx, by, bz)) return;
  const int grp = g.G > 1 ? bz / g.spg : 0;
  const int split = bz - grp * g.spg;
  const float* __restrict__ S = g.G > 1 ? g.gS[grp] : S0;
  const float* __restrict__ Bg = g.G > 1 ? g.gB[grp] : Bg0;
  float* __restrict__ dW = (g.G > 1 && g.slice_stride == 0) ? g.gdW[grp] : dW0;
  const int a0 = by * g.a_tiles * 32;
  const int b0 = bx * WG_BC;
  const int npix = g.PR * g.Ws;
  const int n4s = npix >> 2;
  const int n4b = g.plane_b >> 2;
  const float* zero = cn_zero_line16 + 4 * lane;

  int boff[T], ox[T];
#pragma unroll
  for (int j = 0; j < T; ++j) {
    const int n = j * 32 + l31;
    const int bl = n / 3, t = ky * 3 + (n - bl * 3);
    boff[j] = bl * g.plane_b + (g.offy[t] - g.min_oy) * g.Wb + (g.offx[t] - g.min_ox) + half * g.s;
    ox[j] = g.offx[t] + half * g.s;
  }
  const int aoff = (at * 32 + l31) * g.pitch_s + half;

  f32x16 acc[T];
#pragma unroll
  for (int j = 0; j < T; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int arows = g.a_tiles * 32;
  int chunk = split * g.chunks_per_split;
  int chunk_end = chunk + g.chunks_per_split;
  if (chunk_end > g.total_chunks) chunk_end = g.total_chunks;

  constexpr unsigned WG_IDLE = 0xFFFFFFFFu;
  unsigned so[WG3_KS], bo[WG3_KB];
  {
    const int nsp = arows * n4s, nbp = WG_BC * n4b;
#pragma unroll
    for (int k = 0; k < WG3_KS; ++k) {
      const int p = (wid + WG3_WAVES * k) * 64 + lane;
      const int a = p / n4s, pc = p - a * n4s;
      so[k] = (p < nsp && (a0 + a) < g.A) ? (unsigned)(((long)(a0 + a) * g.scs + 4 * pc) * 4) : WG_IDLE;
    }
#pragma unroll
    for (int k = 0; k < WG3_KB; ++k) {
      const int p = (wid + WG3_WAVES * k) * 64 + lane;
      const int bl = p / n4b, pi = p - bl * n4b;
      bo[k] = (p < nbp && (b0 + bl) < g.Bc) ? (unsigned)(((long)(b0 + bl) * g.bcs + 4 * pi) * 4) : WG_IDLE;
    }
  }
  auto stage = [&](int ck, int buf) {
    float* s_lds = smem + buf * g.buf_stride;
    float* b_lds = s_lds + g.b_lds_off;
    const int n = ck / g.chunks_per_img;
    const int gy0 = (ck - n * g.chunks_per_img) * g.PR;
    const int nsp = arows * n4s, nbp = WG_BC * n4b;
    {
      const int fs0 = gy0 * g.Ws;
      const char* sbase = reinterpret_cast<const char*>(S + (long)n * g.sbs + fs0);
      const bool inside = fs0 + npix <= (int)g.scs;  // wave-uniform: the whole chunk lies inside the plane
#pragma unroll
      for (int k = 0; k < WG3_KS; ++k) {
        if ((wid + WG3_WAVES * k) * 64 < nsp && so[k] != WG_IDLE) {
          const float* src = reinterpret_cast<const float*>(sbase + so[k]);
          if (!inside) {
            const int p = (wid + WG3_WAVES * k) * 64 + lane;
            if (!((fs0 + 4 * (p % n4s) + 3) < (int)g.scs)) src = zero;
          }
          cn_glds16(src, s_lds + (wid + WG3_WAVES * k) * 256);
        }
      }
    }
    {
      const int start = (gy0 * g.s + g.min_oy) * g.Wb + g.min_ox;
      const int f0 = (start >> 2) << 2;
      const char* bbase = reinterpret_cast<const char*>(Bg + (long)n * g.bbs + f0);
      const bool inside = f0 >= 0 && f0 + g.plane_b <= (int)g.bcs;  // wave-uniform: the halo lies inside the plane
#pragma unroll
      for (int k = 0; k < WG3_KB; ++k) {
        if ((wid + WG3_WAVES * k) * 64 < nbp && bo[k] != WG_IDLE) {
          const float* src = reinterpret_cast<const float*>(bbase + bo[k]);
          if (!inside) {
            const int p = (wid + WG3_WAVES * k) * 64 + lane;
            const int fq = f0 + 4 * (p % n4b);
            if (!(fq >= 0 && (fq + 3) < (int)g.bcs)) src = zero;
          }
          cn_glds16(src, b_lds + (wid + WG3_WAVES * k) * 256);
        }
      }
    }
  };

  int cur = 0;
  if (chunk < chunk_end && g.nbuf == 2) stage(chunk, 0);
  for (; chunk < chunk_end; ++chunk) {
    if (g.nbuf == 1) {
      __syncthreads();
      stage(chunk, 0);
    }
    __syncthreads();
    // (The barrier puts the three waves of a SIMD in the same phase. Skewing them -- the waves of kernel row ky start
    // ky * 512 / 1024 / 2048 / 4096 cycles late -- was measured: 239 -> 245 / 245 / 249 / 260 us at 128 -> 128, 8 x 100^2;
    // moving a wave's DMA issue into its group pipeline costs registers the three-waves budget does not have.)
    // (Round 5, measured with tools/wgrad_bench.py at 8 x 100^2, same box: (a) the next chunk's nine DMA instructions
    // spread through this chunk's MFMA loop, three per iteration, 168 registers without scratch once the source pointers
    // were kept from being hoisted: 128 -> 128 240 -> 280 us, 160 -> 128 270-292 -> 346 us -- a DMA issue inside the
    // loop stalls the wave's in-order MFMA stream; (b) progress-based issue priority, s_setprio 3..0 by quarter of the
    // row so that the wave that is behind is served first: the three waves of a SIMD then finish their row together and
    // the chunk takes exactly as long: 243.0 -> 242.2 us, 160 -> 128 268.5 -> 297.4 us. Neither kept. Under this kernel the
    // shader clock reads ~1.97 GHz (s_memtime against the event time of the launch), not 2.4: its 0.62 of the nominal
    // f32 peak is ~0.75 of what the matrix pipes can deliver at that clock.)
    if (g.nbuf == 2 && chunk + 1 < chunk_end) stage(chunk + 1, cur ^ 1);
    const float* s_lds = smem + cur * g.buf_stride;
    const float* b_lds = s_lds + g.b_lds_off;
    const int gy0c = (chunk % g.chunks_per_img) * g.PR;
    const int start = (gy0c * g.s + g.min_oy) * g.Wb + g.min_ox;
    const int sh = start - ((start >> 2) << 2);
    const int nq_row = g.Ws >> 1;
    const int q_lo = (nq_row * kp) / kparts, q_hi = (nq_row * (kp + 1)) / kparts;
    const int seg0 = min(max(g.mq_lo, q_lo), q_hi), seg1 = max(min(g.mq_hi, q_hi), seg0);
    for (int r = 0; r < g.PR; ++r) {
      const float* ap = s_lds + aoff + r * g.Ws;
      const float* bp[T];
#pragma unroll
      for (int j = 0; j < T; ++j) bp[j] = b_lds + boff[j] + sh + (r * S_) * g.Wb;
      int q = q_lo;
      for (; q < seg0; ++q) {  // leading masked pairs
        const float av = ap[2 * q];
        const int cs = 2 * q * S_;
        float bv[T];
#pragma unroll
        for (int j = 0; j < T; ++j) bv[j] = bp[j][cs];
#pragma unroll
        for (int j = 0; j < T; ++j) {
          const float b_ = ((unsigned)(cs + ox[j]) < (unsigned)g.Wb) ? bv[j] : 0.f;
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b_, acc[j], 0, 0, 0);
        }
      }
      {  // interior pairs: groups of GS k-steps with the operands double-buffered in registers.
         // (A conflict-free A operand -- one ds_read_b128 per lane for four k-steps + two v_permlane32_swap, instead of
         // four dwords at a row pitch that puts 32 couts on 8 banks -- was built and measured: 239 -> 251 us at 128 -> 128,
         // 855 -> 899 at 480 -> 128, step 388 -> 383 chips/s: the bank conflicts are not what idles the pipe. Note for the
         // next attempt: __builtin_bit_cast(unsigned, v[i]) of an ext_vector ELEMENT expression yields element 0 / undef
         // with this compiler; copy the element to a scalar first.)
        constexpr int GS = 3;
        const int ng = (seg1 - q) / GS;
        if (ng > 0) {
          float a0[GS], a1[GS], b0[GS][T], b1[GS][T];
#define WG3_LOADG(qq_, a_, b_)                                                   \
  {                                                                              \
    const float* apq_ = ap + 2 * (qq_);                                          \
    _Pragma("unroll") for (int u = 0; u < GS; ++u) a_[u] = apq_[2 * u];          \
    _Pragma("unroll") for (int j = 0; j < T; ++j) {                              \
      const float* bq_ = bp[j] + 2 * (qq_) * S_;                                 \
      _Pragma("unroll") for (int u = 0; u < GS; ++u) b_[u][j] = bq_[2 * u * S_]; \
    }                                                                            \
  }
#define WG3_MFMAG(a_, b_)                                                                                  \
  {                                                                                                        \
    _Pragma("unroll") for (int u = 0; u < GS; ++u)                                                         \
        _Pragma("unroll") for (int j = 0; j < T; ++j)                                                      \
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[u], b_[u][j], acc[j], 0, 0, 0);               \
  }
          WG3_LOADG(q, a0, b0);
          int i = 0;
          for (; i + 2 <= ng; i += 2) {
            WG3_LOADG(q + GS * (i + 1), a1, b1);
            WG3_MFMAG(a0, b0);
            if (i + 2 < ng) WG3_LOADG(q + GS * (i + 2), a0, b0);
            WG3_MFMAG(a1, b1);
          }
          if (i < ng) WG3_MFMAG(a0, b0);
#undef WG3_LOADG
#undef WG3_MFMAG
          q += GS * ng;
        }
      }
      for (; q < seg1; ++q) {
        const float av = ap[2 * q];
        float bv[T];
#pragma unroll
        for (int j = 0; j < T; ++j) bv[j] = bp[j][2 * q * S_];
#pragma unroll
        for (int j = 0; j < T; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[j], acc[j], 0, 0, 0);
      }
      for (; q < q_hi; ++q) {  // trailing masked pairs
        const float av = ap[2 * q];
        const int cs = 2 * q * S_;
        float bv[T];
#pragma unroll
        for (int j = 0; j < T; ++j) bv[j] = bp[j][cs];
#pragma unroll
        for (int j = 0; j < T; ++j) {
          const float b_ = ((unsigned)(cs + ox[j]) < (unsigned)g.Wb) ? bv[j] : 0.f;
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b_, acc[j], 0, 0, 0);
        }
      }
    }
    if (g.nbuf == 2) cur ^= 1;
  }

  // the k-part waves of a (cout tile, kernel row) hold partial sums of the SAME dW tile: add them up through LDS
  if (kparts > 1) {
    __syncthreads();
    float* red = smem;
    const int slot = ky * g.a_tiles + at;  // (cout tile, kernel row) within the block
    if (kp > 0) {
      float* rp = red + (long)((kp - 1) * g.a_tiles * 3 + slot) * (T * 16 * 64) + lane;
#pragma unroll
      for (int j = 0; j < T; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) rp[(j * 16 + r) * 64] = acc[j][r];
    }
    __syncthreads();
    if (kp > 0) return;
    for (int k = 1; k < kparts; ++k) {
      const float* rp = red + (long)((k - 1) * g.a_tiles * 3 + slot) * (T * 16 * 64) + lane;
#pragma unroll
      for (int j = 0; j < T; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] += rp[(j * 16 + r) * 64];
    }
  }
  float* dWs = dW + (long)bz * g.slice_stride;
#pragma unroll
  for (int j = 0; j < T; ++j) {
    const int n = j * 32 + l31;
    const int bl = n / 3, t = ky * 3 + (n - bl * 3);
    const bool bok = (b0 + bl) < g.Bc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int a = a0 + at * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (bok && a < g.A) {
        float* d = (g.slice_stride != 0 ? dWs : dW) + (long)a * g.sa + (long)(b0 + bl) * 9 + t;
        if (g.slice_stride != 0) *d = acc[j][r];
        else atomicAdd(d, acc[j][r]);
      }
    }
  }
}

// dW_g[i] += sum_s part[g * nslices + s][i]; grid = (ceil(n / 256), 1 or 16, groups): with 16 slice groups,
// 16-way atomics per address
struct CnWgradReduceArgs {
  const float* part; long slice_stride; int nslices; long n;
  float* dW[4];
};
__global__ __launch_bounds__(256) void cn_wgrad_reduce_kernel(const CnWgradReduceArgs a) {
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= a.n) return;
  const float* part = a.part + (long)blockIdx.z * a.nslices * a.slice_stride;
  float* dW = a.dW[blockIdx.z];
  float s0 = 0.f, s1 = 0.f;
  int k = blockIdx.y;
  for (; k + (int)gridDim.y < a.nslices; k += 2 * gridDim.y) {
    s0 += part[(long)k * a.slice_stride + i];
    s1 += part[(long)(k + gridDim.y) * a.slice_stride + i];
  }
  if (k < a.nslices) s0 += part[(long)k * a.slice_stride + i];
  if (gridDim.y == 1) dW[i] += s0 + s1;
  else atomicAdd(dW + i, s0 + s1);
}

template <int T>
static int cn_wgrad_launch_vec(const float* S, const float* Bg, float* dW, CnWgradGeom g, float* ws, long ws_floats,
                               hipStream_t stream) {
  if (g.scs % 4 || g.bcs % 4 || g.sbs % 4 || g.bbs % 4 || (g.Ws & 1)) return CN_ERR_ARG;
  g.Wsp = g.Ws;
  g.PR = 128 / g.Ws;
  if (g.PR < 1) g.PR = 1;
  if (g.PR > g.Hs) g.PR = g.Hs;
  // a chunk is PR whole rows = PR*Ws floats and must be whole 16-byte pieces; rows past the image in the last
  // chunk fall outside the (zero-tailed) plane and are zero-filled by the per-piece bound check
  while (g.PR > 1 && (g.PR * g.Ws) % 4 != 0) --g.PR;
  if ((g.PR * g.Ws) % 4 != 0 && g.Ws % 2 == 0 && 2 * g.Ws <= 256) g.PR = 2;
  if ((g.PR * g.Ws) % 4 != 0 || g.PR * g.Ws > 256) return CN_ERR_ARG;
  int max_oy = g.min_oy, max_ox = g.min_ox;
  for (int t = 0; t < g.T; ++t) {
    if (g.offy[t] > max_oy) max_oy = g.offy[t];
    if (g.offx[t] > max_ox) max_ox = g.offx[t];
  }
  g.rows_b = (g.PR - 1) * g.s + (max_oy - g.min_oy) + 1;
  g.pitch_b = g.Wb;
  // last read: (rows_b-1)*Wb + (Ws-1)*s + (max_ox-min_ox) + sh(<=3)
  g.plane_b = ((g.rows_b - 1) * g.Wb + (g.Ws - 1) * g.s + (max_ox - g.min_ox) + 4 + 3) / 4 * 4;
  if (WG_BC * g.plane_b > WG_KB * 4 * 256) return CN_ERR_ARG;
  g.pitch_s = g.PR * g.Ws;  // dense image (DMA pieces run across rows)
  g.a_tiles = g.A > 32 ? 2 : 1;
  size_t lds;
  for (;;) {
    g.b_lds_off = (g.a_tiles * 32 * g.pitch_s + 255) / 256 * 256;  // whole DMA instructions
    g.buf_stride = g.b_lds_off + (WG_BC * g.plane_b + 255) / 256 * 256;
    lds = (size_t)g.buf_stride * sizeof(float);
    if ((lds <= 160 * 1024 && g.a_tiles * 32 * g.pitch_s <= WG_KS * 4 * 256) || g.a_tiles == 1) break;
    g.a_tiles >>= 1;
  }
  if (lds > 160 * 1024 || g.a_tiles * 32 * g.pitch_s > WG_KS * 4 * 256) return CN_ERR_ARG;
  g.nbuf = (2 * lds <= 160 * 1024) ? 2 : 1;
  lds *= g.nbuf;
  {  // room for the in-block reduction over the k-part waves: [kparts-1][a_tiles][T*16][64] floats
    const size_t red = (size_t)(4 / g.a_tiles - 1) * g.a_tiles * T * 16 * 64 * sizeof(float);
    if (red > lds) lds = red;
  }
  g.chunks_per_img = (g.Hs + g.PR - 1) / g.PR;
  g.total_chunks = g.N * g.chunks_per_img;
  const int gx = (g.Bc + WG_BC - 1) / WG_BC, gy = (g.A + g.a_tiles * 32 - 1) / (g.a_tiles * 32);
  if (g.G < 1) {  // single set given through the arguments
    g.G = 1;
    g.gS[0] = S; g.gB[0] = Bg; g.gdW[0] = dW;
  }
  for (int i = 0; i < g.G; ++i)
    if ((reinterpret_cast<uintptr_t>(g.gS[i]) & 15) || (reinterpret_cast<uintptr_t>(g.gB[i]) & 15)) return CN_ERR_ARG;
  // resident blocks on the chip. (Sizing the grids for 224 / 192 / 160 of the 256 CUs, to leave whole CUs to the compute
  // stream's kernels -- which cannot share a SIMD with a weight-gradient wave: 393 of 512 registers per lane -- was
  // measured in round 5: fp32 386.4 -> 382.2 / 373.2 / 354.9 chips/s. The side stream is as long as the compute stream.)
  // (More, shorter-lived blocks -- 512 / 1024 / 2048 per launch, so that the compute stream's workgroups get CUs at every
  // block boundary instead of once per launch -- cut the slowdown of a bilinear adjoint next to this kernel from 9.5x to
  // 5.5x / 3.6x / 2.8x (tools/corun.py) and made the fp32 step SLOWER: 387.0 -> 373.5 / 357.6 / 342.7 chips/s, as did
  // single-buffered staging (half the LDS: 374.5). The slices and per-block epilogues cost more machine time than the
  // compute stream wins back: the step is bound by the sum of the work, not by who waits for whom.)
  const int slots = (lds * 2 <= 160 * 1024) ? 512 : 256;
  int splits = slots / (gx * gy * g.G);                   // never spill into a second, mostly idle round
  if (splits > g.total_chunks) splits = g.total_chunks;
  if (splits < 1) splits = 1;
  g.chunks_per_split = (g.total_chunks + splits - 1) / splits;
  splits = (g.total_chunks + g.chunks_per_split - 1) / g.chunks_per_split;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)cn_wgrad_vec_kernel<T, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    (void)hipFuncSetAttribute((const void*)cn_wgrad_vec_kernel<T, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr_set = true;
  }
  if (g.s != 1 && g.s != 2) return CN_ERR_ARG;
  {
    // pair q covers columns 2q, 2q+1: masked iff 2q*s + min_ox < 0 or (2q+1)*s + max_ox >= Wb
    const int nq_row = g.Ws / 2;
    int lo = (-g.min_ox + 2 * g.s - 1) / (2 * g.s);
    if (lo < 0) lo = 0;
    int hi_num = g.Wb - max_ox - g.s;  // first masked pair = ceil(hi_num / (2s))
    int hi = hi_num <= 0 ? 0 : (hi_num + 2 * g.s - 1) / (2 * g.s);
    if (lo > nq_row) lo = nq_row;
    if (hi > nq_row) hi = nq_row;
    if (hi < lo) hi = lo;
    g.mq_lo = lo;
    g.mq_hi = hi;
  }
  cn_prof_name("cn_wgrad_vec_kernel<%d, %d>", T, g.s == 1 ? 1 : 2);
  cn_prof_desc("wgrad_vec<%d> G%d N%d A%d %dx%d Bc%d %dx%d s%d grid%dx%dx%d nbuf%d", T, g.G, g.N, g.A, g.Hs, g.Ws, g.Bc,
               g.Hb, g.Wb, g.s, gx, gy, splits, g.nbuf);
  g.spg = splits;
  g.grid_x = gx; g.grid_y = gy; g.grid_z = splits * g.G;
  const dim3 grid(cn_xcd_grid((long)gx * gy * splits * g.G));
  // partial-slice mode: >= 32 adds per dW address (slices must fit the caller's workspace)
  const int kparts = 4 / g.a_tiles;
  const long dw_floats = (long)g.A * g.sa;
  const long nslices = splits;  // per group (the k-part waves of a block are summed in LDS)
  float* out = g.gdW[0];
  g.slice_stride = 0;
  if (nslices >= 16 && ws != nullptr && nslices * g.G * dw_floats <= ws_floats) {
    g.slice_stride = dw_floats;
    out = ws;
  }
  static const bool twelve = getenv("CN_WGRAD3") == nullptr || atoi(getenv("CN_WGRAD3")) != 0;  // A/B switch
  if (T == 9 && twelve) cn_prof_name("cn_wgrad_vec3_kernel<%d>", g.s == 1 ? 1 : 2);
  cn_prof_bytes(4.0 * g.G * ((double)g.N * g.A * g.Hs * g.Ws + (double)g.N * g.Bc * g.Hb * g.Wb + (double)dw_floats));
  cn_prof_before(stream);
  if (T == 9 && twelve) {  // 3 x 3: twelve waves per block, one kernel row of taps per wave (cn_wgrad_vec3_kernel)
    static bool attr3 = false;
    if (!attr3) {
      (void)hipFuncSetAttribute((const void*)cn_wgrad_vec3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)cn_wgrad_vec3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr3 = true;
    }
    if (g.s == 1)
      CN_LAUNCH((cn_wgrad_vec3_kernel<1>), grid, dim3(WG3_WAVES * 64), lds, stream, g.gS[0], g.gB[0], out, g);
    else
      CN_LAUNCH((cn_wgrad_vec3_kernel<2>), grid, dim3(WG3_WAVES * 64), lds, stream, g.gS[0], g.gB[0], out, g);
  } else if (g.s == 1)
    CN_LAUNCH((cn_wgrad_vec_kernel<T, 1>), grid, dim3(256), lds, stream, g.gS[0], g.gB[0], out, g);
  else
    CN_LAUNCH((cn_wgrad_vec_kernel<T, 2>), grid, dim3(256), lds, stream, g.gS[0], g.gB[0], out, g);
  cn_prof_after(stream, T == 9 ? 2 : 3, g.flops * g.G);  // the contraction kernel alone
  if (g.slice_stride != 0) {
    // deferred: a sink on this thread takes the G sums (one record per group) and runs them later in one batched launch
    CnSliceSum js[4] = {};
    for (int i = 0; i < g.G; ++i) {
      js[i].part = ws + (long)i * nslices * g.slice_stride; js[i].dw = g.gdW[i]; js[i].slice_stride = g.slice_stride;
      js[i].n = dw_floats; js[i].nslices = (int)nslices; js[i].kind = 0;
    }
    if (cn_ss_push(js, g.G)) return cn_check_launch();
    CnWgradReduceArgs ra = {ws, g.slice_stride, (int)nslices, dw_floats, {g.gdW[0], g.gdW[1], g.gdW[2], g.gdW[3]}};
    CN_LAUNCH(cn_wgrad_reduce_kernel,
                       dim3((unsigned)((dw_floats + 255) / 256), nslices > 128 ? 16 : 1, g.G), dim3(256), 0, stream,
                       ra);
  }
  return cn_check_launch();
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
    g.buf_stride = (g.b_lds_off + WG_BC * g.plane_b + 3) / 4 * 4;
    lds = (size_t)g.buf_stride * sizeof(float);
    if (lds <= 160 * 1024 || g.a_tiles == 1) break;
    g.a_tiles >>= 1;
  }
  if (lds > 160 * 1024) return CN_ERR_LDS;
  g.nbuf = (2 * lds <= 160 * 1024) ? 2 : 1;
  lds *= g.nbuf;
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
  cn_prof_name("cn_wgrad_kernel<%d>", T);
  cn_prof_desc("wgrad_dw<%d> N%d A%d %dx%d Bc%d %dx%d s%d grid%dx%dx%d", T, g.N, g.A, g.Hs, g.Ws, g.Bc, g.Hb, g.Wb, g.s,
               gx, gy, splits);
  g.grid_x = gx; g.grid_y = gy; g.grid_z = splits;
  cn_prof_before(stream);
  CN_LAUNCH((cn_wgrad_kernel<T>), dim3(cn_xcd_grid((long)gx * gy * splits)), dim3(256), lds, stream, S, Bg,
                     dW, g);
  cn_prof_after(stream, T == 9 ? 2 : 3, g.flops);
  return cn_check_launch();
}

// Repack [B][C][H][W] planes into [B][C][cs] with row pitch Wp >= W (zero columns) and a zero tail up to cs:
// gives odd-sized tensors (25x25, 13x13, 99x99 ...) 16-byte aligned planes / even widths for the DMA variant.
struct CnPadSet {
  const float* src; long sbs; long scs;
  float* dst; int C, H, W, Wp, cs;
};
#define CN_PAD_MAX_SETS 8
struct CnPadArgs {
  CnPadSet set[CN_PAD_MAX_SETS];
  int cbegin[CN_PAD_MAX_SETS + 1];  // blockIdx.y ranges of the sets
  int nsets;
};

// All operands of one (grouped) weight gradient in ONE launch: blockIdx.y walks the channels of the sets in turn.
__global__ __launch_bounds__(256) void cn_pad_planes_kernel(const CnPadArgs a) {
  int si = 0;
#pragma unroll 1
  for (int i = 1; i < a.nsets; ++i)
    if ((int)blockIdx.y >= a.cbegin[i]) si = i;
  const CnPadSet& p = a.set[si];
  const int c = blockIdx.y - a.cbegin[si], b = blockIdx.z;
  const float* sp = p.src + b * p.sbs + (long)c * p.scs;
  float* dp = p.dst + ((long)b * p.C + c) * p.cs;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < p.cs; i += gridDim.x * 256) {
    const int r = i / p.Wp, col = i - r * p.Wp;
    dp[i] = (r < p.H && col < p.W) ? sp[r * p.W + col] : 0.f;
  }
}

static int cn_pad_planes(CnPadArgs& a, int B, hipStream_t stream) {
  if (a.nsets == 0) return CN_OK;
  int cs = 0, ctot = 0;
  for (int i = 0; i < a.nsets; ++i) {
    a.cbegin[i] = ctot;
    ctot += a.set[i].C;
    if (a.set[i].cs > cs) cs = a.set[i].cs;
  }
  a.cbegin[a.nsets] = ctot;
  int bx = (cs + 1023) / 1024;
  if (bx < 1) bx = 1;
  CN_LAUNCH(cn_pad_planes_kernel, dim3(bx, ctot, B), dim3(256), 0, stream, a);
  return cn_check_launch();
}

// Generic entry: dW_g[a][b][t] += sum S_g[n,a,gy,gx] * Bg_g[n,b,gy*s+offy[t],gx*s+offx[t]] for G (<= 4) sets of
// identical geometry in ONE launch (G == 1: the plain weight gradient).
// ws (optional, ws_floats floats): scratch for aligned / even-width copies of odd-sized operands and partial slices.
static int cn_wgrad_generic_g(int G, const float* const* Ss, long sbs, int A, int Hs, int Ws, const float* const* Bgs,
                              long bbs, int Bc, int Hb, int Wb, int s, int KH, int KW, int dil, int pad,
                              float* const* dWs, int N, float* ws, long ws_floats, hipStream_t stream) {
  if (G < 1 || G > 4) return CN_ERR_ARG;
  CnWgradGeom g = {};
  g.N = N; g.A = A; g.Hs = Hs; g.Ws = Ws; g.sbs = sbs; g.Bc = Bc; g.Hb = Hb; g.Wb = Wb; g.bbs = bbs; g.s = s;
  g.scs = (long)Hs * Ws; g.bcs = (long)Hb * Wb;
  g.T = KH * KW;
  if (N <= 0 || A <= 0 || Bc <= 0 || Hs <= 0 || Ws <= 0) return CN_OK;
  g.flops = 2.0 * N * Hs * Ws * (double)A * Bc * g.T;
  for (int ky = 0; ky < KH; ++ky)
    for (int kx = 0; kx < KW; ++kx) {
      g.offy[ky * KW + kx] = ky * dil - pad;
      g.offx[ky * KW + kx] = kx * dil - pad;
    }
  g.min_oy = -pad; g.min_ox = -pad;
  g.sa = (long)Bc * g.T;
  if (g.T != 1 && g.T != 9) return CN_ERR_ARG;
  g.G = G;
  for (int i = 0; i < G; ++i) { g.gS[i] = Ss[i]; g.gB[i] = Bgs[i]; g.gdW[i] = dWs[i]; }
  const bool ws_ok = ws != nullptr && (reinterpret_cast<uintptr_t>(ws) & 15) == 0;
  // 16-byte DMA variant when the alignment preconditions hold ...
  int rc = g.T == 1 ? cn_wgrad_launch_vec<1>(Ss[0], Bgs[0], dWs[0], g, ws_ok ? ws : nullptr, ws_floats, stream)
                    : cn_wgrad_launch_vec<9>(Ss[0], Bgs[0], dWs[0], g, ws_ok ? ws : nullptr, ws_floats, stream);
  if (rc != CN_ERR_ARG) return rc;
  // ... else through aligned copies in the caller's workspace (one streaming pass over small tensors) ...
  if (ws_ok) {
    const int Wsp = (Ws + 1) & ~1;
    const long scs = ((long)Hs * Wsp + 3) / 4 * 4, bcs = ((long)Hb * Wb + 3) / 4 * 4;
    const long need_s = (long)N * A * scs, need_b = (long)N * Bc * bcs;
    CnWgradGeom gp = g;
    CnPadArgs pa = {};
    float* w = ws;
    long need = 0;
    bool any_s = false, any_b = false;
    for (int i = 0; i < G; ++i) {
      const bool s_ok = (g.scs % 4 == 0) && (sbs % 4 == 0) && !(Ws & 1) && !(reinterpret_cast<uintptr_t>(Ss[i]) & 15);
      const bool b_ok = (g.bcs % 4 == 0) && (bbs % 4 == 0) && !(reinterpret_cast<uintptr_t>(Bgs[i]) & 15);
      any_s |= !s_ok;
      any_b |= !b_ok;
    }
    // an operand class is copied for every group or for none (the launch has one stride set)
    for (int i = 0; i < G && any_s; ++i) {
      pa.set[pa.nsets++] = {Ss[i], sbs, (long)Hs * Ws, w, A, Hs, Ws, Wsp, (int)scs};
      gp.gS[i] = w; w += need_s; need += need_s;
    }
    for (int i = 0; i < G && any_b; ++i) {
      int same = -1;  // branches that share their input: one copy
      for (int j = 0; j < i; ++j)
        if (Bgs[j] == Bgs[i]) same = j;
      if (same >= 0) { gp.gB[i] = gp.gB[same]; continue; }
      pa.set[pa.nsets++] = {Bgs[i], bbs, (long)Hb * Wb, w, Bc, Hb, Wb, Wb, (int)bcs};
      gp.gB[i] = w; w += need_b; need += need_b;
    }
    if (any_s) { gp.Ws = Wsp; gp.scs = scs; gp.sbs = (long)A * scs; }
    if (any_b) { gp.bcs = bcs; gp.bbs = (long)Bc * bcs; }
    if (need <= ws_floats) {
      const int r = cn_pad_planes(pa, N, stream);
      if (r != CN_OK) return r;
      float* wrest = ws + (need + 3) / 4 * 4;  // what the aligned copies left over serves the partial slices
      const long nrest = ws_floats - (need + 3) / 4 * 4;
      rc = g.T == 1 ? cn_wgrad_launch_vec<1>(gp.gS[0], gp.gB[0], dWs[0], gp, wrest, nrest, stream)
                    : cn_wgrad_launch_vec<9>(gp.gS[0], gp.gB[0], dWs[0], gp, wrest, nrest, stream);
      if (rc != CN_ERR_ARG) return rc;
    }
  }
  // ... else the dword variant, set by set.
  for (int i = 0; i < G; ++i) {
    CnWgradGeom g1 = g;
    g1.G = 0;
    rc = g.T == 1 ? cn_wgrad_launch_t<1>(Ss[i], Bgs[i], dWs[i], g1, stream)
                  : cn_wgrad_launch_t<9>(Ss[i], Bgs[i], dWs[i], g1, stream);
    if (rc != CN_OK) return rc;
  }
  return CN_OK;
}

static int cn_wgrad_generic(const float* S, long sbs, int A, int Hs, int Ws, const float* Bg, long bbs, int Bc,
                            int Hb, int Wb, int s, int KH, int KW, int dil, int pad, float* dW, int N, float* ws,
                            long ws_floats, hipStream_t stream) {
  return cn_wgrad_generic_g(1, &S, sbs, A, Hs, Ws, &Bg, bbs, Bc, Hb, Wb, s, KH, KW, dil, pad, &dW, N, ws, ws_floats,
                            stream);
}

// Grouped Conv2d weight gradient: G (<= 4) convolutions of identical geometry (same padding / dilation) in ONE launch
// -- the dilation branches of ResidualAConv. xs / dys / dws: HOST arrays of G device pointers (xs may repeat one
// input). ACCUMULATES into dws.
extern "C" int cn_conv2d_bwd_weight_grouped_f32(int G, const float* const* xs, long xbs, const float* const* dys,
                                                long dybs, float* const* dws, int B, int Cin, int Hin, int Win,
                                                int Cout, int KH, int KW, int stride, int pad, int dil, float* ws,
                                                long ws_floats, void* stream) {
  if (stride < 1) return CN_ERR_ARG;  // before it is divided by
  const int Hout = (Hin + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  const int Wout = (Win + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  return cn_wgrad_generic_g(G, dys, dybs, Cout, Hout, Wout, xs, xbs, Cin, Hin, Win, stride, KH, KW, dil, pad, dws, B,
                            ws, ws_floats, (hipStream_t)stream);
}

// Conv2d: dw [Cout][Cin][KH][KW] += x (*) dy. NOTE accumulates: zero dw first for a fresh gradient.
extern "C" int cn_conv2d_bwd_weight_f32(const float* x, long xbs, const float* dy, long dybs, float* dw, int B,
                                        int Cin, int Hin, int Win, int Cout, int KH, int KW, int stride, int pad,
                                        int dil, float* ws, long ws_floats, void* stream) {
  if (stride < 1) return CN_ERR_ARG;  // before it is divided by
  const int Hout = (Hin + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  const int Wout = (Win + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  return cn_wgrad_generic(dy, dybs, Cout, Hout, Wout, x, xbs, Cin, Hin, Win, stride, KH, KW, dil, pad, dw, B, ws,
                          ws_floats, (hipStream_t)stream);
}

// ConvTranspose2d: dw [Cin][Cout][KH][KW] += x (small grid) (*) dy (gathered at stride s).
extern "C" int cn_conv_transpose2d_bwd_weight_f32(const float* x, long xbs, const float* dy, long dybs, float* dw,
                                                  int B, int Cin, int Hin, int Win, int Cout, int KH, int KW,
                                                  int stride, int pad, int out_pad, float* ws, long ws_floats,
                                                  void* stream) {
  if (out_pad < 0 || (out_pad > 0 && out_pad >= stride)) return CN_ERR_ARG;
  const int Hout = (Hin - 1) * stride - 2 * pad + KH + out_pad;  // dy lives on the output_padding grid (cn_conv.hip)
  const int Wout = (Win - 1) * stride - 2 * pad + KW + out_pad;
  return cn_wgrad_generic(x, xbs, Cin, Hin, Win, dy, dybs, Cout, Hout, Wout, stride, KH, KW, 1, pad, dw, B, ws,
                          ws_floats, (hipStream_t)stream);
}
