// bf16 weight gradients for gfx950 (MI355X): v_mfma_f32_32x32x16_bf16 with the PIXELS as the contraction index.
//
// Weight gradients of nn.Conv2d / nn.ConvTranspose2d on the mixed-precision path (activations and activation
// gradients bf16 NHWC, parameter gradients fp32 in the flat gradient buffer; reference: autograd of
// /root/reference/src/cultionet/nn/modules/convolution.py:45-120 under precision="16-mixed", model.py:168-186).
// Both layers are the same contraction over a pixel grid g:
//     dW[cP][cQ][t] += sum_{b, g}  P[b][g][cP] * Q[b][g*s + off_t][cQ]
//   Conv2d:           P = dy (cP = cout), Q = x  (cQ = cin),  off_t = k*dil - pad      -> dW laid out [Cout][Cin][T]
//   ConvTranspose2d:  P = x  (cP = cin),  Q = dy (cQ = cout), off_t = k - pad          -> dW laid out [Cin][Cout][T]
// i.e. dW index = (cP * CQ + cQ) * T + t in both cases.
//
// NHWC makes both operands pixel-major ([pixel][channel] rows in LDS), while an MFMA fragment wants 8 consecutive k
// (= pixels) per lane: gfx950's transposing LDS read ds_read_b64_tr_b16 delivers exactly that (4 pixel rows x 16
// channels per 16-lane group, column-major), and because every row of a transposed block is addressed by its own
// lane, a tap is again just a row offset into the halo image -- no alignment constraints, no shifted copies.
//
// Block = 4 waves = 64 cP x 64 cQ x all T taps, T*16 accumulator registers per wave; it walks a contiguous range of
// pixel tiles (split-K over tiles) and stores its partial [T][64][64] fp32 slice into the workspace;
// cn_bwgrad_reduce_kernel sums the slices into dW (+=).
//   1x1: wave = one 32 x 32 quadrant.
//   3x3: wave = ALL FOUR quadrants of two taps + one quadrant of the centre tap (9 MFMAs per k-step either way). With a
//   quadrant x 9 taps per wave every k-step read 2 + 18 transposed fragments for 9 MFMAs: 4 waves x 20 x 4 cycles = 320
//   cycles of LDS for 288 cycles of matrix pipe -- the multiplication phase was LDS-bound (2888 of a tile's 5090 cycles,
//   tools/bwgrad_stamps.py). Two P fragments x (2 taps x 2 Q fragments) + the centre's = 14 reads per k-step: 224.
#include "cn_bf16.h"
#include "cn_profile.h"
#include "cn_slicesum.h"

#define CNW_PITCH 192  // bytes per pixel row in LDS (64 channels = 128 B + 64 B pad): conflict-free transposed reads
#define CNW_MAX_TAPS 9

// Diagnostic build (-DCNBW_STAMP): s_memtime stamps of wave 0 of one block (tools/bwgrad_stamps.py).
#ifdef CNBW_STAMP
__device__ unsigned long long cnbw_stamps[256];
#define CNBW_ST() do { if (do_stamp && stamp_i < 256) cnbw_stamps[stamp_i++] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int cn_bwgrad_read_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(cnbw_stamps), sizeof(unsigned long long) * 256) == hipSuccess ? 0 : -2;
}
#else
#define CNBW_ST() do { } while (0)
#endif

struct CnBWgGeom {
  const bf16_t* P;
  const bf16_t* Q;
  long ldp, ldq;
  int B, CP, CQ;
  int Hg, Wg;      // pixel grid (= spatial size of P)
  int Hq, Wq;      // spatial size of Q
  int s;           // Q pixel = g*s + off
  int T;
  int doff[CNW_MAX_TAPS];  // LDS byte offset of each tap inside the Q halo image
  int qy_off, qx_off;      // Q coordinate of halo pixel (0,0) relative to tile origin * s
  int IH, IW;
  int TH, TW;
  int tiles_x, tiles_per_img, ntiles;
  int nsplit, tiles_per_split;
  int nbp, nbq;    // 64-channel blocks over cP / cQ
  float* part;     // [nsplit][T][nbp*64][nbq*64]
  int total;
};

__device__ __forceinline__ bf16x4 cnw_tr(const unsigned char* lds_addr) {
  typedef short s4 __attribute__((ext_vector_type(4)));
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s4*)(const_cast<unsigned char*>(lds_addr)));
}

// NQ = 16-byte pieces of the Q halo image per thread that are register-prefetched (0: synchronous staging loop for
// large halos, e.g. strided layers).
template <int T, int NQ, bool FULL>
__global__ __launch_bounds__(256, (T > 4 ? 1 : 2)) void cn_bwgrad_kernel(const CnBWgGeom g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wid >> 1, wq = wid & 1;
  int L;
  {
    const int lin = blockIdx.x;
    const int per = (g.total + 7) >> 3;
    L = (lin & 7) * per + (lin >> 3);
    if (L >= g.total) return;
  }
  // L = (split * nbp + bp) * nbq + bq : the blocks of one split (same pixels) are neighbours on one XCD
  const int bq = L % g.nbq;
  const int bp = (L / g.nbq) % g.nbp;
  const int split = L / (g.nbq * g.nbp);
  const int npix = g.TH * g.TW;
  const int IW = g.IW;
  unsigned char* ldsP = lds;
  unsigned char* ldsQ = lds + 128 * CNW_PITCH;

  // transposed-read geometry: within a 16-lane group, lane 4q+p addresses row q, columns 4p..4p+3
  const int g16 = lane >> 4;
  const int rq = (lane & 15) >> 2, rp = lane & 3;
  const int colb = (16 * (g16 & 1) + 4 * rp) * 2;  // byte offset of this lane's 4 columns inside a 32-channel tile
  const int kh8 = 8 * (g16 >> 1);                   // k offset of the lane's half
  // per k-step (16 pixels): Q-image byte offset of the lane's row for the two reads (k = kh8 + rq, kh8 + 4 + rq),
  // column offset folded in; out-of-tile pixels are clamped (their P rows are zero)
  int qrow[16];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int m = ks * 16 + kh8 + 4 * e + rq;
      const int ty = m / g.TW, tx = m - ty * g.TW;
      qrow[ks * 2 + e] = (m < npix ? ((ty * g.s) * IW + tx * g.s) * CNW_PITCH : 0) + wq * 64 + colb + 128 * CNW_PITCH;
    }
  const int prow0 = (kh8 + rq) * CNW_PITCH + wp * 64 + colb;
  // 3x3: fragment 0 of either operand is the wave's OWN quadrant (wp / wq: the centre tap's), fragment 1 the other one
  const int pflip = 64 - 128 * wp, qflip = 64 - 128 * wq;
  const int ta = wid < 2 ? 2 * wid : 2 * wid + 1;  // taps ta, ta + 1 (0 1 | 2 3 | 5 6 | 7 8) and a quarter of tap 4
  constexpr int TB = T == 9 ? 5 : T;               // Q fragments per k-step and set
  int doff[TB];
  if constexpr (T == 9) {
    doff[0] = g.doff[ta]; doff[1] = g.doff[ta] + qflip; doff[2] = g.doff[ta + 1]; doff[3] = g.doff[ta + 1] + qflip;
    doff[4] = g.doff[4];
  } else {
#pragma unroll
    for (int t = 0; t < T; ++t) doff[t] = g.doff[t];
  }

  f32x16 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;

  const int t_begin = split * g.tiles_per_split;
  const int t_end = t_begin + g.tiles_per_split < g.ntiles ? t_begin + g.tiles_per_split : g.ntiles;
  const int cp0 = bp * 64, cq0 = bq * 64;
  const int nq_pieces = g.IH * IW * 8;  // 16-byte pieces of the Q image (8 per pixel)
  const int ksteps = (npix + 15) >> 4;

  // per-thread staging descriptors (tile independent): P piece i = pixel (tid + 256 i) / 8, Q piece likewise
  constexpr int NQR = NQ > 0 ? NQ : 1;
  // One wave per SIMD pays ~4-5 cycles for EVERY instruction with the matrix pipe idle, and the tile-dependent address
  // arithmetic of the 12 prefetch loads (two divisions, 64-bit multiplies, five compares per piece: ~450 instructions)
  // took 2.5k of a tile's 6.7k cycles (tools/bwgrad_stamps.py). Everything that does not depend on the tile is folded
  // into per-lane BYTE offsets once; per tile there is one scalar base per operand and two add + compare pairs per piece.
  // Statically dead pieces (beyond the tile / the channel count / the halo image) get a row that fails every test.
  int pm[4], pty[4], ptx[4];
  unsigned pob[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    pm[i] = (tid + i * 256) >> 3;
    pty[i] = pm[i] / g.TW;
    ptx[i] = pm[i] - pty[i] * g.TW;
  }
  const int c8 = (tid & 7) * 8;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    pob[i] = (unsigned)((((long)pty[i] * g.Wg + ptx[i]) * g.ldp + c8) * 2);
    if (!(pm[i] < npix && cp0 + c8 < g.CP)) pty[i] = 1 << 24;
  }
  int qhy[NQR], qhx[NQR];
  unsigned qob[NQR];
  if (NQ > 0) {
#pragma unroll
    for (int i = 0; i < NQR; ++i) {
      const int p = (tid + i * 256) >> 3;
      qhy[i] = p / IW;
      qhx[i] = p - qhy[i] * IW;
      qob[i] = (unsigned)((((long)qhy[i] * g.Wq + qhx[i]) * g.ldq + c8) * 2);
      if (!(tid + i * 256 < nq_pieces && cq0 + c8 < g.CQ)) qhy[i] = 1 << 24;
    }
  }
  // synchronous large-halo staging (NQ == 0): this lane's first halo pixel and its step of 32 pixels, as (row, column)
  const int sy0 = (tid >> 3) / IW, sx0 = (tid >> 3) - sy0 * IW;
  const int d32 = 32 / IW, r32 = 32 - d32 * IW;
  constexpr unsigned CNW_OOB = 0x80000000u;
  // fetch cursor: the tile the next fetch() loads (tiles are fetched in order), kept as (image, tile row, tile column)
  const int tiles_y = g.tiles_per_img / g.tiles_x;
  int fb = t_begin / g.tiles_per_img;
  int fty = (t_begin - fb * g.tiles_per_img) / g.tiles_x;
  int ftx = t_begin - fb * g.tiles_per_img - fty * g.tiles_x;
  // Register prefetch sets A (and, for the 1x1 kernels, B: two tiles of distance).
  // (Tried for the 3x3 kernels: two LDS images with the ds_write_b128 of tile t+1 and the buffer loads of tile t+2
  // interleaved between the MFMAs of tile t, one barrier per tile -- 122 -> 132 us at 128->128, 100^2: the writes share
  // the in-order LGKM counter with the transposed reads, so every read wait also drains the slow stores.)
  u32x4 pvA[4], qvA[NQR], pvB[4], qvB[NQR];
  // Buffer loads: a lane whose piece is out of range gets an offset beyond num_records and the hardware returns zeros --
  // no EXEC masking, no zero initialisation of the destination registers (4 v_mov per piece).
  auto fetch = [&](int, u32x4 (&pv)[4], u32x4 (&qv)[NQR]) {
    const int gy0 = fty * g.TH, gx0 = ftx * g.TW;
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(g.P + (((long)fb * g.Hg + gy0) * g.Wg + gx0) * g.ldp + cp0), 0, 0x7fffffff, 0x00020000);
    const int ylim = g.Hg - gy0, xlim = g.Wg - gx0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned off = (pty[i] < ylim && ptx[i] < xlim) ? pob[i] : CNW_OOB;
      pv[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, off, 0, 0));
    }
    if (NQ > 0) {
      const int qy0 = gy0 * g.s + g.qy_off, qx0 = gx0 * g.s + g.qx_off;
      // qy0 / qx0 may be negative (halo above / left of the image): the base then points before the plane and is only
      // dereferenced by lanes whose pixel passes the range test
      const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<bf16_t*>(g.Q + (((long)fb * g.Hq + qy0) * g.Wq + qx0) * g.ldq + cq0), 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int i = 0; i < NQR; ++i) {
        const unsigned off =
            ((unsigned)(qhy[i] + qy0) < (unsigned)g.Hq && (unsigned)(qhx[i] + qx0) < (unsigned)g.Wq) ? qob[i] : CNW_OOB;
        qv[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(qrs, off, 0, 0));
      }
    }
    if (++ftx == g.tiles_x) {
      ftx = 0;
      if (++fty == tiles_y) { fty = 0; ++fb; }
    }
  };
  // The same fetch in pieces, for the kernels that issue it BETWEEN the MFMAs of the running tile (TRICKLE below): the
  // scalar part (cursor, two buffer resources, limits) and twelve independent (offset select + load) pieces. `live` =
  // there is a next tile; otherwise every lane's offset is out of range (zeros, no memory access, no branch).
  __amdgpu_buffer_rsrc_t tprs, tqrs;
  int tylim = 0, txlim = 0, tqy0 = 0, tqx0 = 0, thq = 0;
  auto fetch_setup = [&](bool live) __attribute__((always_inline)) {
    const int gy0 = fty * g.TH, gx0 = ftx * g.TW;
    tprs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(g.P + (((long)fb * g.Hg + gy0) * g.Wg + gx0) * g.ldp + cp0), 0, 0x7fffffff, 0x00020000);
    tylim = live ? g.Hg - gy0 : 0;
    txlim = g.Wg - gx0;
    tqy0 = gy0 * g.s + g.qy_off;
    tqx0 = gx0 * g.s + g.qx_off;
    thq = live ? g.Hq : 0;
    tqrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(g.Q + (((long)fb * g.Hq + tqy0) * g.Wq + tqx0) * g.ldq + cq0), 0, 0x7fffffff, 0x00020000);
    if (++ftx == g.tiles_x) {
      ftx = 0;
      if (++fty == tiles_y) { fty = 0; ++fb; }
    }
  };
  auto fetch_p = [&](int i, u32x4 (&pv)[4]) __attribute__((always_inline)) {
    const unsigned off = (pty[i] < tylim && ptx[i] < txlim) ? pob[i] : CNW_OOB;
    pv[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(tprs, off, 0, 0));
  };
  auto fetch_q = [&](int i, u32x4 (&qv)[NQR]) __attribute__((always_inline)) {
    const unsigned off =
        ((unsigned)(qhy[i] + tqy0) < (unsigned)thq && (unsigned)(qhx[i] + tqx0) < (unsigned)g.Wq) ? qob[i] : CNW_OOB;
    qv[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(tqrs, off, 0, 0));
  };
  auto store = [&](int tile, const u32x4 (&pv)[4], const u32x4 (&qv)[NQR]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(ldsP + pm[i] * CNW_PITCH + c8 * 2) = pv[i];
    if (NQ > 0) {
#pragma unroll
      for (int i = 0; i < NQR; ++i)  // unconditional: the LDS image is sized for all NQ * 256 pieces (dead ones hold zeros)
        *reinterpret_cast<u32x4*>(ldsQ + ((tid + i * 256) >> 3) * CNW_PITCH + c8 * 2) = qv[i];
    } else {
      // Large halo (stride 2, transposed, dilation >= 5): staged synchronously, four pieces in flight per lane. The
      // lane's halo pixel walks by 32 pixels per piece (incremental row / column, no division), loads are bounds-checked
      // buffer loads (zeros for the out-of-image halo), so a piece costs ~10 instructions instead of ~45.
      const int b = tile / g.tiles_per_img;
      const int tl = tile - b * g.tiles_per_img;
      const int tyi = tl / g.tiles_x, txi = tl - tyi * g.tiles_x;
      const int qy0 = tyi * g.TH * g.s + g.qy_off, qx0 = txi * g.TW * g.s + g.qx_off;
      const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<bf16_t*>(g.Q + (((long)b * g.Hq + qy0) * g.Wq + qx0) * g.ldq + cq0), 0, 0x7fffffff, 0x00020000);
      const bool chan_ok = cq0 + c8 < g.CQ;
      int hy = sy0, hx = sx0;
#pragma unroll 1
      for (int q = tid; q < nq_pieces; q += 1024) {
        u32x4 v[4];
        int pp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool in = chan_ok && q + 256 * u < nq_pieces && (unsigned)(hy + qy0) < (unsigned)g.Hq &&
                          (unsigned)(hx + qx0) < (unsigned)g.Wq;
          const unsigned off = in ? (unsigned)((hy * g.Wq + hx) * (int)g.ldq + c8) * 2u : CNW_OOB;
          v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(qrs, off, 0, 0));
          pp[u] = (q + 256 * u) >> 3;
          hy += d32;
          hx += r32;
          if (hx >= IW) { hx -= IW; ++hy; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (q + 256 * u < nq_pieces) *reinterpret_cast<u32x4*>(ldsQ + pp[u] * CNW_PITCH + c8 * 2) = v[u];
      }
    }
  };

  // k-steps are software-pipelined by hand over two static fragment sets: the 2 + 2T transposed reads of k-step
  // ks+1 are issued before the T MFMAs of k-step ks (the compiler alone keeps one read pair ahead, which exposes
  // the LDS latency in front of every MFMA: one wave per SIMD has nothing else to run)
#define CNW_READ(A_, B_, KS_)                                                                   \
  {                                                                                             \
    _Pragma("unroll") for (int a = 0; a < NA; ++a) {                                            \
      const bf16x4 pa_ = cnw_tr(lds + prow0 + a * pflip + (KS_) * 16 * CNW_PITCH);              \
      const bf16x4 pb_ = cnw_tr(lds + prow0 + a * pflip + ((KS_) * 16 + 4) * CNW_PITCH);        \
      A_[a] = bf16x8{pa_[0], pa_[1], pa_[2], pa_[3], pb_[0], pb_[1], pb_[2], pb_[3]};           \
    }                                                                                           \
    _Pragma("unroll") for (int t = 0; t < TB; ++t) {                                            \
      const bf16x4 qa_ = cnw_tr(lds + qrow[(KS_) * 2] + doff[t]);                               \
      const bf16x4 qb_ = cnw_tr(lds + qrow[(KS_) * 2 + 1] + doff[t]);                           \
      B_[t] = bf16x8{qa_[0], qa_[1], qa_[2], qa_[3], qb_[0], qb_[1], qb_[2], qb_[3]};           \
    }                                                                                           \
  }
  // 3x3 accumulators: [0..3] tap ta, [4..7] tap ta + 1, each (P frag, Q frag) = (0,0) (0,1) (1,0) (1,1); [8] centre
#define CNW_MMA(A_, B_)                                                                         \
  {                                                                                             \
    if constexpr (T == 9) {                                                                     \
      _Pragma("unroll") for (int u = 0; u < 8; ++u)                                             \
          acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[(u >> 1) & 1], B_[(u >> 2) * 2 + (u & 1)], acc[u], 0, 0, 0); \
      acc[8] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[0], B_[4], acc[8], 0, 0, 0);          \
    } else {                                                                                    \
      _Pragma("unroll") for (int t = 0; t < T; ++t)                                             \
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[0], B_[t], acc[t], 0, 0, 0);      \
    }                                                                                           \
  }
  constexpr int NA = T == 9 ? 2 : 1;
  bf16x8 AX[NA], AY[NA], BX[TB], BY[TB];
  // TRICKLE (3x3, register-prefetched halo, full tiles): the next tile's fetch -- ~135 instructions that cost 1060 of a
  // tile's 5090 cycles with the matrix pipe idle (tools/bwgrad_stamps.py) -- is issued in six parts of two pieces inside
  // the MFMA shadows of k-steps 0-5 instead of in front of the multiplication.
  constexpr bool TRICKLE = FULL && T == 9 && NQ == 8;
  auto compute = [&](bool live) __attribute__((always_inline)) {
    CNW_READ(AX, BX, 0);
    if constexpr (TRICKLE) {
#define CNW_INTERLEAVE_F()                                               \
  {                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < T; ++i_) {                   \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); /* MFMA */      \
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); /* DS read */   \
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); /* VALU */      \
      if (i_ == 3 || i_ == 7)                                            \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); /* VMEM */    \
    }                                                                    \
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                   \
    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                   \
  }
#define CNW_INTERLEAVE()                                                 \
  {                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < T; ++i_) {                   \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 \
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                 \
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                 \
    }                                                                    \
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                   \
    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                   \
  }
      fetch_setup(live);
#pragma unroll
      for (int ks = 0; ks < 8; ks += 2) {
        __builtin_amdgcn_sched_barrier(0);
        CNW_READ(AY, BY, ks + 1);
        CNW_MMA(AX, BX);
        if (ks < 6) {
          if (ks == 0) { fetch_p(0, pvA); fetch_p(1, pvA); }
          else { fetch_q(2 * ks - 4, qvA); fetch_q(2 * ks - 3, qvA); }   // ks 2: q0 q1, ks 4: q4 q5
          CNW_INTERLEAVE_F();
        } else {
          CNW_INTERLEAVE();
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 2 < 8) CNW_READ(AX, BX, ks + 2);
        CNW_MMA(AY, BY);
        if (ks < 6) {
          if (ks == 0) { fetch_p(2, pvA); fetch_p(3, pvA); }
          else { fetch_q(2 * ks - 2, qvA); fetch_q(2 * ks - 1, qvA); }   // ks 2: q2 q3, ks 4: q6 q7
          CNW_INTERLEAVE_F();
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#undef CNW_INTERLEAVE
#undef CNW_INTERLEAVE_F
    } else if (FULL) {
      // Full tiles (8 k-steps, no conditionals => one basic block per k-step): the transposed reads and address adds
      // of k-step ks+1 are INTERLEAVED with the MFMAs of k-step ks, two reads + two VALU in each MFMA's 32-cycle
      // shadow.  One wave per SIMD has nobody to hide behind: a phase-separated order leaves the matrix pipe idle for
      // the ~300 cycles of read issue per k-step.
#define CNW_INTERLEAVE()                                                 \
  {                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < T; ++i_) {                   \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); /* MFMA */      \
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); /* DS read */   \
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); /* VALU */      \
    }                                                                    \
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                   \
    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                   \
  }
#pragma unroll
      for (int ks = 0; ks < 8; ks += 2) {
        __builtin_amdgcn_sched_barrier(0);
        CNW_READ(AY, BY, ks + 1);
        CNW_MMA(AX, BX);
        CNW_INTERLEAVE();
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 2 < 8) CNW_READ(AX, BX, ks + 2);
        CNW_MMA(AY, BY);
        if (ks + 2 < 8) CNW_INTERLEAVE();
        __builtin_amdgcn_sched_barrier(0);
      }
#undef CNW_INTERLEAVE
    } else {
#pragma unroll
      for (int ks = 0; ks < 8; ks += 2) {
        if (ks < ksteps) {
          if (ks + 1 < ksteps) CNW_READ(AY, BY, ks + 1);
          __builtin_amdgcn_sched_barrier(0);
          CNW_MMA(AX, BX);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (ks + 1 < ksteps) {
          if (ks + 2 < ksteps) CNW_READ(AX, BX, ks + 2);
          __builtin_amdgcn_sched_barrier(0);
          CNW_MMA(AY, BY);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };
  if constexpr (T == 1) {
    // 1x1: one MFMA per k-step -- a tile is ~1k cycles of work, far less than the global latency: two tiles in flight
    // (73 -> 46 us at 128->128, 100^2). The 3x3 kernels lose 6 % with the doubled loop body and keep one set.
    if (t_begin < t_end) fetch(t_begin, pvA, qvA);
    if (t_begin + 1 < t_end) fetch(t_begin + 1, pvB, qvB);
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; tile += 2) {
      __syncthreads();  // previous tile's reads are done
      store(tile, pvA, qvA);
      __syncthreads();
      if (tile + 2 < t_end) fetch(tile + 2, pvA, qvA);  // global loads fly over two tiles of multiplication
      compute(false);
      if (tile + 1 < t_end) {
        __syncthreads();
        store(tile + 1, pvB, qvB);
        __syncthreads();
        if (tile + 3 < t_end) fetch(tile + 3, pvB, qvB);
        compute(false);
      }
    }
  } else {
#ifdef CNBW_STAMP
    const bool do_stamp = T == 9 && NQ == 8 && FULL && blockIdx.x == 16 && tid == 0;
    int stamp_i = 0;
#endif
    CNBW_ST();
    if (t_begin < t_end) fetch(t_begin, pvA, qvA);
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; ++tile) {
      CNBW_ST();  // tile top
      __syncthreads();  // previous tile's reads are done
      CNBW_ST();  // barrier 1
      store(tile, pvA, qvA);
      CNBW_ST();  // LDS stores issued (includes the wait for the prefetched registers)
      __syncthreads();
      CNBW_ST();  // barrier 2
      if constexpr (!TRICKLE) {
        if (tile + 1 < t_end) fetch(tile + 1, pvA, qvA);  // next tile's global loads fly while this one is multiplied
      }
      CNBW_ST();  // fetch issued
      compute(tile + 1 < t_end);
      CNBW_ST();  // multiplied
    }
  }
#undef CNW_READ
#undef CNW_MMA
  // partial slice: part[split][t][cP][cQ] (lanes contiguous along cQ)
  const int r = lane & 31, h = lane >> 5;
  const long CPp = (long)g.nbp * 64, CQp = (long)g.nbq * 64;
  float* out = g.part + (long)split * T * CPp * CQp;
#pragma unroll
  for (int u = 0; u < T; ++u) {
    int t = u, tp = wp, tq = wq;  // tap and quadrant of accumulator u
    if constexpr (T == 9) {
      if (u < 8) {
        t = ta + (u >> 2);
        tp = wp ^ ((u >> 1) & 1);
        tq = wq ^ (u & 1);
      } else {
        t = 4;
      }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int cp = cp0 + tp * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
      out[((long)t * CPp + cp) * CQp + cq0 + tq * 32 + r] = acc[u][j];
    }
  }
}

// dw[(cP*CQ + cQ)*T + t] += sum_split part[split][t][cP][cQ]
// 64 outputs per block (consecutive cq: coalesced slice reads), the splits dealt over the block's 4 waves with 4
// independent loads in flight per thread, combined through LDS.
__global__ __launch_bounds__(256) void cn_bwgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                              int nsplit, int T, int CP, int CQ, long CPp, long CQp) {
  __shared__ float red[4][64];
  const long n = (long)CP * CQ * T;
  const long slice = (long)T * CPp * CQp;
  const int lane = threadIdx.x & 63, kg = threadIdx.x >> 6;
  for (long i0 = blockIdx.x * 64L; i0 < n; i0 += (long)gridDim.x * 64) {
    const long i = i0 + lane;  // i = (t*CP + cp)*CQ + cq
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int cq = 0, cp = 0, t = 0;
    if (i < n) {
      cq = (int)(i % CQ);
      const long r = i / CQ;
      cp = (int)(r % CP);
      t = (int)(r / CP);
      const float* p = part + ((long)t * CPp + cp) * CQp + cq;
      int k = kg;
      for (; k + 12 < nsplit; k += 16) {
        s0 += p[k * slice];
        s1 += p[(k + 4) * slice];
        s2 += p[(k + 8) * slice];
        s3 += p[(k + 12) * slice];
      }
      for (; k < nsplit; k += 4) s0 += p[k * slice];
    }
    red[kg][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (kg == 0 && i < n)
      dw[((long)cp * CQ + cq) * T + t] += (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    __syncthreads();
  }
}

static void cnw_pick_tile(int Hg, int Wg, int s, int span, int& TH, int& TW) {
  static const int cand[][2] = {{5, 25}, {8, 16}, {4, 32}, {16, 8}, {2, 64}, {10, 12}, {6, 20}, {9, 14}, {11, 11},
                                {13, 9}, {7, 18}, {1, 128}, {4, 16}, {8, 8}, {2, 32}, {4, 8}, {2, 16}};
  double best = 1e300;
  TH = 4; TW = 8;
  for (auto& c : cand) {
    const int th = c[0], tw = c[1];
    const long ih = (long)(th - 1) * s + span + 1, iw = (long)(tw - 1) * s + span + 1;
    if (ih * iw > 600) continue;
    const long tiles = (long)((Hg + th - 1) / th) * ((Wg + tw - 1) / tw);
    const double cost = (double)tiles * ((th * tw + 15) / 16) * 16.0 + 0.05 * tiles * ih * iw;
    if (cost < best) { best = cost; TH = th; TW = tw; }
  }
}

static int cnw_plan(CnBWgGeom& g, int KH, int KW, int pad, int dil) {
  const int T = KH * KW;
  if (T > CNW_MAX_TAPS || g.s < 1 || dil < 1) return CN_ERR_ARG;
  g.T = T;
  int mn_y = 0, mx_y = 0, mn_x = 0, mx_x = 0;
  int dys[CNW_MAX_TAPS], dxs[CNW_MAX_TAPS];
  for (int ky = 0; ky < KH; ++ky)
    for (int kx = 0; kx < KW; ++kx) {
      const int t = ky * KW + kx;
      dys[t] = ky * dil - pad; dxs[t] = kx * dil - pad;
      if (t == 0) { mn_y = mx_y = dys[t]; mn_x = mx_x = dxs[t]; }
      mn_y = dys[t] < mn_y ? dys[t] : mn_y; mx_y = dys[t] > mx_y ? dys[t] : mx_y;
      mn_x = dxs[t] < mn_x ? dxs[t] : mn_x; mx_x = dxs[t] > mx_x ? dxs[t] : mx_x;
    }
  const int span = (mx_y - mn_y) > (mx_x - mn_x) ? (mx_y - mn_y) : (mx_x - mn_x);
  cnw_pick_tile(g.Hg, g.Wg, g.s, span, g.TH, g.TW);
  g.qy_off = mn_y; g.qx_off = mn_x;
  g.IH = (g.TH - 1) * g.s + (mx_y - mn_y) + 1;
  g.IW = (g.TW - 1) * g.s + (mx_x - mn_x) + 1;
  for (int t = 0; t < T; ++t) g.doff[t] = ((dys[t] - mn_y) * g.IW + (dxs[t] - mn_x)) * CNW_PITCH;
  g.tiles_x = (g.Wg + g.TW - 1) / g.TW;
  g.tiles_per_img = g.tiles_x * ((g.Hg + g.TH - 1) / g.TH);
  g.ntiles = g.tiles_per_img * g.B;
  g.nbp = (g.CP + 63) / 64;
  g.nbq = (g.CQ + 63) / 64;
  const int nb = g.nbp * g.nbq;
  // T = 9 keeps 144 accumulator registers per wave: one block per CU => one full round of 256 blocks; the 1x1
  // kernel runs two blocks per CU
  const int want = T > 4 ? 256 : 512;
  int nsplit = (want + nb - 1) / nb;
  if (nsplit > g.ntiles) nsplit = g.ntiles;
  if (nsplit < 1) nsplit = 1;
  g.tiles_per_split = (g.ntiles + nsplit - 1) / nsplit;
  g.nsplit = (g.ntiles + g.tiles_per_split - 1) / g.tiles_per_split;
  g.total = g.nsplit * nb;
  return CN_OK;
}

static long cnw_ws_floats(const CnBWgGeom& g) { return (long)g.nsplit * g.T * g.nbp * 64 * g.nbq * 64; }

static int cnw_run(CnBWgGeom& g, float* dw, float* ws, long ws_floats, hipStream_t stream) {
  if (g.ntiles <= 0 || g.CP <= 0 || g.CQ <= 0) return CN_OK;
  // tile prefetches address an image through buffer resources with 32-bit byte offsets: refuse images beyond 2 GiB
  if ((long)g.Hg * g.Wg * g.ldp * 2 >= (1L << 31) || (long)g.Hq * g.Wq * g.ldq * 2 >= (1L << 31)) return CN_ERR_ARG;
  // shrink the split until the partial slices fit the workspace
  while (cnw_ws_floats(g) > ws_floats && g.nsplit > 1) {
    g.tiles_per_split *= 2;
    g.nsplit = (g.ntiles + g.tiles_per_split - 1) / g.tiles_per_split;
    g.total = g.nsplit * g.nbp * g.nbq;
  }
  if (ws == nullptr || cnw_ws_floats(g) > ws_floats) return CN_ERR_ARG;
  g.part = ws;
  const int nq = (g.IH * g.IW * 8 + 255) / 256;  // Q pieces per thread: <= 8 are register-prefetched
  // register-prefetched halo images are stored without a per-piece predicate: room for every thread's NQ pieces
  const int q_pix = nq <= 8 ? (nq <= 4 && g.T == 1 ? 4 : 8) * 32 : (nq <= 16 && g.T == 9 ? 16 * 32 : g.IH * g.IW);
  const size_t shmem = (size_t)(128 + (q_pix > g.IH * g.IW ? q_pix : g.IH * g.IW)) * CNW_PITCH;
  if (shmem > 160 * 1024) return CN_ERR_LDS;
  const dim3 grid(cn_xcd_grid(g.total)), block(256);
  const double flops = 2.0 * g.B * (double)g.Hg * g.Wg * g.CP * g.CQ * g.T;
  {
    const int nqn = (g.IH * g.IW * 8 + 255) / 256;
    cn_prof_name("cn_bwgrad_kernel<%d, %d, %s>", g.T,
                 nqn <= 8 ? (g.T == 1 && nqn <= 4 ? 4 : 8) : (nqn <= 16 && g.T == 9 ? 16 : 0),
                 g.TH * g.TW > 112 ? "true" : "false");
  }
  cn_prof_desc("bwgrad B%d %dx%d %dx%d T%d s%d split%d", g.B, g.Hg, g.Wg, g.CP, g.CQ, g.T, g.s, g.nsplit);
  cn_prof_bytes(2.0 * g.B * ((double)g.Hg * g.Wg * g.CP + (double)g.Hq * g.Wq * g.CQ) + 4.0 * g.T * g.CP * g.CQ);
  cn_prof_before(stream);
#define CNW_GO1(T_, NQ_, F_)                                                                                   \
  do {                                                                                                         \
    if (shmem > 64 * 1024)                                                                                     \
      (void)hipFuncSetAttribute((const void*)cn_bwgrad_kernel<T_, NQ_, F_>,                                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);                       \
    CN_LAUNCH((cn_bwgrad_kernel<T_, NQ_, F_>), grid, block, shmem, stream, g);                        \
  } while (0)
  const bool full = g.TH * g.TW > 112;  // all 8 k-steps of 16 pixels are live
#define CNW_GO(T_, NQ_)                                                                                        \
  do {                                                                                                         \
    if (full) CNW_GO1(T_, NQ_, true); else CNW_GO1(T_, NQ_, false);                                            \
  } while (0)
  switch (g.T) {
    case 1: if (nq <= 4) CNW_GO(1, 4); else if (nq <= 8) CNW_GO(1, 8); else CNW_GO(1, 0); break;
    case 9:  // dilations 2-4 of a 5x25 tile need 9-14 halo pieces per thread: still register-prefetched
      if (nq <= 8) CNW_GO(9, 8); else if (nq <= 16) CNW_GO(9, 16); else CNW_GO(9, 0);
      break;
    default: return CN_ERR_ARG;
  }
#undef CNW_GO
#undef CNW_GO1
  cn_prof_after(stream, 5, flops);  // the contraction kernel alone
  const long n = (long)g.CP * g.CQ * g.T;
  {  // deferred: a sink on this thread takes the sum and runs it later in one batched launch (cn_slicesum.h)
    CnSliceSum j = {};
    j.part = g.part; j.dw = dw; j.slice_stride = (long)g.T * g.nbp * 64 * g.nbq * 64; j.n = n; j.nslices = g.nsplit;
    j.kind = 1; j.T = g.T; j.CP = g.CP; j.CQ = g.CQ; j.CPp = g.nbp * 64; j.CQp = g.nbq * 64;
    if (cn_ss_push(&j, 1)) return cn_check_launch();
  }
  const int rb = (int)((n + 63) / 64 < 8192 ? (n + 63) / 64 : 8192);
  CN_LAUNCH(cn_bwgrad_reduce_kernel, dim3(rb), dim3(256), 0, stream, g.part, dw, g.nsplit, g.T, g.CP, g.CQ,
                     (long)g.nbp * 64, (long)g.nbq * 64);
  return cn_check_launch();
}

// Scratch floats that always suffice for the weight gradient of one layer (partial slices of the pixel split).
extern "C" long cn_bwgrad_workspace_floats(int B, int Cin, int Hin, int Win, int Cout, int KH, int KW, int stride,
                                           int pad, int dil, int transposed) {
  CnBWgGeom g = {};
  g.B = B;
  if (!transposed) {
    g.CP = Cout; g.CQ = Cin; g.s = stride;
    g.Hg = (Hin + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
    g.Wg = (Win + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  } else {
    g.CP = Cin; g.CQ = Cout; g.s = stride; g.Hg = Hin; g.Wg = Win;
  }
  if (cnw_plan(g, KH, KW, pad, dil) != CN_OK) return -1;
  return cnw_ws_floats(g);
}

// Conv2d: dw [Cout][Cin][KH][KW] (fp32) += ...; x bf16 NHWC [B,Hin,Win,Cin] (ldx), dy bf16 NHWC [B,Hout,Wout,Cout].
extern "C" int cn_conv2d_bwd_weight_bf16(const void* x, long ldx, const void* dy, long lddy, float* dw, int B, int Cin,
                                         int Hin, int Win, int Cout, int KH, int KW, int stride, int pad, int dil,
                                         float* ws, long ws_floats, void* stream) {
  if (stride < 1) return CN_ERR_ARG;
  CnBWgGeom g = {};
  g.P = (const bf16_t*)dy; g.ldp = lddy; g.Q = (const bf16_t*)x; g.ldq = ldx;
  g.B = B; g.CP = Cout; g.CQ = Cin; g.s = stride;
  g.Hg = (Hin + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  g.Wg = (Win + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  g.Hq = Hin; g.Wq = Win;
  const int rc = cnw_plan(g, KH, KW, pad, dil);
  if (rc != CN_OK) return rc;
  return cnw_run(g, dw, ws, ws_floats, (hipStream_t)stream);
}

// ConvTranspose2d: dw [Cin][Cout][KH][KW] += ...; x [B,Hin,Win,Cin], dy [B,Hout,Wout,Cout], Hout = (Hin-1)*s - 2p + K.
extern "C" int cn_conv_transpose2d_bwd_weight_bf16(const void* x, long ldx, const void* dy, long lddy, float* dw,
                                                   int B, int Cin, int Hin, int Win, int Cout, int KH, int KW,
                                                   int stride, int pad, float* ws, long ws_floats, void* stream) {
  if (stride < 1) return CN_ERR_ARG;
  CnBWgGeom g = {};
  g.P = (const bf16_t*)x; g.ldp = ldx; g.Q = (const bf16_t*)dy; g.ldq = lddy;
  g.B = B; g.CP = Cin; g.CQ = Cout; g.s = stride;
  g.Hg = Hin; g.Wg = Win;
  g.Hq = (Hin - 1) * stride - 2 * pad + KH; g.Wq = (Win - 1) * stride - 2 * pad + KW;
  const int rc = cnw_plan(g, KH, KW, pad, 1);
  if (rc != CN_OK) return rc;
  return cnw_run(g, dw, ws, ws_floats, (hipStream_t)stream);
}
