// bf16 implicit-GEMM convolution for gfx950 (MI355X): v_mfma_f32_32x32x16_bf16, fp32 accumulation.
//
// Mixed-precision path of BASELINE configs[2] (the reference's default precision="16-mixed",
// /root/reference/src/cultionet/model.py:168-186): activations and activation gradients are bf16 in HBM,
// laid out NHWC ([B][H][W][C], `ld` = elements between consecutive pixels, so a channel slice of a concat buffer is
// consumed / produced in place), parameters stay fp32 masters and are re-packed to bf16 MFMA fragments once per
// optimizer step. One kernel serves nn.Conv2d forward, its backward-data, nn.ConvTranspose2d forward and its
// backward-data (convolution.py:45-120): a launch is a set of CLASSES, each a stride-1 walk over a logical pixel
// grid with its own tap list
//     input pixel = g*is + d[t]        output pixel = g*os + o0
// (a plain conv has one class; a strided scatter has s*s parity classes: no zero-stuffing, no wasted MACs).
//
// Why NHWC for bf16 (the fp32 path is NCHW): a bf16 MFMA operand is 8 CONSECUTIVE k (input channels) per lane. With
// channels innermost a lane's fragment is one ds_read_b128 from a [halo pixel][32 channels] LDS image and a tap is a
// row offset into that image -- no im2col, no transposes, every tap aligned.
//
// Block = 256 threads = 4 waves; tile = TH x TW logical pixels (<= 128, four 32-pixel MFMA columns) x 32*WN output
// channels. A = weights (rows = cout), B = pixels (cols): the accumulator's lane index is the pixel and its registers
// hold 4 consecutive couts x 4 groups, i.e. 8-byte packed bf16 stores straight into NHWC. Wave w owns cout tile
// w % WN and the pixel columns {w / WN + i * (4 / WN)}: its weight fragments come straight from global memory
// (packed in fragment order: one 16-byte load per lane serves WN MFMAs, no cross-wave reuse exists to stage for),
// the pixel image is staged once per 32-channel chunk and shared by the four waves and all taps.
#include <cstdlib>
#include "cn_bf16.h"
#include "cn_profile.h"
#include "cn_ticket.h"

#define CNB_MAX_TAPS 9
#define CNB_MAX_CLASSES 16
#define CNB_MAX_GROUPS 4

struct CnBClass {
  int ntaps;
  int wt[CNB_MAX_TAPS];    // tap index in the packed weights
  int doff[CNB_MAX_TAPS];  // LDS byte offset of the tap inside the halo image
  int oy0, ox0;            // output coordinate offset
  int Hg, Wg;              // logical grid
  int tiles_x, tiles_per_img;
  int block_begin;         // first logical block of this class
  int iy_off, ix_off;      // input coordinate of halo pixel (0,0) relative to (tile origin * is)
  int IH, IW;              // halo image size in pixels
  int grp;
  unsigned mgIW;           // ceil(2^32 / IW): x / IW == umulhi(x, mgIW) for x < 65536 (0 when IW == 1)
  int rowpad;              // bytes added per halo row in LDS (see cnb_rowpad): keeps ds_read_b128 conflict-free
                           // across the tile-row wraps of a 32-pixel column
};

struct CnBGeom {
  const bf16_t* x[CNB_MAX_GROUPS];
  const bf16_t* wp[CNB_MAX_GROUPS];
  const float* bias[CNB_MAX_GROUPS];
  void* y[CNB_MAX_GROUPS];
  long ldx, ldy;
  long y_bs;      // out_kind 1: batch stride of the f32 NCHW output (elements)
  int B, Cin, Hin, Win, Cout, Hout, Wout;
  int is, os;
  int TH, TW;
  unsigned mgTW;  // ceil(2^32 / TW) (0 when TW == 1)
  int KS;         // 16-channel k-steps (ceil(Cin / 16))
  int NT;         // 32-cout tiles (ceil(Cout / 32))
  int nblk_n;     // cout blocks (of 32*WN) per pixel tile
  int out_kind;   // 0: bf16 NHWC   1: f32 NCHW (thin head convolutions)
  int accumulate;
  int act;        // fused eval epilogue: 1 = SiLU applied to conv + bias BEFORE the residual / accumulate add
  const bf16_t* res[CNB_MAX_GROUPS];  // fused eval epilogue: y = res + act(conv + bias) (bf16 NHWC, pixel stride ldres,
  long ldres;                         // same spatial size as y); nullptr: none. res == y is the in-place accumulate.
  int total;      // logical blocks
  int interleave; // != 0: every class has the same number of tiles and logical block = (tile * ncls + class) * nblk_n + nb,
                  // classes in order of descending taps (parity classes of a strided scatter: see cn_conv_geom.h)
  int ncls;
  float* stats[CNB_MAX_GROUPS];  // nullable (all or none): per group and pixel tile {sum, sum of squares}[Cout] of the
                  // fp32 results, rows [tile][2][Cout]
                  // (plain stores, one row per tile: same-address float atomics serialise at ~200 ns each and
                  // 2560 tiles x 4 waves of them made a 128->128 conv at 100x100 nine times slower)
  // In-launch BatchNorm statistics (fin_cnt nullable): the launch FINISHES its per-tile rows itself -- a two-level
  // last-block ticket per (group, cout block): the last of every 16 tiles sums their rows into a group row (fp64, tile
  // order), the last group finisher sums the group rows (group order) and writes mean / rstd / running statistics of
  // its 32 * WN channels. One dependent launch (cn_bbn_group_finalize_kernel, 9 us + its dispatch gap) fewer per
  // BatchNorm; for launches of <= CN_BNWS_CONV_MAX_TILES tiles per group (a last block reads <= 16 + 63 rows).
  int* fin_cnt;       // [G * nblk_n][CN_T2_COUNTERS], zero between launches (head of the grouped-BatchNorm workspace)
  double* fin_grows;  // [G * nblk_n][CN_T2_COUNTERS - 1][64 * WN]
  float* fin_mean[CNB_MAX_GROUPS];
  float* fin_rstd[CNB_MAX_GROUPS];
  float* fin_rmean[CNB_MAX_GROUPS];  // nullable: no running statistics
  float* fin_rvar[CNB_MAX_GROUPS];
  float fin_eps, fin_momentum;
  int fin_tiles;      // tiles of one group
  CnBClass cls[CNB_MAX_CLASSES];
};

// Diagnostic build (-DCNB_STAMP): s_memtime stamps of one wave of two blocks, read back with cn_bconv_read_stamps
// (tools/bconv_stamps.py). Never compiled into the shipped library.
#ifdef CNB_STAMP
__device__ unsigned long long cnb_stamps[2 * 128];
#define CNB_ST(i) do { if (do_stamp) cnb_stamps[stamp_slot * 128 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int cn_bconv_read_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(cnb_stamps), sizeof(unsigned long long) * 256) == hipSuccess ? 0 : -2;
}
#else
#define CNB_ST(i) do { } while (0)
#endif
// Diagnostic build (-DCNB_TRACE): start / end (s_memrealtime, 100 MHz) and HW_ID of EVERY block of the last launch
// (tools/bconv_trace.py: how many blocks a CU really holds over a launch). Never compiled into the shipped library.
#ifdef CNB_TRACE
#define CNB_TRACE_MAX 16384
__device__ unsigned long long cnb_trace[3 * CNB_TRACE_MAX];
extern "C" int cn_bconv_read_trace(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(cnb_trace), sizeof(unsigned long long) * 3 * n) == hipSuccess ? 0 : -2;
}
#define CNB_TRACE_BEGIN() const unsigned long long trace_t0 = __builtin_amdgcn_s_memrealtime()
#define CNB_TRACE_END()                                                                         \
  do {                                                                                          \
    if (threadIdx.x == 0 && L < CNB_TRACE_MAX) {                                                \
      cnb_trace[3 * L] = trace_t0;                                                              \
      cnb_trace[3 * L + 1] = __builtin_amdgcn_s_memrealtime();                                  \
      cnb_trace[3 * L + 2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));       \
    }                                                                                           \
  } while (0)
#else
#define CNB_TRACE_BEGIN() do { } while (0)
#define CNB_TRACE_END() do { } while (0)
#endif

// x / d for 0 <= x < 65536 with mg = ceil(2^32 / d) precomputed on the host (two instructions instead of ~40)
__device__ __forceinline__ int cnb_div(int x, unsigned mg) { return mg ? (int)__umulhi((unsigned)x, mg) : x; }
static inline unsigned cnb_magic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d); }

__device__ __forceinline__ f32x16 cnb_mfma(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// KSC = 16-channel k-steps per staged chunk (2 / 4 / 8 => 32 / 64 / 128 channels per LDS image): fewer, longer
// chunks mean fewer barrier + staging phases per MFMA (a 1x1 conv has only one tap per chunk to amortise them over).
template <int KSC>
struct CnbPitch { static constexpr int value = KSC * 32 + 16; };  // bytes per halo pixel: channels + 16 B pad
                                                                  // (5, 9, 17 sixteen-byte slots: coprime with 16)

// Three blocks per CU (<= 168 VGPRs): measured against the two-blocks variant that double-buffered the pixel
// fragments and register-prefetched the staging (209-237 VGPRs), thread-level parallelism wins on every 3x3 shape
// (128->128 @100^2: 767 -> 842 TFLOP/s, @50^2: 531 -> 647): while one block stages or stores, two others multiply.
//
// Round 5, built, parity-green and REMOVED: a "wide wave" variant (two 32-cout tiles per wave: block = 2 cout waves x 2
// pixel waves, 256 pixels x 128 couts, 128 accumulator registers, two blocks per CU) on the premise that the step loop
// is LDS-bound (four waves reading the same pixel fragments: 128 B per cycle at the full matrix rate) -- a fragment read
// once would feed twice the MFMAs. In-kernel stamps (tools/bconv_stamps.py, 32 x 128 -> 128 x 100^2) say otherwise: the
// loop ran at 26.9 ticks per MFMA and SIMD against 24.2 for this kernel (halving the LDS bytes per MFMA bought
// nothing: the loop is not LDS-bound), while the staging of a 256-pixel halo in register-bounded phases cost 32 % of
// a block's life (13 % here) with only two blocks per CU to hide it: 119 us against 104-107 us per launch, bf16 step
// 2089 against 2140 chips/s. (Its first build also kept all 128 accumulators in SCRATCH: an un-unrolled loop over the
// wave's cout tiles in the epilogue left acc[u] dynamically indexed -- 349 us.) profiles/r05_bconv_wide_stamps.txt.
template <int WN, int NP, int KSC, int MPW_ = 4>
__global__ __launch_bounds__(256, 3) void cn_bconv_kernel(const CnBGeom g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int WM = 4 / WN;   // waves along the pixel columns
  constexpr int MPW = MPW_;    // 32-pixel columns per wave: every weight fragment (one 16-byte global load per
                               // lane) feeds MPW MFMAs; the block tile is 32 * MPW * WM pixels x 32 * WN couts.
                               // MPW = 4 everywhere but on small planes: (WN, MPW) = (2, 2) keeps the 128-pixel tile
                               // (5 x 25 covers 25 / 50 / 100 exactly) and halves the couts per block, i.e. doubles
                               // the blocks of a launch that would leave half of the 256 CUs idle
  constexpr int PITCH = CnbPitch<KSC>::value;
  constexpr int PPP = KSC * 2;  // 16-byte pieces per pixel
  constexpr int KPAIRS = KSC / 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: keeps the weight-fragment
                                                             // offsets in SGPRs (no waterfall loops around buffer loads)
  const int r = lane & 31, h = lane >> 5;
  const int wn = wid % WN, wm = wid / WN;

  // ---- logical block (XCD-aware order: consecutive logical blocks share an L2) ----
  // (A persistent variant -- 768 resident blocks walking the tile list -- was measured: no gain from skipping the
  // workgroup relaunch, and the loop-carried state pushed the kernel over its 168-VGPR budget into spills.)
  int L;
  {
    const int lin = blockIdx.x;
    const int per = (g.total + 7) >> 3;
    L = (lin & 7) * per + (lin >> 3);
    if (L >= g.total) return;
  }
#ifdef CNB_STAMP
  const bool do_stamp = (L == 0 || L == g.total / 2) && tid == 0;
  const int stamp_slot = L == 0 ? 0 : 1;
  int stamp_i = 8;
#endif
  CNB_ST(0);
  CNB_TRACE_BEGIN();
  int ci = 0;
  if (g.interleave) {
    ci = (L / g.nblk_n) % g.ncls;
  } else {
#pragma unroll 1
    for (int c = 1; c < g.ncls; ++c)
      if (L >= g.cls[c].block_begin) ci = c;
  }
  const CnBClass& k = g.cls[ci];
  const int grp = k.grp;
  const int local = g.interleave ? L : L - k.block_begin;
  const int nb = local % g.nblk_n;
  const int tile = g.interleave ? (local / g.nblk_n) / g.ncls : local / g.nblk_n;
  const int b = tile / k.tiles_per_img;
  const int tl = tile - b * k.tiles_per_img;
  const int tyi = tl / k.tiles_x, txi = tl - tyi * k.tiles_x;
  const int gy0 = tyi * g.TH, gx0 = txi * g.TW;  // logical tile origin
  const int ntile = nb * WN + wn;                // this wave's 32-cout tile
  const bool n_live = ntile < g.NT;
  const int IW = k.IW;
  const int npix = g.TH * g.TW;

  // ---- staging descriptors: piece q = 16 bytes (8 channels) of one halo pixel ----
  const bf16_t* __restrict__ xg = g.x[grp];
  int gpix[NP];  // input pixel index (b, iy, ix) of piece tid + i*256, -1 outside the image / past the image
  const int npieces = k.IH * IW * PPP;
  const int iy0 = gy0 * g.is + k.iy_off, ix0 = gx0 * g.is + k.ix_off;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int q = tid + i * 256;
    const int p = q / PPP;
    const int hy = cnb_div(p, k.mgIW), hx = p - hy * IW;
    const int iy = iy0 + hy, ix = ix0 + hx;
    const bool ok = q < npieces && iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Win;
    gpix[i] = ok ? (b * g.Hin + iy) * g.Win + ix : -1;
  }
  const int s8 = (tid % PPP) * 8;  // channel offset of this thread's pieces inside a chunk (256 % PPP == 0)

  // ---- this wave's pixel columns: LDS base of each lane's pixel ----
  int pbase[MPW];
#pragma unroll
  for (int i = 0; i < MPW; ++i) {
    const int m = (wm + i * WM) * 32 + r;
    const int ty = cnb_div(m, g.mgTW), tx = m - ty * g.TW;
    pbase[i] = m < npix ? ((ty * g.is) * IW + tx * g.is) * PITCH + (NP <= 6 ? (ty * g.is) * k.rowpad : 0) + h * 16 : h * 16;
  }

  f32x16 acc[MPW];
#pragma unroll
  for (int i = 0; i < MPW; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

  const int KS = g.KS, NT = g.NT;
  const int nchunks = (KS + KSC - 1) / KSC;
  const int ntaps = k.ntaps;
  // tap tables live in one VGPR each (lane t holds tap t) and are fetched with v_readlane: a scalar load inside the
  // tap loop would need s_waitcnt lgkmcnt(0), which also drains the LDS reads deliberately kept in flight
  const int doff_v = lane < CNB_MAX_TAPS ? k.doff[lane < CNB_MAX_TAPS ? lane : 0] : 0;
  const int wt_v = lane < CNB_MAX_TAPS ? k.wt[lane < CNB_MAX_TAPS ? lane : 0] : 0;
  const int spc = ntaps * KPAIRS;     // steps per chunk: (tap, k-pair)
  const int nsteps = nchunks * spc;

  CNB_ST(1);
  if (ntaps > 0) {
    // Weight fragments: buffer loads with the whole fragment address in SGPRs (one VMEM instruction, no per-lane
    // address arithmetic): byte offset of (tap, kstep) = ((wt * KS + kstep) * NT + ntile) * 1024, + lane * 16.
    // Out-of-range k-steps (Cin not a multiple of the chunk) and dead cout tiles are CLAMPED to valid fragments
    // instead of masked: the pixel image holds zeros for channels >= Cin, and dead tiles never store.
    const unsigned long long wbase = reinterpret_cast<unsigned long long>(g.wp[grp]);
    const unsigned wlo = __builtin_amdgcn_readfirstlane((unsigned)wbase);
    const unsigned whi = __builtin_amdgcn_readfirstlane((unsigned)(wbase >> 32));
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<bf16_t*>(((unsigned long long)whi << 32) | wlo), 0, 0x7fffffff, 0x00020000);
    const int lane16 = lane * 16;
    const int nt_c = n_live ? ntile : NT - 1;
    // step (ch, t, kp): k-steps ch*KSC + 2*kp and + 1
    auto wload = [&](int ch, int t, int kp, int half) -> bf16x8 {
      int ks = ch * KSC + kp * 2 + half;
      ks = ks < KS ? ks : KS - 1;
      const int wt = __builtin_amdgcn_readlane(wt_v, t);
      const int soff = ((wt * KS + ks) * NT + nt_c) * 1024;
      // Inline asm on purpose: hipcc waits vmcnt(0) for compiler-visible loads carried around a loop, i.e. for the
      // fragments issued a few instructions earlier (a full L2 round trip per 8 MFMAs). These loads are waited for by
      // hand with a COUNTED s_waitcnt (CNB_WAIT_A) that leaves the other register set's two loads in flight.
      u32x4 v;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(lane16), "s"(wrsrc), "s"(soff));
      return __builtin_bit_cast(bf16x8, v);
    };
    // The loop walks the linear step sequence (chunk, tap, k-pair), two steps per iteration with static register
    // names for the two weight-fragment sets (each re-loaded for step s+2 right after its MFMAs have issued). The 8
    // pixel fragments of a step are read from LDS in one batch (8 distinct register quads, so the 8 MFMAs wait with
    // counted lgkmcnt instead of read, wait, MFMA, read, wait, ... through one quad as the compiler schedules it).
    bf16x8 X[2 * MPW];
    bf16x8 ax0, ax1, ay0, ay1;
    int ch = 0, t = 0, kp = 0;      // current step
    int ch2 = 0, t2 = 0, kp2 = 0;   // two steps ahead (weight prefetch)
    auto adv = [&](int& c_, int& t_, int& k_) {
      if (KPAIRS > 1) {
        if (++k_ < KPAIRS) return;
        k_ = 0;
      }
      if (++t_ == ntaps) { t_ = 0; ++c_; }
    };
    ax0 = wload(0, 0, 0, 0); ax1 = wload(0, 0, 0, 1);
    adv(ch2, t2, kp2);
    {
      const int cq = ch2 < nchunks ? ch2 : nchunks - 1;
      ay0 = wload(cq, t2, kp2, 0); ay1 = wload(cq, t2, kp2, 1);
    }
    adv(ch2, t2, kp2);

    // chunk staging: global -> registers -> padded LDS image, in at most two phases of five 16-byte pieces per thread
    // (bounds the registers in flight); the other two blocks of the CU multiply meanwhile
#define CNB_PHASE(I0, I1)                                                                               \
  {                                                                                                     \
    u32x4 sv[(I1) - (I0) > 0 ? (I1) - (I0) : 1];                                                        \
    _Pragma("unroll") for (int i = (I0); i < (I1); ++i) {                                               \
      u32x4 v = {0u, 0u, 0u, 0u};                                                                       \
      if (gpix[i] >= 0 && cc + s8 < g.Cin)                                                              \
        v = *reinterpret_cast<const u32x4*>(xg + (long)gpix[i] * g.ldx + s8 + cc);                      \
      sv[i - (I0)] = v;                                                                                 \
    }                                                                                                   \
    if ((I0) == 0) __syncthreads(); /* the previous chunk's reads are done */                           \
    _Pragma("unroll") for (int i = (I0); i < (I1); ++i) {                                               \
      const int q = tid + i * 256;                                                                      \
      if (q < npieces) {                                                                                \
        const int p_ = q / PPP;                                                                         \
        *reinterpret_cast<u32x4*>(lds + p_ * PITCH + (NP <= 6 ? cnb_div(p_, k.mgIW) * k.rowpad : 0) + (q % PPP) * 16) = sv[i - (I0)]; \
      }                                                                                                 \
    }                                                                                                   \
  }
#define CNB_STAGE()                                                                                     \
  {                                                                                                     \
    const int cc = ch * (KSC * 16);                                                                     \
    constexpr int H1 = NP > 6 ? 5 : NP; /* NP <= 6: one phase (one exposed global latency per chunk instead of two) */                                                                 \
    CNB_ST(100 + ch * 4);                                                                               \
    CNB_PHASE(0, H1);                                                                                   \
    CNB_ST(101 + ch * 4);                                                                               \
    if (NP > H1) CNB_PHASE(H1, NP);                                                                     \
    CNB_ST(102 + ch * 4);                                                                               \
    /* A piece past the halo image is loaded (masked) but never stored, so the compiler's scoreboard leaves the  \
       chunk with that load "pending" into a register the step loop re-uses for pixel fragments -- and answers \
       with s_waitcnt vmcnt(0) in front of every tap's ds_reads, draining the hand-counted weight prefetch too. \
       An explicit vmcnt(0) here (once per chunk) clears the scoreboard. */                                     \
    __builtin_amdgcn_s_waitcnt(0x0F70);                                                                 \
    __syncthreads();                                                                                    \
    CNB_ST(103 + ch * 4);                                                                               \
  }
#define CNB_READ(BUF, T_, KP_)                                                                     \
  {                                                                                                \
    const int toff_ = __builtin_amdgcn_readlane(doff_v, T_) + (KP_) * 64;                          \
    _Pragma("unroll") for (int i = 0; i < MPW; ++i) {                                              \
      BUF[i] = *reinterpret_cast<const bf16x8*>(lds + pbase[i] + toff_);                           \
      BUF[MPW + i] = *reinterpret_cast<const bf16x8*>(lds + pbase[i] + toff_ + 32);                \
    }                                                                                              \
  }
#define CNB_HALF(CUR, OTH, A0, A1)                                                                 \
  {                                                                                                \
    if (t == 0 && kp == 0) CNB_STAGE();                                                            \
    CNB_READ(CUR, t, kp);                                                                          \
    /* weight fragments of this step: everything but the two newest loads (the other set's) has landed */ \
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                               \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    _Pragma("unroll") for (int i = 0; i < MPW; ++i) acc[i] = cnb_mfma(A0, CUR[i], acc[i]);         \
    _Pragma("unroll") for (int i = 0; i < MPW; ++i) acc[i] = cnb_mfma(A1, CUR[MPW + i], acc[i]);   \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    {                                                                                              \
      const int cq = ch2 < nchunks ? ch2 : nchunks - 1;                                            \
      A0 = wload(cq, t2, kp2, 0);                                                                  \
      A1 = wload(cq, t2, kp2, 1);                                                                  \
    }                                                                                              \
    adv(ch2, t2, kp2);                                                                             \
    adv(ch, t, kp);                                                                                \
    CNB_STEP_STAMP();                                                                              \
  }
#ifdef CNB_STAMP
#define CNB_STEP_STAMP() do { if (stamp_i < 96) { CNB_ST(stamp_i); ++stamp_i; } } while (0)
#else
#define CNB_STEP_STAMP() do { } while (0)
#endif
    CNB_ST(2);
    if constexpr (KPAIRS >= 2) {
      // Nested (chunk, tap, k-pair) loops with the k-pairs unrolled: the register set of a step is its k-pair's parity
      // (compile time), the tap's LDS offset is fetched once per tap, the prefetch target two steps ahead is "same tap,
      // k-pair + 2" or "next tap, k-pair + 2 - KPAIRS" with the next (tap, chunk) worked out once per tap, and a weight
      // fragment's scalar offset is one v_readlane of a per-tap base table + k-step * stride. The linear-sequence form
      // below spent ~45 scalar instructions per step of 8 MFMAs on this bookkeeping (two (chunk, tap, k-pair) counters
      // advanced with branches, five scalar multiply-adds per fragment): PMC 8 SALU per MFMA, the CU's one scalar unit
      // as busy as the matrix pipes.
      const int NTK = NT * 1024;  // bytes between consecutive k-steps of one tap
      const int wtb_v = lane < CNB_MAX_TAPS ? (k.wt[lane < CNB_MAX_TAPS ? lane : 0] * KS * NT + nt_c) * 1024 : 0;
      auto wl = [&](int chq, int tq, int kpq, int half) -> bf16x8 {
        int ks = chq * KSC + kpq * 2 + half;
        ks = ks < KS ? ks : KS - 1;
        const int soff = __builtin_amdgcn_readlane(wtb_v, tq) + ks * NTK;
        u32x4 v;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(lane16), "s"(wrsrc), "s"(soff));
        return __builtin_bit_cast(bf16x8, v);
      };
      // (the prologue above loaded step 0 into ax and step 1 into ay: (0, 0, 0) and (0, 0, 1) for KPAIRS >= 2)
#pragma unroll 1
      for (ch = 0; ch < nchunks; ++ch) {
        CNB_STAGE();
#pragma unroll 1
        for (t = 0; t < ntaps; ++t) {
          int tn = t + 1, chn = ch;
          if (tn == ntaps) {
            tn = 0;
            chn = ch + 1 < nchunks ? ch + 1 : ch;  // past the end: a clamped, never consumed prefetch
          }
          const int toff_t = __builtin_amdgcn_readlane(doff_v, t);
#pragma unroll
          for (int kq = 0; kq < KPAIRS; ++kq) {
#pragma unroll
            for (int i = 0; i < MPW; ++i) {
              X[i] = *reinterpret_cast<const bf16x8*>(lds + pbase[i] + toff_t + kq * 64);
              X[MPW + i] = *reinterpret_cast<const bf16x8*>(lds + pbase[i] + toff_t + kq * 64 + 32);
            }
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if ((kq & 1) == 0) {
#pragma unroll
              for (int i = 0; i < MPW; ++i) acc[i] = cnb_mfma(ax0, X[i], acc[i]);
#pragma unroll
              for (int i = 0; i < MPW; ++i) acc[i] = cnb_mfma(ax1, X[MPW + i], acc[i]);
            } else {
#pragma unroll
              for (int i = 0; i < MPW; ++i) acc[i] = cnb_mfma(ay0, X[i], acc[i]);
#pragma unroll
              for (int i = 0; i < MPW; ++i) acc[i] = cnb_mfma(ay1, X[MPW + i], acc[i]);
            }
            __builtin_amdgcn_sched_barrier(0);
            const bool same = kq + 2 < KPAIRS;
            const int cq = same ? ch : chn, tq = same ? t : tn, kpq = same ? kq + 2 : kq + 2 - KPAIRS;
            if ((kq & 1) == 0) {
              ax0 = wl(cq, tq, kpq, 0);
              ax1 = wl(cq, tq, kpq, 1);
            } else {
              ay0 = wl(cq, tq, kpq, 0);
              ay1 = wl(cq, tq, kpq, 1);
            }
            CNB_STEP_STAMP();
          }
        }
      }
    } else {
#pragma unroll 1
      for (int s = 0; s < nsteps; s += 2) {
        CNB_HALF(X, X, ax0, ax1);
        if (s + 1 < nsteps) CNB_HALF(X, X, ay0, ay1);
      }
    }
    // The last two steps prefetched (clamped) fragments nobody consumes. The compiler does not know that an inline-asm
    // load writes its destination LATER: it is free to reuse those registers at once -- e.g. for the epilogue's store
    // addresses -- and a late-landing fragment then corrupts a pointer (seen as a GPU memory access fault, only when a
    // concurrent stream delayed the loads). Drain them before any register is recycled.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef CNB_PHASE
#undef CNB_STAGE
#undef CNB_READ
#undef CNB_HALF
  }

  // ---- epilogue: lane = pixel, registers = 4 groups of 4 consecutive couts ----
  CNB_ST(3);
  const float* __restrict__ bias = g.bias[grp];
  const int n0 = ntile * 32;
  float bsum[16];
  if (bias != nullptr) {  // wave-uniform: ConvBlock2d is bias-free (BatchNorm follows), skip the 16 predicated loads
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int n = n0 + (j & 3) + 8 * (j >> 2) + 4 * h;
      bsum[j] = (n_live && n < g.Cout) ? bias[n] : 0.f;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 16; ++j) bsum[j] = 0.f;
  }
  float s1[16], s2[16];
  float* const stats_g = g.stats[grp];
  if (stats_g != nullptr) {
#pragma unroll
    for (int j = 0; j < 16; ++j) s1[j] = s2[j] = 0.f;
  }
  // One loop per output mode, the mode tested OUTSIDE the pixel-group loop: with the tests inside, every group began at
  // a join of the read-modify-write paths and the compiler's waitcnt scoreboard answered with s_waitcnt vmcnt(0) -- each
  // group waited for the previous group's stores to reach memory (~1.2k cycles per group, 13 % of a tile's life).
#define CNB_GROUP_HEAD(i)                                                                      \
  CNB_ST(96 + (i));                                                                            \
  const int m = (wm + (i) * WM) * 32 + r;                                                      \
  const int ty = cnb_div(m, g.mgTW), tx = m - ty * g.TW;                                       \
  const int gy = gy0 + ty, gx = gx0 + tx;                                                      \
  const bool ok = n_live && m < npix && gy < k.Hg && gx < k.Wg;                                \
  const int oy = gy * g.os + k.oy0, ox = gx * g.os + k.ox0;                                    \
  if (stats_g != nullptr) {                                                                    \
    _Pragma("unroll") for (int j = 0; j < 16; ++j) {                                           \
      const float v = ok ? acc[i][j] + bsum[j] : 0.f;                                          \
      s1[j] += v;                                                                              \
      s2[j] += v * v;                                                                          \
    }                                                                                          \
  }
  if (g.out_kind == 0 && (g.Cout & 7) == 0) {
    // A pixel's 32 couts are split over its two half-wave lanes in 4-cout groups (lane i: 8q..8q+3, lane i+32:
    // 8q+4..8q+7). v_permlane32_swap trades group q of the upper half for group q+1 of the lower half, so every
    // lane ends up with 8 CONSECUTIVE couts: two 16-byte stores per pixel column instead of four 8-byte ones.
    const bool accum = g.accumulate != 0;
    const bf16_t* __restrict__ resb = g.res[grp];
    if (g.act != 0) {  // fused eval epilogue (BatchNorm folded into weights / bias): SiLU before the residual add
#pragma unroll
      for (int i = 0; i < MPW; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const float v = acc[i][j] + bsum[j];
          acc[i][j] = v / (1.0f + __expf(-v));
        }
#pragma unroll
      for (int j = 0; j < 16; ++j) bsum[j] = 0.f;  // the bias is inside the activation now
    }
#pragma unroll
    for (int i = 0; i < MPW; ++i) {
      CNB_GROUP_HEAD(i)
      const long opix = ((long)b * g.Hout + oy) * g.Wout + ox;
      bf16_t* yp = reinterpret_cast<bf16_t*>(g.y[grp]) + opix * g.ldy + n0;
#pragma unroll
      for (int q = 0; q < 4; q += 2) {
        unsigned a0 = cn_pack_bf16(acc[i][4 * q] + bsum[4 * q], acc[i][4 * q + 1] + bsum[4 * q + 1]);
        unsigned a1 = cn_pack_bf16(acc[i][4 * q + 2] + bsum[4 * q + 2], acc[i][4 * q + 3] + bsum[4 * q + 3]);
        unsigned b0 = cn_pack_bf16(acc[i][4 * q + 4] + bsum[4 * q + 4], acc[i][4 * q + 5] + bsum[4 * q + 5]);
        unsigned b1 = cn_pack_bf16(acc[i][4 * q + 6] + bsum[4 * q + 6], acc[i][4 * q + 7] + bsum[4 * q + 7]);
        const auto sw0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto sw1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        const int n = n0 + 8 * (q + h);
        if (!ok || n >= g.Cout) continue;
        u32x4* dst = reinterpret_cast<u32x4*>(yp + 8 * (q + h));
        u32x4 pk = {sw0[0], sw1[0], sw0[1], sw1[1]};
        if (accum || resb != nullptr) {
          float ov[8], nv[8];
          if (resb != nullptr) cn_unpack8(*reinterpret_cast<const u32x4*>(resb + opix * g.ldres + n0 + 8 * (q + h)), ov);
          else cn_unpack8(*dst, ov);
          cn_unpack8(pk, nv);
#pragma unroll
          for (int e = 0; e < 8; ++e) nv[e] += ov[e];
          pk = cn_pack8(nv);
        }
        *dst = pk;
      }
    }
  } else if (g.out_kind == 0) {
#pragma unroll
    for (int i = 0; i < MPW; ++i) {
      CNB_GROUP_HEAD(i)
      if (!ok) continue;
      bf16_t* yp = reinterpret_cast<bf16_t*>(g.y[grp]) + (((long)b * g.Hout + oy) * g.Wout + ox) * g.ldy + n0 + 4 * h;
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // ragged couts: element-wise
        const int n = n0 + 8 * q + 4 * h;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n + e >= g.Cout) continue;
          float v = acc[i][4 * q + e] + bsum[4 * q + e];
          if (g.accumulate) v += cn_bf16_to_f32(yp[8 * q + e]);
          yp[8 * q + e] = cn_f32_to_bf16(v);
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < MPW; ++i) {
      CNB_GROUP_HEAD(i)
      if (!ok) continue;
      float* yp = reinterpret_cast<float*>(g.y[grp]) + (long)b * g.y_bs + (long)oy * g.Wout + ox;
      const long cs = (long)g.Hout * g.Wout;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int n = n0 + (j & 3) + 8 * (j >> 2) + 4 * h;
        if (n >= g.Cout) continue;
        const float v = acc[i][j] + bsum[j];
        if (g.accumulate) yp[n * cs] += v; else yp[n * cs] = v;
      }
    }
  }
#undef CNB_GROUP_HEAD
  CNB_ST(4);
  if (stats_g != nullptr) {
    // Per-cout sums over this wave's pixels = sums over the LANES of 32 values per lane: transposed through LDS
    // (each lane writes its 32 partials as one padded row, then sums ONE column over its half-wave's 32 rows): ~100
    // instructions per wave instead of 160 cross-lane shuffles + 160 adds.
    __syncthreads();  // every wave is done with the pixel image
    float* red = reinterpret_cast<float*>(lds) + wid * (64 * 33);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      red[lane * 33 + j] = s1[j];
      red[lane * 33 + 16 + j] = s2[j];
    }
    const int jj = lane & 31, hh = lane >> 5;
    float tot = 0.f;
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) tot += red[(hh * 32 + rr) * 33 + jj];
    if (WM > 1) {
      // WM waves share a cout tile (each saw a part of the tile's pixels): combine them in LDS in a FIXED order -- no
      // float atomics (run-to-run identical sums), no zero-fill of the rows by the launcher (a hipMemsetAsync in front
      // of every such launch sat on the critical path with ~10-30 us of dispatch latency)
      __syncthreads();  // every wave has read its transposed partials
      float* comb = reinterpret_cast<float*>(lds);
      comb[wid * 64 + lane] = tot;
      __syncthreads();
      tot = 0.f;
#pragma unroll
      for (int m = 0; m < WM; ++m) tot += comb[(m * WN + wn) * 64 + lane];
    }
    const int j = jj & 15;
    const int n = n0 + (j & 3) + 8 * (j >> 2) + 4 * hh;
    if (n_live && n < g.Cout && wm == 0) {  // wid = wm * WN + wn: the wm == 0 wave of each cout tile stores the row
      float* row = stats_g + (long)tile * 2 * g.Cout + (jj >= 16 ? g.Cout : 0) + n;
      if (g.fin_cnt != nullptr) __hip_atomic_store(row, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else *row = tot;
    }
    if (g.fin_cnt != nullptr) {  // block-uniform
      constexpr int W = 64 * WN;  // this block's columns: {sum, sum of squares} x its 32 * WN couts
      const int dom = grp * g.nblk_n + nb;
      int* cnt = g.fin_cnt + dom * CN_T2_COUNTERS;
      double* grows = g.fin_grows + (long)dom * (CN_T2_COUNTERS - 1) * W;
      const int ntl = g.fin_tiles, ngr = (ntl + CN_T2_GROUP - 1) / CN_T2_GROUP;
      const int grpi = tile / CN_T2_GROUP, t0 = grpi * CN_T2_GROUP;
      const int gn = t0 + CN_T2_GROUP <= ntl ? CN_T2_GROUP : ntl - t0;
      int* s_flag = reinterpret_cast<int*>(lds);
      double* tot2 = reinterpret_cast<double*>(lds + 16);
      const int stat = tid / (32 * WN), cc = tid - stat * (32 * WN);
      const int nc = nb * (32 * WN) + cc;
      const long coff = (long)stat * g.Cout + (nc < g.Cout ? nc : g.Cout - 1);  // dead columns re-read the last channel
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // the row is on its way to memory; the transposition scratch is dead
      if (tid == 0) {
        const int kq = __hip_atomic_fetch_add(cnt + grpi, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_flag = (kq == gn - 1);
        if (kq == gn - 1) __hip_atomic_store(cnt + grpi, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      if (*s_flag) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (tid < W) {
          float v[CN_T2_GROUP];
#pragma unroll
          for (int i = 0; i < CN_T2_GROUP; ++i)  // clamped, never predicated: all 16 loads in flight together
            v[i] = __hip_atomic_load(stats_g + (long)(t0 + (i < gn ? i : gn - 1)) * 2 * g.Cout + coff, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
          double sd = 0.0;
#pragma unroll
          for (int i = 0; i < CN_T2_GROUP; ++i) sd += i < gn ? (double)v[i] : 0.0;
          __hip_atomic_store(grows + (long)grpi * W + tid, sd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
          const int kq = __hip_atomic_fetch_add(cnt + CN_T2_COUNTERS - 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          *s_flag = (kq == ngr - 1);
          if (kq == ngr - 1) __hip_atomic_store(cnt + CN_T2_COUNTERS - 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (*s_flag) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          if (tid < W) {
            double sd = 0.0;
            for (int k0 = 0; k0 < ngr; k0 += CN_T2_GROUP) {
              double v[CN_T2_GROUP];
#pragma unroll
              for (int i = 0; i < CN_T2_GROUP; ++i)
                v[i] = __hip_atomic_load(grows + (long)(k0 + i < ngr ? k0 + i : ngr - 1) * W + tid, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
              for (int i = 0; i < CN_T2_GROUP; ++i) sd += k0 + i < ngr ? v[i] : 0.0;
            }
            tot2[tid] = sd;
          }
          __syncthreads();
          if (tid < 32 * WN && nc < g.Cout) {  // (stat == 0 here: nc is this thread's channel)
            const double count = (double)g.B * g.Hout * g.Wout;
            const double md = tot2[tid] / count;
            double var = tot2[32 * WN + tid] / count - md * md;
            if (var < 0.0) var = 0.0;
            g.fin_mean[grp][nc] = (float)md;
            g.fin_rstd[grp][nc] = (float)(1.0 / sqrt(var + (double)g.fin_eps));
            if (g.fin_rmean[grp] != nullptr) {
              const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
              g.fin_rmean[grp][nc] = (1.f - g.fin_momentum) * g.fin_rmean[grp][nc] + g.fin_momentum * (float)md;
              g.fin_rvar[grp][nc] = (1.f - g.fin_momentum) * g.fin_rvar[grp][nc] + g.fin_momentum * (float)unbiased;
            }
          }
        }
      }
    }
  }
  CNB_ST(5);
  CNB_TRACE_END();
}

// ------------------------------------------------------------------------------------------------------------
// packed weights: fragment order [tap][kstep][ntile][lane][8], element j of lane l = w[k = kstep*16 + 8*(l>>5) + j]
// [n = ntile*32 + (l&31)] of tap t (zero beyond K / N)
// ------------------------------------------------------------------------------------------------------------
struct CnBPackDesc {
  const float* w;
  bf16_t* wp;
  int T, K, N, KS, NT;
  int pad;
  long sk, sn, st;
  const float* nscale;  // nullable: per-n (cout) factor folded into the packed copy (eval-mode BatchNorm)
};

// The batched per-step repack (fp32 masters -> bf16 MFMA fragments, every layer in one launch). A block stages a tile
// of 32 (slow source dimension) x 32 (mid dimension) x T floats through LDS, so the reads are runs of 32 * T
// contiguous floats per source row and the writes whole 1 KB fragment sets; the per-fragment strided gather this
// replaces ran at ~1 TB/s effective: 0.53 ms per step at hidden 64 (5 % of the reference-default step), 0.11 ms at 32.
#define CNB_PK_PITCH (32 * CNB_MAX_TAPS + 1)
__global__ __launch_bounds__(256) void cn_bpack_kernel(const CnBPackDesc* __restrict__ descs) {
  __shared__ float tile[32 * CNB_PK_PITCH];
  const CnBPackDesc d = descs[blockIdx.y];
  const bool n_slow = d.sn > d.sk;  // which of (k, n) is the slowest source dimension
  const long ss = n_slow ? d.sn : d.sk, sm = n_slow ? d.sk : d.sn;
  if (d.T > CNB_MAX_TAPS || d.st != 1 || sm != d.T || d.nscale != nullptr) {  // generic strides: plain gather
    const long total = (long)d.T * d.KS * d.NT * 64;  // 16-byte fragments
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
      const int l = (int)(i & 63);
      long q = i >> 6;
      const int nt = (int)(q % d.NT);
      q /= d.NT;
      const int ks = (int)(q % d.KS);
      const int t = (int)(q / d.KS);
      const int n = nt * 32 + (l & 31);
      const int k0 = ks * 16 + 8 * (l >> 5);
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[j] = (n < d.N && k0 + j < d.K) ? d.w[(k0 + j) * d.sk + n * d.sn + t * d.st] : 0.f;
      if (d.nscale != nullptr && n < d.N) {
        const float sc = d.nscale[n];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= sc;
      }
      u32x4 o = {cn_pack_bf16(v[0], v[1]), cn_pack_bf16(v[2], v[3]), cn_pack_bf16(v[4], v[5]), cn_pack_bf16(v[6], v[7])};
      reinterpret_cast<u32x4*>(d.wp)[i] = o;
    }
    return;
  }
  const int T = d.T;
  const int S = n_slow ? d.N : d.K, M = n_slow ? d.K : d.N;
  const int k32 = (d.KS + 1) >> 1;            // tiles of 32 along k (two 16-deep k-steps each)
  const int ts = n_slow ? d.NT : k32, tm = n_slow ? k32 : d.NT;
  const int run = 32 * T;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int tl = blockIdx.x; tl < ts * tm; tl += gridDim.x) {
    const int s0 = (tl / tm) * 32, m0 = (tl % tm) * 32;
    __syncthreads();
    // one wave per source row (a contiguous run of 32 * T <= 288 floats), 8 rows per wave: all 8 x 5 loads of a
    // thread are issued before the first LDS store (clamped, never predicated -- as a loop of load -> store pairs the
    // tile took 36 memory latencies: 258 us per step at hidden 64)
    {
      constexpr int NJ = (32 * CNB_MAX_TAPS + 63) / 64;
      float v[8][NJ];
      const long mlim = (long)M * T;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int sr = wave + 4 * i;
        const long rowo = (long)(s0 + sr < S ? s0 + sr : S - 1) * ss;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
          const long mi = (long)m0 * T + lane + 64 * jj;
          v[i][jj] = d.w[rowo + (mi < mlim ? mi : mlim - 1)];
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int sr = wave + 4 * i;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
          const int j = lane + 64 * jj;
          const long mi = (long)m0 * T + j;
          if (j < run) tile[sr * CNB_PK_PITCH + j] = (s0 + sr < S && mi < mlim) ? v[i][jj] : 0.f;
        }
      }
    }
    __syncthreads();
    const int k0 = n_slow ? m0 : s0, n0 = n_slow ? s0 : m0;
    const int nt = n0 >> 5;
    // fragment sets (tap t, k-step half h) of the tile: lane l holds k = k0 + 16 h + 8 (l >> 5) + j, n = n0 + (l & 31)
    for (int f = wave; f < 2 * T; f += 4) {
      const int t = f >> 1, h = f & 1;
      const int ks = (k0 >> 4) + h;
      if (ks >= d.KS) continue;
      const int kl = 16 * h + 8 * (lane >> 5), nl = lane & 31;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[j] = n_slow ? tile[nl * CNB_PK_PITCH + (kl + j) * T + t] : tile[(kl + j) * CNB_PK_PITCH + nl * T + t];
      u32x4 o = {cn_pack_bf16(v[0], v[1]), cn_pack_bf16(v[2], v[3]), cn_pack_bf16(v[4], v[5]), cn_pack_bf16(v[6], v[7])};
      reinterpret_cast<u32x4*>(d.wp)[(((long)t * d.KS + ks) * d.NT + nt) * 64 + lane] = o;
    }
  }
}

extern "C" long cn_bconv_packed_elems(int T, int K, int N) {
  return (long)T * ((K + 15) / 16) * ((N + 31) / 32) * 512;
}

// descs: DEVICE array of n 72-byte records {const float* w; bf16* wp; int T, K, N, KS, NT, pad; long sk, sn, st;
// const float* nscale (nullable)}
extern "C" int cn_pack_weights_batched_bf16(const void* descs, int n, void* stream) {
  if (n <= 0) return CN_OK;
  CN_LAUNCH(cn_bpack_kernel, dim3(64, n), dim3(256), 0, (hipStream_t)stream, (const CnBPackDesc*)descs);
  return cn_check_launch();
}

__global__ __launch_bounds__(256) void cn_bpack_one_kernel(const CnBPackDesc d) {
  const long total = (long)d.T * d.KS * d.NT * 64;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int l = (int)(i & 63);
    long q = i >> 6;
    const int nt = (int)(q % d.NT);
    q /= d.NT;
    const int ks = (int)(q % d.KS);
    const int t = (int)(q / d.KS);
    const int n = nt * 32 + (l & 31);
    const int k0 = ks * 16 + 8 * (l >> 5);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      v[j] = (n < d.N && k0 + j < d.K) ? d.w[(k0 + j) * d.sk + n * d.sn + t * d.st] : 0.f;
    if (d.nscale != nullptr && n < d.N) {
      const float sc = d.nscale[n];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= sc;
    }
    u32x4 o = {cn_pack_bf16(v[0], v[1]), cn_pack_bf16(v[2], v[3]), cn_pack_bf16(v[4], v[5]), cn_pack_bf16(v[6], v[7])};
    reinterpret_cast<u32x4*>(d.wp)[i] = o;
  }
}

extern "C" int cn_pack_weights_bf16(const float* w, void* wp, int T, int K, int N, long sk, long sn, long st,
                                    void* stream) {
  CnBPackDesc d = {w, (bf16_t*)wp, T, K, N, (K + 15) / 16, (N + 31) / 32, 0, sk, sn, st};
  const long total = (long)d.T * d.KS * d.NT * 64;
  const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  CN_LAUNCH(cn_bpack_one_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d);
  return cn_check_launch();
}

// The same pack with a per-n (cout) factor folded in: W'[n] = W[n] * nscale[n] -- eval-mode BatchNorm folded into the
// convolution that feeds it (nscale = gamma / sqrt(running_var + eps), from cn_bn_fold_f32).
extern "C" int cn_pack_weights_scaled_bf16(const float* w, const float* nscale, void* wp, int T, int K, int N, long sk,
                                           long sn, long st, void* stream) {
  CnBPackDesc d = {w, (bf16_t*)wp, T, K, N, (K + 15) / 16, (N + 31) / 32, 0, sk, sn, st, nscale};
  const long total = (long)d.T * d.KS * d.NT * 64;
  const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  CN_LAUNCH(cn_bpack_one_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d);
  return cn_check_launch();
}

// ------------------------------------------------------------------------------------------------------------
// host side: classes, tiles, launch
// ------------------------------------------------------------------------------------------------------------
static inline int cnb_floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

// Pixel tile (TH x TW <= max_pix = 128 / 256 / 512 for WN = 4 / 2 / 1) for a logical grid: exact covers first
// (5x25, 10x25 and 20x25 cover 100, 50 and 25 exactly).
static void cnb_pick_tile(int Hg, int Wg, int is, int span, int max_pix, int& TH, int& TW) {
  static const int cand[][2] = {{5, 25}, {8, 16}, {4, 32}, {16, 8}, {2, 64}, {10, 12}, {6, 20}, {9, 14}, {11, 11},
                                {13, 9}, {7, 18}, {3, 42}, {1, 128}, {4, 16}, {8, 8}, {2, 32}, {4, 8}, {2, 16},
                                {10, 25}, {16, 16}, {8, 32}, {13, 19}, {12, 21}, {14, 18}, {4, 64},
                                {20, 25}, {16, 32}, {22, 23}, {13, 39}, {8, 64}, {32, 16}};
  double best = 1e300;
  TH = 4; TW = 8;
  for (auto& c : cand) {
    const int th = c[0], tw = c[1];
    if (th * tw > max_pix) continue;
    const long ih = (long)(th - 1) * is + span + 1, iw = (long)(tw - 1) * is + span + 1;
    if (ih * iw > 640) continue;  // staging budget: 10 pieces per thread, 51 KB of LDS
    const long tiles = (long)((Hg + th - 1) / th) * ((Wg + tw - 1) / tw);
    // cost ~ MFMA columns issued (every block runs all max_pix / 32 columns) with a small penalty for halo bytes
    const double cost = (double)tiles * max_pix + 0.05 * tiles * ih * iw;
    if (cost < best) { best = cost; TH = th; TW = tw; }
  }
}

// Host request for the in-launch statistics finalize (cn_conv2d_fwd_grouped_bnstats_bf16); `done` reports the decision.
struct CnBFinReq {
  float* const* means; float* const* rstds; float* const* rmeans; float* const* rvars;
  float momentum, eps;
  float* ws; long ws_floats;  // the grouped-BatchNorm workspace (zero head: cn_ticket.h)
  int done;
};

static int cnb_launch(CnBGeom& g, int G, hipStream_t stream, double flops, CnBFinReq* fin = nullptr) {
  // tiles: one (TH, TW) for the launch, from the largest class grid and the widest tap span
  int span = 0, Hg = 1, Wg = 1;
  for (int c = 0; c < g.ncls; ++c) {
    CnBClass& k = g.cls[c];
    int mn_y = 0, mx_y = 0, mn_x = 0, mx_x = 0;
    for (int t = 0; t < k.ntaps; ++t) {
      const int dy = k.doff[t] >> 16, dx = (short)(k.doff[t] & 0xffff);  // packed (dy, dx) from the builders
      if (t == 0) { mn_y = mx_y = dy; mn_x = mx_x = dx; }
      mn_y = dy < mn_y ? dy : mn_y; mx_y = dy > mx_y ? dy : mx_y;
      mn_x = dx < mn_x ? dx : mn_x; mx_x = dx > mx_x ? dx : mx_x;
    }
    k.iy_off = mn_y; k.ix_off = mn_x;
    const int sp = (mx_y - mn_y) > (mx_x - mn_x) ? (mx_y - mn_y) : (mx_x - mn_x);
    span = sp > span ? sp : span;
    Hg = k.Hg > Hg ? k.Hg : Hg; Wg = k.Wg > Wg ? k.Wg : Wg;
    // stash the spans for the second pass
    k.IH = mx_y - mn_y; k.IW = mx_x - mn_x;
  }
  g.KS = (g.Cin + 15) / 16;
  g.NT = (g.Cout + 31) / 32;
  {  // weight fragments are fetched through a buffer resource with a 32-bit scalar byte offset: refuse packs >= 2 GiB
    int wt_max = 0;
    for (int c = 0; c < g.ncls; ++c)
      for (int t = 0; t < g.cls[c].ntaps; ++t) wt_max = g.cls[c].wt[t] > wt_max ? g.cls[c].wt[t] : wt_max;
    if ((long)(wt_max + 1) * g.KS * g.NT * 1024 >= (1L << 31)) return CN_ERR_ARG;
  }
  int WN = g.NT >= 4 ? 4 : (g.NT >= 2 ? 2 : 1);
  int MPWv = 4;
  cnb_pick_tile(Hg, Wg, g.is, span, 128 * (4 / WN), g.TH, g.TW);
  if (WN == 4 && getenv("CN_BCONV_NO_HALF") == nullptr) {
    // small planes (25x25 at batch 32: 160 tiles x 1 cout block on 256 CUs): two 64-cout blocks per pixel tile instead
    long blocks = 0;
    for (int c = 0; c < g.ncls; ++c)
      blocks += (long)((g.cls[c].Wg + g.TW - 1) / g.TW) * ((g.cls[c].Hg + g.TH - 1) / g.TH) * g.B * ((g.NT + 3) / 4);
    if (blocks <= 224) { WN = 2; MPWv = 2; }
  }
  g.nblk_n = (g.NT + WN - 1) / WN;
  // halo sizes, then the chunk depth: as many 16-channel k-steps per staged image as the staging budget allows
  // (10 pieces of 16 bytes per thread), 1x1 launches preferring the deepest (one tap per chunk to amortise the
  // barrier + staging phase over), 3x3 launches 64 channels
  int max_pix = 0, max_taps = 0;
  for (int c = 0; c < g.ncls; ++c) {
    CnBClass& k = g.cls[c];
    const int sy = k.IH, sx = k.IW;
    k.IH = (g.TH - 1) * g.is + sy + 1;
    k.IW = (g.TW - 1) * g.is + sx + 1;
    max_pix = k.IH * k.IW > max_pix ? k.IH * k.IW : max_pix;
    max_taps = k.ntaps > max_taps ? k.ntaps : max_taps;
  }
  int KSC = 2;
  {
    const int pref[3] = {max_taps <= 1 ? 8 : 4, max_taps <= 1 ? 4 : 2, 2};
    for (int i = 0; i < 3; ++i) {
      const int c = pref[i];
      if (c > 2 && c / 2 >= g.KS) continue;  // deeper than the input is wide
      if (((long)max_pix * c * 2 + 255) / 256 <= 10) { KSC = c; break; }
    }
  }
  const int pitch = KSC * 32 + 16;
  g.mgTW = cnb_magic(g.TW);
  for (int c = 0; c < g.ncls; ++c) g.cls[c].mgIW = cnb_magic(g.cls[c].IW);
  const int np = (int)(((long)max_pix * KSC * 2 + 255) / 256);
  if (np > 10) return CN_ERR_LDS;
  // A wave's ds_read_b128 covers 32 consecutive tile pixels m (lane groups of 16, one 256-byte bank row per group):
  // with a pixel pitch of s = pitch/16 (odd) sixteen-byte slots the slot of pixel m is s*m mod 16, all distinct within
  // a group -- as long as consecutive m are consecutive in LDS. A 32-pixel column wraps over tile rows (TW = 25), where
  // the halo image jumps by IW - TW pixels and two lanes of a group land on one slot (PMC: 48 % of the LDS cycles of
  // the 3x3 layers were conflict cycles). rowpad shifts every halo row by the slots that restore s*m mod 16:
  // rowpad = (-s * (IW - TW)) mod 16 slots. Unit input stride only (a stride-2 image has an even slot step anyway).
  // Measured: SQ_LDS_BANK_CONFLICT 11.8M -> 1.5M per 128->128 launch at 32x100^2, LDS busy 36 % -> 23 %; the kernel's
  // duration did not move (the LDS was not the limiter), the headroom is there for the next restructuring.
  int max_rowpad_bytes = 0;
  for (int c = 0; c < g.ncls; ++c) {
    CnBClass& k = g.cls[c];
    k.rowpad = 0;
    if (g.is == 1 && k.IH > 1 && np <= 6) {  // the big-halo (NP = 10) instantiations sit at the VGPR limit: no pad there
      const int s16 = (pitch / 16) & 15;
      const int slots = ((-(s16 * (k.IW - g.TW))) % 16 + 16) % 16;
      k.rowpad = slots * 16;
    }
    if (k.IH * k.rowpad > max_rowpad_bytes) max_rowpad_bytes = k.IH * k.rowpad;
  }
  long total = 0;
  for (int c = 0; c < g.ncls; ++c) {
    CnBClass& k = g.cls[c];
    for (int t = 0; t < k.ntaps; ++t) {
      const int dy = k.doff[t] >> 16, dx = (short)(k.doff[t] & 0xffff);
      k.doff[t] = ((dy - k.iy_off) * k.IW + (dx - k.ix_off)) * pitch + (dy - k.iy_off) * k.rowpad;
    }
    k.tiles_x = (k.Wg + g.TW - 1) / g.TW;
    k.tiles_per_img = k.tiles_x * ((k.Hg + g.TH - 1) / g.TH);
    k.block_begin = (int)total;
    total += (long)k.tiles_per_img * g.B * g.nblk_n;
  }
  if (total <= 0) return CN_OK;
  if (total > 0x7fffff00L) return CN_ERR_ARG;
  g.total = (int)total;
  if (g.interleave) {  // (asked for by cnb_scatter) granted when the classes' tile counts agree
    for (int c = 1; c < g.ncls; ++c)
      if (g.cls[c].tiles_per_img != g.cls[0].tiles_per_img) g.interleave = 0;
    if (g.ncls < 2 || g.stats[0] != nullptr) g.interleave = 0;
  }
  size_t shmem = (size_t)max_pix * pitch + max_rowpad_bytes;
  if (g.stats[0] != nullptr && shmem < 4 * 64 * 33 * sizeof(float)) shmem = 4 * 64 * 33 * sizeof(float);
  const dim3 grid(cn_xcd_grid(total)), block(256);
  if (g.stats[0] != nullptr && g.ncls != G) return CN_ERR_ARG;  // one class per group: a tile's row has ONE writer
  g.fin_cnt = nullptr;
  if (fin != nullptr) {
    fin->done = 0;
    static const bool on = getenv("CN_CONV_BNFIN") == nullptr || atoi(getenv("CN_CONV_BNFIN")) != 0;  // A/B switch
    const long tiles = (long)g.cls[0].tiles_per_img * g.B;
    const long domains = (long)G * g.nblk_n;
    const long grow_doubles = domains * (CN_T2_COUNTERS - 1) * 64 * WN;
    if (on && g.stats[0] != nullptr && tiles <= CN_BNWS_CONV_MAX_TILES && domains <= CN_BNWS_CONV_DOMAINS &&
        fin->ws != nullptr && (reinterpret_cast<uintptr_t>(fin->ws) & 7) == 0 &&
        fin->ws_floats >= CN_BNWS_HEAD_INTS + 2 * grow_doubles) {
      g.fin_cnt = reinterpret_cast<int*>(fin->ws) + CN_BNWS_CONV_OFF;
      g.fin_grows = reinterpret_cast<double*>(fin->ws + CN_BNWS_HEAD_INTS);
      g.fin_tiles = (int)tiles;
      g.fin_eps = fin->eps; g.fin_momentum = fin->momentum;
      for (int i = 0; i < G; ++i) {
        g.fin_mean[i] = fin->means[i]; g.fin_rstd[i] = fin->rstds[i];
        g.fin_rmean[i] = fin->rmeans ? fin->rmeans[i] : nullptr;
        g.fin_rvar[i] = fin->rvars ? fin->rvars[i] : nullptr;
      }
      fin->done = 1;
    }
  }
  const int NPv = np <= 4 ? 4 : (np <= 6 ? 6 : 10);
  if (MPWv == 2)
    cn_prof_name("cn_bconv_kernel<%d, %d, %d, 2>", WN, (KSC == 2 ? NPv : (KSC == 4 ? (NPv <= 6 ? 6 : 10) : 10)), KSC);
  else
    cn_prof_name("cn_bconv_kernel<%d, %d, %d, 4>", WN, (KSC == 2 ? NPv : (KSC == 4 ? (NPv <= 6 ? 6 : 10) : 10)), KSC);
  cn_prof_desc("bconv B%d %dx%d %d->%d cls%d taps%d s%d/%d", g.B, g.Hin, g.Win, g.Cin, g.Cout, g.ncls, g.cls[0].ntaps,
               g.is, g.os);
  {  // bf16 operands / results, every group its own tensors; the packed weights of all classes once
    int taps_all = 0;
    for (int c = 0; c < g.ncls; ++c) taps_all += g.cls[c].ntaps;
    cn_prof_bytes(2.0 * G * ((double)g.B * g.Hin * g.Win * g.Cin + (double)g.B * g.Hout * g.Wout * g.Cout * (g.out_kind ? 2 : 1)) +
                  2.0 * (double)taps_all * g.Cin * g.Cout);
  }
  cn_prof_before(stream);
#define CNB_GO3(WN_, NP_, KSC_)                                                                                  \
  do {                                                                                                           \
    if (shmem > 64 * 1024)                                                                                       \
      (void)hipFuncSetAttribute((const void*)cn_bconv_kernel<WN_, NP_, KSC_>,                                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);                         \
    CN_LAUNCH((cn_bconv_kernel<WN_, NP_, KSC_>), grid, block, shmem, stream, g);                        \
  } while (0)
#define CNB_GO3H(NP_, KSC_)                                                                                      \
  do {                                                                                                           \
    if (shmem > 64 * 1024)                                                                                       \
      (void)hipFuncSetAttribute((const void*)cn_bconv_kernel<2, NP_, KSC_, 2>,                                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);                         \
    CN_LAUNCH((cn_bconv_kernel<2, NP_, KSC_, 2>), grid, block, shmem, stream, g);                                \
  } while (0)
#define CNB_GO(NP_, KSC_)                                                                                        \
  do {                                                                                                           \
    if (MPWv == 2) CNB_GO3H(NP_, KSC_);                                                                          \
    else if (WN == 4) CNB_GO3(4, NP_, KSC_); else if (WN == 2) CNB_GO3(2, NP_, KSC_); else CNB_GO3(1, NP_, KSC_); \
  } while (0)
  if (KSC == 2) {
    if (NPv == 4) CNB_GO(4, 2); else if (NPv == 6) CNB_GO(6, 2); else CNB_GO(10, 2);
  } else if (KSC == 4) {
    if (NPv <= 6) CNB_GO(6, 4); else CNB_GO(10, 4);
  } else {
    CNB_GO(10, 8);
  }
#undef CNB_GO
#undef CNB_GO3
#undef CNB_GO3H
  cn_prof_after(stream, 4, flops);
  return cn_check_launch();
}

static inline int cnb_pack_d(int dy, int dx) { return (dy << 16) | (dx & 0xffff); }

// Gather form: in = o*stride + k*dil - pad (Conv2d forward, ConvTranspose2d backward-data).
static int cnb_gather(int G, const bf16_t* const* xs, long ldx, const bf16_t* const* wps, const float* const* biases,
                      void* const* ys, long ldy, long y_bs, int B, int Cin, int Hin, int Win, int Cout, int Hout,
                      int Wout, int KH, int KW, int stride, const int* pads, const int* dils, int accumulate,
                      int out_kind, float* const* stats, hipStream_t stream, int act = 0,
                      const bf16_t* const* ress = nullptr, long ldres = 0, CnBFinReq* fin = nullptr) {
  if (G < 1 || G > CNB_MAX_GROUPS || KH * KW > CNB_MAX_TAPS || stride < 1) return CN_ERR_ARG;
  if (Hout <= 0 || Wout <= 0 || B <= 0) return CN_OK;
  CnBGeom g = {};
  for (int i = 0; i < G; ++i) {
    g.x[i] = xs[i]; g.wp[i] = wps[i]; g.bias[i] = biases ? biases[i] : nullptr; g.y[i] = ys[i];
    g.res[i] = ress ? ress[i] : nullptr;
    g.stats[i] = stats ? stats[i] : nullptr;
  }
  g.act = act; g.ldres = ldres;
  g.ldx = ldx; g.ldy = ldy; g.y_bs = y_bs;
  g.B = B; g.Cin = Cin; g.Hin = Hin; g.Win = Win; g.Cout = Cout; g.Hout = Hout; g.Wout = Wout;
  g.is = stride; g.os = 1; g.out_kind = out_kind; g.accumulate = accumulate;
  g.ncls = G;
  for (int i = 0; i < G; ++i) {
    if (dils[i] < 1) return CN_ERR_ARG;
    CnBClass& k = g.cls[i];
    k.grp = i; k.Hg = Hout; k.Wg = Wout; k.oy0 = 0; k.ox0 = 0; k.ntaps = KH * KW;
    for (int ky = 0; ky < KH; ++ky)
      for (int kx = 0; kx < KW; ++kx) {
        const int t = ky * KW + kx;
        k.doff[t] = cnb_pack_d(ky * dils[i] - pads[i], kx * dils[i] - pads[i]);
        k.wt[t] = t;
      }
  }
  const double flops = 2.0 * B * Hout * Wout * (double)Cout * Cin * KH * KW * G;
  return cnb_launch(g, G, stream, flops, fin);
}

// Scatter form: out[o] = bias + sum_k src[(o + pad - k*dil)/s] W[k] where divisible (Conv2d backward-data,
// ConvTranspose2d forward), as s*s parity classes in one launch.
static int cnb_scatter(int G, const bf16_t* const* srcs, long lds_, const bf16_t* const* wps,
                       const float* const* biases, void* const* outs, long ldo, int B, int Csrc, int Hs, int Ws,
                       int Cdst, int Ho, int Wo, int KH, int KW, int stride, const int* pads, const int* dils,
                       int accumulate, float* stats, hipStream_t stream) {
  if (G < 1 || G > CNB_MAX_GROUPS || KH * KW > CNB_MAX_TAPS || stride < 1 || G * stride * stride > CNB_MAX_CLASSES)
    return CN_ERR_ARG;
  if (Ho <= 0 || Wo <= 0 || B <= 0) return CN_OK;
  CnBGeom g = {};
  for (int i = 0; i < G; ++i) {
    g.x[i] = srcs[i]; g.wp[i] = wps[i]; g.bias[i] = biases ? biases[i] : nullptr; g.y[i] = outs[i];
  }
  g.ldx = lds_; g.ldy = ldo;
  g.B = B; g.Cin = Csrc; g.Hin = Hs; g.Win = Ws; g.Cout = Cdst; g.Hout = Ho; g.Wout = Wo;
  g.is = 1; g.os = stride; g.out_kind = 0; g.accumulate = accumulate;
  (void)stats;  // (strided scatters have s*s classes per group: no per-tile statistics rows)
  int nc = 0;
  double macs = 0.0;
  for (int gi = 0; gi < G; ++gi) {
    const int pad = pads[gi], dil = dils[gi];
    if (dil < 1) return CN_ERR_ARG;
    for (int py = 0; py < stride; ++py)
      for (int px = 0; px < stride; ++px) {
        CnBClass& k = g.cls[nc];
        k = CnBClass{};
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
            k.doff[nt] = cnb_pack_d(cnb_floordiv(ny, stride), cnb_floordiv(nx, stride));
            k.wt[nt] = ky * KW + kx;
            ++nt;
          }
        }
        k.ntaps = nt;
        macs += (double)k.Hg * k.Wg * nt;
        ++nc;
      }
  }
  g.ncls = nc;
  if (nc == 0) return CN_OK;
  // heavy parity classes first, interleaved per cell tile when their tile counts agree (round 6, as in cn_conv.hip)
  for (int i = 1; i < nc; ++i) {
    const CnBClass key = g.cls[i];
    int j = i - 1;
    while (j >= 0 && g.cls[j].ntaps < key.ntaps) { g.cls[j + 1] = g.cls[j]; --j; }
    g.cls[j + 1] = key;
  }
  // (batch 32, 128 -> 128: 50^2 -> 99^2 63.1 -> 57.6 us, 25^2 -> 49^2 24.8 -> 23.0, the stride-4 25^2 -> 97^2 47.1 -> 37.1 us
  // alone; bf16 step 2203.6 -> 2210.1 chips/s same box)
  g.interleave = stride > 1 ? 1 : 0;
  return cnb_launch(g, G, stream, 2.0 * B * macs * Csrc * Cdst);
}

// Rows of the per-tile BatchNorm statistics cn_conv2d_fwd_bf16 writes ([rows][2][Cout] floats; no zero-fill needed).
extern "C" int cn_conv2d_stats_rows_bf16(int B, int Hin, int Win, int Cout, int KH, int KW, int stride, int pad,
                                         int dil) {
  if (stride < 1 || dil < 1) return -1;
  const int Hout = (Hin + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  const int Wout = (Win + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  if (Hout <= 0 || Wout <= 0) return 0;
  const int NT = (Cout + 31) / 32;
  const int WN = NT >= 4 ? 4 : (NT >= 2 ? 2 : 1);
  int TH, TW;
  const int span = dil * ((KH > KW ? KH : KW) - 1);
  cnb_pick_tile(Hout, Wout, stride, span, 128 * (4 / WN), TH, TW);
  return B * ((Hout + TH - 1) / TH) * ((Wout + TW - 1) / TW);
}

// ---- C ABI ---------------------------------------------------------------------------------------------------
// nn.Conv2d forward (convolution.py:71-120). x bf16 NHWC [B,Hin,Win,>=Cin] (ldx), wp from cn_pack_weights_bf16 with
// K = Cin, N = Cout; y bf16 NHWC (out_kind 0, ldy) or f32 NCHW (out_kind 1, batch stride y_bs). stats (nullable):
// 2*Cout floats, zeroed by the caller, receive the per-channel sum / sum of squares of the fp32 results
// (BatchNorm batch statistics without re-reading y).
extern "C" int cn_conv2d_fwd_bf16(const void* x, long ldx, const void* wp, const float* bias, void* y, long ldy,
                                  long y_bs, int B, int Cin, int Hin, int Win, int Cout, int KH, int KW, int stride,
                                  int pad, int dil, int accumulate, int out_kind, float* stats, void* stream) {
  if (stride < 1) return CN_ERR_ARG;
  const int Hout = (Hin + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  const int Wout = (Win + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  const bf16_t* xs = (const bf16_t*)x;
  const bf16_t* ws = (const bf16_t*)wp;
  return cnb_gather(1, &xs, ldx, &ws, &bias, &y, ldy, y_bs, B, Cin, Hin, Win, Cout, Hout, Wout, KH, KW, stride, &pad,
                    &dil, accumulate, out_kind, stats != nullptr ? &stats : nullptr, (hipStream_t)stream);
}

// G (<= 4) convolutions of one shape in one launch (the dilation branches of ResidualAConv, convolution.py:376-395).
// Eval-mode ConvBlock2d in ONE launch (convolution.py:71-120 with BatchNorm in inference mode): the caller folds the
// running statistics into the packed weights (cn_pack_weights_scaled_bf16: W' = W * gamma / sqrt(var + eps) per cout)
// and into `bias` (beta - mean * gamma / sqrt(var + eps)); y = res + act(conv(x, W') + bias), act 0 = identity,
// 1 = SiLU; res (nullable) bf16 NHWC with pixel stride ldres -- the ResUNet-a running sum (convolution.py:376-395).
// Cout must be a multiple of 8 (bf16 NHWC fast path). No BatchNorm launch, no second pass over the activation.
extern "C" int cn_conv2d_fwd_fused_bf16(const void* x, long ldx, const void* wp, const float* bias, const void* res,
                                        long ldres, void* y, long ldy, int B, int Cin, int Hin, int Win, int Cout, int KH,
                                        int KW, int stride, int pad, int dil, int act, void* stream) {
  if (stride < 1 || (Cout & 7) || act < 0 || act > 1) return CN_ERR_ARG;
  const int Hout = (Hin + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  const int Wout = (Win + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  const bf16_t* xs = (const bf16_t*)x;
  const bf16_t* ws = (const bf16_t*)wp;
  const bf16_t* rs = (const bf16_t*)res;
  return cnb_gather(1, &xs, ldx, &ws, &bias, &y, ldy, 0, B, Cin, Hin, Win, Cout, Hout, Wout, KH, KW, stride, &pad,
                    &dil, 0, 0, nullptr, (hipStream_t)stream, act, &rs, ldres);
}

extern "C" int cn_conv2d_fwd_grouped_bf16(int G, const void* const* xs, long ldx, const void* const* wps,
                                          const float* const* biases, void* const* ys, long ldy, int B, int Cin,
                                          int Hin, int Win, int Cout, int KH, int KW, int stride, const int* pads,
                                          const int* dils, int accumulate, float* const* stats, void* stream) {
  if (stride < 1 || G < 1 || G > CNB_MAX_GROUPS) return CN_ERR_ARG;
  const int Hout = (Hin + 2 * pads[0] - dils[0] * (KH - 1) - 1) / stride + 1;
  const int Wout = (Win + 2 * pads[0] - dils[0] * (KW - 1) - 1) / stride + 1;
  for (int i = 1; i < G; ++i)
    if ((Hin + 2 * pads[i] - dils[i] * (KH - 1) - 1) / stride + 1 != Hout ||
        (Win + 2 * pads[i] - dils[i] * (KW - 1) - 1) / stride + 1 != Wout)
      return CN_ERR_ARG;
  return cnb_gather(G, (const bf16_t* const*)xs, ldx, (const bf16_t* const*)wps, biases, ys, ldy, 0, B, Cin, Hin, Win,
                    Cout, Hout, Wout, KH, KW, stride, pads, dils, accumulate, 0, stats, (hipStream_t)stream);
}

// The same launch (bias-free: ConvBlock2d, convolution.py:71-120) that also FINISHES the BatchNorm batch statistics of
// its G outputs when it can: *finalized = 1 -> means / rstds (and, when given, the running statistics with `momentum`)
// are written by the launch and the caller goes straight to cn_bn_act_group_fwd_bf16(..., conv_rows = -1);
// *finalized = 0 -> only the per-tile rows in `stats` were written (more than 1008 tiles per convolution, or a workspace
// too small) and the caller passes them to cn_bn_act_group_fwd_bf16 as before. bn_ws: the grouped-BatchNorm workspace
// of the launch stream (cn_bn_group_workspace_floats_bf16(G, Cout) floats, zero-filled once; its head holds the tickets).
extern "C" int cn_conv2d_fwd_grouped_bnstats_bf16(int G, const void* const* xs, long ldx, const void* const* wps,
                                                  void* const* ys, long ldy, int B, int Cin, int Hin, int Win, int Cout,
                                                  int KH, int KW, int stride, const int* pads, const int* dils,
                                                  float* const* stats, float* const* means, float* const* rstds,
                                                  float* const* running_means, float* const* running_vars,
                                                  float momentum, float eps, float* bn_ws, long bn_ws_floats,
                                                  int* finalized, void* stream) {
  if (stride < 1 || G < 1 || G > CNB_MAX_GROUPS || stats == nullptr || means == nullptr || rstds == nullptr ||
      finalized == nullptr || (running_means == nullptr) != (running_vars == nullptr))
    return CN_ERR_ARG;
  const int Hout = (Hin + 2 * pads[0] - dils[0] * (KH - 1) - 1) / stride + 1;
  const int Wout = (Win + 2 * pads[0] - dils[0] * (KW - 1) - 1) / stride + 1;
  for (int i = 1; i < G; ++i)
    if ((Hin + 2 * pads[i] - dils[i] * (KH - 1) - 1) / stride + 1 != Hout ||
        (Win + 2 * pads[i] - dils[i] * (KW - 1) - 1) / stride + 1 != Wout)
      return CN_ERR_ARG;
  CnBFinReq fin = {means, rstds, running_means, running_vars, momentum, eps, bn_ws, bn_ws_floats, 0};
  const int rc = cnb_gather(G, (const bf16_t* const*)xs, ldx, (const bf16_t* const*)wps, nullptr, ys, ldy, 0, B, Cin, Hin,
                            Win, Cout, Hout, Wout, KH, KW, stride, pads, dils, 0, 0, stats, (hipStream_t)stream, 0, nullptr,
                            0, &fin);
  *finalized = fin.done;
  return rc;
}

// Conv2d backward-data: dx [B,Hin,Win,Cin] (+)= scatter(dy [B,Hout,Wout,Cout]); wp_t packed with K = Cout, N = Cin.
extern "C" int cn_conv2d_bwd_data_bf16(const void* dy, long lddy, const void* wp_t, void* dx, long lddx, int B, int Cin,
                                       int Hin, int Win, int Cout, int KH, int KW, int stride, int pad, int dil,
                                       int accumulate, void* stream) {
  if (stride < 1) return CN_ERR_ARG;
  const int Hout = (Hin + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  const int Wout = (Win + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  const bf16_t* s = (const bf16_t*)dy;
  const bf16_t* w = (const bf16_t*)wp_t;
  return cnb_scatter(1, &s, lddy, &w, nullptr, &dx, lddx, B, Cout, Hout, Wout, Cin, Hin, Win, KH, KW, stride, &pad,
                     &dil, accumulate, nullptr, (hipStream_t)stream);
}

extern "C" int cn_conv2d_bwd_data_grouped_bf16(int G, const void* const* dys, long lddy, const void* const* wps_t,
                                               void* const* dxs, long lddx, int B, int Cin, int Hin, int Win, int Cout,
                                               int KH, int KW, int stride, const int* pads, const int* dils,
                                               int accumulate, void* stream) {
  if (stride < 1 || G < 1 || G > CNB_MAX_GROUPS) return CN_ERR_ARG;
  const int Hout = (Hin + 2 * pads[0] - dils[0] * (KH - 1) - 1) / stride + 1;
  const int Wout = (Win + 2 * pads[0] - dils[0] * (KW - 1) - 1) / stride + 1;
  return cnb_scatter(G, (const bf16_t* const*)dys, lddy, (const bf16_t* const*)wps_t, nullptr, dxs, lddx, B, Cout, Hout,
                     Wout, Cin, Hin, Win, KH, KW, stride, pads, dils, accumulate, nullptr, (hipStream_t)stream);
}

// nn.ConvTranspose2d forward (convolution.py:45-68): y [B,Hout,Wout,Cout], Hout = (Hin-1)*s - 2*pad + K.
extern "C" int cn_conv_transpose2d_fwd_bf16(const void* x, long ldx, const void* wp, const float* bias, void* y,
                                            long ldy, int B, int Cin, int Hin, int Win, int Cout, int KH, int KW,
                                            int stride, int pad, int accumulate, void* stream) {
  const int Hout = (Hin - 1) * stride - 2 * pad + KH;
  const int Wout = (Win - 1) * stride - 2 * pad + KW;
  const bf16_t* s = (const bf16_t*)x;
  const bf16_t* w = (const bf16_t*)wp;
  const int dil = 1;
  return cnb_scatter(1, &s, ldx, &w, &bias, &y, ldy, B, Cin, Hin, Win, Cout, Hout, Wout, KH, KW, stride, &pad, &dil,
                     accumulate, nullptr, (hipStream_t)stream);
}

// ConvTranspose2d backward-data: dx [B,Hin,Win,Cin] (+)= conv_stride_s(dy); wp_t packed with K = Cout, N = Cin.
extern "C" int cn_conv_transpose2d_bwd_data_bf16(const void* dy, long lddy, const void* wp_t, void* dx, long lddx,
                                                 int B, int Cin, int Hin, int Win, int Cout, int KH, int KW,
                                                 int stride, int pad, int accumulate, void* stream) {
  const int Hout = (Hin - 1) * stride - 2 * pad + KH;
  const int Wout = (Win - 1) * stride - 2 * pad + KW;
  const bf16_t* s = (const bf16_t*)dy;
  const bf16_t* w = (const bf16_t*)wp_t;
  const int dil = 1;
  return cnb_gather(1, &s, lddy, &w, nullptr, &dx, lddx, 0, B, Cout, Hout, Wout, Cin, Hin, Win, KH, KW, stride, &pad,
                    &dil, accumulate, 0, nullptr, (hipStream_t)stream);
}
