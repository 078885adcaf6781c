// PreTimeReduction (reference: models/nunet.py:18-105) as ONE fused kernel family for gfx950.
//
//   x [B, C, T, H, W]  ->  for k in {3, 5}:  Conv3d(C -> C, (k,1,1)) -> BatchNorm3d -> SiLU
//                                            -> Conv3d(C -> Cout, (T-k+1,1,1)) -> BatchNorm2d -> SiLU
//                          LayerNorm_Cout(branch3 + branch5)                       -> y [B, Cout, H, W]
//
// Everything between the BatchNorm reductions is per pixel: 36 inputs (C = 3, T = 12), ~2.4k multiply-adds, 32 outputs.
// Rounds 1-3 ran the stage as ~40 launches of the generic kernels (banded 1x1 contractions through the MFMA path at 19
// TFLOP/s, BatchNorm3d as a 3-channel view, BatchNorm2d, LayerNorm, conversions): 1.15 ms of the 16 ms mixed-precision
// step and ~0.6 ms of the fp32 step for 0.1 % of the FLOPs. Here the chain is RECOMPUTED from x in every pass instead of
// being stored (x is 144 bytes per pixel; the intermediates would be 900): three forward passes (BatchNorm3d
// statistics, BatchNorm2d statistics, output) and three backward passes (LayerNorm / BatchNorm2d sums, second-conv
// weight gradients + BatchNorm3d sums, first-conv weight gradients), each ONE launch that finishes its own cross-block
// reduction through the two-level last-block ticket of cn_ticket.h (no finalize launches, fixed summation order).
// Inference is the output pass alone. The input needs no gradient (it is the data).
//
// Third version of the work decomposition (round 4; the first two, measured: DESIGN section 4c). A WAVE owns 32
// pixels: lane l works on pixel l % 32 and on the entries (time steps) of parity l / 32 of the first convolutions, so
// the pair of activations a[2s], a[2s + 1] a wave produces per step IS the B operand (K = 2) of
// v_mfma_f32_32x32x2_f32, and the second convolution -- the one dense contraction of the stage, 73 % of its multiply-adds
// -- runs on the matrix pipe with the weights as A operand (one conflict-free LDS read per lane and step): the
// accumulator comes out as [32 pixels] x [16 output channels per lane], 16 registers per branch instead of one per
// output. The backward contractions use the same pipe: da = W^T dr with the accumulator registers of dr as B operand
// as they are, and dWb = dr a^T over the pixels with both operands read transposed out of LDS. Column sums over pixels
// (statistics, parameter gradients) are DPP half-wave reductions. All weights and per-channel constants are staged once
// per block in LDS; blocks are persistent over the pixel tiles.
#include <cstdlib>
#include <type_traits>
#include "cn_bf16.h"
#include "cn_ticket.h"
#include "cn_profile.h"

#define PT_MAX_BLOCKS 512      // persistent blocks of the generic kernel (two per CU)
#define PT_MAX_BLOCKS_REG 1008 // ... of the register variant (up to four per CU; the ticket's limit)
#define PT_MAX_C 8
#define PT_PXB 128  // pixels per block tile: 4 waves x 32
#define PT_KP 5     // weight row pitch of the first convolutions in LDS (k <= 5)
#define PT_LP 33    // pitch of the per-wave [row][32 pixels] transposition tiles (conflict-free both ways)

struct CnPtBranch {
  const float* wa;   // [C][C][k]       Conv3d(C -> C, (k,1,1)).weight
  const float* wb;   // [Cout][C][Tp]   Conv3d(C -> Cout, (Tp,1,1)).weight
  const float* g3; const float* b3; float* rm3; float* rv3; float* mean3; float* rstd3;  // BatchNorm3d(C)
  const float* g2; const float* b2; float* rm2; float* rv2; float* mean2; float* rstd2;  // BatchNorm2d(Cout)
  float* dwa; float* dwb; float* dg3; float* db3; float* dg2; float* db2;                // gradients (accumulated)
  int k, Tp;
};

struct CnPtArgs {
  CnPtBranch br[2];
  const float* x; long xbs;
  const float* gL; const float* bL; float* dgL; float* dbL;
  void* y; long y_stride;         // out_kind 0: fp32 NCHW, batch stride; 1: bf16 NHWC, pixel stride
  const void* dy; long dy_stride;
  int out_kind;
  int B, C, T, HW, Cout;
  long P;
  int training;
  float eps3, eps2, mom3, mom2, epsL;
  float* coef2;  // [2 branches][2][Cout]
  float* coef3;  // [2 branches][2][C]
  float* dz;     // [E][P]
  float* img;    // the table image (pt_lds layout up to .xs): weights in operand order, per-channel constants
  CnTicket2 tk;
  int ntiles;
};

// SiLU and its derivative on the hardware exp / rcp (1 ulp each: ~2e-7 relative).
__device__ __forceinline__ float pt_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float pt_silu(float x) { return x * pt_sigmoid(x); }
__device__ __forceinline__ float pt_silu_grad(float x) {
  const float s = pt_sigmoid(x);
  return s * (1.0f + x * (1.0f - s));
}
__device__ __forceinline__ f32x16 pt_mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// Sum over the 32 lanes of each half-wave (DPP, no LDS): the totals land in lanes 31 and 63.
__device__ __forceinline__ float pt_half_sum(float v) {
  v = cn_dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = cn_dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = cn_dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = cn_dpp_add<0x118, 0xf>(v);  // row_shr:8  -> lane 15 of each row holds the row total
  v = cn_dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  return v;
}
// Sums EACH of the 16 values of v over the 32 lanes of a half-wave with a halving butterfly: at every step a lane keeps
// one half of its values and hands the other half to the partner that keeps those (16 -> 8 -> 4 -> 2 -> 1 values, partner
// distance 1, 2, 4, 8 inside the 16-lane row, then 16 across the rows). Returns, in lane l, the half-wave total of
// value l % 16. ~55 vector instructions for the 16 sums; one DPP chain per value (5 dependent steps with their wait
// states, a predicated LDS add each) was ~210.
template <int CTRL, int BANKS>
__device__ __forceinline__ float pt_dpp_into(float old, float src) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xf, BANKS, false));
}
__device__ __forceinline__ float pt_hsum16(const f32x16& vin, int lane) {
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
  float v[16], w[8], y[4], z[2];
  // (the elements become opaque scalars first: `b ? vec[1] : vec[0]` is canonicalised into the DYNAMIC index vec[b],
  // which the backend lowers to a chain of 16 compares and selects per access -- 2100 instructions a call)
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    v[j] = vin[j];
    if (j & 1) asm volatile("" : "+v"(v[j]));  // (one opaque element per pair is enough to keep the select a select)
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {  // partner lane ^ 1: quad_perm [1, 0, 3, 2]
    const float keep = b0 ? v[2 * k + 1] : v[2 * k], give = b0 ? v[2 * k] : v[2 * k + 1];
    w[k] = keep + pt_dpp_into<0xB1, 0xf>(0.f, give);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {  // lane ^ 2: quad_perm [2, 3, 0, 1]
    const float keep = b1 ? w[2 * k + 1] : w[2 * k], give = b1 ? w[2 * k] : w[2 * k + 1];
    y[k] = keep + pt_dpp_into<0x4E, 0xf>(0.f, give);
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {  // lane ^ 4: row_shl:4 into banks 0, 2 and row_shr:4 into banks 1, 3
    const float keep = b2 ? y[2 * k + 1] : y[2 * k], give = b2 ? y[2 * k] : y[2 * k + 1];
    z[k] = keep + pt_dpp_into<0x114, 0xa>(pt_dpp_into<0x104, 0x5>(0.f, give), give);
  }
  float r;
  {  // lane ^ 8: row_shl:8 into banks 0, 1 and row_shr:8 into banks 2, 3
    const float keep = b3 ? z[1] : z[0], give = b3 ? z[0] : z[1];
    r = keep + pt_dpp_into<0x118, 0xc>(pt_dpp_into<0x108, 0x3>(0.f, give), give);
  }
  return r + __shfl_xor(r, 16, 64);  // the other row of the half-wave
}
// accumulator register j of lane (pixel, half) of a 32x32 tile holds output row 8 (j / 4) + 4 half + j % 4
__device__ __forceinline__ int pt_row(int j, int half) { return 8 * (j >> 2) + 4 * half + (j & 3); }

__host__ __device__ static inline int pt_ceil32(int v) { return (v + 31) & ~31; }
__host__ __device__ static inline int pt_steps(int C, int T) { return C * ((T - 2 + 1) / 2) + C * ((T - 4 + 1) / 2); }
__host__ __device__ static inline int pt_nvals(int PASS, int C, int Cout, int CMAX) {
  switch (PASS) {
    case 0: return 4 * C;
    case 1: return 4 * Cout;
    case 3: return 6 * Cout;
    case 4: return 4 * C;
    case 5: return 2 * C * CMAX * PT_KP;
    default: return 0;
  }
}
// LDS carve (floats) -- host and device compute the same offsets
struct PtLds {
  int wa, c3, k3, wA, wB, c2, ln, k2, xs, as_, drs, lacc, total;
};
__host__ __device__ static inline PtLds pt_lds(int PASS, int C, int T, int Cout, int CMAX, bool reg = false) {
  const int E3 = C * (T - 2), E5 = C * (T - 4);
  const int CP = pt_ceil32(Cout);
  const int EP = pt_ceil32(E3) + pt_ceil32(E5);
  PtLds l;
  int o = 0;
  l.wa = o; o += (2 * C * CMAX * PT_KP + 3) & ~3;
  l.c3 = o; o += 2 * C * 4;
  l.k3 = o; o += (2 * C * 2 + 3) & ~3;
  // (the tables up to here form the IMAGE every block copies from global memory in one batched pass; its layout does
  // not depend on PASS, only how much of it a pass copies: pt_img_floats)
  l.wA = o; o += pt_steps(C, T) * 2 * CP;
  l.c2 = o; o += 2 * CP * 4;
  l.ln = o; o += 2 * CP;
  l.k2 = o; o += 2 * CP * 2;
  l.wB = o; if (PASS == 4) o += CP * EP;
  o = (o + 3) & ~3;
  l.xs = o; if (!reg) o += C * (T + 6) * PT_PXB;  // (T + PT_TPAD rows per channel; the register variant has no x tile)
  // per wave: a[entry][pixel] (later da[entry][pixel]) and dr[branch][cout][pixel]. Register variant: ONE slab per wave,
  // a-rows packed to EP3 + E5 with dr right behind, so the 32-row tile accesses of the shorter branch run over into the
  // wave's own (dead or finite) dr rows instead of needing padded rows: 81 KB, two blocks per CU
  l.as_ = o; if (PASS == 4) o += reg ? 4 * (pt_ceil32(E3) + E5 + 2 * CP) * PT_LP : 4 * EP * PT_LP;
  l.drs = o; if (PASS == 4 && !reg) o += 4 * 2 * CP * PT_LP;
  l.lacc = o; o += (4 * pt_nvals(PASS, C, CP, CMAX) + 3) & ~3;  // (per-channel rows at pitch CP: see NVL in the kernel)
  l.total = o;
  return l;
}

// Builds the table image (once per forward / backward call). mode 0: training forward (batch statistics do not exist
// yet: the finishing blocks of PASS 0 / 1 write c3 / c2), 1: inference (running statistics), 2: backward (saved
// statistics; PASS 3 / 4 finishers add k2 / k3).
template <int CMAX>
__global__ __launch_bounds__(256) void cn_pretime_pack_kernel(const CnPtArgs a, int mode) {
  const int C = a.C, T = a.T, Cout = a.Cout;
  const int CP = pt_ceil32(Cout);
  const int T3 = T - 2, T5 = T - 4, E3 = C * T3, E5 = C * T5;
  const int EP3 = pt_ceil32(E3), EP = EP3 + pt_ceil32(E5);
  const int NS3 = C * ((T3 + 1) >> 1);
  const PtLds L = pt_lds(4, C, T, Cout, CMAX);
  float* img = a.img;
  const int gt = blockIdx.x * 256 + threadIdx.x, gn = gridDim.x * 256;
  for (int i = gt; i < 2 * C * CMAX * PT_KP; i += gn) {
    const int dt = i % PT_KP;
    int q = i / PT_KP;
    const int c = q % CMAX;
    q /= CMAX;
    const int cp = q % C, brn = q / C;
    const int k = brn ? 5 : 3;
    img[L.wa + i] = (c < C && dt < k) ? a.br[brn].wa[(cp * C + c) * k + dt] : 0.f;
  }
  for (int i = gt; i < 2 * C; i += gn) {
    const int brn = i / C, cp = i - brn * C;
    const CnPtBranch& r = a.br[brn];
    float mu = 0.f, rho = 1.f;
    if (mode == 1) { mu = r.rm3[cp]; rho = 1.0f / sqrtf(r.rv3[cp] + a.eps3); }
    if (mode == 2) {
      if (a.training) { mu = r.mean3[cp]; rho = r.rstd3[cp]; }
      else { mu = r.rm3[cp]; rho = 1.0f / sqrtf(r.rv3[cp] + a.eps3); }
    }
    img[L.c3 + i * 4 + 0] = rho;
    img[L.c3 + i * 4 + 1] = -mu * rho;
    img[L.c3 + i * 4 + 2] = r.g3[cp];
    img[L.c3 + i * 4 + 3] = r.b3[cp];
    img[L.k3 + i * 2 + 0] = 0.f;
    img[L.k3 + i * 2 + 1] = 0.f;
  }
  // A operands of the second convolutions: wA[step][half][cout] = wb[cout][cp][2 s + half] (0 beyond the row / Cout)
  for (int i = gt; i < pt_steps(C, T) * 2 * CP; i += gn) {
    const int o = i % CP;
    int q = i / CP;
    const int hf = q & 1;
    int gs = q >> 1;
    const int brn = gs >= NS3 ? 1 : 0;
    gs -= brn ? NS3 : 0;
    const int Tp = brn ? T5 : T3;
    const int nsr = (Tp + 1) >> 1;
    const int cp = gs / nsr, tp = 2 * (gs - cp * nsr) + hf;
    img[L.wA + i] = (o < Cout && tp < Tp) ? a.br[brn].wb[((long)o * C + cp) * Tp + tp] : 0.f;
  }
  // wB[cout][entry slot] (branch 3 entries at [0, E3), branch 5 at [EP3, EP3 + E5), zeros elsewhere)
  for (int i = gt; i < CP * EP; i += gn) {
    const int es = i % EP, o = i / EP;
    const int brn = es >= EP3 ? 1 : 0;
    const int loc = es - (brn ? EP3 : 0);
    const int Eb = brn ? E5 : E3;
    img[L.wB + i] = (o < Cout && loc < Eb) ? a.br[brn].wb[(long)o * Eb + loc] : 0.f;
  }
  for (int i = gt; i < 2 * CP; i += gn) {
    const int brn = i / CP, o = i - brn * CP;
    const CnPtBranch& r = a.br[brn];
    float mu = 0.f, rho = 0.f, g2 = 0.f, b2 = 0.f;
    if (o < Cout) {
      g2 = r.g2[o]; b2 = r.b2[o];
      if (mode == 1 || (mode == 2 && !a.training)) { mu = r.rm2[o]; rho = 1.0f / sqrtf(r.rv2[o] + a.eps2); }
      else if (mode == 2) { mu = r.mean2[o]; rho = r.rstd2[o]; }
    }
    img[L.c2 + i * 4 + 0] = rho;
    img[L.c2 + i * 4 + 1] = -mu * rho;
    img[L.c2 + i * 4 + 2] = g2;
    img[L.c2 + i * 4 + 3] = b2;
    img[L.k2 + i * 2 + 0] = 0.f;
    img[L.k2 + i * 2 + 1] = 0.f;
  }
  for (int o = gt; o < CP; o += gn) {
    img[L.ln + o] = o < Cout ? a.gL[o] : 0.f;
    img[L.ln + CP + o] = o < Cout ? a.bL[o] : 0.f;
  }
}

// floats of the table image a pass needs in LDS (PASS 4 also takes wB, the last table)
__host__ __device__ static inline int pt_img_floats(int PASS, int C, int T, int Cout, int CMAX) {
  const PtLds l = pt_lds(4, C, T, Cout, CMAX);
  return PASS == 4 ? l.xs : l.wB;
}

#define PT_TPAD 6  // pad rows per channel of the x tile: window reads past the last time step need no clamp

// One Conv3d(C -> C, (K,1,1)) stack over a pixel, two time steps per wave step: lane half h works on tp = 2 s + h.
// The K-deep window xw[c][.] of every input channel slides along time in registers: per (channel, step) ONE pointer
// add and two LDS reads with immediate offsets (the first version rebuilt every address -- multiply, clamp, select --
// and spent three quarters of the pass's vector instructions on it). xs is [C][T + PT_TPAD][128 pixels].
// entry(cp, tp, live, h, xw): h = convolution value at tp (finite garbage when !live: tp beyond the row).
// Diagnostic build (-DPT_STAMP): s_memtime stamps of wave 0 of block 0 of the LAST pass launched, read back with
// cn_pretime_read_stamps (tools/pretime_stamps.py). Never compiled into the shipped library.
#ifdef PT_STAMP
__device__ unsigned long long pt_stamps[32];
#define PT_ST(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) pt_stamps[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int cn_pretime_read_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(pt_stamps), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -2;
}
#else
#define PT_ST(i) do { } while (0)
#endif

// Phase fence of the register variant: the scheduler otherwise pulls the LDS table reads of LATER phases of the tile
// body (per-channel constants, transposed operands) above the current one to cover their latency -- tens of registers
// each -- and the pass no longer fits the register budget of two or three blocks per CU. Costs one exposed LDS round
// trip per phase, which the other resident waves cover.
#ifndef PT_FENCES
#define PT_FENCES 1
#endif
#define PT_PHASE() do { if (REG && PT_FENCES) asm volatile("" ::: "memory"); } while (0)

extern __shared__ __attribute__((aligned(16))) float pt_smem[];  // the one dynamic LDS array of the pass kernels
#define SM(i) pt_smem[(i)]

template <int K, int CMAX, class FB, class FE, class FR>
__device__ __forceinline__ void pt_rows(int wl, int xs, int C, int T, int pcol, int half, FB&& row_begin, FE&& entry,
                                        FR&& row_end) {
  const int Tp = T - K + 1;
  const int NS = (Tp + 1) >> 1;
  const int TS = T + PT_TPAD;
  for (int cp = 0; cp < C; ++cp) {
    // NO per-channel branches: channels c >= C carry zero weights (the table image pads them) and re-read channel 0,
    // so every loop below is straight-line code -- with `if (c < C)` around each channel the compiler emitted one basic
    // block per channel, each ending in s_waitcnt lgkmcnt(0): C exposed LDS round trips per step, ~1000 cycles a step.
    float w[CMAX][K], xw[CMAX][K];
    int xp[CMAX];  // LDS offsets (every shared-memory access indexes the one __shared__ array: a float* threaded
                   // through lambdas and arrays degraded to FLAT loads with 64-bit address arithmetic)
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      xp[c] = xs + ((c < C ? c : 0) * TS + half) * PT_PXB + pcol;  // &x[c][half][pixel]
#pragma unroll
      for (int dt = 0; dt < K; ++dt) {
        w[c][dt] = SM(wl + ((cp * CMAX + c) * PT_KP + dt));
        xw[c][dt] = SM(xp[c] + dt * PT_PXB);
      }
    }
    row_begin(cp);
    for (int s = 0; s < NS; ++s) {
      const int tp = 2 * s + half;
      // the two new window values per channel for the NEXT step are fetched first: their LDS latency hides under this
      // step's arithmetic (pad rows beyond T: never a clamp)
      float n0[CMAX], n1[CMAX];
#pragma unroll
      for (int c = 0; c < CMAX; ++c) {
        n0[c] = SM(xp[c] + K * PT_PXB);        // x[c][tp + K]
        n1[c] = SM(xp[c] + (K + 1) * PT_PXB);  // x[c][tp + K + 1]
        xp[c] += 2 * PT_PXB;
      }
      float h0 = 0.f, h1 = 0.f;  // two partial sums: half the length of the dependent FMA chain
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
#pragma unroll
        for (int dt = 0; dt < K; ++dt) {
          if ((c + dt) & 1) h1 += w[c][dt] * xw[c][dt];
          else h0 += w[c][dt] * xw[c][dt];
        }
      entry(cp, tp, tp < Tp, h0 + h1, xw);
#pragma unroll
      for (int c = 0; c < CMAX; ++c) {
#pragma unroll
        for (int dt = 0; dt + 2 < K; ++dt) xw[c][dt] = xw[c][dt + 2];
        xw[c][K - 2] = n0[c];
        xw[c][K - 1] = n1[c];
      }
    }
    row_end(cp);
  }
}

// The same stack for a COMPILE-TIME (C, T) (round 5): the pixel's whole C x T cube sits in registers, loaded straight
// from global memory and pre-shifted by the lane's entry parity -- xh[c][j] = x[c][half + j] -- so that every window
// index below is a constant and the loops over output channels and steps unroll completely: no x tile in LDS, no staging
// barrier (the waves of a block never meet inside the tile loop), no window shifting (a third of the generic loop's
// vector instructions were v_mov), no padded fourth channel at C = 3. Same callbacks as pt_rows; xw[c][dt] is a view of
// xh, renamed away by the compiler.
template <int K, int CC, int TT, int CMAX, bool UCP, class FB, class FE, class FR>
__device__ __forceinline__ void pt_rows_reg(int wl, const float (&xh)[CC][TT], int half, FB&& row_begin, FE&& entry,
                                            FR&& row_end) {
  constexpr int Tp = TT - K + 1;
  constexpr int NS = (Tp + 1) >> 1;
  // (the loop over output channels stays ROLLED: fully unrolled, the scheduler hoisted every LDS operand of the C x NS
  // steps to the top of one giant block -- 256 VGPRs and 776 bytes of scratch in the output pass; UCP: PASS 5 unrolls it
  // to keep its sums in registers across tiles)
#pragma unroll UCP ? CC : 1
  for (int cp = 0; cp < CC; ++cp) {
    float w[CC][K];
#pragma unroll
    for (int c = 0; c < CC; ++c)
#pragma unroll
      for (int dt = 0; dt < K; ++dt) w[c][dt] = SM(wl + ((cp * CMAX + c) * PT_KP + dt));
    row_begin(cp);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      float xw[CC][K];
      float h0 = 0.f, h1 = 0.f;
#pragma unroll
      for (int c = 0; c < CC; ++c)
#pragma unroll
        for (int dt = 0; dt < K; ++dt) {
          xw[c][dt] = xh[c][2 * s + dt];
          if ((c + dt) & 1) h1 += w[c][dt] * xw[c][dt];
          else h0 += w[c][dt] * xw[c][dt];
        }
      const bool live = (2 * s + 1 < Tp) ? true : half == 0;
      entry(cp, 2 * s + half, live, h0 + h1, xw);
    }
    row_end(cp);
  }
}

// PASS 0: BatchNorm3d statistics   1: BatchNorm2d statistics   2: output (training or inference)
// PASS 3: backward sums of LayerNorm / BatchNorm2d   4: dW of the second convolutions + BatchNorm3d sums (+ dz scratch)
// PASS 5: dW of the first convolutions
// CMAX: compile-time bound of the input channels (4 or 8). MT: 32-channel tiles of Cout (1 or 2). NE: 32-entry tiles per
// branch (PASS 4 only; 1..3). CC, TT: compile-time (C, T) of the register variant (pt_rows_reg), 0 = the generic kernel.
template <int PASS, int CMAX, int MT, int NE, int CC = 0, int TT = 0>
#ifndef PT_MINB
#define PT_MINB 1
#endif
#ifndef PT_REG_MINB
#define PT_REG_MINB 2
#endif
#ifndef PT_REG_MINB5
#define PT_REG_MINB5 2  // (three blocks per CU: 168 registers + 84 bytes of scratch -- 49 MB of scratch writes per launch in the
                        // FETCH / WRITE counters -- and 72.2 us against 69.5 at two: 190 registers, no scratch)
#endif
#ifndef PT_REG_MINB4
#define PT_REG_MINB4 2  // (PASS 4 at two blocks per CU: 256 registers with 60 bytes of scratch, 81 KB of LDS; 193 -> 145 us at batch 32)
#endif
// (PASS 3 at C <= 4, Cout <= 32 fits 251 VGPRs without its 48 AGPR copies: two blocks per CU instead of one)
// (the output pass at three blocks per CU -- 168 VGPRs, 128 bytes of scratch -- measured 1-4 % faster: not worth the spills)
// (register variant, measured per pass at batch 32: PASS 3 at three blocks per CU spills 68 bytes and is slower, 67 -> 75 us;
// the C = 4 inference cube at three spills 64 bytes and is faster, 239 -> 202 us)
__global__ __launch_bounds__(256, CC > 0 ? (PASS == 4 ? PT_REG_MINB4 : PASS == 5 ? PT_REG_MINB5 : CC == 4 ? 3 : PT_REG_MINB) : (PASS == 3 && MT == 1 && CMAX == 4) ? 2 : PT_MINB)
void cn_pretime_kernel(const CnPtArgs a) {
  constexpr bool REG = CC > 0;
  constexpr int NC = REG ? CC : CMAX;  // input channels the unrolled loops walk (the generic kernel pads to CMAX)
  float* const lds = pt_smem;  // (only for the block-cooperative copy at the top and the parked doubles at the end)
  __shared__ int s_flag;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;
  const int C = REG ? CC : a.C, T = REG ? TT : a.T, Cout = a.Cout, HW = a.HW;
  constexpr int CP = 32 * MT;
  const int pcol = wid * 32 + l32;  // pixel column inside the block tile
  const int T3 = T - 2, T5 = T - 4;
  const int E3 = C * T3, E5 = C * T5, E = E3 + E5;
  const int EP3 = pt_ceil32(E3), EP = EP3 + pt_ceil32(E5);
  const int NS3 = C * ((T3 + 1) >> 1);  // wave steps of branch 3
  const PtLds L = pt_lds(PASS, C, T, Cout, CMAX, REG);
  const int wa_l = L.wa;
  const int c3_l = L.c3;
  const int k3_l = L.k3;
  const int wA_l = L.wA;
  const int wB_l = L.wB;
  const int c2_l = L.c2;
  const int ln_l = L.ln;
  const int k2_l = L.k2;
  const int xs = L.xs;
  const int AR = REG ? EP3 + E5 : EP;  // a-rows of a wave's slab
  const int as_w = REG ? L.as_ + wid * (AR + 2 * CP) * PT_LP : L.as_ + wid * EP * PT_LP;
  const int drs_w = REG ? as_w + AR * PT_LP : L.drs + wid * 2 * CP * PT_LP;
  const int lacc = L.lacc;
  const int NV = pt_nvals(PASS, C, Cout, CMAX);  // values of the block's ticket row
  // ... and of a wave's LDS accumulators: PASS 1 / 3 keep their per-channel rows at pitch CP there, so that the
  // butterfly's one LDS add per lane needs no `channel < Cout` predicate (the padded slots are never copied out)
  const int NVL = pt_nvals(PASS, C, CP, CMAX);
  const float vN = 1.0f / (float)Cout;

  PT_ST(0);
  // ---- the table image -> LDS, once per block: 16-byte loads, four in flight per thread (building the tables in
  // every block -- index arithmetic plus dependent scalar-indexed loads -- was ~25 us at the head of each block) ----
  {
    const int n4 = (pt_img_floats(PASS, C, T, Cout, CMAX) + 3) >> 2;
    const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(a.img);
    f32x4* dst = reinterpret_cast<f32x4*>(lds);
    for (int i0 = tid; i0 < n4; i0 += 256 * 4) {
      f32x4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int k = i0 + i * 256; v[i] = src[k < n4 ? k : n4 - 1]; }
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int k = i0 + i * 256; if (k < n4) dst[k] = v[i]; }
    }
  }
  for (int i = tid; i < 4 * NVL; i += 256) SM(lacc + (i)) = 0.f;
  if (!REG)
    for (int i = tid; i < C * PT_TPAD * PT_PXB; i += 256) {  // the pad rows of the x tile (never written again)
      const int c = i / (PT_TPAD * PT_PXB), r = i - c * (PT_TPAD * PT_PXB);
      SM(xs + ((c * (T + PT_TPAD) + T) * PT_PXB + r)) = 0.f;
    }
  if (REG) __syncthreads();  // the staged tables become visible; the waves do not meet again before the final reduction

  const int my = lacc + wid * NVL;  // this wave's accumulators (LDS offset)
  auto wsum = [&](int v, float val) {  // full-wave sum (both halves belong to the same value)
    // (ds_add_f32, no return value: nothing waits for it -- a read-modify-write here exposed one LDS round trip per
    // VALUE, 100-200 of them per tile; the address belongs to this wave alone, so the order of the adds is the program's)
    const float t = cn_wave_sum_to_lane63(val);
    if (lane == 63) atomicAdd(&SM(my + (v)), t);
  };
  // per-half sums of the 16 accumulator registers of a lane (channels 32 mt + pt_row(j, half)) into row `row` of the
  // wave's accumulators: lane l < 16 of each half ends up with the total of register l
  auto hsum16 = [&](int row, int mt, const f32x16& vals, int half_, int l32_) {
    const float t = pt_hsum16(vals, l32_);
    if (l32_ < 16) atomicAdd(&SM(my + (row * CP + mt * 32 + pt_row(l32_, half_))), t);
  };

  f32x16 accW[2][MT][NE];  // PASS 4: dWb accumulators (cout x entry tiles), summed over the wave's tiles
  if (PASS == 4) {
#pragma unroll
    for (int brn = 0; brn < 2; ++brn)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int et = 0; et < NE; ++et)
#pragma unroll
          for (int j = 0; j < 16; ++j) accW[brn][mt][et][j] = 0.f;
  }

  // PASS 5, register variant: the first-convolution weight-gradient sums of ALL output channels stay in registers across
  // the block's tiles (72 at C = 3) and are reduced over the wave once, after the loop: per tile that was 72 wave sums of
  // 6 DPP steps + an LDS add each, as many vector instructions as the multiply-adds they finish
  float gp3[REG && PASS == 5 ? CC : 1][NC][3], gp5[REG && PASS == 5 ? CC : 1][NC][5];
  if (REG && PASS == 5) {
#pragma unroll
    for (int cp = 0; cp < CC; ++cp)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) gp3[cp][c][dt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 5; ++dt) gp5[cp][c][dt] = 0.f;
      }
  }
  PT_ST(1);
  const int ntb = (int)((a.P + PT_PXB - 1) / PT_PXB);
  int st_i = 2;
  for (int tile = blockIdx.x; tile < ntb; tile += gridDim.x) {
    if (st_i < 26) { PT_ST(st_i); ++st_i; }
    const long p = (long)tile * PT_PXB + pcol;
    const bool valid = p < a.P;
    const float vm = valid ? 1.f : 0.f;
    const int b = valid ? (int)(p / HW) : 0;
    const int l = valid ? (int)(p - (long)b * HW) : 0;
    const unsigned pofs = (unsigned)(valid ? p : 0) * 4u;  // byte offset of the pixel inside a dz row (P < 2^30: host check)
    // The lane's coordinates are made OPAQUE once per tile: every table address of the body is lane part + constant, all
    // of them loop-invariant, and the compiler hoisted ~100 of them out of the tile loop as separate registers (PASS 4:
    // 464 bytes of scratch at two blocks per CU). Recomputed per tile they fold into the instructions' offset fields.
    int hv_ = half, lv_ = l32;
    asm volatile("" : "+v"(hv_), "+v"(lv_));
    {
    const int half = hv_, l32 = lv_;
    float xh[REG ? CC : 1][REG ? TT : 1];
    auto load_x = [&]() {
      // (nothing writes LDS inside this loop any more, so the compiler hoisted every table read of the tile body --
      // ~300 loop-invariant registers, 400-800 bytes of scratch -- out of it: the clobber keeps them where they are used)
      asm volatile("" ::: "memory");
      // the pixel's cube -> registers, shifted by the lane's parity: xh[c][j] = x[c][half + j] (j = T - 1 of the odd
      // parity would be row T: it re-reads row T - 1, a value only dead entries touch); C T loads in flight per lane
      // (addresses: a UNIFORM row pointer per load -- scalar arithmetic -- plus one 32-bit per-lane byte offset, the
      // global_load saddr form; with per-lane 64-bit pointers the compiler kept all 36 of them live: 72 registers)
      const unsigned vo1 = (unsigned)(((long)b * a.xbs + l) * 4);
      const unsigned vo = vo1 + (unsigned)half * (unsigned)HW * 4u;
      const char* xb = reinterpret_cast<const char*>(a.x);
#pragma unroll
      for (int c = 0; c < CC; ++c)
#pragma unroll
        for (int j = 0; j < TT; ++j) {
          const char* rowp = xb + (long)(c * TT + j) * HW * 4;
          const float v = *reinterpret_cast<const float*>(rowp + (j < TT - 1 ? vo : vo1));
          xh[c][j] = valid ? v : 0.f;
        }
    };
    if constexpr (REG) {
      load_x();
    } else {
    __syncthreads();  // previous tile's LDS reads are done (first time: the staged weights become visible)
    {
      // x tile: the two halves of a wave load alternate rows of the wave's 32 pixels; twelve loads in flight per lane
      const float* xp = a.x + (long)b * a.xbs + l;
      const int CT = C * T;
      for (int row0 = half; row0 < CT; row0 += 24) {
        float v[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          const int row = row0 + 2 * i;
          v[i] = xp[(long)(row < CT ? row : CT - 1) * HW];
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          const int row = row0 + 2 * i;
          const int c = row / T;  // (row = c * T + t -> LDS row c * (T + PT_TPAD) + t)
          if (row < CT) SM(xs + ((row + c * PT_TPAD) * PT_PXB + pcol)) = valid ? v[i] : 0.f;
        }
      }
    }
    __syncthreads();
    }
    // the first-convolution stack of this build: register variant or LDS-window variant
    auto rows = [&](auto kc, int wl, auto&& rb, auto&& en, auto&& re) {
      constexpr int K = decltype(kc)::value;
      if constexpr (REG) pt_rows_reg<K, CC, TT, CMAX, PASS == 5>(wl, xh, half, rb, en, re);
      else pt_rows<K, CMAX>(wl, xs, C, T, pcol, half, rb, en, re);
    };
    if (st_i < 26) { PT_ST(st_i); ++st_i; }

    if (PASS == 0) {
      float s = 0.f, q = 0.f;
      int slot = 0;
      auto rb = [&](int) { s = 0.f; q = 0.f; };
      auto en = [&](int, int, bool live, float h, auto&) { if (live) { s += h; q += h * h; } };
      auto re = [&](int) { wsum(slot, s); wsum(slot + 1, q); slot += 2; };
      rows(std::integral_constant<int, 3>{}, wa_l, rb, en, re);
      rows(std::integral_constant<int, 5>{}, wa_l + C * CMAX * PT_KP, rb, en, re);
      continue;
    }
    if (PASS == 5) {
      // dh = g3 rho (dz - c0 - hh c1); dWa[cp][c][dt] += sum_px sum_tp dh * x[c][tp + dt] (the window IS x[c][tp + dt])
      int ebase = 0, vbase = 0;
      auto branch = [&](auto kc, int wl, int brn, auto& gp) {
        constexpr int K = decltype(kc)::value;
        float g[NC][K];
        float rho = 0.f, off = 0.f, g3 = 0.f, c0 = 0.f, c1 = 0.f;
        const int Tp = T - K + 1;
        auto rb = [&](int cp) {
          const f32x4 cc = *reinterpret_cast<const f32x4*>(&SM(c3_l + (brn * C + cp) * 4));
          rho = cc[0]; off = cc[1]; g3 = cc[2];
          c0 = SM(k3_l + ((brn * C + cp) * 2)); c1 = SM(k3_l + ((brn * C + cp) * 2 + 1));
#pragma unroll
          for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int dt = 0; dt < K; ++dt) g[c][dt] = 0.f;
        };
        auto en = [&](int cp, int tp, bool live, float h, auto& xw) {
          const float hh = h * rho + off;
          const int e = ebase + cp * Tp + (live ? tp : 0);
          // (uniform row pointer + 32-bit lane offset: the saddr form, as for x)
          const float* dzp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.dz) + (long)e * a.P * 4 + pofs);
          const float dzv = (valid && live) ? *dzp : 0.f;
          const float dh = (valid && live) ? g3 * rho * (dzv - c0 - hh * c1) : 0.f;
#pragma unroll
          for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int dt = 0; dt < K; ++dt) {
              if constexpr (REG) gp[cp][c][dt] += dh * xw[c][dt];  // (cp is a constant here: the stack is unrolled)
              else g[c][dt] += dh * xw[c][dt];
            }
        };
        auto re = [&](int cp) {
          if constexpr (!REG) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
              for (int dt = 0; dt < K; ++dt) wsum(vbase + (cp * CMAX + c) * PT_KP + dt, g[c][dt]);
          }
        };
        rows(kc, wl, rb, en, re);
        ebase += C * Tp;
        vbase += C * CMAX * PT_KP;
      };
      branch(std::integral_constant<int, 3>{}, wa_l, 0, gp3);
      branch(std::integral_constant<int, 5>{}, wa_l + C * CMAX * PT_KP, 1, gp5);
      continue;
    }

    // ---- PASS 1..4: first convolutions -> SiLU(BatchNorm3d) -> second convolutions on the matrix pipe ----
    f32x16 acc[2][MT];
#pragma unroll
    for (int brn = 0; brn < 2; ++brn)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[brn][mt][j] = 0.f;
    {
      float rho = 0.f, off = 0.f, g3 = 0.f, b3 = 0.f;
      int gs = 0;
      auto branch = [&](auto kc, int wl, auto bc, int slot0) {
        constexpr int K = decltype(kc)::value;
        constexpr int brn = decltype(bc)::value;
        const int Tp = T - K + 1;
        auto rb = [&](int cp) {
          const f32x4 cc = *reinterpret_cast<const f32x4*>(&SM(c3_l + (brn * C + cp) * 4));
          rho = cc[0]; off = cc[1]; g3 = cc[2]; b3 = cc[3];
        };
        auto en = [&](int cp, int tp, bool live, float h, auto&) {
          const float aval = live ? pt_silu(g3 * (h * rho + off) + b3) : 0.f;
          if (PASS == 4 && live) SM(as_w + ((slot0 + cp * Tp + tp) * PT_LP + l32)) = aval * vm;
          const int wv = wA_l + (gs * 2 + half) * CP + l32;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[brn][mt] = pt_mfma(SM(wv + (mt * 32)), aval, acc[brn][mt]);
          ++gs;
        };
        auto re = [&](int) {};
        rows(kc, wl, rb, en, re);
      };
      branch(std::integral_constant<int, 3>{}, wa_l, std::integral_constant<int, 0>{}, 0);
      branch(std::integral_constant<int, 5>{}, wa_l + C * CMAX * PT_KP, std::integral_constant<int, 1>{}, EP3);
    }
    if (st_i < 26) { PT_ST(st_i); ++st_i; }
    // this lane: pixel l32, output channels o(mt, j) = 32 mt + pt_row(j, half)
    if (PASS == 1) {
#pragma unroll
      for (int brn = 0; brn < 2; ++brn)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
        {
          f32x16 v1, v2;
#pragma unroll
          for (int j = 0; j < 16; ++j) { v1[j] = acc[brn][mt][j] * vm; v2[j] = v1[j] * v1[j]; }
          hsum16(brn * 2, mt, v1, half, l32);
          hsum16(brn * 2 + 1, mt, v2, half, l32);
        }
      continue;
    }
    PT_PHASE();
    // ---- BatchNorm2d + SiLU, branch sum, LayerNorm over Cout ----
    // acc <- rhat (normalised); vv = affine BatchNorm output; u = sum of the activations (0 for padded channels)
    f32x16 vv[2][MT], u[MT];
    float m = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int o = mt * 32 + pt_row(j, half);
        float uj = 0.f;
#pragma unroll
        for (int brn = 0; brn < 2; ++brn) {
          const f32x4 cc = *reinterpret_cast<const f32x4*>(&SM(c2_l + (brn * CP + o) * 4));
          acc[brn][mt][j] = acc[brn][mt][j] * cc[0] + cc[1];
          vv[brn][mt][j] = cc[2] * acc[brn][mt][j] + cc[3];
          uj += pt_silu(vv[brn][mt][j]);
        }
        uj = o < Cout ? uj : 0.f;
        u[mt][j] = uj;
        m += uj;
        if ((j & 3) == 3) PT_PHASE();
      }
    m += __shfl_xor(m, 32, 64);
    m *= vN;
    float var = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int o = mt * 32 + pt_row(j, half);
        const float d = o < Cout ? u[mt][j] - m : 0.f;
        var += d * d;
      }
    var += __shfl_xor(var, 32, 64);
    const float rL = 1.0f / sqrtf(var * vN + a.epsL);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int o = mt * 32 + pt_row(j, half);
        u[mt][j] = o < Cout ? (u[mt][j] - m) * rL : 0.f;  // u <- uhat
      }
    PT_PHASE();
    if (PASS == 2) {
      if (valid) {
        if (a.out_kind == 0) {
          float* yp = reinterpret_cast<float*>(a.y) + (long)b * a.y_stride + l;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              const int o = mt * 32 + pt_row(j, half);
              if (o < Cout) yp[(long)o * HW] = SM(ln_l + (o)) * u[mt][j] + SM(ln_l + (CP + o));
            }
        } else {
          bf16_t* yp = reinterpret_cast<bf16_t*>(a.y) + p * a.y_stride;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // four consecutive channels 32 mt + 8 q + 4 half + (0..3): one 8-byte store
              const int o = mt * 32 + 8 * q + 4 * half;
              if (o < Cout) {
                u32x2 pk;
                pk[0] = cn_pack_bf16(SM(ln_l + (o)) * u[mt][4 * q] + SM(ln_l + (CP + o)), SM(ln_l + (o + 1)) * u[mt][4 * q + 1] + SM(ln_l + (CP + o + 1)));
                pk[1] = cn_pack_bf16(SM(ln_l + (o + 2)) * u[mt][4 * q + 2] + SM(ln_l + (CP + o + 2)),
                                     SM(ln_l + (o + 3)) * u[mt][4 * q + 3] + SM(ln_l + (CP + o + 3)));
                *reinterpret_cast<u32x2*>(yp + o) = pk;
              }
            }
        }
      }
      continue;
    }
    PT_PHASE();
    // ---- backward: LayerNorm, SiLU, BatchNorm2d ----
    f32x16 dyv[MT];
    if (a.out_kind == 0) {
      const float* dp = reinterpret_cast<const float*>(a.dy) + (long)b * a.dy_stride + l;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int o = mt * 32 + pt_row(j, half);
          dyv[mt][j] = (valid && o < Cout) ? dp[(long)(o < Cout ? o : 0) * HW] : 0.f;
        }
    } else {
      const bf16_t* dp = reinterpret_cast<const bf16_t*>(a.dy) + (valid ? p : 0) * a.dy_stride;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int o = mt * 32 + 8 * q + 4 * half;
          const u32x2 pk = *reinterpret_cast<const u32x2*>(dp + (o < Cout ? o : 0));
          const float on = (valid && o < Cout) ? 1.f : 0.f;
          dyv[mt][4 * q] = cn_bf16_lo(pk[0]) * on;
          dyv[mt][4 * q + 1] = cn_bf16_hi(pk[0]) * on;
          dyv[mt][4 * q + 2] = cn_bf16_lo(pk[1]) * on;
          dyv[mt][4 * q + 3] = cn_bf16_hi(pk[1]) * on;
        }
    }
    PT_PHASE();
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int o = mt * 32 + pt_row(j, half);
        const float gj = dyv[mt][j] * SM(ln_l + (o));
        s1 += gj;
        s2 += gj * u[mt][j];
      }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    s1 *= vN; s2 *= vN;
    if (PASS == 3) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        f32x16 du, t1, t2;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int o = mt * 32 + pt_row(j, half);
          du[j] = o < Cout ? rL * (dyv[mt][j] * SM(ln_l + (o)) - s1 - u[mt][j] * s2) : 0.f;
        }
#pragma unroll
        for (int brn = 0; brn < 2; ++brn) {
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            t1[j] = du[j] * pt_silu_grad(vv[brn][mt][j]);
            t2[j] = t1[j] * acc[brn][mt][j];
          }
          hsum16(brn * 2, mt, t1, half, l32);
          hsum16(brn * 2 + 1, mt, t2, half, l32);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) t1[j] = dyv[mt][j] * u[mt][j];
        hsum16(4, mt, t1, half, l32);
        hsum16(5, mt, dyv[mt], half, l32);
      }
      continue;
    }
    PT_PHASE();
    // ---- PASS 4 ----
    // dr (kept in vv), and transposed into LDS [cout][pixel] for the dWb contraction
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int o = mt * 32 + pt_row(j, half);
        const float du = o < Cout ? rL * (dyv[mt][j] * SM(ln_l + (o)) - s1 - u[mt][j] * s2) : 0.f;
#pragma unroll
        for (int brn = 0; brn < 2; ++brn) {
          const float dv = du * pt_silu_grad(vv[brn][mt][j]);
          const f32x4 cc = *reinterpret_cast<const f32x4*>(&SM(c2_l + (brn * CP + o) * 4));
          const float c0 = SM(k2_l + ((brn * CP + o) * 2)), c1 = SM(k2_l + ((brn * CP + o) * 2 + 1));
          const float dr = cc[2] * cc[0] * (dv - c0 - acc[brn][mt][j] * c1) * vm;
          vv[brn][mt][j] = dr;
          SM(drs_w + ((brn * CP + o) * PT_LP + l32)) = dr;
        }
      }
    PT_PHASE();
    // dWb[cout][entry] += sum_px dr[cout][px] a[entry][px]: K = pixels, two per step; A = dr^T, B = a^T out of LDS
#pragma unroll
    for (int brn = 0; brn < 2; ++brn) {
      const int ab = as_w + (brn ? EP3 : 0) * PT_LP;
      const int db = drs_w + brn * CP * PT_LP;
      const int net = ((brn ? E5 : E3) + 31) >> 5;
      for (int t = 0; t < 16; ++t) {
        const int px = 2 * t + half;
#pragma unroll
        for (int et = 0; et < NE; ++et) {
          if (et < net) {
            const float bv = SM(ab + ((et * 32 + l32) * PT_LP + px));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
              accW[brn][mt][et] = pt_mfma(SM(db + ((mt * 32 + l32) * PT_LP + px)), bv, accW[brn][mt][et]);
          }
        }
      }
    }
    PT_PHASE();
    // da[entry][px] = sum_cout wb[cout][entry] dr[cout][px]: K = couts, the accumulator registers of dr ARE the B
    // operands (k-step (mt, j): couts 32 mt + pt_row(j, 0) and + pt_row(j, 1)); result transposed through LDS into
    // the (pixel, entry parity) layout of the first convolutions (it replaces a[][] in as_w)
#pragma unroll
    for (int brn = 0; brn < 2; ++brn) {
      const int net = ((brn ? E5 : E3) + 31) >> 5;
#pragma unroll
      for (int et = 0; et < NE; ++et) {
        if (et < net) {
          f32x16 da;
#pragma unroll
          for (int j = 0; j < 16; ++j) da[j] = 0.f;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              const int o = mt * 32 + pt_row(j, half);
              da = pt_mfma(SM(wB_l + (o * EP + (brn ? EP3 : 0) + et * 32 + l32)), vv[brn][mt][j], da);
            }
#pragma unroll
          for (int j = 0; j < 16; ++j)
            SM(as_w + (((brn ? EP3 : 0) + et * 32 + pt_row(j, half)) * PT_LP + l32)) = da[j];
        }
      }
    }
    PT_PHASE();
    // dz = da * silu'(z), BatchNorm3d sums, dz -> scratch
    // (loading the cube AGAIN here instead of keeping its 36 registers live through the matrix-pipe phases made the
    // allocation worse -- 192 bytes of scratch against 60 at two blocks per CU -- and was dropped)
    {
      float rho = 0.f, off = 0.f, g3 = 0.f, b3 = 0.f, s = 0.f, q = 0.f;
      int slot = 0, ebase = 0;
      auto branch = [&](auto kc, int wl, int brn, int slot0) {
        constexpr int K = decltype(kc)::value;
        const int Tp = T - K + 1;
        auto rb = [&](int cp) {
          const f32x4 cc = *reinterpret_cast<const f32x4*>(&SM(c3_l + (brn * C + cp) * 4));
          rho = cc[0]; off = cc[1]; g3 = cc[2]; b3 = cc[3];
          s = 0.f; q = 0.f;
        };
        auto en = [&](int cp, int tp, bool live, float h, auto&) {
          if (live) {
            const float hh = h * rho + off;
            const int loc = cp * Tp + tp;
            const float dzv = SM(as_w + ((slot0 + loc) * PT_LP + l32)) * pt_silu_grad(g3 * hh + b3);
            s += dzv;
            q += dzv * hh;
            if (valid) *reinterpret_cast<float*>(reinterpret_cast<char*>(a.dz) + (long)(ebase + loc) * a.P * 4 + pofs) = dzv;
          }
        };
        auto re = [&](int) { wsum(slot, s); wsum(slot + 1, q); slot += 2; };
        rows(kc, wl, rb, en, re);
        ebase += C * Tp;
      };
      branch(std::integral_constant<int, 3>{}, wa_l, 0, 0);
      branch(std::integral_constant<int, 5>{}, wa_l + C * CMAX * PT_KP, 1, EP3);
    }
    }
  }

  if (REG && PASS == 5) {
#pragma unroll
    for (int cp = 0; cp < CC; ++cp)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) wsum((cp * CMAX + c) * PT_KP + dt, gp3[cp][c][dt]);
#pragma unroll
        for (int dt = 0; dt < 5; ++dt) wsum(C * CMAX * PT_KP + (cp * CMAX + c) * PT_KP + dt, gp5[cp][c][dt]);
      }
  }
  // ---- block row -> two-level last-block reduction -> finish ----
  PT_ST(30);
  if (PASS == 2) return;
  const int blk = blockIdx.x;
  __syncthreads();
  {
    const int base = PASS == 4 ? Cout * E : 0;
    for (int v = tid; v < NV; v += 256) {
      int lv = v;
      if (PASS == 1 || PASS == 3) { const int row = v / Cout; lv = row * CP + (v - row * Cout); }
      cn_t2_store(a.tk, blk, base + v,
                  (SM(lacc + (lv)) + SM(lacc + (NVL + lv))) + (SM(lacc + (2 * NVL + lv)) + SM(lacc + (3 * NVL + lv))));
    }
  }
  if (PASS == 4) {
    // the four waves' dWb tiles -> LDS (wave-private slabs over the dead x / a / dr regions) -> summed in wave order
    const int slab = L.xs;
    const int W4 = Cout * E;
    __syncthreads();
    for (int w2 = 0; w2 < 4; ++w2) {
      if (wid == w2) {
#pragma unroll
        for (int brn = 0; brn < 2; ++brn) {
          const int Eb = brn ? E5 : E3;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int et = 0; et < NE; ++et)
#pragma unroll
              for (int j = 0; j < 16; ++j) {
                const int o = mt * 32 + pt_row(j, half), e = et * 32 + l32;
                if (o < Cout && e < Eb) {
                  const int col = (brn ? Cout * E3 : 0) + o * Eb + e;
                  SM(slab + (col)) = w2 == 0 ? accW[brn][mt][et][j] : SM(slab + (col)) + accW[brn][mt][et][j];
                }
              }
        }
      }
      __syncthreads();
    }
    for (int col = tid; col < W4; col += 256) cn_t2_store(a.tk, blk, col, SM(slab + (col)));
  }
  // finish (the last-arriving block): statistics / coefficients / parameter gradients
  const double cntP = (double)a.P;
  cn_t2_reduce_fn(a.tk, blk, &s_flag, [&](int col, double tot) {
    if (PASS == 0 || PASS == 1 || PASS == 3) {
      reinterpret_cast<double*>(lds)[col] = tot;  // {stat 0, stat 1} pairs sit in different columns: park, finish below
    } else if (PASS == 4) {
      if (col < Cout * E) {
        const int brn = col >= Cout * E3 ? 1 : 0;
        a.br[brn].dwb[col - (brn ? Cout * E3 : 0)] += (float)tot;
      } else {
        reinterpret_cast<double*>(lds)[col - Cout * E] = tot;
      }
    } else {  // PASS 5: value (brn, cp, c, dt) at [(brn*C + cp)*CMAX + c]*PT_KP + dt
      const int dt = col % PT_KP;
      int q = col / PT_KP;
      const int c = q % CMAX;
      q /= CMAX;
      const int cp = q % C, brn = q / C;
      const int k = brn ? 5 : 3;
      if (c < C && dt < k) a.br[brn].dwa[(cp * C + c) * k + dt] += (float)tot;
    }
  });
  if (PASS == 5 || !s_flag) return;
  __syncthreads();
  const double* tot = reinterpret_cast<const double*>(lds);
  if (PASS == 0 || PASS == 4) {
    for (int i = tid; i < 2 * C; i += 256) {
      const int brn = i / C, cp = i - brn * C;
      const CnPtBranch& r = a.br[brn];
      const double v0 = tot[(brn * C + cp) * 2], v1 = tot[(brn * C + cp) * 2 + 1];
      const double cnt = cntP * (double)(brn ? T5 : T3);
      if (PASS == 0) {
        const double md = v0 / cnt;
        double var = v1 / cnt - md * md;
        if (var < 0.0) var = 0.0;
        r.mean3[cp] = (float)md;
        r.rstd3[cp] = (float)(1.0 / sqrt(var + (double)a.eps3));
        {  // the next passes' table image
          const PtLds LI = pt_lds(4, C, T, Cout, CMAX);
          const float rho = (float)(1.0 / sqrt(var + (double)a.eps3));
          a.img[LI.c3 + (brn * C + cp) * 4 + 0] = rho;
          a.img[LI.c3 + (brn * C + cp) * 4 + 1] = -(float)md * rho;
        }
        if (r.rm3 != nullptr) {
          const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
          r.rm3[cp] = (1.f - a.mom3) * r.rm3[cp] + a.mom3 * (float)md;
          r.rv3[cp] = (1.f - a.mom3) * r.rv3[cp] + a.mom3 * (float)unb;
        }
      } else {
        a.coef3[(brn * 2) * C + cp] = a.training ? (float)(v0 / cnt) : 0.f;
        a.coef3[(brn * 2 + 1) * C + cp] = a.training ? (float)(v1 / cnt) : 0.f;
        {
          const PtLds LI = pt_lds(4, C, T, Cout, CMAX);
          a.img[LI.k3 + (brn * C + cp) * 2 + 0] = a.training ? (float)(v0 / cnt) : 0.f;
          a.img[LI.k3 + (brn * C + cp) * 2 + 1] = a.training ? (float)(v1 / cnt) : 0.f;
        }
        r.dg3[cp] += (float)v1;
        r.db3[cp] += (float)v0;
      }
    }
  } else {  // PASS 1 / 3: columns [(brn*2 + stat)*Cout + o] (+ LayerNorm parameter gradients in PASS 3)
    for (int i = tid; i < 2 * Cout; i += 256) {
      const int brn = i / Cout, o = i - brn * Cout;
      const CnPtBranch& r = a.br[brn];
      const double v0 = tot[(brn * 2) * Cout + o], v1 = tot[(brn * 2 + 1) * Cout + o];
      if (PASS == 1) {
        const double md = v0 / cntP;
        double var = v1 / cntP - md * md;
        if (var < 0.0) var = 0.0;
        r.mean2[o] = (float)md;
        r.rstd2[o] = (float)(1.0 / sqrt(var + (double)a.eps2));
        {
          const PtLds LI = pt_lds(4, C, T, Cout, CMAX);
          const float rho = (float)(1.0 / sqrt(var + (double)a.eps2));
          a.img[LI.c2 + (brn * CP + o) * 4 + 0] = rho;
          a.img[LI.c2 + (brn * CP + o) * 4 + 1] = -(float)md * rho;
        }
        if (r.rm2 != nullptr) {
          const double unb = cntP > 1.0 ? var * cntP / (cntP - 1.0) : var;
          r.rm2[o] = (1.f - a.mom2) * r.rm2[o] + a.mom2 * (float)md;
          r.rv2[o] = (1.f - a.mom2) * r.rv2[o] + a.mom2 * (float)unb;
        }
      } else {
        a.coef2[(brn * 2) * Cout + o] = a.training ? (float)(v0 / cntP) : 0.f;
        a.coef2[(brn * 2 + 1) * Cout + o] = a.training ? (float)(v1 / cntP) : 0.f;
        {
          const PtLds LI = pt_lds(4, C, T, Cout, CMAX);
          a.img[LI.k2 + (brn * CP + o) * 2 + 0] = a.training ? (float)(v0 / cntP) : 0.f;
          a.img[LI.k2 + (brn * CP + o) * 2 + 1] = a.training ? (float)(v1 / cntP) : 0.f;
        }
        r.dg2[o] += (float)v1;
        r.db2[o] += (float)v0;
      }
    }
    if (PASS == 3)
      for (int o = tid; o < Cout; o += 256) {
        a.dgL[o] += (float)tot[4 * Cout + o];
        a.dbL[o] += (float)tot[5 * Cout + o];
      }
  }
}

// ---- host side -----------------------------------------------------------------------------------------------------
static inline int pt_cmax(int C) { return C <= 4 ? 4 : 8; }
static inline int pt_row_width(int PASS, int C, int T, int Cout) {
  const int E = C * (T - 2) + C * (T - 4);
  switch (PASS) {
    case 0: return 4 * C;
    case 1: return 4 * Cout;
    case 3: return 6 * Cout;
    case 4: return Cout * E + 4 * C;
    case 5: return 2 * C * pt_cmax(C) * PT_KP;
    default: return 0;
  }
}
static inline bool pt_supported(int C, int T, int Cout) {
  return C >= 1 && C <= PT_MAX_C && T >= 5 && Cout >= 8 && Cout <= 64 && (Cout & 7) == 0;
}
static inline int pt_ne(int C, int T) { return (pt_ceil32(C * (T - 2)) >> 5); }  // entry tiles of the longer branch
// (C, T) with a register-variant instantiation (pt_rows_reg): the reference's default cube; everything else runs the
// generic kernel. CN_PRETIME_REG=0 forces the generic kernel (A/B).
// The inference cube of the scene predictor (C = 4, T = 25: 100 registers of x) has the output pass only.
static inline bool pt_reg(int PASS, int C, int T, int Cout) {
  static const int on = [] { const char* e = getenv("CN_PRETIME_REG"); return e ? atoi(e) : 1; }();
  if (!on || Cout > 32) return false;
  return (C == 3 && T == 12) || (PASS == 2 && C == 4 && T == 25);
}
static inline size_t pt_shmem(int PASS, int C, int T, int Cout, bool reg = false) {
  size_t sh = (size_t)pt_lds(PASS, C, T, Cout, pt_cmax(C), reg).total * 4;
  if (PASS == 4) {  // the end-of-kernel slab of the dWb tiles overlays the x / a / dr regions
    const PtLds L = pt_lds(PASS, C, T, Cout, pt_cmax(C), reg);
    const size_t slab = (size_t)(L.xs + Cout * (C * (T - 2) + C * (T - 4))) * 4;
    sh = sh < slab ? slab : sh;
  }
  const size_t park = (size_t)(6 * Cout > 4 * C ? 6 * Cout : 4 * C) * 8;  // the finish phase parks doubles at the head
  return sh < park ? park : sh;
}
// floats: [counters 64][ticket body for the widest row][coef2][coef3][dz]
static inline long pt_off_body() { return CN_T2_COUNTERS; }
static inline long pt_body_floats(int C, int T, int Cout) {
  int w = 0;
  for (int ps = 0; ps < 6; ++ps) { const int r = pt_row_width(ps, C, T, Cout); w = r > w ? r : w; }
  return (cn_t2_body_floats(PT_MAX_BLOCKS_REG, w) + 63) / 64 * 64;
}
extern "C" long cn_pretime_workspace_floats(int B, int C, int T, int HW, int Cout, int with_backward) {
  if (!pt_supported(C, T, Cout)) return -1;
  for (int ps = 0; ps < (with_backward ? 6 : 3); ++ps)  // every pass the caller may launch must fit the 160 KiB LDS
    if (pt_shmem(ps, C, T, Cout) > 160 * 1024) return -1;
  // dWb accumulator registers of the gradient pass: up to three 32-entry tiles per branch at Cout <= 32; the two-tile
  // Cout = 64 instantiation spills (1 KB per lane) and is left to the generic path for now
  if (with_backward && (pt_ne(C, T) > 3 || Cout > 32)) return -1;
  const long E = (long)C * (T - 2) + (long)C * (T - 4);
  long n = pt_off_body() + pt_body_floats(C, T, Cout) + 4L * pt_ceil32(Cout) + 4L * C + 128 +
           ((pt_lds(4, C, T, Cout, pt_cmax(C)).xs + 63) / 64 * 64);
  if (with_backward) n += E * (long)B * HW;
  return n;
}

static int pt_fill(CnPtArgs& a, const float* x, long xbs, const void* const* params, float* const* stats, int B, int C,
                   int T, int HW, int Cout, int training, const float* bn, float eps_ln, float* ws, long ws_floats,
                   int with_backward) {
  if (!pt_supported(C, T, Cout) || B <= 0 || HW <= 0) return CN_ERR_ARG;
  if ((long)B * HW >= (1L << 30)) return CN_ERR_ARG;  // (32-bit byte offsets of a pixel inside a row of the dz scratch)
  if (ws == nullptr || ws_floats < cn_pretime_workspace_floats(B, C, T, HW, Cout, with_backward)) return CN_ERR_ARG;
  a.x = x; a.xbs = xbs; a.B = B; a.C = C; a.T = T; a.HW = HW; a.Cout = Cout; a.P = (long)B * HW;
  a.training = training; a.eps3 = bn[0]; a.mom3 = bn[1]; a.eps2 = bn[2]; a.mom2 = bn[3]; a.epsL = eps_ln;
  a.ntiles = 0;
  a.coef2 = ws + pt_off_body() + pt_body_floats(C, T, Cout);
  a.coef3 = a.coef2 + 4L * pt_ceil32(Cout);
  a.img = a.coef3 + 64;
  a.dz = a.img + (pt_lds(4, C, T, Cout, pt_cmax(C)).xs + 63) / 64 * 64;
  for (int brn = 0; brn < 2; ++brn) {
    CnPtBranch& r = a.br[brn];
    const void* const* p = params + brn * 10;
    r.wa = (const float*)p[0]; r.wb = (const float*)p[1];
    r.g3 = (const float*)p[2]; r.b3 = (const float*)p[3]; r.rm3 = (float*)p[4]; r.rv3 = (float*)p[5];
    r.g2 = (const float*)p[6]; r.b2 = (const float*)p[7]; r.rm2 = (float*)p[8]; r.rv2 = (float*)p[9];
    r.mean3 = stats[brn * 4 + 0]; r.rstd3 = stats[brn * 4 + 1]; r.mean2 = stats[brn * 4 + 2]; r.rstd2 = stats[brn * 4 + 3];
    r.k = brn ? 5 : 3; r.Tp = T - r.k + 1;
    if (!training && (r.rm3 == nullptr || r.rv3 == nullptr || r.rm2 == nullptr || r.rv2 == nullptr)) return CN_ERR_ARG;
  }
  a.gL = (const float*)params[20]; a.bL = (const float*)params[21];
  return CN_OK;
}

template <int PASS>
static int pt_launch(CnPtArgs a, float* ws, hipStream_t stream) {
  const int ntb = (int)((a.P + PT_PXB - 1) / PT_PXB);
  // (the register variant addresses x with 32-bit per-lane byte offsets)
  const bool reg = pt_reg(PASS, a.C, a.T, a.Cout) && (long)a.B * a.xbs * 4 < (1L << 32);
  // (PASS 3 / 4 of the register variant and the C = 4 output pass hold two blocks per CU, PASS 5 three, the others four)
  const int maxb = !reg ? PT_MAX_BLOCKS : PASS == 4 ? 256 * PT_REG_MINB4 : PASS == 5 ? 256 * PT_REG_MINB5
                   : PASS == 3 ? 256 * PT_REG_MINB : a.C == 4 ? 768 : PT_MAX_BLOCKS_REG;
  // persistent blocks: weights are staged once per block; the register variant takes equal shares (2500 tiles on 768
  // blocks would be 4 rounds for 3.26 tiles of work: 834 blocks x 3)
  const int nblk = ntb <= maxb ? ntb : reg ? (ntb + (ntb + maxb - 1) / maxb - 1) / ((ntb + maxb - 1) / maxb) : maxb;
  const size_t shmem = pt_shmem(PASS, a.C, a.T, a.Cout, reg);
  if (shmem > 160 * 1024) return CN_ERR_LDS;
  const int W = pt_row_width(PASS, a.C, a.T, a.Cout);
  if (W > 0) a.tk = cn_t2_carve(reinterpret_cast<int*>(ws), ws + pt_off_body(), nblk, W);
  const int MT = a.Cout > 32 ? 2 : 1;
  const int NE = PASS == 4 ? pt_ne(a.C, a.T) : 1;
#define PT_GO(CM, MT_, NE_)                                                                                    \
  do {                                                                                                         \
    if (shmem > 64 * 1024)                                                                                     \
      (void)hipFuncSetAttribute((const void*)cn_pretime_kernel<PASS, CM, MT_, NE_>,                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);                       \
    CN_LAUNCH((cn_pretime_kernel<PASS, CM, MT_, NE_>), dim3(nblk), dim3(256), shmem, stream, a);              \
  } while (0)
#define PT_GO_MT(CM)                                                                                           \
  do {                                                                                                         \
    if constexpr (PASS == 4) {                                                                                 \
      if (MT == 2) PT_GO(CM, 2, 1);                                                                            \
      else if (NE == 1) PT_GO(CM, 1, 1); else if (NE == 2) PT_GO(CM, 1, 2); else PT_GO(CM, 1, 3);              \
    } else {                                                                                                   \
      if (MT == 2) PT_GO(CM, 2, 1); else PT_GO(CM, 1, 1);                                                      \
    }                                                                                                          \
  } while (0)
  const bool prof = cn_prof_on();  // per-pass times for tools/pretime_bench.py (kind 6: not a contraction kernel)
  if (prof) { cn_prof_name("cn_pretime_kernel<%d%s>", PASS, reg ? ", reg" : ""); cn_prof_before(stream); }
  if (reg && a.C == 3) {  // Cout <= 32, C T = 36: one cout tile, one entry tile per branch
    if (shmem > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)cn_pretime_kernel<PASS, 4, 1, 1, 3, 12>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)shmem);
    CN_LAUNCH((cn_pretime_kernel<PASS, 4, 1, 1, 3, 12>), dim3(nblk), dim3(256), shmem, stream, a);
  } else if (reg) {
    if constexpr (PASS == 2) CN_LAUNCH((cn_pretime_kernel<2, 4, 1, 1, 4, 25>), dim3(nblk), dim3(256), shmem, stream, a);
  } else if (pt_cmax(a.C) == 4) PT_GO_MT(4); else PT_GO_MT(8);
#undef PT_GO_MT
#undef PT_GO
  if (prof) cn_prof_after(stream, 6, 0.0);
  return CN_OK;
}

// params: HOST array of 22 device pointers: per branch (k = 3, then k = 5) {wa, wb, gamma3, beta3, running_mean3,
// running_var3, gamma2, beta2, running_mean2, running_var2}, then {ln_gamma, ln_beta}. stats: HOST array of 8 device
// pointers: per branch {mean3 [C], rstd3 [C], mean2 [Cout], rstd2 [Cout]} (written in training mode, read by backward).
// bn: HOST {eps3, momentum3, eps2, momentum2}. y: out_kind 0 fp32 NCHW (y_stride = batch stride), 1 bf16 NHWC (pixel
// stride). Returns CN_ERR_ARG for shapes outside the fused kernel (the caller keeps its generic path).
extern "C" int cn_pretime_fwd_f32(const float* x, long xbs, const void* const* params, float* const* stats, void* y,
                                  long y_stride, int out_kind, int B, int C, int T, int HW, int Cout, int training,
                                  const float* bn, float eps_ln, float* ws, long ws_floats, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  CnPtArgs a = {};
  int rc = pt_fill(a, x, xbs, params, stats, B, C, T, HW, Cout, training, bn, eps_ln, ws, ws_floats, 0);
  if (rc != CN_OK) return rc;
  a.y = y; a.y_stride = y_stride; a.out_kind = out_kind;
  if (pt_cmax(C) == 4) CN_LAUNCH(cn_pretime_pack_kernel<4>, dim3(8), dim3(256), 0, stream, a, training ? 0 : 1);
  else CN_LAUNCH(cn_pretime_pack_kernel<8>, dim3(8), dim3(256), 0, stream, a, training ? 0 : 1);
  if (training) {
    if ((rc = pt_launch<0>(a, ws, stream)) != CN_OK) return rc;
    if ((rc = pt_launch<1>(a, ws, stream)) != CN_OK) return rc;
  }
  if ((rc = pt_launch<2>(a, ws, stream)) != CN_OK) return rc;
  return cn_check_launch();
}

// grads: HOST array of 14 device pointers: per branch {dwa, dwb, dgamma3, dbeta3, dgamma2, dbeta2}, then
// {d ln_gamma, d ln_beta}; all ACCUMULATED. dy in the layout of y.
extern "C" int cn_pretime_bwd_f32(const float* x, long xbs, const void* const* params, float* const* stats,
                                  const void* dy, long dy_stride, int out_kind, float* const* grads, int B, int C, int T,
                                  int HW, int Cout, int training, const float* bn, float eps_ln, float* ws,
                                  long ws_floats, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  CnPtArgs a = {};
  int rc = pt_fill(a, x, xbs, params, stats, B, C, T, HW, Cout, training, bn, eps_ln, ws, ws_floats, 1);
  if (rc != CN_OK) return rc;
  a.dy = dy; a.dy_stride = dy_stride; a.out_kind = out_kind;
  for (int brn = 0; brn < 2; ++brn) {
    CnPtBranch& r = a.br[brn];
    float* const* g = grads + brn * 6;
    r.dwa = g[0]; r.dwb = g[1]; r.dg3 = g[2]; r.db3 = g[3]; r.dg2 = g[4]; r.db2 = g[5];
  }
  a.dgL = grads[12]; a.dbL = grads[13];
  if (pt_cmax(C) == 4) CN_LAUNCH(cn_pretime_pack_kernel<4>, dim3(8), dim3(256), 0, stream, a, 2);
  else CN_LAUNCH(cn_pretime_pack_kernel<8>, dim3(8), dim3(256), 0, stream, a, 2);
  if ((rc = pt_launch<3>(a, ws, stream)) != CN_OK) return rc;
  if ((rc = pt_launch<4>(a, ws, stream)) != CN_OK) return rc;
  if ((rc = pt_launch<5>(a, ws, stream)) != CN_OK) return rc;
  return cn_check_launch();
}
