// PreTimeReduction (reference: models/nunet.py:18-105) as ONE fused kernel family for gfx950.
//
//   x [B, C, T, H, W]  ->  for k in {3, 5}:  Conv3d(C -> C, (k,1,1)) -> BatchNorm3d -> SiLU
//                                            -> Conv3d(C -> Cout, (T-k+1,1,1)) -> BatchNorm2d -> SiLU
//                          LayerNorm_Cout(branch3 + branch5)                       -> y [B, Cout, H, W]
//
// Everything between the two BatchNorm reductions is per pixel: 36 inputs (C = 3, T = 12), ~2.4k multiply-adds, 32
// outputs. Rounds 1-3 ran the stage as ~40 launches of the generic kernels (banded 1x1 contractions through the MFMA
// path at 19 TFLOP/s, BatchNorm3d as a 3-channel view, BatchNorm2d, LayerNorm, conversions): 1.15 ms of the 16 ms
// mixed-precision step and ~0.6 ms of the fp32 step for 0.1 % of the FLOPs. Here the chain is RECOMPUTED from x in
// every pass instead of being stored (x is 144 bytes per pixel; the intermediates would be 900): three forward passes
// (BatchNorm3d statistics, BatchNorm2d statistics, output) and three backward passes (LayerNorm / BatchNorm2d sums,
// second-conv weight gradients + BatchNorm3d sums, first-conv weight gradients), each ONE launch that finishes its own
// cross-block reduction through the two-level last-block ticket of cn_ticket.h (no finalize launches, fixed summation
// order). Inference is the output pass alone. The input needs no gradient (it is the data).
//
// Work decomposition (second version; the first -- a wave per 8 output channels, the first convolutions' activations
// exchanged through LDS, weights by scalar loads inside the loops -- ran at 1-2 % of the vector peak: every entry paid
// a scalar-load round trip, two barriers per tile, 8 waves per CU): a THREAD owns a pixel and NO (<= 32) output
// channels of both branches in registers, a wave = 64 consecutive pixels, no barrier inside a tile when one wave covers
// all of Cout (NOG = Cout / NO = 1; Cout = 64 runs two waves per pixel group that only meet for the LayerNorm sums).
// All weights and per-channel constants are staged ONCE per block in LDS and read as wave-uniform broadcasts; the x
// tile is staged per tile ([row][pixel]: conflict-free per-lane reads) and walked with a k-deep register window per
// input channel (one LDS read per (channel, time step) instead of k). Column sums over pixels (statistics, parameter
// gradients) are DPP wave reductions into lane 63, accumulated per wave in LDS, combined per block at the end.
#include <cstdlib>
#include <type_traits>
#include "cn_bf16.h"
#include "cn_ticket.h"

#define PT_MAX_BLOCKS 512
#define PT_MAX_C 8

struct CnPtBranch {
  const float* wa;   // [C][C][k]       Conv3d(C -> C, (k,1,1)).weight
  const float* wb;   // [Cout][C][Tp]   Conv3d(C -> Cout, (Tp,1,1)).weight
  const float* wbt;  // [C*Tp][Cout]    transposed copy (workspace; cn_pretime_pack_kernel)
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
  CnTicket2 tk;
  int ntiles;
};

__global__ void cn_pretime_pack_kernel(const CnPtArgs a, float* wbt0, float* wbt1) {
  for (int brn = 0; brn < 2; ++brn) {
    const CnPtBranch& r = a.br[brn];
    float* dst = brn ? wbt1 : wbt0;
    const int E = a.C * r.Tp;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < E * a.Cout; i += gridDim.x * blockDim.x) {
      const int loc = i / a.Cout, o = i - loc * a.Cout;
      dst[i] = r.wb[(long)o * E + loc];
    }
  }
}


// SiLU and its derivative on the hardware exp / rcp (1 ulp each: ~2e-7 relative; the reference's own fp32 SiLU is no
// closer to the real function). The per-entry activation is a third of the pass's vector instructions otherwise.
__device__ __forceinline__ float pt_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float pt_silu(float x) { return x * pt_sigmoid(x); }
__device__ __forceinline__ float pt_silu_grad(float x) {
  const float s = pt_sigmoid(x);
  return s * (1.0f + x * (1.0f - s));
}
// NO wave-uniform weights w[0..NO) sitting one per lane (lane j holds w[j]) -> acc[j] += w[j] * v. One per-lane LDS read
// serves a whole entry; v_readlane moves each weight into a scalar operand (a 16-byte broadcast read of LDS costs 8
// LDS cycles per 4 weights and wave, which made the LDS -- shared by the CU's four SIMDs -- the bound).
template <int NO>
__device__ __forceinline__ void pt_fma_lane_weights(float wlane, float v, float* acc) {
  const int wb = __float_as_int(wlane);
  const f32x2 v2 = {v, v};
#pragma unroll
  for (int j = 0; j < NO; j += 2) {  // packed fp32 FMA (v_pk_fma_f32): two outputs per vector instruction
    const f32x2 w2 = {__int_as_float(__builtin_amdgcn_readlane(wb, j)), __int_as_float(__builtin_amdgcn_readlane(wb, j + 1))};
    f32x2 r2 = {acc[j], acc[j + 1]};
    r2 = __builtin_elementwise_fma(w2, v2, r2);
    acc[j] = r2[0];
    acc[j + 1] = r2[1];
  }
}
template <int NO>
__device__ __forceinline__ float pt_dot_lane_weights(float wlane, const float* v) {
  const int wb = __float_as_int(wlane);
  float d = 0.f;
#pragma unroll
  for (int j = 0; j < NO; ++j) d += __int_as_float(__builtin_amdgcn_readlane(wb, j)) * v[j];
  return d;
}

#define PT_KP 5  // weight row pitch of the first convolutions in LDS (k <= 5)

// LDS carve (floats) -- host and device compute the same offsets
struct PtLds {
  int wa, c3, k3, wbt, c2, ln, k2, xs, as_, drs, dacc, ex, lacc, total;
};
__host__ __device__ static inline int pt_ceil4(int v) { return (v + 3) & ~3; }
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
__host__ __device__ static inline PtLds pt_lds(int PASS, int C, int T, int Cout, int CMAX, int PXB) {
  const int E3 = C * (T - 2), E5 = C * (T - 4);
  PtLds l;
  int o = 0;
  l.wa = o; o += pt_ceil4(2 * C * CMAX * PT_KP);
  l.c3 = o; o += 2 * C * 4;
  l.k3 = o; o += pt_ceil4(2 * C * 2);
  l.wbt = o; if (PASS >= 1 && PASS <= 4) o += (E3 + E5) * Cout;
  l.c2 = o; if (PASS >= 2 && PASS <= 4) o += 2 * Cout * 4;
  l.ln = o; if (PASS >= 2 && PASS <= 4) o += 2 * Cout;
  l.k2 = o; if (PASS == 4) o += 4 * Cout;
  l.xs = o; o += C * T * PXB;
  l.as_ = o; if (PASS == 4) o += (pt_ceil4(E3) + pt_ceil4(E5)) * PXB;
  l.drs = o; if (PASS == 4) o += 2 * Cout * PXB;
  l.dacc = o; if (PASS == 4) o += (E3 + E5) * PXB;
  l.ex = o; if (PASS >= 2 && PASS <= 4) o += 2 * 4 * 64;
  l.lacc = o; o += pt_ceil4(4 * pt_nvals(PASS, C, Cout, CMAX));
  l.total = o;
  return l;
}

// One Conv3d(C -> C, (K,1,1)) stack over a pixel: rows cp = 0..C-1, entries tp = 0..T-K. The K-deep window xw[c][.]
// of every input channel slides along time in registers: one LDS read per (c, tp). entry(cp, tp, h, xw) gets the
// convolution value and the window (x[c][tp .. tp+K-1], valid for c < C).
template <int K, int CMAX, class FB, class FE, class FR>
__device__ __forceinline__ void pt_rows(const float* __restrict__ wl, const float* __restrict__ xs, int C, int T,
                                        int PXB, int pc, FB&& row_begin, FE&& entry, FR&& row_end) {
  const int Tp = T - K + 1;
  for (int cp = 0; cp < C; ++cp) {
    float w[CMAX][K], xw[CMAX][K];
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
#pragma unroll
      for (int dt = 0; dt < K; ++dt) {
        w[c][dt] = wl[(cp * CMAX + c) * PT_KP + dt];
        xw[c][dt] = 0.f;
      }
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) {
#pragma unroll
        for (int dt = 1; dt < K; ++dt) xw[c][dt] = xs[(c * T + dt - 1) * PXB + pc];
      }
    row_begin(cp);
    for (int tp = 0; tp < Tp; ++tp) {
      float h = 0.f;
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < C) {
#pragma unroll
          for (int dt = 0; dt + 1 < K; ++dt) xw[c][dt] = xw[c][dt + 1];
          xw[c][K - 1] = xs[(c * T + tp + K - 1) * PXB + pc];
#pragma unroll
          for (int dt = 0; dt < K; ++dt) h += w[c][dt] * xw[c][dt];
        }
      entry(cp, tp, h, xw);
    }
    row_end(cp);
  }
}

// PASS 0: BatchNorm3d statistics   1: BatchNorm2d statistics   2: output (training or inference)
// PASS 3: backward sums of LayerNorm / BatchNorm2d   4: dW of the second convolutions + BatchNorm3d sums (+ dz scratch)
// PASS 5: dW of the first convolutions
// CMAX: compile-time bound of the input channels (4 or 8). NO: output channels per thread (8, 16 or 32).
template <int PASS, int CMAX, int NO>
__global__ __launch_bounds__(256) void cn_pretime_kernel(const CnPtArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int s_flag;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int C = a.C, T = a.T, Cout = a.Cout, HW = a.HW;
  const int nthr = blockDim.x;
  const int NOG = Cout / NO;             // waves per pixel group
  const int NPG = (nthr >> 6) / NOG;     // pixel groups (of 64 pixels) per block
  const int PXB = 64 * NPG;
  const int pxg = wid / NOG, og = wid - pxg * NOG;
  const int pc = pxg * 64 + lane;   // pixel column inside the block tile
  const int o0 = og * NO;
  const int wl_lane = lane < NO ? lane : NO - 1;  // lane j < NO fetches weight j of an entry
  const int T3 = T - 2, T5 = T - 4;
  const int E3 = C * T3, E5 = C * T5, E = E3 + E5;
  const int S5 = pt_ceil4(E3);
  const PtLds L = pt_lds(PASS, C, T, Cout, CMAX, PXB);
  float* wa_l = lds + L.wa;
  float* c3_l = lds + L.c3;
  float* k3_l = lds + L.k3;
  float* wbt_l = lds + L.wbt;
  float* c2_l = lds + L.c2;
  float* ln_l = lds + L.ln;
  float* k2_l = lds + L.k2;
  float* xs = lds + L.xs;
  float* as_ = lds + L.as_;
  float* drs = lds + L.drs;
  float* dacc = lds + L.dacc;
  float* ex = lds + L.ex;
  float* lacc = lds + L.lacc;
  const int NV = pt_nvals(PASS, C, Cout, CMAX);
  const float vN = 1.0f / (float)Cout;

  // ---- stage weights and per-channel constants once per block ----
  for (int i = tid; i < 2 * C * CMAX * PT_KP; i += nthr) {
    const int dt = i % PT_KP;
    int q = i / PT_KP;
    const int c = q % CMAX;
    q /= CMAX;
    const int cp = q % C, brn = q / C;
    const int k = brn ? 5 : 3;
    wa_l[i] = (c < C && dt < k) ? a.br[brn].wa[(cp * C + c) * k + dt] : 0.f;
  }
  for (int i = tid; i < 2 * C; i += nthr) {
    const int brn = i / C, cp = i - brn * C;
    const CnPtBranch& r = a.br[brn];
    float mu = 0.f, rho = 1.f;
    if (PASS > 0) {
      if (a.training) { mu = r.mean3[cp]; rho = r.rstd3[cp]; }
      else { mu = r.rm3[cp]; rho = 1.0f / sqrtf(r.rv3[cp] + a.eps3); }
    }
    c3_l[i * 4 + 0] = rho;
    c3_l[i * 4 + 1] = -mu * rho;
    c3_l[i * 4 + 2] = r.g3[cp];
    c3_l[i * 4 + 3] = r.b3[cp];
    if (PASS == 5) {
      k3_l[i * 2 + 0] = a.coef3[(brn * 2) * C + cp];
      k3_l[i * 2 + 1] = a.coef3[(brn * 2 + 1) * C + cp];
    }
  }
  if (PASS >= 1 && PASS <= 4) {
    // (the two transposed tables are contiguous in the workspace: one copy, eight loads in flight per thread -- a
    // load-wait-store loop was ~15 us of dependent round trips at the head of EVERY block)
    const float* __restrict__ src = a.br[0].wbt;
    const int n = E * Cout;
    for (int i0 = tid; i0 < n; i0 += nthr * 8) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { const int k = i0 + i * nthr; v[i] = src[k < n ? k : n - 1]; }
#pragma unroll
      for (int i = 0; i < 8; ++i) { const int k = i0 + i * nthr; if (k < n) wbt_l[k] = v[i]; }
    }
  }
  if (PASS >= 2 && PASS <= 4) {
    for (int i = tid; i < 2 * Cout; i += nthr) {
      const int brn = i / Cout, o = i - brn * Cout;
      const CnPtBranch& r = a.br[brn];
      float mu, rho;
      if (a.training) { mu = r.mean2[o]; rho = r.rstd2[o]; }
      else { mu = r.rm2[o]; rho = 1.0f / sqrtf(r.rv2[o] + a.eps2); }
      c2_l[i * 4 + 0] = rho;
      c2_l[i * 4 + 1] = -mu * rho;
      c2_l[i * 4 + 2] = r.g2[o];
      c2_l[i * 4 + 3] = r.b2[o];
      if (PASS == 4) {
        k2_l[i * 2 + 0] = a.coef2[(brn * 2) * Cout + o];
        k2_l[i * 2 + 1] = a.coef2[(brn * 2 + 1) * Cout + o];
      }
    }
    for (int o = tid; o < Cout; o += nthr) { ln_l[o] = a.gL[o]; ln_l[Cout + o] = a.bL[o]; }
  }
  for (int i = tid; i < 4 * NV; i += nthr) lacc[i] = 0.f;
  if (PASS == 4) {
    for (int i = E3 * PXB + tid; i < S5 * PXB; i += nthr) as_[i] = 0.f;
    for (int i = (S5 + E5) * PXB + tid; i < (S5 + pt_ceil4(E5)) * PXB; i += nthr) as_[i] = 0.f;
  }

  float* my = lacc + wid * NV;  // this wave's accumulators (lane 63 adds the wave totals)
  auto wsum = [&](int v, float val) {
    const float t = cn_wave_sum_to_lane63(val);
    if (lane == 63) my[v] += t;
  };

  float AW[8];  // PASS 4: this thread's dWb tile (2 couts x 4 entries), summed over the block's tiles
#pragma unroll
  for (int j = 0; j < 8; ++j) AW[j] = 0.f;
  float AW2[8], AW3[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) AW2[j] = AW3[j] = 0.f;

  const int ntb = (int)((a.P + PXB - 1) / PXB);  // block tiles
  for (int tile = blockIdx.x; tile < ntb; tile += gridDim.x) {
    const long p = (long)tile * PXB + pc;
    const bool valid = p < a.P;
    const float vm = valid ? 1.f : 0.f;
    const int b = valid ? (int)(p / HW) : 0;
    const int l = valid ? (int)(p - (long)b * HW) : 0;
    __syncthreads();  // previous tile's LDS reads are done (and, first time, the staged weights are visible below)
    {
      // x tile: rows dealt to the NOG waves of the pixel group (every wave of a group has the same pixels). TWELVE
      // loads in flight per thread (clamped row index, never a predicated load): one row at a time was a chain of
      // C * T dependent HBM round trips per tile -- most of the first version's run time.
      const float* xp = a.x + (long)b * a.xbs + l;
      const int CT = C * T;
      for (int row0 = og; row0 < CT; row0 += NOG * 12) {
        float v[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          const int row = row0 + i * NOG;
          v[i] = xp[(long)(row < CT ? row : CT - 1) * HW];
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          const int row = row0 + i * NOG;
          if (row < CT) xs[row * PXB + pc] = valid ? v[i] : 0.f;
        }
      }
    }
    __syncthreads();

    if (PASS == 0) {
      float s = 0.f, q = 0.f;
      int slot = 0;
      auto rb = [&](int) { s = 0.f; q = 0.f; };
      auto en = [&](int, int, float h, auto&) { s += h; q += h * h; };
      auto re = [&](int) { wsum(slot, s); wsum(slot + 1, q); slot += 2; };
      if (og == 0) {  // (one wave per pixel group: the other waves of a Cout > NO group would count the pixels twice)
        pt_rows<3, CMAX>(wa_l, xs, C, T, PXB, pc, rb, en, re);
        pt_rows<5, CMAX>(wa_l + C * CMAX * PT_KP, xs, C, T, PXB, pc, rb, en, re);
      }
      continue;
    }
    if (PASS == 5) {
      // dh = g3 rho (dz - c0 - hh c1); dWa[cp][c][dt] += sum_px sum_tp dh * x[c][tp + dt] (the window IS x[c][tp + dt])
      if (og == 0) {
        int ebase = 0, vbase = 0;
        auto branch = [&](auto kc, const float* wl, int brn) {
          constexpr int K = decltype(kc)::value;
          float g[CMAX][K];
          float rho = 0.f, off = 0.f, g3 = 0.f, c0 = 0.f, c1 = 0.f;
          const int Tp = T - K + 1;
          auto rb = [&](int cp) {
            const f32x4 cc = *reinterpret_cast<const f32x4*>(c3_l + (brn * C + cp) * 4);
            rho = cc[0]; off = cc[1]; g3 = cc[2];
            c0 = k3_l[(brn * C + cp) * 2]; c1 = k3_l[(brn * C + cp) * 2 + 1];
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
#pragma unroll
              for (int dt = 0; dt < K; ++dt) g[c][dt] = 0.f;
          };
          auto en = [&](int cp, int tp, float h, auto& xw) {
            const float hh = h * rho + off;
            const int e = ebase + cp * Tp + tp;
            const float dzv = valid ? a.dz[(long)e * a.P + p] : 0.f;
            const float dh = g3 * rho * (dzv - c0 - hh * c1) * vm;
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
#pragma unroll
              for (int dt = 0; dt < K; ++dt) g[c][dt] += dh * xw[c][dt];
          };
          auto re = [&](int cp) {
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
#pragma unroll
              for (int dt = 0; dt < K; ++dt) wsum(vbase + (cp * CMAX + c) * PT_KP + dt, g[c][dt]);
          };
          pt_rows<K, CMAX>(wl, xs, C, T, PXB, pc, rb, en, re);
          ebase += C * Tp;
          vbase += C * CMAX * PT_KP;
        };
        branch(std::integral_constant<int, 3>{}, wa_l, 0);
        branch(std::integral_constant<int, 5>{}, wa_l + C * CMAX * PT_KP, 1);
      }
      continue;
    }

    // ---- PASS 1..4: both convolution stacks -> r[branch][own outputs] ----
    float r_[2][NO];
#pragma unroll
    for (int brn = 0; brn < 2; ++brn)
#pragma unroll
      for (int j = 0; j < NO; ++j) r_[brn][j] = 0.f;
    {
      float rho = 0.f, off = 0.f, g3 = 0.f, b3 = 0.f;
      auto branch = [&](auto kc, const float* wl, int brn, const float* wb_l, int slot0) {
        constexpr int K = decltype(kc)::value;
        const int Tp = T - K + 1;
        auto rb = [&](int cp) {
          const f32x4 cc = *reinterpret_cast<const f32x4*>(c3_l + (brn * C + cp) * 4);
          rho = cc[0]; off = cc[1]; g3 = cc[2]; b3 = cc[3];
        };
        auto en = [&](int cp, int tp, float h, auto&) {
          const float aval = pt_silu(g3 * (h * rho + off) + b3);
          const int loc = cp * Tp + tp;
          if (PASS == 4 && og == 0) as_[(slot0 + loc) * PXB + pc] = aval;
          pt_fma_lane_weights<NO>(wb_l[loc * Cout + o0 + wl_lane], aval, r_[brn]);
        };
        auto re = [&](int) {};
        pt_rows<K, CMAX>(wl, xs, C, T, PXB, pc, rb, en, re);
      };
      branch(std::integral_constant<int, 3>{}, wa_l, 0, wbt_l, 0);
      branch(std::integral_constant<int, 5>{}, wa_l + C * CMAX * PT_KP, 1, wbt_l + E3 * Cout, S5);
    }
    if (PASS == 1) {
#pragma unroll
      for (int brn = 0; brn < 2; ++brn)
#pragma unroll
        for (int j = 0; j < NO; ++j) {
          const float v = r_[brn][j] * vm;
          wsum((brn * 2) * Cout + o0 + j, v);
          wsum((brn * 2 + 1) * Cout + o0 + j, v * v);
        }
      continue;
    }
    // ---- BatchNorm2d + SiLU, branch sum, LayerNorm over Cout ----
    // r_ <- rhat (normalised); vv = affine BatchNorm output; u = sum of the activations
    float vv[2][NO], u[NO];
#pragma unroll
    for (int j = 0; j < NO; ++j) u[j] = 0.f;
#pragma unroll
    for (int brn = 0; brn < 2; ++brn)
#pragma unroll
      for (int j = 0; j < NO; ++j) {
        const f32x4 cc = *reinterpret_cast<const f32x4*>(c2_l + (brn * Cout + o0 + j) * 4);
        r_[brn][j] = r_[brn][j] * cc[0] + cc[1];
        vv[brn][j] = cc[2] * r_[brn][j] + cc[3];
        u[j] += pt_silu(vv[brn][j]);
      }
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < NO; ++j) m += u[j];
    if (NOG > 1) {
      ex[wid * 64 + lane] = m;
      __syncthreads();
      m = 0.f;
      for (int w2 = 0; w2 < NOG; ++w2) m += ex[(pxg * NOG + w2) * 64 + lane];
    }
    m *= vN;
    float var = 0.f;
#pragma unroll
    for (int j = 0; j < NO; ++j) { const float d = u[j] - m; var += d * d; }
    if (NOG > 1) {
      ex[(4 + wid) * 64 + lane] = var;
      __syncthreads();
      var = 0.f;
      for (int w2 = 0; w2 < NOG; ++w2) var += ex[(4 + pxg * NOG + w2) * 64 + lane];
    }
    const float rL = 1.0f / sqrtf(var * vN + a.epsL);
#pragma unroll
    for (int j = 0; j < NO; ++j) u[j] = (u[j] - m) * rL;  // u <- uhat
    if (PASS == 2) {
      if (valid) {
        if (a.out_kind == 0) {
          float* yp = reinterpret_cast<float*>(a.y) + (long)b * a.y_stride + (long)o0 * HW + l;
#pragma unroll
          for (int j = 0; j < NO; ++j) yp[(long)j * HW] = ln_l[o0 + j] * u[j] + ln_l[Cout + o0 + j];
        } else {
          bf16_t* yp = reinterpret_cast<bf16_t*>(a.y) + p * a.y_stride + o0;
#pragma unroll
          for (int j8 = 0; j8 < NO / 8; ++j8) {
            float yv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) yv[i] = ln_l[o0 + j8 * 8 + i] * u[j8 * 8 + i] + ln_l[Cout + o0 + j8 * 8 + i];
            *reinterpret_cast<u32x4*>(yp + j8 * 8) = cn_pack8(yv);
          }
        }
      }
      continue;
    }
    // ---- backward: LayerNorm, SiLU, BatchNorm2d ----
    float dyv[NO];
    if (a.out_kind == 0) {
      const float* dp = reinterpret_cast<const float*>(a.dy) + (long)b * a.dy_stride + (long)o0 * HW + l;
#pragma unroll
      for (int j = 0; j < NO; ++j) dyv[j] = valid ? dp[(long)j * HW] : 0.f;
    } else {
      const bf16_t* dp = reinterpret_cast<const bf16_t*>(a.dy) + (valid ? p : 0) * a.dy_stride + o0;
#pragma unroll
      for (int j8 = 0; j8 < NO / 8; ++j8) {
        float t8[8];
        cn_unpack8(*reinterpret_cast<const u32x4*>(dp + j8 * 8), t8);
#pragma unroll
        for (int i = 0; i < 8; ++i) dyv[j8 * 8 + i] = t8[i] * vm;
      }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NO; ++j) {
      const float gj = dyv[j] * ln_l[o0 + j];
      s1 += gj;
      s2 += gj * u[j];
    }
    if (NOG > 1) {
      __syncthreads();  // (the variance partials in ex[] have been read)
      ex[wid * 64 + lane] = s1;
      ex[(4 + wid) * 64 + lane] = s2;
      __syncthreads();
      s1 = s2 = 0.f;
      for (int w2 = 0; w2 < NOG; ++w2) { s1 += ex[(pxg * NOG + w2) * 64 + lane]; s2 += ex[(4 + pxg * NOG + w2) * 64 + lane]; }
    }
    s1 *= vN; s2 *= vN;
    if (PASS == 3) {
#pragma unroll
      for (int j = 0; j < NO; ++j) {
        const float du = rL * (dyv[j] * ln_l[o0 + j] - s1 - u[j] * s2);
#pragma unroll
        for (int brn = 0; brn < 2; ++brn) {
          const float dv = du * pt_silu_grad(vv[brn][j]);
          wsum((brn * 2) * Cout + o0 + j, dv);
          wsum((brn * 2 + 1) * Cout + o0 + j, dv * r_[brn][j]);
        }
        wsum(4 * Cout + o0 + j, dyv[j] * u[j]);
        wsum(5 * Cout + o0 + j, dyv[j]);
      }
      continue;
    }
    // ---- PASS 4 ----
    // dr (kept in vv) -> LDS for the dWb contraction
#pragma unroll
    for (int j = 0; j < NO; ++j) {
      const float du = rL * (dyv[j] * ln_l[o0 + j] - s1 - u[j] * s2);
#pragma unroll
      for (int brn = 0; brn < 2; ++brn) {
        const float dv = du * pt_silu_grad(vv[brn][j]);
        const f32x4 cc = *reinterpret_cast<const f32x4*>(c2_l + (brn * Cout + o0 + j) * 4);
        const float c0 = k2_l[(brn * Cout + o0 + j) * 2], c1 = k2_l[(brn * Cout + o0 + j) * 2 + 1];
        const float dr = cc[2] * cc[0] * (dv - c0 - r_[brn][j] * c1) * vm;
        vv[brn][j] = dr;
        drs[(brn * Cout + o0 + j) * PXB + pc] = dr;
      }
    }
    // da[e] = sum_o wb[o][e] dr[o]: this thread's NO outputs; the NOG waves of a pixel group add up through LDS
    for (int i = og; i < E; i += NOG) dacc[i * PXB + pc] = 0.f;
    if (NOG > 1) __syncthreads();
    for (int e = 0; e < E; ++e) {
      const int brn = e >= E3 ? 1 : 0;
      const float wl_ = wbt_l[e * Cout + o0 + wl_lane];  // (wbt_l rows: branch 3 then 5)
      const float da = brn == 0 ? pt_dot_lane_weights<NO>(wl_, vv[0]) : pt_dot_lane_weights<NO>(wl_, vv[1]);
      if (NOG > 1) atomicAdd(&dacc[e * PXB + pc], da);  // (ds_add_f32: one address per lane, waves of a group in turn)
      else dacc[e * PXB + pc] = da;
    }
    __syncthreads();  // as_ / drs / dacc complete for the whole block tile
    // dz = da * silu'(z), BatchNorm3d sums, dz -> scratch (one wave per pixel group)
    if (og == 0) {
      float rho = 0.f, off = 0.f, g3 = 0.f, b3 = 0.f, s = 0.f, q = 0.f;
      int slot = 0, ebase = 0;
      auto branch = [&](auto kc, const float* wl, int brn) {
        constexpr int K = decltype(kc)::value;
        const int Tp = T - K + 1;
        auto rb = [&](int cp) {
          const f32x4 cc = *reinterpret_cast<const f32x4*>(c3_l + (brn * C + cp) * 4);
          rho = cc[0]; off = cc[1]; g3 = cc[2]; b3 = cc[3];
          s = 0.f; q = 0.f;
        };
        auto en = [&](int cp, int tp, float h, auto&) {
          const float hh = h * rho + off;
          const int e = ebase + cp * Tp + tp;
          const float dzv = dacc[e * PXB + pc] * pt_silu_grad(g3 * hh + b3);
          s += dzv;
          q += dzv * hh;
          if (valid) a.dz[(long)e * a.P + p] = dzv;
        };
        auto re = [&](int) { wsum(slot, s); wsum(slot + 1, q); slot += 2; };
        pt_rows<K, CMAX>(wl, xs, C, T, PXB, pc, rb, en, re);
        ebase += C * Tp;
      };
      branch(std::integral_constant<int, 3>{}, wa_l, 0);
      branch(std::integral_constant<int, 5>{}, wa_l + C * CMAX * PT_KP, 1);
    }
    {
      // dWb[brn][o][loc] += sum_px dr[brn][o][px] * a[brn][loc][px]: a thread owns 2 couts x 4 entries (x 3 rounds)
      const int half = Cout >> 1;
      const int nt3 = half * (pt_ceil4(E3) >> 2), nt5 = half * (pt_ceil4(E5) >> 2);
      auto tile_mac = [&](int ti, float* acc8) {
        if (ti >= nt3 + nt5) return;
        const int brn = ti >= nt3 ? 1 : 0;
        const int tl = ti - (brn ? nt3 : 0);
        const int o2 = tl % half, e4 = tl / half;
        const float* d0 = drs + (brn * Cout + 2 * o2) * PXB;
        const float* a0 = as_ + ((brn ? S5 : 0) + e4 * 4) * PXB;
        for (int q4 = 0; q4 < PXB / 4; ++q4) {
          const f32x4 da_ = *reinterpret_cast<const f32x4*>(d0 + q4 * 4);
          const f32x4 db_ = *reinterpret_cast<const f32x4*>(d0 + PXB + q4 * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(a0 + i * PXB + q4 * 4);
            acc8[i] += da_[0] * av[0] + da_[1] * av[1] + da_[2] * av[2] + da_[3] * av[3];
            acc8[4 + i] += db_[0] * av[0] + db_[1] * av[1] + db_[2] * av[2] + db_[3] * av[3];
          }
        }
      };
      tile_mac(tid, AW);
      tile_mac(tid + nthr, AW2);
      tile_mac(tid + 2 * nthr, AW3);
    }
  }

  // ---- block row -> two-level last-block reduction -> finish ----
  if (PASS == 2) return;
  const int blk = blockIdx.x;
  __syncthreads();
  {
    // the four waves' accumulators -> one row
    const int base = PASS == 4 ? Cout * E : 0;
    for (int v = tid; v < NV; v += nthr)
    {
      float t = 0.f;
      for (int w2 = 0; w2 < (nthr >> 6); ++w2) t += lacc[w2 * NV + v];
      cn_t2_store(a.tk, blk, base + v, t);
    }
  }
  if (PASS == 4) {
    const int half = Cout >> 1;
    const int nt3 = half * (pt_ceil4(E3) >> 2), nt5 = half * (pt_ceil4(E5) >> 2);
    auto tile_store = [&](int ti, const float* acc8) {
      if (ti >= nt3 + nt5) return;
      const int brn = ti >= nt3 ? 1 : 0;
      const int tl = ti - (brn ? nt3 : 0);
      const int o2 = tl % half, e4 = tl / half;
      const int Eb = brn ? E5 : E3;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int loc = e4 * 4 + i;
        if (loc < Eb) {
          cn_t2_store(a.tk, blk, (brn ? Cout * E3 : 0) + (2 * o2) * Eb + loc, acc8[i]);
          cn_t2_store(a.tk, blk, (brn ? Cout * E3 : 0) + (2 * o2 + 1) * Eb + loc, acc8[4 + i]);
        }
      }
    };
    tile_store(tid, AW);
    tile_store(tid + nthr, AW2);
    tile_store(tid + 2 * nthr, AW3);
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
    for (int i = tid; i < 2 * C; i += nthr) {
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
        if (r.rm3 != nullptr) {
          const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
          r.rm3[cp] = (1.f - a.mom3) * r.rm3[cp] + a.mom3 * (float)md;
          r.rv3[cp] = (1.f - a.mom3) * r.rv3[cp] + a.mom3 * (float)unb;
        }
      } else {
        a.coef3[(brn * 2) * C + cp] = a.training ? (float)(v0 / cnt) : 0.f;
        a.coef3[(brn * 2 + 1) * C + cp] = a.training ? (float)(v1 / cnt) : 0.f;
        r.dg3[cp] += (float)v1;
        r.db3[cp] += (float)v0;
      }
    }
  } else {  // PASS 1 / 3: columns [(brn*2 + stat)*Cout + o] (+ LayerNorm parameter gradients in PASS 3)
    for (int i = tid; i < 2 * Cout; i += nthr) {
      const int brn = i / Cout, o = i - brn * Cout;
      const CnPtBranch& r = a.br[brn];
      const double v0 = tot[(brn * 2) * Cout + o], v1 = tot[(brn * 2 + 1) * Cout + o];
      if (PASS == 1) {
        const double md = v0 / cntP;
        double var = v1 / cntP - md * md;
        if (var < 0.0) var = 0.0;
        r.mean2[o] = (float)md;
        r.rstd2[o] = (float)(1.0 / sqrt(var + (double)a.eps2));
        if (r.rm2 != nullptr) {
          const double unb = cntP > 1.0 ? var * cntP / (cntP - 1.0) : var;
          r.rm2[o] = (1.f - a.mom2) * r.rm2[o] + a.mom2 * (float)md;
          r.rv2[o] = (1.f - a.mom2) * r.rv2[o] + a.mom2 * (float)unb;
        }
      } else {
        a.coef2[(brn * 2) * Cout + o] = a.training ? (float)(v0 / cntP) : 0.f;
        a.coef2[(brn * 2 + 1) * Cout + o] = a.training ? (float)(v1 / cntP) : 0.f;
        r.dg2[o] += (float)v1;
        r.db2[o] += (float)v0;
      }
    }
    if (PASS == 3)
      for (int o = tid; o < Cout; o += nthr) {
        a.dgL[o] += (float)tot[4 * Cout + o];
        a.dbL[o] += (float)tot[5 * Cout + o];
      }
  }
}

// ---- host side -----------------------------------------------------------------------------------------------------
static inline int pt_cmax(int C) { return C <= 4 ? 4 : 8; }
// output channels per thread: 32 where the registers allow it (forward, statistics), 16 in the gradient passes that
// keep dy / dr / vv per output as well
static inline int pt_no(int PASS, int Cout) {
  const int cap = (PASS == 3 || PASS == 4) ? 16 : 32;
  return Cout < cap ? Cout : cap;
}
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
  return C >= 1 && C <= PT_MAX_C && T >= 5 && (Cout == 8 || Cout == 16 || Cout == 32 || Cout == 64);
}
static inline size_t pt_shmem_npg(int PASS, int C, int T, int Cout, int npg) {
  size_t sh = (size_t)pt_lds(PASS, C, T, Cout, pt_cmax(C), 64 * npg).total * 4;
  const size_t park = (size_t)(6 * Cout > 4 * C ? 6 * Cout : 4 * C) * 8;  // the finish phase parks doubles at the head
  return sh < park ? park : sh;
}
// pixel groups (waves' worth of 64 pixels) per block: as many as 256 threads hold, fewer if the LDS image (x tile,
// and in PASS 4 the activation / dr / da tiles of the block) would not fit; 0 = does not fit at all
static inline int pt_npg(int PASS, int C, int T, int Cout) {
  const int nog = Cout / pt_no(PASS, Cout);
  for (int npg = 4 / nog; npg >= 1; npg >>= 1)
    if (pt_shmem_npg(PASS, C, T, Cout, npg) <= 156 * 1024) return npg;
  return 0;
}
// floats: [counters 64][ticket body for the widest row][wbt x2][coef2][coef3][dz]
static inline long pt_off_body() { return CN_T2_COUNTERS; }
static inline long pt_body_floats(int C, int T, int Cout) {
  int w = 0;
  for (int ps = 0; ps < 6; ++ps) { const int r = pt_row_width(ps, C, T, Cout); w = r > w ? r : w; }
  return (cn_t2_body_floats(PT_MAX_BLOCKS, w) + 63) / 64 * 64;
}
extern "C" long cn_pretime_workspace_floats(int B, int C, int T, int HW, int Cout, int with_backward) {
  if (!pt_supported(C, T, Cout)) return -1;
  for (int ps = 0; ps < (with_backward ? 6 : 3); ++ps)  // every pass the caller may launch must fit the 160 KiB LDS
    if (pt_npg(ps, C, T, Cout) == 0) return -1;
  if (with_backward) {
    const int nt = (Cout / 2) * ((pt_ceil4(C * (T - 2)) + pt_ceil4(C * (T - 4))) / 4);
    const int nthr = 64 * pt_npg(4, C, T, Cout) * (Cout / pt_no(4, Cout));
    if (nt > 3 * nthr) return -1;  // dWb tiles: three per thread
  }
  const long E = (long)C * (T - 2) + (long)C * (T - 4);
  long n = pt_off_body() + pt_body_floats(C, T, Cout) + E * Cout + 4L * Cout + 4L * C + 64;
  if (with_backward) n += E * (long)B * HW;
  return n;
}

static int pt_fill(CnPtArgs& a, const float* x, long xbs, const void* const* params, float* const* stats, int B, int C,
                   int T, int HW, int Cout, int training, const float* bn, float eps_ln, float* ws, long ws_floats,
                   int with_backward) {
  if (!pt_supported(C, T, Cout) || B <= 0 || HW <= 0) return CN_ERR_ARG;
  if ((long)B * HW >= (1L << 31) - 64) return CN_ERR_ARG;
  if (ws == nullptr || ws_floats < cn_pretime_workspace_floats(B, C, T, HW, Cout, with_backward)) return CN_ERR_ARG;
  a.x = x; a.xbs = xbs; a.B = B; a.C = C; a.T = T; a.HW = HW; a.Cout = Cout; a.P = (long)B * HW;
  a.training = training; a.eps3 = bn[0]; a.mom3 = bn[1]; a.eps2 = bn[2]; a.mom2 = bn[3]; a.epsL = eps_ln;
  a.ntiles = 0;
  const long E3 = (long)C * (T - 2), E5 = (long)C * (T - 4);
  float* wbt0 = ws + pt_off_body() + pt_body_floats(C, T, Cout);
  float* wbt1 = wbt0 + E3 * Cout;
  a.coef2 = wbt1 + E5 * Cout;
  a.coef3 = a.coef2 + 4L * Cout;
  a.dz = a.coef3 + 4L * C + (64 - (4 * C) % 64) % 64;
  for (int brn = 0; brn < 2; ++brn) {
    CnPtBranch& r = a.br[brn];
    const void* const* p = params + brn * 10;
    r.wa = (const float*)p[0]; r.wb = (const float*)p[1];
    r.g3 = (const float*)p[2]; r.b3 = (const float*)p[3]; r.rm3 = (float*)p[4]; r.rv3 = (float*)p[5];
    r.g2 = (const float*)p[6]; r.b2 = (const float*)p[7]; r.rm2 = (float*)p[8]; r.rv2 = (float*)p[9];
    r.mean3 = stats[brn * 4 + 0]; r.rstd3 = stats[brn * 4 + 1]; r.mean2 = stats[brn * 4 + 2]; r.rstd2 = stats[brn * 4 + 3];
    r.wbt = brn ? wbt1 : wbt0;
    r.k = brn ? 5 : 3; r.Tp = T - r.k + 1;
    if (!training && (r.rm3 == nullptr || r.rv3 == nullptr || r.rm2 == nullptr || r.rv2 == nullptr)) return CN_ERR_ARG;
  }
  a.gL = (const float*)params[20]; a.bL = (const float*)params[21];
  return CN_OK;
}

template <int PASS>
static int pt_launch(CnPtArgs a, float* ws, hipStream_t stream) {
  const int NO = pt_no(PASS, a.Cout);
  const int npg = pt_npg(PASS, a.C, a.T, a.Cout);
  if (npg == 0) return CN_ERR_LDS;
  const int PXB = 64 * npg;
  const int nthr = 64 * npg * (a.Cout / NO);
  const int ntb = (int)((a.P + PXB - 1) / PXB);
  const int cap = PT_MAX_BLOCKS;  // persistent blocks (the per-block weight staging is paid once per block, not per tile)
  const int nblk = ntb < cap ? ntb : cap;
  const size_t shmem = pt_shmem_npg(PASS, a.C, a.T, a.Cout, npg);
  const int W = pt_row_width(PASS, a.C, a.T, a.Cout);
  if (W > 0) a.tk = cn_t2_carve(reinterpret_cast<int*>(ws), ws + pt_off_body(), nblk, W);
#define PT_GO(CM, NO_)                                                                                         \
  do {                                                                                                         \
    if (shmem > 64 * 1024)                                                                                     \
      (void)hipFuncSetAttribute((const void*)cn_pretime_kernel<PASS, CM, NO_>,                                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);                       \
    CN_LAUNCH((cn_pretime_kernel<PASS, CM, NO_>), dim3(nblk), dim3(nthr), shmem, stream, a);                   \
  } while (0)
#define PT_GO_NO(CM)                                                                                           \
  do {                                                                                                         \
    if (NO == 8) PT_GO(CM, 8); else if (NO == 16) PT_GO(CM, 16); else PT_GO(CM, 32);                           \
  } while (0)
  if (pt_cmax(a.C) == 4) PT_GO_NO(4); else PT_GO_NO(8);
#undef PT_GO_NO
#undef PT_GO
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
  CN_LAUNCH(cn_pretime_pack_kernel, dim3(8), dim3(256), 0, stream, a, const_cast<float*>(a.br[0].wbt),
            const_cast<float*>(a.br[1].wbt));
  if (training) {
    if ((rc = pt_launch<0>(a, ws, stream)) != CN_OK) return rc;
    if ((rc = pt_launch<1>(a, ws, stream)) != CN_OK) return rc;
  }
  if ((rc = pt_launch<2>(a, ws, stream)) != CN_OK) return rc;
  return cn_check_launch();
}

// grads: HOST array of 14 device pointers: per branch {dwa, dwb, dgamma3, dbeta3, dgamma2, dbeta2}, then
// {d ln_gamma, d ln_beta}; all ACCUMULATED. dy in the layout of y. The workspace must be the one of the forward call
// (with_backward = 1 sizing) only for its transposed weights, which are re-made here.
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
  CN_LAUNCH(cn_pretime_pack_kernel, dim3(8), dim3(256), 0, stream, a, const_cast<float*>(a.br[0].wbt),
            const_cast<float*>(a.br[1].wbt));
  if ((rc = pt_launch<3>(a, ws, stream)) != CN_OK) return rc;
  if ((rc = pt_launch<4>(a, ws, stream)) != CN_OK) return rc;
  if ((rc = pt_launch<5>(a, ws, stream)) != CN_OK) return rc;
  return cn_check_launch();
}
