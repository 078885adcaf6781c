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
// Work decomposition: a block = 64 pixels (the lanes of a wave) x NOG = Cout / 8 waves; wave `og` owns output
// channels [8 og, 8 og + 8) of both branches, so every weight a wave touches is wave-uniform (scalar loads, SGPR
// operands) and the bf16 NHWC output is one 16-byte store per lane. The x tile and the C*(T-2) + C*(T-4) activations of
// the first convolutions live in LDS, shared by the block's waves (computed once per block, entry e by wave e % NOG).
#include <cstdlib>
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

// LDS carve (floats) -- the host computes the same sizes (pt_shmem_bytes)
struct PtLds {
  int xs, as_, hs, drs, red, acc, total;
};
__host__ __device__ static inline int pt_ceil4(int v) { return (v + 3) & ~3; }
__host__ __device__ static inline PtLds pt_lds(int PASS, int C, int T, int Cout, int nthr) {
  const int E3 = C * (T - 2), E5 = C * (T - 4);
  const int ES = pt_ceil4(E3) + pt_ceil4(E5);
  PtLds l;
  int o = 0;
  l.xs = o; o += C * T * 64;
  l.as_ = o; if (PASS >= 1 && PASS <= 4) o += ES * 64;
  l.hs = o; if (PASS >= 4) o += (E3 + E5) * 64;
  l.drs = o; if (PASS == 4) o += 2 * Cout * 64;
  l.red = o; if (PASS >= 2 && PASS <= 4) o += 2 * (Cout / 8) * 64;
  l.acc = o;
  if (PASS == 0 || PASS == 4) o += 4 * C * nthr;
  if (PASS == 5) o += 8 * C * C * 64;
  l.total = o;
  return l;
}

// PASS 0: BatchNorm3d statistics   1: BatchNorm2d statistics   2: output (training or inference)
// PASS 3: backward sums of LayerNorm / BatchNorm2d   4: dW of the second convolutions + BatchNorm3d sums (+ dz scratch)
// PASS 5: dW of the first convolutions
template <int PASS, int MAXIT>
__global__ __launch_bounds__(512) void cn_pretime_kernel(const CnPtArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int s_flag;
  const int tid = threadIdx.x, px = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // = output group og; provably wave-uniform
  const int nthr = blockDim.x, NOG = nthr >> 6;
  const int C = a.C, T = a.T, Cout = a.Cout, HW = a.HW;
  const int T3 = T - 2, T5 = T - 4;
  const int E3 = C * T3, E5 = C * T5, E = E3 + E5;
  const int S5 = pt_ceil4(E3);  // LDS slot of branch 5's first entry (both branches padded to 4 entries: dW tiles)
  const PtLds L = pt_lds(PASS, C, T, Cout, nthr);
  float* xs = lds + L.xs;
  float* as_ = lds + L.as_;
  float* hs = lds + L.hs;
  float* drs = lds + L.drs;
  float* red = lds + L.red;
  float* acc = lds + L.acc;
  const float vN = 1.0f / (float)Cout;

  // ---- per-kernel initialisation of LDS accumulators / pad rows ----
  if (PASS == 0 || PASS == 4)
    for (int i = tid; i < 4 * C * nthr; i += nthr) acc[i] = 0.f;
  if (PASS == 5)
    for (int i = tid; i < 8 * C * C * 64; i += nthr) acc[i] = 0.f;
  if (PASS >= 1 && PASS <= 4) {
    for (int i = E3 * 64 + tid; i < S5 * 64; i += nthr) as_[i] = 0.f;
    for (int i = (S5 + E5) * 64 + tid; i < (S5 + pt_ceil4(E5)) * 64; i += nthr) as_[i] = 0.f;
  }

  // register accumulators across the block's tiles
  float A2[2][2][8];   // PASS 1: {sum r, sum r^2}; PASS 3: {sum dv, sum dv*rhat} per branch and own channel
  float AL[2][8];      // PASS 3: {sum dy*uhat, sum dy}
  float AW[MAXIT][8];  // PASS 4: dWb tiles (2 couts x 4 entries)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) A2[i][s][j] = 0.f;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) AL[s][j] = 0.f;
#pragma unroll
  for (int it = 0; it < MAXIT; ++it)
#pragma unroll
    for (int j = 0; j < 8; ++j) AW[it][j] = 0.f;

  // per-thread constants of the own 8 output channels (wave-uniform values: scalar loads)
  const int o0 = wid * 8;

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const long p = (long)tile * 64 + px;
    const bool valid = p < a.P;
    const float vm = valid ? 1.f : 0.f;
    const int b = valid ? (int)(p / HW) : 0;
    const int l = valid ? (int)(p - (long)b * HW) : 0;
    __syncthreads();  // the previous tile's LDS reads are done
    {
      const float* xp = a.x + (long)b * a.xbs + l;
      for (int row = wid; row < C * T; row += NOG) {
        const float v = xp[(long)row * HW];  // (clamped pixel for the ragged tail: always a valid address)
        xs[row * 64 + px] = valid ? v : 0.f;
      }
    }
    __syncthreads();
    // ---- first convolutions, entry by entry (wave-uniform entry index) ----
    for (int e = wid; e < E; e += NOG) {
      const int brn = e >= E3 ? 1 : 0;
      const CnPtBranch& r = a.br[brn];
      const int loc = e - (brn ? E3 : 0);
      const int Tp = brn ? T5 : T3, k = brn ? 5 : 3;
      const int cp = loc / Tp, tp = loc - cp * Tp;
      float h = 0.f;
      for (int c = 0; c < C; ++c) {
        const float* w = r.wa + (cp * C + c) * k;
        const float* xr = xs + (c * T + tp) * 64 + px;
        for (int dt = 0; dt < k; ++dt) h += w[dt] * xr[dt * 64];
      }
      if (PASS == 0) {
        const int slot = (brn * C + cp) * 2;
        acc[slot * nthr + tid] += h;
        acc[(slot + 1) * nthr + tid] += h * h;
        continue;
      }
      float mu, rho;
      if (a.training) { mu = r.mean3[cp]; rho = r.rstd3[cp]; }
      else { mu = r.rm3[cp]; rho = 1.0f / sqrtf(r.rv3[cp] + a.eps3); }
      const float hh = (h - mu) * rho;
      const int slot_e = brn ? S5 + loc : loc;
      if (PASS <= 4) {
        const float z = r.g3[cp] * hh + r.b3[cp];
        as_[slot_e * 64 + px] = cn_silu(z);
      }
      if (PASS == 4) hs[e * 64 + px] = hh;
      if (PASS == 5) {
        const float dzv = valid ? a.dz[(long)e * a.P + p] : 0.f;
        const float c0 = a.coef3[(brn * 2) * C + cp], c1 = a.coef3[(brn * 2 + 1) * C + cp];
        hs[e * 64 + px] = r.g3[cp] * rho * (dzv - c0 - hh * c1) * vm;  // dh
      }
    }
    if (PASS == 0) continue;
    __syncthreads();
    if (PASS == 5) {
      // dWa[brn][cp][c][dt] += sum_px sum_tp dh[brn][cp][tp] * x[c][tp + dt]: output q owned by wave q % NOG
      const int n3 = C * C * 3, nout = 8 * C * C;
      for (int q = wid; q < nout; q += NOG) {
        const int brn = q >= n3 ? 1 : 0;
        const int ql = q - (brn ? n3 : 0);
        const int k = brn ? 5 : 3, Tp = brn ? T5 : T3;
        const int dt = ql % k;
        const int cc = ql / k;  // cp * C + c
        const int cp = cc / C, c = cc - cp * C;
        const float* dh = hs + ((brn ? E3 : 0) + cp * Tp) * 64 + px;
        const float* xr = xs + (c * T + dt) * 64 + px;
        float v = 0.f;
        for (int tp = 0; tp < Tp; ++tp) v += dh[tp * 64] * xr[tp * 64];
        acc[q * 64 + px] += v;
      }
      continue;
    }
    // ---- second convolutions: the wave's 8 output channels of both branches ----
    float r_[2][8];
#pragma unroll
    for (int brn = 0; brn < 2; ++brn) {
#pragma unroll
      for (int j = 0; j < 8; ++j) r_[brn][j] = 0.f;
      const int Eb = brn ? E5 : E3;
      const float* av = as_ + (brn ? S5 : 0) * 64 + px;
      const float* w = a.br[brn].wbt + o0;
      for (int loc = 0; loc < Eb; ++loc) {
        const float aval = av[loc * 64];
#pragma unroll
        for (int j = 0; j < 8; ++j) r_[brn][j] += w[(long)loc * Cout + j] * aval;
      }
    }
    if (PASS == 1) {
#pragma unroll
      for (int brn = 0; brn < 2; ++brn)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = r_[brn][j] * vm;
          A2[brn][0][j] += v;
          A2[brn][1][j] += v * v;
        }
      continue;
    }
    // ---- BatchNorm2d + SiLU, branch sum, LayerNorm over the Cout channels of the pixel ----
    float rh[2][8], vv[2][8], u[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) u[j] = 0.f;
#pragma unroll
    for (int brn = 0; brn < 2; ++brn) {
      const CnPtBranch& r = a.br[brn];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float mu, rho;
        if (a.training) { mu = r.mean2[o0 + j]; rho = r.rstd2[o0 + j]; }
        else { mu = r.rm2[o0 + j]; rho = 1.0f / sqrtf(r.rv2[o0 + j] + a.eps2); }
        rh[brn][j] = (r_[brn][j] - mu) * rho;
        vv[brn][j] = r.g2[o0 + j] * rh[brn][j] + r.b2[o0 + j];
        u[j] += cn_silu(vv[brn][j]);
      }
    }
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) part += u[j];
    red[wid * 64 + px] = part;
    __syncthreads();
    float m = 0.f;
    for (int w2 = 0; w2 < NOG; ++w2) m += red[w2 * 64 + px];
    m *= vN;
    float q2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = u[j] - m; q2 += d * d; }
    red[(NOG + wid) * 64 + px] = q2;
    __syncthreads();
    float var = 0.f;
    for (int w2 = 0; w2 < NOG; ++w2) var += red[(NOG + w2) * 64 + px];
    const float rL = 1.0f / sqrtf(var * vN + a.epsL);
    float uh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) uh[j] = (u[j] - m) * rL;
    if (PASS == 2) {
      float yv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) yv[j] = a.gL[o0 + j] * uh[j] + a.bL[o0 + j];
      if (valid) {
        if (a.out_kind == 0) {
          float* yp = reinterpret_cast<float*>(a.y) + (long)b * a.y_stride + (long)o0 * HW + l;
#pragma unroll
          for (int j = 0; j < 8; ++j) yp[(long)j * HW] = yv[j];
        } else {
          *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(a.y) + p * a.y_stride + o0) = cn_pack8(yv);
        }
      }
      continue;
    }
    // ---- backward: LayerNorm, SiLU, BatchNorm2d ----
    float dyv[8];
    if (a.out_kind == 0) {
      const float* dp = reinterpret_cast<const float*>(a.dy) + (long)b * a.dy_stride + (long)o0 * HW + l;
#pragma unroll
      for (int j = 0; j < 8; ++j) dyv[j] = valid ? dp[(long)j * HW] : 0.f;
    } else {
      const long pc = valid ? p : 0;
      cn_unpack8(*reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(a.dy) + pc * a.dy_stride + o0), dyv);
#pragma unroll
      for (int j = 0; j < 8; ++j) dyv[j] *= vm;
    }
    float gg[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      gg[j] = dyv[j] * a.gL[o0 + j];
      s1 += gg[j];
      s2 += gg[j] * uh[j];
    }
    __syncthreads();  // (the variance sums in red[] have been read)
    red[wid * 64 + px] = s1;
    red[(NOG + wid) * 64 + px] = s2;
    __syncthreads();
    float mg = 0.f, mgu = 0.f;
    for (int w2 = 0; w2 < NOG; ++w2) { mg += red[w2 * 64 + px]; mgu += red[(NOG + w2) * 64 + px]; }
    mg *= vN; mgu *= vN;
    float dv[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float du = rL * (gg[j] - mg - uh[j] * mgu);
#pragma unroll
      for (int brn = 0; brn < 2; ++brn) dv[brn][j] = du * cn_silu_grad(vv[brn][j]);
    }
    if (PASS == 3) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int brn = 0; brn < 2; ++brn) {
          A2[brn][0][j] += dv[brn][j];
          A2[brn][1][j] += dv[brn][j] * rh[brn][j];
        }
        AL[0][j] += dyv[j] * uh[j];
        AL[1][j] += dyv[j];
      }
      continue;
    }
    // ---- PASS 4: dr -> LDS; dWb tiles; da -> dz (scratch) + BatchNorm3d sums ----
#pragma unroll
    for (int brn = 0; brn < 2; ++brn) {
      const CnPtBranch& r = a.br[brn];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float c0 = a.coef2[(brn * 2) * Cout + o0 + j], c1 = a.coef2[(brn * 2 + 1) * Cout + o0 + j];
        float rho;
        if (a.training) rho = r.rstd2[o0 + j];
        else rho = 1.0f / sqrtf(r.rv2[o0 + j] + a.eps2);
        drs[(brn * Cout + o0 + j) * 64 + px] = r.g2[o0 + j] * rho * (dv[brn][j] - c0 - rh[brn][j] * c1) * vm;
      }
    }
    __syncthreads();
    {
      // dWb[brn][o][loc] += sum_px dr[brn][o][px] * a[brn][loc][px]: a thread owns 2 couts x 4 entries per iteration
      const int half = Cout >> 1;
      const int nt3 = half * (pt_ceil4(E3) >> 2), nt5 = half * (pt_ceil4(E5) >> 2);
#pragma unroll
      for (int it = 0; it < MAXIT; ++it) {
        const int ti = tid + it * nthr;
        if (ti >= nt3 + nt5) break;
        const int brn = ti >= nt3 ? 1 : 0;
        const int tl = ti - (brn ? nt3 : 0);
        const int o2 = tl % half, e4 = tl / half;
        const float* d0 = drs + (brn * Cout + 2 * o2) * 64;
        const float* a0 = as_ + ((brn ? S5 : 0) + e4 * 4) * 64;
#pragma unroll 4
        for (int q4 = 0; q4 < 16; ++q4) {
          const f32x4 da_ = *reinterpret_cast<const f32x4*>(d0 + q4 * 4);
          const f32x4 db_ = *reinterpret_cast<const f32x4*>(d0 + 64 + q4 * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(a0 + i * 64 + q4 * 4);
            AW[it][i] += da_[0] * av[0] + da_[1] * av[1] + da_[2] * av[2] + da_[3] * av[3];
            AW[it][4 + i] += db_[0] * av[0] + db_[1] * av[1] + db_[2] * av[2] + db_[3] * av[3];
          }
        }
      }
    }
    for (int e = wid; e < E; e += NOG) {
      const int brn = e >= E3 ? 1 : 0;
      const CnPtBranch& r = a.br[brn];
      const int loc = e - (brn ? E3 : 0);
      const int Tp = brn ? T5 : T3;
      const int cp = loc / Tp;
      const float* w = r.wbt + (long)loc * Cout;
      const float* dr = drs + brn * Cout * 64 + px;
      float da = 0.f;
      for (int o = 0; o < Cout; ++o) da += w[o] * dr[o * 64];
      const float hh = hs[e * 64 + px];
      const float dzv = da * cn_silu_grad(r.g3[cp] * hh + r.b3[cp]);
      const int slot = (brn * C + cp) * 2;
      acc[slot * nthr + tid] += dzv;
      acc[(slot + 1) * nthr + tid] += dzv * hh;
      if (valid) a.dz[(long)e * a.P + p] = dzv;
    }
  }

  // ---- block row -> two-level last-block reduction -> finish ----
  if (PASS == 2) return;
  const int blk = blockIdx.x;
  __syncthreads();
  if (PASS == 0 || PASS == 4) {
    // LDS accumulators [slot][thread]: wave sums, then the NOG wave partials per slot
    const int nslot = 4 * C;
    float* wsum = xs;  // (x tile no longer needed) [nslot][NOG]
    for (int s = 0; s < nslot; ++s) {
      const float v = cn_wave_sum(acc[s * nthr + tid]);
      if (px == 0) wsum[s * NOG + wid] = v;
    }
    __syncthreads();
    const int base = PASS == 4 ? Cout * E : 0;
    if (tid < nslot) {
      float v = 0.f;
      for (int w2 = 0; w2 < NOG; ++w2) v += wsum[tid * NOG + w2];
      cn_t2_store(a.tk, blk, base + tid, v);
    }
  }
  if (PASS == 1 || PASS == 3) {
#pragma unroll
    for (int brn = 0; brn < 2; ++brn)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = cn_wave_sum(A2[brn][s][j]);
          if (px == 0) cn_t2_store(a.tk, blk, (brn * 2 + s) * Cout + o0 + j, v);
        }
    if (PASS == 3) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = cn_wave_sum(AL[s][j]);
          if (px == 0) cn_t2_store(a.tk, blk, (4 + s) * Cout + o0 + j, v);
        }
    }
  }
  if (PASS == 4) {
    const int half = Cout >> 1;
    const int nt3 = half * (pt_ceil4(E3) >> 2), nt5 = half * (pt_ceil4(E5) >> 2);
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int ti = tid + it * nthr;
      if (ti >= nt3 + nt5) break;
      const int brn = ti >= nt3 ? 1 : 0;
      const int tl = ti - (brn ? nt3 : 0);
      const int o2 = tl % half, e4 = tl / half;
      const int Eb = brn ? E5 : E3;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int loc = e4 * 4 + i;
        if (loc < Eb) {
          cn_t2_store(a.tk, blk, (brn ? Cout * E3 : 0) + (2 * o2) * Eb + loc, AW[it][i]);
          cn_t2_store(a.tk, blk, (brn ? Cout * E3 : 0) + (2 * o2 + 1) * Eb + loc, AW[it][4 + i]);
        }
      }
    }
  }
  if (PASS == 5) {
    const int nout = 8 * C * C;
    for (int q = wid; q < nout; q += NOG) {
      const float v = cn_wave_sum(acc[q * 64 + px]);
      if (px == 0) cn_t2_store(a.tk, blk, q, v);
    }
  }
  // finish (the last-arriving block): statistics / coefficients / parameter gradients
  const double cntP = (double)a.P;
  cn_t2_reduce_fn(a.tk, blk, &s_flag, [&](int col, double tot) {
    if (PASS == 0 || PASS == 1 || PASS == 3) {
      // pairs {stat 0, stat 1} are finished together below (they sit in different columns): park the totals
      reinterpret_cast<double*>(lds)[col] = tot;
    } else if (PASS == 4) {
      if (col < Cout * E) {
        const int brn = col >= Cout * E3 ? 1 : 0;
        a.br[brn].dwb[col - (brn ? Cout * E3 : 0)] += (float)tot;
      } else {
        reinterpret_cast<double*>(lds)[col - Cout * E] = tot;
      }
    } else {  // PASS 5
      const int n3 = C * C * 3;
      const int brn = col >= n3 ? 1 : 0;
      a.br[brn].dwa[col - (brn ? n3 : 0)] += (float)tot;
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
static inline int pt_row_width(int PASS, int C, int T, int Cout) {
  const int E = C * (T - 2) + C * (T - 4);
  switch (PASS) {
    case 0: return 4 * C;
    case 1: return 4 * Cout;
    case 3: return 6 * Cout;
    case 4: return Cout * E + 4 * C;
    case 5: return 8 * C * C;
    default: return 0;
  }
}
static inline int pt_blocks(long P) {
  const long tiles = (P + 63) / 64;
  return (int)(tiles < PT_MAX_BLOCKS ? tiles : PT_MAX_BLOCKS);
}
static inline bool pt_supported(int C, int T, int Cout) {
  return C >= 1 && C <= PT_MAX_C && T >= 5 && Cout >= 8 && Cout <= 64 && (Cout & 7) == 0;
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
    if ((size_t)pt_lds(ps, C, T, Cout, 8 * Cout).total * 4 > 160 * 1024) return -1;
  if (with_backward) {
    const int nt = (Cout / 2) * ((pt_ceil4(C * (T - 2)) + pt_ceil4(C * (T - 4))) / 4);
    if ((nt + 8 * Cout - 1) / (8 * Cout) > 4) return -1;
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
  a.ntiles = (int)((a.P + 63) / 64);
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
  const int nthr = 8 * a.Cout;
  const int nblk = PASS == 2 ? (a.ntiles < 2048 ? a.ntiles : 2048) : pt_blocks(a.P);
  const PtLds L = pt_lds(PASS, a.C, a.T, a.Cout, nthr);
  size_t shmem = (size_t)L.total * 4;
  const int W = pt_row_width(PASS, a.C, a.T, a.Cout);
  // the finish phase parks up to 6 * Cout (or 4 * C) doubles at the head of the LDS
  const size_t park = (size_t)(6 * a.Cout > 4 * a.C ? 6 * a.Cout : 4 * a.C) * 8;
  if (shmem < park) shmem = park;
  if (shmem > 160 * 1024) return CN_ERR_LDS;
  if (W > 0) a.tk = cn_t2_carve(reinterpret_cast<int*>(ws), ws + pt_off_body(), nblk, W);
  int maxit = 1;
  if (PASS == 4) {
    const int E3 = a.C * (a.T - 2), E5 = a.C * (a.T - 4);
    const int nt = (a.Cout / 2) * ((pt_ceil4(E3) + pt_ceil4(E5)) / 4);
    maxit = (nt + nthr - 1) / nthr;
    if (maxit > 4) return CN_ERR_ARG;
  }
#define PT_GO(MI)                                                                                              \
  do {                                                                                                         \
    if (shmem > 64 * 1024)                                                                                     \
      (void)hipFuncSetAttribute((const void*)cn_pretime_kernel<PASS, MI>,                                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);                       \
    CN_LAUNCH((cn_pretime_kernel<PASS, MI>), dim3(nblk), dim3(nthr), shmem, stream, a);                       \
  } while (0)
  if constexpr (PASS != 4) {
    PT_GO(1);
  } else {
    if (maxit == 1) PT_GO(1);
    else if (maxit == 2) PT_GO(2);
    else if (maxit == 3) PT_GO(3);
    else PT_GO(4);
  }
#undef PT_GO
  return CN_OK;
}

// params: HOST array of 22 device pointers: per branch (k = 3, then k = 5) {wa, wb, gamma3, beta3, running_mean3,
// running_var3, gamma2, beta2, running_mean2, running_var2}, then {ln_gamma, ln_beta}. stats: HOST array of 8 device
// pointers: per branch {mean3 [C], rstd3 [C], mean2 [Cout], rstd2 [Cout]} (written in training mode, read by backward).
// bn: HOST {eps3, momentum3, eps2, momentum2}. y: out_kind 0 fp32 NCHW (y_stride = batch stride), 1 bf16 NHWC (pixel
// stride). Returns CN_ERR_ARG for shapes outside the fused kernel (the caller keeps its generic path), CN_ERR_LDS if
// the LDS image of a pass does not fit.
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
