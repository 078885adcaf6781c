// BatchNorm (+SiLU, +residual) and LayerNorm over channels on bf16 NHWC activations (gfx950, HBM-bound).
//
// Mixed-precision path (BASELINE configs[2]; the reference's precision="16-mixed", model.py:168-186): activations
// bf16 [pixels][C] with a pixel stride `ld` (channel slices of concat buffers in place), statistics, parameters and
// parameter gradients fp32 (reference ops: nn.BatchNorm2d + SiLU of ConvBlock2d, nn/modules/convolution.py:71-120;
// nn.LayerNorm around NeighborhoodAttention2D, convolution.py:338-353).
// A thread owns ONE 8-channel group (16 bytes) and walks pixels, so its per-channel constants and partial sums live
// in registers and every access is a full 16-byte lane load of a contiguous pixel row.
#include <cstdlib>
#include "cn_bf16.h"
#include "cn_ticket.h"

#define BBN_MAX_BLOCKS 512

// ---- per-channel sums over pixels ---------------------------------------------------------------------------
// MODE 0: {sum x, sum x^2}                      (BatchNorm forward statistics)
// MODE 1: {sum dz, sum dz*xhat}, dz = dy*act'(z) (BatchNorm backward)
// part[(blk*2 + k) * C + c]
template <int MODE>
__global__ __launch_bounds__(256) void cn_bbn_partial_kernel(const bf16_t* __restrict__ x, long ldx,
                                                            const bf16_t* __restrict__ dy, long lddy,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, long P, int C, int act,
                                                            long rows_per_block, float* __restrict__ part) {
  __shared__ float red[256 * 16];
  const int C8 = C >> 3;
  const int R = 256 / C8;
  const int tid = threadIdx.x;
  const int row = tid / C8, cg = tid - row * C8;
  const bool live = row < R;
  float a1[8], a2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a1[j] = a2[j] = 0.f;
  float m[8], rs[8], ga[8], be[8];
  if (MODE == 1 && live) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      m[j] = mean[cg * 8 + j]; rs[j] = rstd[cg * 8 + j]; ga[j] = gamma[cg * 8 + j]; be[j] = beta[cg * 8 + j];
    }
  }
  const long p0 = blockIdx.x * rows_per_block;
  const long p1 = p0 + rows_per_block < P ? p0 + rows_per_block : P;
  if (live) {
    // two pixels per iteration: both 16-byte loads (four in MODE 1) are in flight before either is consumed
    auto accum = [&](const u32x4& xr, const u32x4& dr) {
      float xv[8];
      cn_unpack8(xr, xv);
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { a1[j] += xv[j]; a2[j] += xv[j] * xv[j]; }
      } else {
        float dv[8];
        cn_unpack8(dr, dv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = (xv[j] - m[j]) * rs[j];
          float dz = dv[j];
          if (act == 1) dz *= cn_silu_grad(ga[j] * xh + be[j]);
          a1[j] += dz;
          a2[j] += dz * xh;
        }
      }
    };
    const u32x4 z4 = {0u, 0u, 0u, 0u};
    long p = p0 + row;
    for (; p + R < p1; p += 2 * R) {
      const u32x4 xa = *reinterpret_cast<const u32x4*>(x + p * ldx + cg * 8);
      const u32x4 xb = *reinterpret_cast<const u32x4*>(x + (p + R) * ldx + cg * 8);
      u32x4 da = z4, db = z4;
      if (MODE == 1) {
        da = *reinterpret_cast<const u32x4*>(dy + p * lddy + cg * 8);
        db = *reinterpret_cast<const u32x4*>(dy + (p + R) * lddy + cg * 8);
      }
      accum(xa, da);
      accum(xb, db);
    }
    if (p < p1) {
      const u32x4 xa = *reinterpret_cast<const u32x4*>(x + p * ldx + cg * 8);
      u32x4 da = z4;
      if (MODE == 1) da = *reinterpret_cast<const u32x4*>(dy + p * lddy + cg * 8);
      accum(xa, da);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[tid * 16 + j] = a1[j]; red[tid * 16 + 8 + j] = a2[j]; }
  __syncthreads();
  for (int idx = tid; idx < C8 * 16; idx += 256) {
    const int g2 = idx >> 4, j = idx & 15;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += red[(r * C8 + g2) * 16 + j];
    part[((long)blockIdx.x * 2 + (j >> 3)) * C + g2 * 8 + (j & 7)] = s;
  }
}

// Per-channel fp64 combine of the partial-sum rows: one WAVE per channel (4 channels per 256-thread block), lane l
// walks rows l, l+64, ... in four independent chains, then a shuffle tree. Rows come from the statistics pass
// (<= 512) or from the convolution epilogue (one per pixel tile: 2560 at batch 32 x 100^2), so the serial depth is what
// matters: 16 lanes per channel took 17.5 us for 2560 rows, one thread per channel 80 us for 512.
#define BBN_FIN_CH 4  // channels per finalize block
__device__ __forceinline__ void bbn_combine(const float* __restrict__ part, int nblk, int C, int c, int sub, double& s,
                                            double& ss) {
  s = 0.0; ss = 0.0;
  if (c < C) {
    double s1 = 0.0, q1 = 0.0, s2 = 0.0, q2 = 0.0, s3 = 0.0, q3 = 0.0;
    int i = sub;
    for (; i + 192 < nblk; i += 256) {
      s += part[((long)i * 2) * C + c]; ss += part[((long)i * 2 + 1) * C + c];
      s1 += part[((long)(i + 64) * 2) * C + c]; q1 += part[((long)(i + 64) * 2 + 1) * C + c];
      s2 += part[((long)(i + 128) * 2) * C + c]; q2 += part[((long)(i + 128) * 2 + 1) * C + c];
      s3 += part[((long)(i + 192) * 2) * C + c]; q3 += part[((long)(i + 192) * 2 + 1) * C + c];
    }
    for (; i < nblk; i += 64) { s += part[((long)i * 2) * C + c]; ss += part[((long)i * 2 + 1) * C + c]; }
    s += s1 + s2 + s3; ss += q1 + q2 + q3;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off, 64); ss += __shfl_xor(ss, off, 64); }
}

// Forward finalize: batch mean / rstd (fp64 combine), running-statistics update (momentum, unbiased variance).
// `part` rows come from the statistics pass or straight from the convolution's epilogue (one row per pixel tile).
__global__ __launch_bounds__(256) void cn_bbn_finalize_fwd_kernel(const float* __restrict__ part, int nblk, int C,
                                                                 double count,
                                                                 float eps, float momentum,
                                                                 float* __restrict__ running_mean,
                                                                 float* __restrict__ running_var,
                                                                 float* __restrict__ mean, float* __restrict__ rstd,
                                                                 int training) {
  const int c = blockIdx.x * BBN_FIN_CH + (threadIdx.x >> 6), sub = threadIdx.x & 63;
  if (!training) {
    if (c < C && sub == 0) {
      mean[c] = running_mean[c];
      rstd[c] = 1.0f / sqrtf(running_var[c] + eps);
    }
    return;
  }
  double s = 0.0, ss = 0.0;
  bbn_combine(part, nblk, C, c, sub, s, ss);
  if (c >= C || sub != 0) return;
  const double md = s / count;
  double var = ss / count - md * md;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)md;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean != nullptr) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)md;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// Backward finalize: coef = {mean(dz), mean(dz*xhat)}; dgamma += sum dz*xhat, dbeta += sum dz.
__global__ __launch_bounds__(256) void cn_bbn_finalize_bwd_kernel(const float* __restrict__ part, int nblk, int C,
                                                                 double count, float* __restrict__ coef,
                                                                 float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, int training) {
  const int c = blockIdx.x * BBN_FIN_CH + (threadIdx.x >> 6), sub = threadIdx.x & 63;
  double s1, s2;
  bbn_combine(part, nblk, C, c, sub, s1, s2);
  if (c >= C || sub != 0) return;
  coef[c] = training ? (float)(s1 / count) : 0.f;
  coef[C + c] = training ? (float)(s2 / count) : 0.f;
  dgamma[c] += (float)s2;
  dbeta[c] += (float)s1;
}

// MODE 0: y = act(gamma*(x-mean)*rstd + beta) (+ res)
// MODE 1: dx (+)= gamma*rstd*(dz - coef0 - xhat*coef1), dz = dy*act'(z)
template <int MODE>
__global__ __launch_bounds__(256) void cn_bbn_apply_kernel(const bf16_t* __restrict__ x, long ldx,
                                                          const bf16_t* __restrict__ dy, long lddy,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ coef,
                                                          const bf16_t* __restrict__ res, long ldr,
                                                          bf16_t* __restrict__ y, long ldy, long P, int C, int act,
                                                          int accumulate, long rows_per_block) {
  const int C8 = C >> 3;
  const int R = 256 / C8;
  const int tid = threadIdx.x;
  const int row = tid / C8, cg = tid - row * C8;
  if (row >= R) return;
  float sc[8], sh[8], m[8], rs[8], c0[8], c1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cg * 8 + j;
    m[j] = mean[c]; rs[j] = rstd[c];
    sc[j] = gamma[c] * rs[j];
    sh[j] = beta[c] - m[j] * sc[j];
    if (MODE == 1) { c0[j] = coef[c]; c1[j] = coef[C + c]; }
  }
  const long p0 = blockIdx.x * rows_per_block;
  const long p1 = p0 + rows_per_block < P ? p0 + rows_per_block : P;
  const u32x4 z4 = {0u, 0u, 0u, 0u};
  auto body = [&](long p, const u32x4& xr, const u32x4& ar, const u32x4& br) {
    float xv[8], o[8];
    cn_unpack8(xr, xv);
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float z = xv[j] * sc[j] + sh[j];
        if (act == 1) z = cn_silu(z);
        o[j] = z;
      }
      if (res != nullptr) {
        float rv[8];
        cn_unpack8(ar, rv);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += rv[j];
      }
    } else {
      float dv[8];
      cn_unpack8(ar, dv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (xv[j] - m[j]) * rs[j];
        float dz = dv[j];
        if (act == 1) dz *= cn_silu_grad(xv[j] * sc[j] + sh[j]);
        o[j] = sc[j] * (dz - c0[j] - xh * c1[j]);
      }
      if (accumulate) {
        float ov[8];
        cn_unpack8(br, ov);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += ov[j];
      }
    }
    *reinterpret_cast<u32x4*>(y + p * ldy + cg * 8) = cn_pack8(o);
  };
  // second operand: residual (MODE 0) or dy (MODE 1); third: the dx being accumulated into
  auto ld2 = [&](long p) -> u32x4 {
    if (MODE == 0) return res != nullptr ? *reinterpret_cast<const u32x4*>(res + p * ldr + cg * 8) : z4;
    return *reinterpret_cast<const u32x4*>(dy + p * lddy + cg * 8);
  };
  auto ld3 = [&](long p) -> u32x4 {
    return (MODE == 1 && accumulate) ? *reinterpret_cast<const u32x4*>(y + p * ldy + cg * 8) : z4;
  };
  long p = p0 + row;
  for (; p + R < p1; p += 2 * R) {  // two pixels per iteration: all loads issued before the first is consumed
    const u32x4 xa = *reinterpret_cast<const u32x4*>(x + p * ldx + cg * 8);
    const u32x4 xb = *reinterpret_cast<const u32x4*>(x + (p + R) * ldx + cg * 8);
    const u32x4 aa = ld2(p), ab = ld2(p + R);
    const u32x4 ba = ld3(p), bb = ld3(p + R);
    body(p, xa, aa, ba);
    body(p + R, xb, ab, bb);
  }
  if (p < p1) {
    const u32x4 xa = *reinterpret_cast<const u32x4*>(x + p * ldx + cg * 8);
    const u32x4 aa = ld2(p);
    const u32x4 ba = ld3(p);
    body(p, xa, aa, ba);
  }
}

// grid of the apply passes: no per-block output rows, so many more (smaller) blocks than the statistics passes
static inline void bbn_apply_grid(long P, int C, int& nblk, long& rows) {
  const int R = 256 / (C >> 3);
  long want = (P + (long)R * 4 - 1) / ((long)R * 4);  // >= 4 row-iterations per block
  if (want > 4096) want = 4096;
  if (want < 1) want = 1;
  rows = (P + want - 1) / want;
  rows = (rows + R - 1) / R * R;
  nblk = (int)((P + rows - 1) / rows);
}

static inline void bbn_grid(long P, int C, int& nblk, long& rows) {
  const int R = 256 / (C >> 3);
  long want = (P + (long)R * 8 - 1) / ((long)R * 8);  // >= 8 row-iterations per block
  if (want > BBN_MAX_BLOCKS) want = BBN_MAX_BLOCKS;
  if (want < 1) want = 1;
  rows = (P + want - 1) / want;
  rows = (rows + R - 1) / R * R;
  nblk = (int)((P + rows - 1) / rows);
}

// Floats of scratch for the calls below: per-block partial sums + backward coefficients.
extern "C" long cn_bn_workspace_floats_bf16(int C) { return (long)BBN_MAX_BLOCKS * 2 * C + 2 * C + 4096 + 128L * C; }

// y = act(bn(x)) (+ res). x, res, y: bf16 [P][C] rows with pixel strides; C % 8 == 0, C <= 2048.
// training: batch statistics (saved to mean / rstd, running stats updated); else running statistics.
// conv_sums (nullable, training only): conv_rows rows of {sum, sum of squares}[C] from cn_conv2d_fwd_bf16's epilogue
// -- skips the statistics pass over x.
extern "C" int cn_bn_act_fwd_bf16(const void* x, long ldx, const float* gamma, const float* beta, float* running_mean,
                                  float* running_var, const void* res, long ldr, void* y, long ldy, float* mean,
                                  float* rstd, float* ws, long P, int C, int training, float momentum, float eps,
                                  int act, const float* conv_sums, int conv_rows, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (P <= 0 || C <= 0) return CN_OK;
  if ((C & 7) || C > 2048) return CN_ERR_ARG;
  if (!training && (running_mean == nullptr || running_var == nullptr)) return CN_ERR_ARG;
  int nblk;
  long rows;
  bbn_grid(P, C, nblk, rows);
  if (training && !(conv_sums != nullptr && conv_rows > 0))
    CN_LAUNCH((cn_bbn_partial_kernel<0>), dim3(nblk), dim3(256), 0, stream, (const bf16_t*)x, ldx, nullptr,
                       0L, nullptr, nullptr, nullptr, nullptr, P, C, act, rows, ws);
  const bool fused = training && conv_sums != nullptr && conv_rows > 0;
  CN_LAUNCH(cn_bbn_finalize_fwd_kernel, dim3((C + BBN_FIN_CH - 1) / BBN_FIN_CH), dim3(256), 0, stream, fused ? conv_sums : ws,
                     fused ? conv_rows : nblk, C, (double)P, eps, momentum, running_mean, running_var, mean, rstd,
                     training);
  int ablk;
  long arows;
  bbn_apply_grid(P, C, ablk, arows);
  CN_LAUNCH((cn_bbn_apply_kernel<0>), dim3(ablk), dim3(256), 0, stream, (const bf16_t*)x, ldx, nullptr, 0L,
                     mean, rstd, gamma, beta, nullptr, (const bf16_t*)res, ldr, (bf16_t*)y, ldy, P, C, act, 0, arows);
  return cn_check_launch();
}

// dx (+)= d act(bn(x)) / dx . dy; dgamma / dbeta are ACCUMULATED. dx nullable (parameter gradients only).
extern "C" int cn_bn_act_bwd_bf16(const void* x, long ldx, const void* dy, long lddy, const float* mean,
                                  const float* rstd, const float* gamma, const float* beta, void* dx, long lddx,
                                  float* dgamma, float* dbeta, float* ws, long P, int C, int training, int act,
                                  int accumulate_dx, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (P <= 0 || C <= 0) return CN_OK;
  if ((C & 7) || C > 2048) return CN_ERR_ARG;
  int nblk;
  long rows;
  bbn_grid(P, C, nblk, rows);
  float* coef = ws + (long)BBN_MAX_BLOCKS * 2 * C;
  CN_LAUNCH((cn_bbn_partial_kernel<1>), dim3(nblk), dim3(256), 0, stream, (const bf16_t*)x, ldx,
                     (const bf16_t*)dy, lddy, mean, rstd, gamma, beta, P, C, act, rows, ws);
  CN_LAUNCH(cn_bbn_finalize_bwd_kernel, dim3((C + BBN_FIN_CH - 1) / BBN_FIN_CH), dim3(256), 0, stream, ws, nblk, C, (double)P,
                     coef, dgamma, dbeta, training);
  if (dx != nullptr) {
    int ablk;
    long arows;
    bbn_apply_grid(P, C, ablk, arows);
    CN_LAUNCH((cn_bbn_apply_kernel<1>), dim3(ablk), dim3(256), 0, stream, (const bf16_t*)x, ldx,
                       (const bf16_t*)dy, lddy, mean, rstd, gamma, beta, coef, nullptr, 0L, (bf16_t*)dx, lddx, P, C,
                       act, accumulate_dx, arows);
  }
  return cn_check_launch();
}

// =====================================================================================================================
// Grouped BatchNorm(+SiLU): the G (<= 4) BatchNorm layers of one ResidualAConv level (convolution.py:376-395) in one
// launch per pass, and a ONE-LAUNCH finalize. Round 3's profile of the mixed-precision step: 84 finalize launches of
// 6-9 us (one wave per channel reading 4-byte columns of the partial rows: every 128-byte line fetched by 32 waves),
// the running sum `res + SiLU(BN_0) + SiLU(BN_1)` read and re-written once per branch, dy of the summed level read
// once per branch and pass. Here:
//   * finalize = coalesced column sums (thread = column of the [rows][2C] partial matrix, a block = a slice of rows,
//     fp64) whose per-block slices are combined by the LAST ARRIVING block of each group (device ticket: slices
//     stored write-through, one relaxed agent-scope fetch-add per block, the last arriver reads the slices with
//     agent-scope loads -- MI355X_MICROARCH.md 'inter-workgroup visibility'; cdna_hip_programming.md 5 'in-launch split-K
//     reduction'); the order of the sums is fixed by the indices, not by the arrival order: bit-reproducible;
//   * forward apply (summed level): y = res + sum_g act(bn_g(x_g)) in ONE pass, accumulated in fp32, rounded once;
//   * backward (summed level): dy is read once per pass for all G branches.
// =====================================================================================================================
#define BBG_MAX 4
#define BBG_FIN_BLOCKS 64   // row slices per group in the finalize
#define BBG_CNT_INTS 512    // ticket counters at the start of the workspace (zero on entry, left zero on exit):
                            // one per (group, 32-channel group): 4 x 64 at most

struct CnBBnGroupArgs {
  const bf16_t* x[BBG_MAX];
  const bf16_t* dy[BBG_MAX];
  const float* gamma[BBG_MAX];
  const float* beta[BBG_MAX];
  float* mean[BBG_MAX];
  float* rstd[BBG_MAX];
  float* running_mean[BBG_MAX];
  float* running_var[BBG_MAX];
  bf16_t* y[BBG_MAX];
  bf16_t* dx[BBG_MAX];
  float* dgamma[BBG_MAX];
  float* dbeta[BBG_MAX];
  const float* rows[BBG_MAX];  // finalize input: [nrows][2][C] fp32 partial rows of group g
  int accumulate_dx[BBG_MAX];
  const bf16_t* res;
  long ldx, lddy, ldy, lddx, ldr;
  long P, rows_per_block;
  int G, C, act, training, nrows;
  long row_pitch;   // floats between consecutive partial rows of one group
  float eps, momentum;
  float* coef;      // [G][2][C]
  double* slices;   // [G][ceil(C/32)][BBG_FIN_BLOCKS][64] finalize scratch
  int* counters;    // [G][ceil(C/32)] tickets
};

// agent-scope (write-through / L2-bypassing) accesses for data handed between workgroups inside one launch
__device__ __forceinline__ void bbg_store_agent(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double bbg_load_agent(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The finalize arithmetic of one channel (shared by the stand-alone finalize kernel and the ticketed partial kernel).
template <int MODE>
__device__ __forceinline__ void bbg_finish_channel(const CnBBnGroupArgs& a, int g, int c, double v0, double v1) {
  const double count = (double)a.P;
  if (MODE == 0) {
    const double md = v0 / count;
    double var = v1 / count - md * md;
    if (var < 0.0) var = 0.0;
    a.mean[g][c] = (float)md;
    a.rstd[g][c] = (float)(1.0 / sqrt(var + (double)a.eps));
    if (a.running_mean[g] != nullptr) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      a.running_mean[g][c] = (1.f - a.momentum) * a.running_mean[g][c] + a.momentum * (float)md;
      a.running_var[g][c] = (1.f - a.momentum) * a.running_var[g][c] + a.momentum * (float)unbiased;
    }
  } else {
    float* coef = a.coef + (long)g * 2 * a.C;
    coef[c] = a.training ? (float)(v0 / count) : 0.f;
    coef[a.C + c] = a.training ? (float)(v1 / count) : 0.f;
    a.dgamma[g][c] += (float)v1;
    a.dbeta[g][c] += (float)v0;
  }
}

// MODE 0 (forward): columns = {sum x, sum x^2}[C]  -> mean / rstd (+ running statistics)
// MODE 1 (backward): columns = {sum dz, sum dz*xhat}[C] -> coef = column / count; dgamma += col1, dbeta += col0
// grid (NB row slices, ceil(C / 32) channel groups, G). A block = 64 columns (32 channels x {stat 0, stat 1}: two
// 128-byte segments per row) x 4 row lanes: every thread has ALL its loads in flight at once (<= 16), so the whole
// reduction is two load latencies deep -- the first version (one thread per column walking 40 rows, then 64 slices, four
// and two loads at a time) took 23 us per launch, all of it dependent round trips.
#define BBG_ROW_LOADS 16  // rows per thread and batch in phase 1
template <int MODE>
__global__ __launch_bounds__(256) void cn_bbn_group_finalize_kernel(const CnBBnGroupArgs a) {
  __shared__ double red[4][64];
  __shared__ int s_last;
  const int g = blockIdx.z, cgp = blockIdx.y, b = blockIdx.x, nb = gridDim.x, ncg = gridDim.y;
  const int q = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int ch = cgp * 32 + (q & 31), stat = q >> 5;
  const bool live = ch < a.C;
  // (a dead channel of a ragged last group reads channel C - 1 and discards it: no predicated loads, see below)
  const float* __restrict__ rows = a.rows[g] + (long)stat * a.C + (live ? ch : a.C - 1);
  const int rpb = (a.nrows + nb - 1) / nb;
  const int r0 = b * rpb, r1 = r0 + rpb < a.nrows ? r0 + rpb : a.nrows;
  const long rp = a.row_pitch;
  double acc = 0.0;
  for (int rb = r0 + rl; rb < r1; rb += 4 * BBG_ROW_LOADS) {
    float v[BBG_ROW_LOADS];
#pragma unroll
    for (int i = 0; i < BBG_ROW_LOADS; ++i) {
      // clamped, never predicated: a per-element "load or zero" makes hipcc branch around every load and wait for it
      const int r = rb + 4 * i;
      v[i] = rows[(long)(r < r1 ? r : r1 - 1) * rp];
    }
#pragma unroll
    for (int i = 0; i < BBG_ROW_LOADS; ++i) acc += (rb + 4 * i < r1) ? (double)v[i] : 0.0;
  }
  red[rl][q] = acc;
  __syncthreads();
  double* slices = a.slices + (((long)g * ncg + cgp) * nb) * 64;  // [nb][64] of this (group, channel group)
  if (rl == 0) bbg_store_agent(slices + (long)b * 64 + q, (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]));
  // publish: the storing wave drains its write-through stores, the block meets, one lane draws the ticket
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int* counter = a.counters + g * ncg + cgp;
  if (threadIdx.x == 0) {
    const int t = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == nb - 1);
    if (t == nb - 1) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
  }
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // (compiler ordering only: the loads below are agent-scope)
  // the last arriver: slices k = rl, rl + 4, ... (fixed order), all loads of a batch in flight together
  double tot = 0.0;
  for (int kb = rl; kb < nb; kb += 4 * BBG_ROW_LOADS) {
    double v[BBG_ROW_LOADS];
#pragma unroll
    for (int i = 0; i < BBG_ROW_LOADS; ++i) {
      const int k = kb + 4 * i;
      v[i] = bbg_load_agent(slices + (long)(k < nb ? k : nb - 1) * 64 + q);
    }
#pragma unroll
    for (int i = 0; i < BBG_ROW_LOADS; ++i) tot += (kb + 4 * i < nb) ? v[i] : 0.0;
  }
  __syncthreads();
  red[rl][q] = tot;
  __syncthreads();
  if (threadIdx.x >= 32 || !live) return;
  // threads 0..31: channel ch, v0 = column of stat 0, v1 = column of stat 1
  const double v0 = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
  const double v1 = (red[0][32 + q] + red[1][32 + q]) + (red[2][32 + q] + red[3][32 + q]);
  bbg_finish_channel<MODE>(a, g, ch, v0, v1);
}

// eval mode: mean / rstd from the running statistics (no batch statistics, no ticket)
__global__ __launch_bounds__(256) void cn_bbn_group_eval_stats_kernel(const CnBBnGroupArgs a) {
  const int g = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < a.C) {
    a.mean[g][c] = a.running_mean[g][c];
    a.rstd[g][c] = 1.0f / sqrtf(a.running_var[g][c] + a.eps);
  }
}

struct CnBBnTickets { CnTicket2 t[BBG_MAX]; };

// Per-channel sums over pixels AND their finalize in one launch: GS groups handled by ONE thread (GS = 2: the summed
// level, dy shared), or group blockIdx.y (GS = 1). Every block stores its row {k}[C] of group g write-through; the
// two-level last-block reduction (cn_t2_reduce, one ticket domain per group) hands the column totals to the last
// arriver, which finishes the channels: MODE 0 mean / rstd / running statistics, MODE 1 backward coefficients and
// dgamma / dbeta. No stand-alone finalize launch.
template <int MODE, int GS>
__global__ __launch_bounds__(256) void cn_bbn_group_partial_kernel(const CnBBnGroupArgs a, const CnBBnTickets tks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem);                    // 256 * 16 floats
  double* tot = reinterpret_cast<double*>(smem + 256 * 16 * 4);    // 2C doubles
  __shared__ int s_flag;
  const int C = a.C, C8 = C >> 3;
  const int R = 256 / C8;
  const int tid = threadIdx.x;
  const int row = tid / C8, cg = tid - row * C8;
  const bool live = row < R;
  const int g0 = GS == 1 ? blockIdx.y : 0;
  float a1[GS][8], a2[GS][8], m[GS][8], rs[GS][8], ga[GS][8], be[GS][8];
#pragma unroll
  for (int q = 0; q < GS; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a1[q][j] = a2[q][j] = 0.f;
      if (MODE == 1 && live) {
        const int c = cg * 8 + j;
        m[q][j] = a.mean[g0 + q][c]; rs[q][j] = a.rstd[g0 + q][c];
        ga[q][j] = a.gamma[g0 + q][c]; be[q][j] = a.beta[g0 + q][c];
      }
    }
  const long p0 = blockIdx.x * a.rows_per_block;
  const long p1 = p0 + a.rows_per_block < a.P ? p0 + a.rows_per_block : a.P;
  if (live) {
    const bool shared_dy = GS > 1;  // the summed level: one dy for every branch
    auto accum = [&](int q, const u32x4& xq, const u32x4& dq) {
      float xv[8];
      cn_unpack8(xq, xv);
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { a1[q][j] += xv[j]; a2[q][j] += xv[j] * xv[j]; }
      } else {
        float dv[8];
        cn_unpack8(dq, dv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = (xv[j] - m[q][j]) * rs[q][j];
          float dz = dv[j];
          if (a.act == 1) dz *= cn_silu_grad(ga[q][j] * xh + be[q][j]);
          a1[q][j] += dz;
          a2[q][j] += dz * xh;
        }
      }
    };
    const u32x4 z4 = {0u, 0u, 0u, 0u};
    long p = p0 + row;
    if (GS == 1) {
      // two pixels per iteration: both 16-byte loads (four in MODE 1) are in flight before either is consumed
      const bf16_t* __restrict__ xp = a.x[g0];
      const bf16_t* __restrict__ dp = a.dy[g0];
      for (; p + R < p1; p += 2 * R) {
        const u32x4 xa = *reinterpret_cast<const u32x4*>(xp + p * a.ldx + cg * 8);
        const u32x4 xb = *reinterpret_cast<const u32x4*>(xp + (p + R) * a.ldx + cg * 8);
        u32x4 da = z4, db = z4;
        if (MODE == 1) {
          da = *reinterpret_cast<const u32x4*>(dp + p * a.lddy + cg * 8);
          db = *reinterpret_cast<const u32x4*>(dp + (p + R) * a.lddy + cg * 8);
        }
        accum(0, xa, da);
        accum(0, xb, db);
      }
    }
    for (; p < p1; p += R) {
      u32x4 xr[GS], dr[GS];
#pragma unroll
      for (int q = 0; q < GS; ++q) {
        xr[q] = *reinterpret_cast<const u32x4*>(a.x[g0 + q] + p * a.ldx + cg * 8);
        dr[q] = z4;
      }
      if (MODE == 1) {
        dr[0] = *reinterpret_cast<const u32x4*>(a.dy[g0] + p * a.lddy + cg * 8);
#pragma unroll
        for (int q = 1; q < GS; ++q)
          dr[q] = shared_dy ? dr[0] : *reinterpret_cast<const u32x4*>(a.dy[g0 + q] + p * a.lddy + cg * 8);
      }
#pragma unroll
      for (int q = 0; q < GS; ++q) accum(q, xr[q], dr[q]);
    }
  }
#pragma unroll
  for (int q = 0; q < GS; ++q) {
    const CnTicket2& tk = tks.t[g0 + q];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[tid * 16 + j] = a1[q][j]; red[tid * 16 + 8 + j] = a2[q][j]; }
    __syncthreads();
    for (int idx = tid; idx < C8 * 16; idx += 256) {
      const int g2 = idx >> 4, j = idx & 15;
      float s = 0.f;
      for (int r = 0; r < R; ++r) s += red[(r * C8 + g2) * 16 + j];
      cn_t2_store(tk, blockIdx.x, (j >> 3) * C + g2 * 8 + (j & 7), s);
    }
    if (cn_t2_reduce(tk, blockIdx.x, &s_flag, tot))
      for (int c = tid; c < C; c += 256) bbg_finish_channel<MODE>(a, g0 + q, c, tot[c], tot[C + c]);
  }
}

// Forward apply. GS >= 1 with a.G == GS summed in the thread: y[0] = res + sum_g act(bn_g(x_g)); GS == 0: plain,
// group = blockIdx.y: y[g] = act(bn_g(x_g)).
template <int GS>
__global__ __launch_bounds__(256) void cn_bbn_group_apply_fwd_kernel(const CnBBnGroupArgs a) {
  constexpr int NG = GS == 0 ? 1 : GS;
  const int C8 = a.C >> 3;
  const int R = 256 / C8;
  const int tid = threadIdx.x;
  const int row = tid / C8, cg = tid - row * C8;
  if (row >= R) return;
  const int g0 = GS == 0 ? blockIdx.y : 0;
  float sc[NG][8], sh[NG][8];
#pragma unroll
  for (int q = 0; q < NG; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cg * 8 + j;
      const float r_ = a.rstd[g0 + q][c];
      sc[q][j] = a.gamma[g0 + q][c] * r_;
      sh[q][j] = a.beta[g0 + q][c] - a.mean[g0 + q][c] * sc[q][j];
    }
  const long p0 = blockIdx.x * a.rows_per_block;
  const long p1 = p0 + a.rows_per_block < a.P ? p0 + a.rows_per_block : a.P;
  const bf16_t* __restrict__ res = a.res;
  bf16_t* __restrict__ y = a.y[g0];
  const u32x4 z4 = {0u, 0u, 0u, 0u};
  auto one = [&](long p, const u32x4* xr, const u32x4& rr) {
    float o[8];
    if (res != nullptr) cn_unpack8(rr, o);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = 0.f;
    }
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      float xv[8];
      cn_unpack8(xr[q], xv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float z = xv[j] * sc[q][j] + sh[q][j];
        if (a.act == 1) z = cn_silu(z);
        o[j] += z;
      }
    }
    *reinterpret_cast<u32x4*>(y + p * a.ldy + cg * 8) = cn_pack8(o);
  };
  long p = p0 + row;
  for (; p + R < p1; p += 2 * R) {  // two pixels in flight: 2 * (NG + 1) sixteen-byte loads before the first use
    u32x4 xa[NG], xb[NG];
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      xa[q] = *reinterpret_cast<const u32x4*>(a.x[g0 + q] + p * a.ldx + cg * 8);
      xb[q] = *reinterpret_cast<const u32x4*>(a.x[g0 + q] + (p + R) * a.ldx + cg * 8);
    }
    const u32x4 ra = res != nullptr ? *reinterpret_cast<const u32x4*>(res + p * a.ldr + cg * 8) : z4;
    const u32x4 rb = res != nullptr ? *reinterpret_cast<const u32x4*>(res + (p + R) * a.ldr + cg * 8) : z4;
    one(p, xa, ra);
    one(p + R, xb, rb);
  }
  if (p < p1) {
    u32x4 xa[NG];
#pragma unroll
    for (int q = 0; q < NG; ++q) xa[q] = *reinterpret_cast<const u32x4*>(a.x[g0 + q] + p * a.ldx + cg * 8);
    const u32x4 ra = res != nullptr ? *reinterpret_cast<const u32x4*>(res + p * a.ldr + cg * 8) : z4;
    one(p, xa, ra);
  }
}

// Backward apply: dx_g (+)= gamma*rstd*(dz - coef0 - xhat*coef1). GS = 2: both branches of a summed level in the
// thread (dy read once); GS = 1: group blockIdx.y.
template <int GS>
__global__ __launch_bounds__(256) void cn_bbn_group_apply_bwd_kernel(const CnBBnGroupArgs a) {
  const int C = a.C, C8 = C >> 3;
  const int R = 256 / C8;
  const int tid = threadIdx.x;
  const int row = tid / C8, cg = tid - row * C8;
  if (row >= R) return;
  const int g0 = GS == 1 ? blockIdx.y : 0;
  float sc[GS][8], sh[GS][8], m[GS][8], rs[GS][8], c0[GS][8], c1[GS][8];
#pragma unroll
  for (int q = 0; q < GS; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cg * 8 + j;
      m[q][j] = a.mean[g0 + q][c]; rs[q][j] = a.rstd[g0 + q][c];
      sc[q][j] = a.gamma[g0 + q][c] * rs[q][j];
      sh[q][j] = a.beta[g0 + q][c] - m[q][j] * sc[q][j];
      c0[q][j] = a.coef[(long)(g0 + q) * 2 * C + c];
      c1[q][j] = a.coef[(long)(g0 + q) * 2 * C + C + c];
    }
  const long p0 = blockIdx.x * a.rows_per_block;
  const long p1 = p0 + a.rows_per_block < a.P ? p0 + a.rows_per_block : a.P;
  for (long p = p0 + row; p < p1; p += R) {
    u32x4 xr[GS], dr[GS], orr[GS];
#pragma unroll
    for (int q = 0; q < GS; ++q) {
      xr[q] = *reinterpret_cast<const u32x4*>(a.x[g0 + q] + p * a.ldx + cg * 8);
      if (q == 0 || GS == 1) dr[q] = *reinterpret_cast<const u32x4*>(a.dy[g0 + q] + p * a.lddy + cg * 8);
      else dr[q] = dr[0];
      if (a.dx[g0 + q] != nullptr && a.accumulate_dx[g0 + q])
        orr[q] = *reinterpret_cast<const u32x4*>(a.dx[g0 + q] + p * a.lddx + cg * 8);
    }
#pragma unroll
    for (int q = 0; q < GS; ++q) {
      if (a.dx[g0 + q] == nullptr) continue;
      float xv[8], dv[8], o[8];
      cn_unpack8(xr[q], xv);
      cn_unpack8(dr[q], dv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (xv[j] - m[q][j]) * rs[q][j];
        float dz = dv[j];
        if (a.act == 1) dz *= cn_silu_grad(xv[j] * sc[q][j] + sh[q][j]);
        o[j] = sc[q][j] * (dz - c0[q][j] - xh * c1[q][j]);
      }
      if (a.accumulate_dx[g0 + q]) {
        float ov[8];
        cn_unpack8(orr[q], ov);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += ov[j];
      }
      *reinterpret_cast<u32x4*>(a.dx[g0 + q] + p * a.lddx + cg * 8) = cn_pack8(o);
    }
  }
}

// Workspace of the grouped calls (floats). HEAD (fixed, whatever G and C are -- the engine hands ONE buffer to calls of
// every shape): BBG_CNT_INTS finalize tickets + BBG_MAX * CN_T2_COUNTERS tickets of the statistics passes, ZERO on
// entry, left zero by every launch. BODY: finalize slices (fp64), backward coefficients, per-block rows.
static inline long bbg_slices_off() { return CN_BNWS_HEAD_INTS; }  // 8-byte aligned (cn_ticket.h: the fixed head)
static inline long bbg_coef_off(int G, int C) {
  return bbg_slices_off() + (long)G * ((C + 31) / 32) * BBG_FIN_BLOCKS * 64 * 2;  // doubles
}
static inline long bbg_part_off(int G, int C) { return (bbg_coef_off(G, C) + (long)G * 2 * C + 63) / 64 * 64; }
static inline long bbg_domain_floats(int C) { return (cn_t2_body_floats(BBN_MAX_BLOCKS, 2 * C) + 63) / 64 * 64; }
extern "C" long cn_bn_group_workspace_floats_bf16(int G, int C) {
  const long own = bbg_part_off(G, C) + (long)G * bbg_domain_floats(C);
  // a convolution that finishes its own statistics (cn_conv2d_fwd_grouped_bnstats_bf16) parks its group rows at the
  // start of the body: (CN_T2_COUNTERS - 1) x G x 2 x (couts padded to a 128-cout block) doubles
  const long conv = bbg_slices_off() + 2L * (CN_T2_COUNTERS - 1) * G * 2 * ((C + 127) / 128 * 128);
  return own > conv ? own : conv;
}
static inline CnBBnTickets bbg_tickets(float* ws, int G, int C, int nblk) {
  CnBBnTickets t = {};
  for (int g = 0; g < G; ++g)
    t.t[g] = cn_t2_carve(reinterpret_cast<int*>(ws) + BBG_CNT_INTS + g * CN_T2_COUNTERS,
                         ws + bbg_part_off(G, C) + (long)g * bbg_domain_floats(C), nblk, 2 * C);
  return t;
}
static inline size_t bbg_partial_shmem(int C) { return 256 * 16 * 4 + (size_t)2 * C * 8; }

// row slices of the finalize: >= 4 rows per block (one per row lane), at most BBG_FIN_BLOCKS
static inline int bbg_fin_blocks(int nrows) {
  int nb = (nrows + 3) / 4;
  if (nb > BBG_FIN_BLOCKS) nb = BBG_FIN_BLOCKS;
  return nb < 1 ? 1 : nb;
}

static void bbg_common(CnBBnGroupArgs& a, int G, const void* const* xs, long ldx, const float* const* gammas,
                       const float* const* betas, float* const* means, float* const* rstds, float* ws, long P, int C,
                       int act, int training) {
  a.G = G; a.C = C; a.P = P; a.act = act; a.training = training; a.ldx = ldx;
  for (int g = 0; g < G; ++g) {
    a.x[g] = (const bf16_t*)xs[g]; a.gamma[g] = gammas[g]; a.beta[g] = betas[g];
    a.mean[g] = means[g]; a.rstd[g] = rstds[g];
  }
  a.counters = reinterpret_cast<int*>(ws);
  a.slices = reinterpret_cast<double*>(ws + bbg_slices_off());
  a.coef = ws + bbg_coef_off(G, C);
}

// ys: G outputs (sum_outputs == 0) or ys[0] = res + sum_g act(bn_g(xs[g])) (sum_outputs != 0; res nullable).
// conv_sums (nullable, training): per-group `stats` rows of the convolution epilogue (conv_rows rows of [2][C]).
extern "C" int cn_bn_act_group_fwd_bf16(int G, const void* const* xs, long ldx, const float* const* gammas,
                                        const float* const* betas, float* const* running_means,
                                        float* const* running_vars, const void* res, long ldr, void* const* ys,
                                        long ldy, float* const* means, float* const* rstds, float* ws, long P, int C,
                                        int training, float momentum, float eps, int act, int sum_outputs,
                                        const float* const* conv_sums, int conv_rows, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (P <= 0 || C <= 0) return CN_OK;
  if (G < 1 || G > BBG_MAX || (C & 7) || C > 2048) return CN_ERR_ARG;
  if (res != nullptr && !sum_outputs) return CN_ERR_ARG;
  CnBBnGroupArgs a = {};
  bbg_common(a, G, xs, ldx, gammas, betas, means, rstds, ws, P, C, act, training);
  a.eps = eps; a.momentum = momentum; a.res = (const bf16_t*)res; a.ldr = ldr; a.ldy = ldy;
  for (int g = 0; g < G; ++g) {
    a.running_mean[g] = running_means ? running_means[g] : nullptr;
    a.running_var[g] = running_vars ? running_vars[g] : nullptr;
    a.y[g] = (bf16_t*)ys[sum_outputs ? 0 : g];
    if (!training && (a.running_mean[g] == nullptr || a.running_var[g] == nullptr)) return CN_ERR_ARG;
  }
  if (training) {
    const bool fused = conv_sums != nullptr && conv_rows > 0;
    if (conv_rows == -1) {
      // means / rstds (and the running statistics) were written by the convolution launch that produced xs
      // (cn_conv2d_fwd_grouped_bnstats_bf16 reported `finalized`): nothing to do before the apply
    } else if (fused) {
      for (int g = 0; g < G; ++g) a.rows[g] = conv_sums[g];
      a.nrows = conv_rows;
      a.row_pitch = 2L * C;
      CN_LAUNCH((cn_bbn_group_finalize_kernel<0>), dim3(bbg_fin_blocks(a.nrows), (C + 31) / 32, G), dim3(256), 0,
                stream, a);
    } else {  // no conv-epilogue rows: a statistics pass over x that finishes itself (ticketed)
      int nblk;
      long rows;
      bbn_grid(P, C, nblk, rows);
      a.rows_per_block = rows;
      CN_LAUNCH((cn_bbn_group_partial_kernel<0, 1>), dim3(nblk, G), dim3(256), bbg_partial_shmem(C), stream, a,
                bbg_tickets(ws, G, C, nblk));
    }
  } else {
    CN_LAUNCH(cn_bbn_group_eval_stats_kernel, dim3((C + 255) / 256, G), dim3(256), 0, stream, a);
  }
  int ablk;
  long arows;
  bbn_apply_grid(P, C, ablk, arows);
  a.rows_per_block = arows;
  if (!sum_outputs) {
    CN_LAUNCH((cn_bbn_group_apply_fwd_kernel<0>), dim3(ablk, G), dim3(256), 0, stream, a);
  } else {
    switch (G) {
      case 1: CN_LAUNCH((cn_bbn_group_apply_fwd_kernel<1>), dim3(ablk), dim3(256), 0, stream, a); break;
      case 2: CN_LAUNCH((cn_bbn_group_apply_fwd_kernel<2>), dim3(ablk), dim3(256), 0, stream, a); break;
      case 3: CN_LAUNCH((cn_bbn_group_apply_fwd_kernel<3>), dim3(ablk), dim3(256), 0, stream, a); break;
      default: CN_LAUNCH((cn_bbn_group_apply_fwd_kernel<4>), dim3(ablk), dim3(256), 0, stream, a); break;
    }
  }
  return cn_check_launch();
}

// dys[g]: gradient of output g (shared_dy != 0: the summed level, every dys[g] is the same tensor and is read once).
// dxs[g] nullable; dgamma / dbeta ACCUMULATED.
extern "C" int cn_bn_act_group_bwd_bf16(int G, const void* const* xs, long ldx, const void* const* dys, long lddy,
                                        const float* const* means, const float* const* rstds,
                                        const float* const* gammas, const float* const* betas, void* const* dxs,
                                        long lddx, const int* accumulate_dx, float* const* dgammas,
                                        float* const* dbetas, float* ws, long P, int C, int training, int act,
                                        void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (P <= 0 || C <= 0) return CN_OK;
  if (G < 1 || G > BBG_MAX || (C & 7) || C > 2048) return CN_ERR_ARG;
  CnBBnGroupArgs a = {};
  bbg_common(a, G, xs, ldx, gammas, betas, const_cast<float* const*>(means), const_cast<float* const*>(rstds), ws, P, C,
             act, training);
  a.lddy = lddy; a.lddx = lddx;
  // dy shared by the two branches of a summed level: the in-thread form (one read of dy for both branches) measured
  // SLOWER than one grid row per branch (partial 49 vs 31 us, apply 42 vs 24 us at 32 x 100^2 x 128: 143-149 VGPRs, three
  // waves per SIMD with one pixel in flight each -- latency-bound, and the second read of dy comes out of the Infinity
  // Cache anyway). Kept behind CN_BBN_SHARED_DY=1 for experiments.
  static const bool allow_shared = getenv("CN_BBN_SHARED_DY") != nullptr;
  bool shared = G == 2 && allow_shared;
  bool any_dx = false;
  for (int g = 0; g < G; ++g) {
    a.dy[g] = (const bf16_t*)dys[g]; a.dx[g] = (bf16_t*)dxs[g]; a.accumulate_dx[g] = accumulate_dx[g];
    a.dgamma[g] = dgammas[g]; a.dbeta[g] = dbetas[g];
    if (dys[g] != dys[0]) shared = false;
    any_dx = any_dx || dxs[g] != nullptr;
  }
  int nblk;
  long rows;
  bbn_grid(P, C, nblk, rows);
  a.rows_per_block = rows;
  // statistics pass + its finalize (coefficients, dgamma / dbeta) in ONE launch: the last-arriving block finishes
  const CnBBnTickets tks = bbg_tickets(ws, G, C, nblk);
  if (shared) CN_LAUNCH((cn_bbn_group_partial_kernel<1, 2>), dim3(nblk), dim3(256), bbg_partial_shmem(C), stream, a, tks);
  else CN_LAUNCH((cn_bbn_group_partial_kernel<1, 1>), dim3(nblk, G), dim3(256), bbg_partial_shmem(C), stream, a, tks);
  if (any_dx) {
    int ablk;
    long arows;
    bbn_apply_grid(P, C, ablk, arows);
    a.rows_per_block = arows;
    if (shared) CN_LAUNCH((cn_bbn_group_apply_bwd_kernel<2>), dim3(ablk), dim3(256), 0, stream, a);
    else CN_LAUNCH((cn_bbn_group_apply_bwd_kernel<1>), dim3(ablk, G), dim3(256), 0, stream, a);
  }
  return cn_check_launch();
}

// ---- LayerNorm over the channel axis (rows of the NHWC image) -------------------------------------------------
// C8 = C/8 lanes share a pixel (C8 a power of two <= 64); statistics by lane shuffles, two-pass in registers.
template <int C8>
__device__ __forceinline__ float ln_group_sum(float v) {
#pragma unroll
  for (int off = C8 >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <int C8>
__global__ __launch_bounds__(256) void cn_bln_fwd_kernel(const bf16_t* __restrict__ x, long ldx,
                                                        const float* __restrict__ w, const float* __restrict__ bi,
                                                        const bf16_t* __restrict__ res, long ldr,
                                                        bf16_t* __restrict__ y, long ldy, long P, float eps) {
  constexpr int R = 256 / C8;
  const int tid = threadIdx.x;
  const int row = tid / C8, cg = tid % C8;
  float wv[8], bv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { wv[j] = w[cg * 8 + j]; bv[j] = bi[cg * 8 + j]; }
  const float invC = 1.0f / (float)(C8 * 8);
  for (long p = (long)blockIdx.x * R + row; p < P; p += (long)gridDim.x * R) {
    float xv[8], o[8];
    cn_unpack8(*reinterpret_cast<const u32x4*>(x + p * ldx + cg * 8), xv);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += xv[j];
    const float mu = ln_group_sum<C8>(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = xv[j] - mu; q += d * d; }
    const float rs = rsqrtf(ln_group_sum<C8>(q) * invC + eps);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (xv[j] - mu) * rs * wv[j] + bv[j];
    if (res != nullptr) {
      float rv[8];
      cn_unpack8(*reinterpret_cast<const u32x4*>(res + p * ldr + cg * 8), rv);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += rv[j];
    }
    *reinterpret_cast<u32x4*>(y + p * ldy + cg * 8) = cn_pack8(o);
  }
}

template <int C8>
__global__ __launch_bounds__(256) void cn_bln_bwd_kernel(const bf16_t* __restrict__ x, long ldx,
                                                        const bf16_t* __restrict__ dy, long lddy,
                                                        const float* __restrict__ w, bf16_t* __restrict__ dx,
                                                        long lddx, float* __restrict__ dw, float* __restrict__ db,
                                                        long P, float eps, int accumulate) {
  __shared__ float red[256 * 16];
  constexpr int R = 256 / C8;
  const int tid = threadIdx.x;
  const int row = tid / C8, cg = tid % C8;
  float wv[8], gw[8], gb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { wv[j] = w[cg * 8 + j]; gw[j] = 0.f; gb[j] = 0.f; }
  const float invC = 1.0f / (float)(C8 * 8);
  for (long p = (long)blockIdx.x * R + row; p < P; p += (long)gridDim.x * R) {
    float xv[8], dv[8], o[8];
    cn_unpack8(*reinterpret_cast<const u32x4*>(x + p * ldx + cg * 8), xv);
    cn_unpack8(*reinterpret_cast<const u32x4*>(dy + p * lddy + cg * 8), dv);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += xv[j];
    const float mu = ln_group_sum<C8>(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = xv[j] - mu; q += d * d; }
    const float rs = rsqrtf(ln_group_sum<C8>(q) * invC + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = (xv[j] - mu) * rs;
      const float gj = dv[j] * wv[j];
      s1 += gj;
      s2 += gj * xh;
      gw[j] += dv[j] * xh;
      gb[j] += dv[j];
      xv[j] = xh;
      dv[j] = gj;
    }
    const float m1 = ln_group_sum<C8>(s1) * invC, m2 = ln_group_sum<C8>(s2) * invC;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = rs * (dv[j] - m1 - xv[j] * m2);
    if (accumulate) {
      float ov[8];
      cn_unpack8(*reinterpret_cast<const u32x4*>(dx + p * lddx + cg * 8), ov);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += ov[j];
    }
    *reinterpret_cast<u32x4*>(dx + p * lddx + cg * 8) = cn_pack8(o);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[tid * 16 + j] = gw[j]; red[tid * 16 + 8 + j] = gb[j]; }
  __syncthreads();
  for (int idx = tid; idx < C8 * 16; idx += 256) {
    const int g2 = idx >> 4, j = idx & 15;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += red[(r * C8 + g2) * 16 + j];
    atomicAdd((j >> 3 ? db : dw) + g2 * 8 + (j & 7), s);
  }
}

#define BLN_DISPATCH(KERNEL, ...)                                                                       \
  switch (C >> 3) {                                                                                     \
    case 1: CN_LAUNCH((KERNEL<1>), grid, dim3(256), 0, stream, __VA_ARGS__); break;           \
    case 2: CN_LAUNCH((KERNEL<2>), grid, dim3(256), 0, stream, __VA_ARGS__); break;           \
    case 4: CN_LAUNCH((KERNEL<4>), grid, dim3(256), 0, stream, __VA_ARGS__); break;           \
    case 8: CN_LAUNCH((KERNEL<8>), grid, dim3(256), 0, stream, __VA_ARGS__); break;           \
    case 16: CN_LAUNCH((KERNEL<16>), grid, dim3(256), 0, stream, __VA_ARGS__); break;         \
    case 32: CN_LAUNCH((KERNEL<32>), grid, dim3(256), 0, stream, __VA_ARGS__); break;         \
    case 64: CN_LAUNCH((KERNEL<64>), grid, dim3(256), 0, stream, __VA_ARGS__); break;         \
    default: return CN_ERR_ARG;                                                                         \
  }

// y = LayerNorm_C(x) * w + b (+ res); C in {8, 16, 32, 64, 128, 256, 512}.
extern "C" int cn_layernorm_c_fwd_bf16(const void* x, long ldx, const float* w, const float* b, const void* res,
                                       long ldr, void* y, long ldy, long P, int C, float eps, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (P <= 0) return CN_OK;
  if (C & 7) return CN_ERR_ARG;
  const int R = 256 / (C >> 3);
  long nb = (P + R - 1) / R;
  if (nb > 4096) nb = 4096;
  const dim3 grid((unsigned)nb);
  BLN_DISPATCH(cn_bln_fwd_kernel, (const bf16_t*)x, ldx, w, b, (const bf16_t*)res, ldr, (bf16_t*)y, ldy, P, eps);
  return cn_check_launch();
}

// dx (+)= ...; dw / db ACCUMULATED with fp32 atomics (one add per block and channel).
extern "C" int cn_layernorm_c_bwd_bf16(const void* x, long ldx, const void* dy, long lddy, const float* w, void* dx,
                                       long lddx, float* dw, float* db, long P, int C, float eps, int accumulate_dx,
                                       void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (P <= 0) return CN_OK;
  if (C & 7) return CN_ERR_ARG;
  const int R = 256 / (C >> 3);
  long nb = (P + (long)R * 8 - 1) / ((long)R * 8);
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  const dim3 grid((unsigned)nb);
  BLN_DISPATCH(cn_bln_bwd_kernel, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, w, (bf16_t*)dx, lddx, dw, db, P, eps,
               accumulate_dx);
  return cn_check_launch();
}

// bias gradients on the bf16 path: out[c] (+)= sum_p x[p][c] in ONE launch: per-block column sums (<= 512 blocks),
// combined by the two-level last-block reduction above. ws: cn_bn_workspace_floats_bf16(C) floats whose leading words
// are the ticket counters: ZERO before the first call (every call leaves them zero).
__global__ __launch_bounds__(256) void cn_bsum_ticket_kernel(const bf16_t* __restrict__ x, long ldx, long P, int C,
                                                            long rows_per_block, const CnTicket2 tk,
                                                            float* __restrict__ out, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem);                 // 256 * 8 floats
  double* tot = reinterpret_cast<double*>(smem + 256 * 8 * 4);  // C doubles
  __shared__ int s_flag;
  const int C8 = C >> 3;
  const int R = 256 / C8;
  const int tid = threadIdx.x;
  const int row = tid / C8, cg = tid - row * C8;
  float a1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a1[j] = 0.f;
  const long p0 = blockIdx.x * rows_per_block;
  const long p1 = p0 + rows_per_block < P ? p0 + rows_per_block : P;
  if (row < R) {
    long p = p0 + row;
    for (; p + R < p1; p += 2 * R) {
      const u32x4 xa = *reinterpret_cast<const u32x4*>(x + p * ldx + cg * 8);
      const u32x4 xb = *reinterpret_cast<const u32x4*>(x + (p + R) * ldx + cg * 8);
      float va[8], vb[8];
      cn_unpack8(xa, va);
      cn_unpack8(xb, vb);
#pragma unroll
      for (int j = 0; j < 8; ++j) a1[j] += va[j] + vb[j];
    }
    if (p < p1) {
      float va[8];
      cn_unpack8(*reinterpret_cast<const u32x4*>(x + p * ldx + cg * 8), va);
#pragma unroll
      for (int j = 0; j < 8; ++j) a1[j] += va[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[tid * 8 + j] = a1[j];
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    const int g2 = c >> 3, j = c & 7;
    float sum = 0.f;
    for (int r = 0; r < R; ++r) sum += red[(r * C8 + g2) * 8 + j];
    cn_t2_store(tk, blockIdx.x, c, sum);
  }
  if (!cn_t2_reduce(tk, blockIdx.x, &s_flag, tot)) return;
  for (int c = tid; c < C; c += 256) out[c] = accumulate ? out[c] + (float)tot[c] : (float)tot[c];
}

extern "C" int cn_channel_sum_bf16(const void* x, long ldx, long P, int C, float* out, int accumulate, float* ws,
                                   void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (P <= 0 || C <= 0) return CN_OK;
  if ((C & 7) || C > 2048) return CN_ERR_ARG;
  int nblk;
  long rows;
  bbn_grid(P, C, nblk, rows);
  const CnTicket2 tk = cn_t2_carve(reinterpret_cast<int*>(ws), ws + CN_T2_COUNTERS, nblk, C);
  const size_t shmem = 256 * 8 * 4 + (size_t)C * 8;
  CN_LAUNCH(cn_bsum_ticket_kernel, dim3(nblk), dim3(256), shmem, stream, (const bf16_t*)x, ldx, P, C, rows, tk, out,
            accumulate);
  return cn_check_launch();
}
