// Tanimoto loss family (forward sums, per-sample coefficients, elementwise backward) and the
// label preparation of get_true_labels, fused. HBM-bound single-pass reductions (gfx950).
//
// Reference: TanimotoComplementLoss / TanimotoDistLoss / CombinedLoss / LossPreprocessing
// (/root/reference/src/cultionet/losses/losses.py:9-340), get_true_labels + calc_loss
// (/root/reference/src/cultionet/models/lightning.py:161-354).
//
// Per sample five running sums suffice (Sy, Syh, Syyh, Syy, Syhyh over C*H*W after masking);
// the complement terms follow algebraically with N = C*H*W because masked pixels are zeroed
// BEFORE the complement (they enter it as (1,1) pairs).
#include "cn_common.h"

// target modes
#define TGT_FLOAT 0   // target_f [B][C][HW] float (regression: bdist)
#define TGT_EQ 1      // labels [B][HW] int64: t = (y == klass)            (true_edge)
#define TGT_RANGE 2   // labels: t = (0 < y < klass)                       (true_crop)
#define TGT_ONEHOT 3  // labels: t[c] = (y == c)                           (LossPreprocessing one-hot, C > 1)
// mask modes
#define MSK_NONE 0
#define MSK_LABEL 1   // m = (labels != -1)   (get_true_labels; identity when no -1 is present)
#define MSK_I64 2     // explicit int64 mask [B][HW]
#define MSK_F32 3     // explicit float mask [B][HW]

__device__ __forceinline__ float ls_target(int mode, const float* tf, const long long* lab, long b, int c, int C,
                                           long HW, long p, int klass) {
  if (mode == TGT_FLOAT) return tf[(b * C + c) * HW + p];
  const long long y = lab[b * HW + p];
  if (mode == TGT_EQ) return y == klass ? 1.f : 0.f;
  if (mode == TGT_RANGE) return (y > 0 && y < klass) ? 1.f : 0.f;
  return y == c ? 1.f : 0.f;
}

__device__ __forceinline__ float ls_mask(int mode, const long long* lab, const void* mk, long b, long HW, long p) {
  if (mode == MSK_NONE) return 1.f;
  if (mode == MSK_LABEL) return lab[b * HW + p] != -1 ? 1.f : 0.f;
  if (mode == MSK_I64) return (float)((const long long*)mk)[b * HW + p];
  return ((const float*)mk)[b * HW + p];
}

// sums[b][5] (double, atomically accumulated: zero first)
__global__ __launch_bounds__(256) void cn_tanimoto_sums_kernel(const float* __restrict__ pred, long pbs,
                                                              const float* __restrict__ tf,
                                                              const long long* __restrict__ lab,
                                                              const void* __restrict__ mk, int tmode, int mmode,
                                                              int klass, int C, long HW, double* __restrict__ sums) {
  __shared__ double scratch[4];
  const long b = blockIdx.y;
  const long n = (long)C * HW;
  double a[5] = {0, 0, 0, 0, 0};
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i / HW);
    const long p = i - c * HW;
    const float m = ls_mask(mmode, lab, mk, b, HW, p);
    const float yh = pred[b * pbs + i] * m;
    const float y = ls_target(tmode, tf, lab, b, c, C, HW, p, klass) * m;
    a[0] += y;
    a[1] += yh;
    a[2] += (double)y * yh;
    a[3] += (double)y * y;
    a[4] += (double)yh * yh;
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const double r = cn_block_sum<double, 256>(a[k], scratch);
    if (threadIdx.x == 0) atomicAdd(sums + b * 5 + k, r);
  }
}

// loss kinds
#define LOSS_COMPLEMENT 0  // TanimotoComplementLoss (depth 5)
#define LOSS_DIST 1        // TanimotoDistLoss
#define LOSS_COMBINED 2    // mean of the two

// f(tpl, sq) and its partials for the two distance definitions
__device__ __forceinline__ void tnm_complement(double tpl, double sq, double smooth, int depth, double& f,
                                               double& dtpl, double& dsq) {
  double den = 0.0, dden_tpl = 0.0, dden_sq = 0.0;
  for (int d = 0; d < depth; ++d) {
    const double a = (double)(1 << d), bb = -(2.0 * a - 1.0);
    const double q = a * sq + bb * tpl + smooth;
    den += 1.0 / q;
    dden_tpl += -bb / (q * q);
    dden_sq += -a / (q * q);
  }
  const double num = tpl + smooth, scale = 1.0 / depth;
  f = 1.0 - num * den * scale;
  dtpl = -(den + num * dden_tpl) * scale;
  dsq = -(num * dden_sq) * scale;
}

__device__ __forceinline__ void tnm_dist(double tpl, double sq, double smooth, double& f, double& dtpl, double& dsq) {
  const double num = tpl + smooth, den = sq - tpl + smooth;
  f = 1.0 - num / den;
  dtpl = -(1.0 / den + num / (den * den));
  dsq = num / (den * den);
}

// One thread per sample: loss_b and coefficients coef[b][4] = {A, Bq, Ac, Bc} so that
//   dL/dyhat_m = A*y + 2*Bq*yh - Ac*(1-y) - 2*Bc*(1-yh)      (all per sample, already /B and *0.5)
// loss_out[0] = mean_b loss_b ; loss_b_out[b] optional.
__global__ void cn_tanimoto_finalize_kernel(const double* __restrict__ sums, int B, double N, int kind, float smooth,
                                            int depth, float* __restrict__ loss_out, float* __restrict__ coef,
                                            float weight, float* __restrict__ total_out) {
  __shared__ double scratch[4];
  double lsum = 0.0;
  for (int b = threadIdx.x; b < B; b += 256) {
    const double Sy = sums[b * 5 + 0], Syh = sums[b * 5 + 1], tpl = sums[b * 5 + 2];
    const double sq = sums[b * 5 + 3] + sums[b * 5 + 4];
    const double tplc = N - Sy - Syh + tpl;                       // sum (1-y)(1-yh)
    const double sqc = 2.0 * N - 2.0 * Sy - 2.0 * Syh + sq;       // sum (1-y)^2 + (1-yh)^2
    double f1 = 0, a1 = 0, b1 = 0, f2 = 0, a2 = 0, b2 = 0, w = 1.0;
    if (kind == LOSS_COMPLEMENT || kind == LOSS_COMBINED) {
      double f, da, db;
      tnm_complement(tpl, sq, smooth, depth, f, da, db); f1 += f; a1 += da; b1 += db;
      tnm_complement(tplc, sqc, smooth, depth, f, da, db); f2 += f; a2 += da; b2 += db;
    }
    if (kind == LOSS_DIST || kind == LOSS_COMBINED) {
      double f, da, db;
      tnm_dist(tpl, sq, smooth, f, da, db); f1 += f; a1 += da; b1 += db;
      tnm_dist(tplc, sqc, smooth, f, da, db); f2 += f; a2 += da; b2 += db;
    }
    if (kind == LOSS_COMBINED) w = 0.5;
    const double lb = 0.5 * (f1 + f2) * w;
    lsum += lb;
    const double sc = 0.5 * w / B;
    coef[b * 4 + 0] = (float)(a1 * sc);
    coef[b * 4 + 1] = (float)(b1 * sc);
    coef[b * 4 + 2] = (float)(a2 * sc);
    coef[b * 4 + 3] = (float)(b2 * sc);
  }
  lsum = cn_block_sum<double, 256>(lsum, scratch);
  if (threadIdx.x == 0) {
    loss_out[0] = (float)(lsum / B);
    if (total_out != nullptr) total_out[0] += weight * (float)(lsum / B);
  }
}

// dpred[b][i] (+)= upstream * m * (A*y + 2*Bq*yh - Ac*(1-y) - 2*Bc*(1-yh)),  yh,y masked
__global__ __launch_bounds__(256) void cn_tanimoto_bwd_kernel(const float* __restrict__ pred, long pbs,
                                                             const float* __restrict__ tf,
                                                             const long long* __restrict__ lab,
                                                             const void* __restrict__ mk, int tmode, int mmode,
                                                             int klass, int C, long HW, const float* __restrict__ coef,
                                                             float upstream, float* __restrict__ dpred, long dbs,
                                                             int accumulate) {
  const long b = blockIdx.y;
  const long n = (long)C * HW;
  const float A = coef[b * 4 + 0], Bq = coef[b * 4 + 1], Ac = coef[b * 4 + 2], Bc = coef[b * 4 + 3];
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i / HW);
    const long p = i - c * HW;
    const float m = ls_mask(mmode, lab, mk, b, HW, p);
    const float yh = pred[b * pbs + i] * m;
    const float y = ls_target(tmode, tf, lab, b, c, C, HW, p, klass) * m;
    float g = upstream * m * (A * y + 2.f * Bq * yh - Ac * (1.f - y) - 2.f * Bc * (1.f - yh));
    if (accumulate) g += dpred[b * dbs + i];
    dpred[b * dbs + i] = g;
  }
}

static dim3 loss_grid(int B, long n) {
  long bx = (n + 1023) / 1024;
  if (bx > 64) bx = 64;
  if (bx < 1) bx = 1;
  return dim3((unsigned)bx, B);
}

// Forward: loss_out[0] = mean over batch; coef [B][4] kept for backward; sums: [B][5] doubles (scratch).
extern "C" int cn_tanimoto_fwd_f32(const float* pred, long pbs, const float* target_f, const long long* labels,
                                   const void* mask, int target_mode, int mask_mode, int klass, int B, int C, long HW,
                                   int loss_kind, float smooth, int depth, double* sums, float* coef, float* loss_out,
                                   float weight, float* total_out /*nullable: += weight*loss*/, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || B > 65535) return CN_ERR_ARG;
  if (hipMemsetAsync(sums, 0, sizeof(double) * 5 * B, stream) != hipSuccess) return CN_ERR_LAUNCH;
  CN_LAUNCH(cn_tanimoto_sums_kernel, loss_grid(B, (long)C * HW), dim3(256), 0, stream, pred, pbs, target_f,
                     labels, mask, target_mode, mask_mode, klass, C, HW, sums);
  CN_LAUNCH(cn_tanimoto_finalize_kernel, dim3(1), dim3(256), 0, stream, sums, B, (double)C * HW, loss_kind,
                     smooth, depth, loss_out, coef, weight, total_out);
  return cn_check_launch();
}

extern "C" int cn_tanimoto_bwd_f32(const float* pred, long pbs, const float* target_f, const long long* labels,
                                   const void* mask, int target_mode, int mask_mode, int klass, int B, int C, long HW,
                                   const float* coef, float upstream, float* dpred, long dbs, int accumulate,
                                   void* stream_) {
  if (B <= 0 || B > 65535) return CN_ERR_ARG;
  CN_LAUNCH(cn_tanimoto_bwd_kernel, loss_grid(B, (long)C * HW), dim3(256), 0, (hipStream_t)stream_, pred, pbs,
                     target_f, labels, mask, target_mode, mask_mode, klass, C, HW, coef, upstream, dpred, dbs,
                     accumulate);
  return cn_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------
// The three main losses of calc_loss (distance, edge, crop: lightning.py:307-339) in ONE launch per pass (round 6). The
// native step ran them head by head -- memset, sums, finalize (x3), a fill of the running total, backward (x3): thirteen
// dependent launches of 5-9 us on tensors of [B,1,100,100], in the one stretch of the step where nothing else runs (the
// turn from forward to backward). Same kernels' arithmetic, head = blockIdx.z; the total is written (w0*l0 + w1*l1 + w2*l2,
// added in head order as the per-head calls did), not accumulated: no fill in front.
// ---------------------------------------------------------------------------------------------------------------
#define CN_LOSS_MAX_HEADS 4
struct CnLossHead {       // 80 bytes; the engine packs a host array of these
  const float* pred;      // [B][C][HW], batch stride pbs
  long pbs;
  const float* tf;        // float target (TGT_FLOAT) or NULL
  const long long* lab;   // labels or NULL
  const void* mk;         // explicit mask or NULL
  float* dpred;           // backward: gradient of pred (batch stride dbs), NULL in forward
  long dbs;
  int tmode, mmode, klass, C;
  float weight;
  int accumulate;         // backward: dpred += instead of =
};
static_assert(sizeof(CnLossHead) == 80, "CnLossHead is an 80-byte record");
struct CnLossHeads { CnLossHead h[CN_LOSS_MAX_HEADS]; int n; };

__global__ __launch_bounds__(256) void cn_tanimoto_multi_sums_kernel(const CnLossHeads hs, long HW,
                                                                    double* __restrict__ sums, int B) {
  __shared__ double scratch[4];
  const CnLossHead& h = hs.h[blockIdx.z];
  const long b = blockIdx.y;
  const long n = (long)h.C * HW;
  double a[5] = {0, 0, 0, 0, 0};
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i / HW);
    const long p = i - c * HW;
    const float m = ls_mask(h.mmode, h.lab, h.mk, b, HW, p);
    const float yh = h.pred[b * h.pbs + i] * m;
    const float y = ls_target(h.tmode, h.tf, h.lab, b, c, h.C, HW, p, h.klass) * m;
    a[0] += y;
    a[1] += yh;
    a[2] += (double)y * yh;
    a[3] += (double)y * y;
    a[4] += (double)yh * yh;
  }
  double* out = sums + ((long)blockIdx.z * B + b) * 5;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const double r = cn_block_sum<double, 256>(a[k], scratch);
    if (threadIdx.x == 0) atomicAdd(out + k, r);
  }
}

__global__ void cn_tanimoto_multi_finalize_kernel(const CnLossHeads hs, const double* __restrict__ sums, int B, long HW,
                                                  int kind, float smooth, int depth, float* __restrict__ loss_out,
                                                  float* __restrict__ coef, float* __restrict__ total_out) {
  __shared__ double scratch[4];
  float total = 0.f;
  for (int hd = 0; hd < hs.n; ++hd) {
    const double N = (double)hs.h[hd].C * HW;
    const double* sm = sums + (long)hd * B * 5;
    float* cf = coef + (long)hd * B * 4;
    double lsum = 0.0;
    for (int b = threadIdx.x; b < B; b += 256) {
      const double Sy = sm[b * 5 + 0], Syh = sm[b * 5 + 1], tpl = sm[b * 5 + 2];
      const double sq = sm[b * 5 + 3] + sm[b * 5 + 4];
      const double tplc = N - Sy - Syh + tpl;
      const double sqc = 2.0 * N - 2.0 * Sy - 2.0 * Syh + sq;
      double f1 = 0, a1 = 0, b1 = 0, f2 = 0, a2 = 0, b2 = 0, w = 1.0;
      if (kind == LOSS_COMPLEMENT || kind == LOSS_COMBINED) {
        double f, da, db;
        tnm_complement(tpl, sq, smooth, depth, f, da, db); f1 += f; a1 += da; b1 += db;
        tnm_complement(tplc, sqc, smooth, depth, f, da, db); f2 += f; a2 += da; b2 += db;
      }
      if (kind == LOSS_DIST || kind == LOSS_COMBINED) {
        double f, da, db;
        tnm_dist(tpl, sq, smooth, f, da, db); f1 += f; a1 += da; b1 += db;
        tnm_dist(tplc, sqc, smooth, f, da, db); f2 += f; a2 += da; b2 += db;
      }
      if (kind == LOSS_COMBINED) w = 0.5;
      lsum += 0.5 * (f1 + f2) * w;
      const double sc = 0.5 * w / B;
      cf[b * 4 + 0] = (float)(a1 * sc);
      cf[b * 4 + 1] = (float)(b1 * sc);
      cf[b * 4 + 2] = (float)(a2 * sc);
      cf[b * 4 + 3] = (float)(b2 * sc);
    }
    lsum = cn_block_sum<double, 256>(lsum, scratch);
    __syncthreads();
    if (threadIdx.x == 0) {
      loss_out[hd] = (float)(lsum / B);
      total += hs.h[hd].weight * (float)(lsum / B);
    }
  }
  if (threadIdx.x == 0 && total_out != nullptr) total_out[0] = total;
}

__global__ __launch_bounds__(256) void cn_tanimoto_multi_bwd_kernel(const CnLossHeads hs, long HW,
                                                                   const float* __restrict__ coef, int B) {
  const CnLossHead& h = hs.h[blockIdx.z];
  const long b = blockIdx.y;
  const long n = (long)h.C * HW;
  const float* cf = coef + ((long)blockIdx.z * B + b) * 4;
  const float A = cf[0], Bq = cf[1], Ac = cf[2], Bc = cf[3];
  const float upstream = h.weight;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i / HW);
    const long p = i - c * HW;
    const float m = ls_mask(h.mmode, h.lab, h.mk, b, HW, p);
    const float yh = h.pred[b * h.pbs + i] * m;
    const float y = ls_target(h.tmode, h.tf, h.lab, b, c, h.C, HW, p, h.klass) * m;
    float g = upstream * m * (A * y + 2.f * Bq * yh - Ac * (1.f - y) - 2.f * Bc * (1.f - yh));
    if (h.accumulate) g += h.dpred[b * h.dbs + i];
    h.dpred[b * h.dbs + i] = g;
  }
}

static int loss_heads(int n, const void* heads, CnLossHeads& hs, int& Cmax) {
  if (n < 1 || n > CN_LOSS_MAX_HEADS || heads == nullptr) return CN_ERR_ARG;
  hs = CnLossHeads{};
  hs.n = n;
  Cmax = 1;
  for (int i = 0; i < n; ++i) {
    hs.h[i] = ((const CnLossHead*)heads)[i];
    if (hs.h[i].pred == nullptr || hs.h[i].C < 1) return CN_ERR_ARG;
    if (hs.h[i].C > Cmax) Cmax = hs.h[i].C;
  }
  return CN_OK;
}

// n (<= 4) losses over tensors of the same batch size and H*W. heads: HOST array of n CnLossHead records. sums:
// n*B*5 doubles of scratch, coef: n*B*4 floats kept for backward, loss_out: n floats (per-head batch means),
// total_out (nullable): = sum_h weight_h * loss_h.
extern "C" int cn_tanimoto_multi_fwd_f32(int n, const void* heads, int B, long HW, int loss_kind, float smooth, int depth,
                                         double* sums, float* coef, float* loss_out, float* total_out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || B > 65535) return CN_ERR_ARG;
  CnLossHeads hs;
  int Cmax;
  const int rc = loss_heads(n, heads, hs, Cmax);
  if (rc != CN_OK) return rc;
  if (hipMemsetAsync(sums, 0, sizeof(double) * 5 * B * n, stream) != hipSuccess) return CN_ERR_LAUNCH;
  dim3 g = loss_grid(B, (long)Cmax * HW);
  g.z = n;
  CN_LAUNCH(cn_tanimoto_multi_sums_kernel, g, dim3(256), 0, stream, hs, HW, sums, B);
  CN_LAUNCH(cn_tanimoto_multi_finalize_kernel, dim3(1), dim3(256), 0, stream, hs, sums, B, HW, loss_kind, smooth, depth,
            loss_out, coef, total_out);
  return cn_check_launch();
}

// dpred_h (+)= weight_h * dL_h/dpred_h for the n heads (records as in the forward call, with dpred / dbs / accumulate set).
extern "C" int cn_tanimoto_multi_bwd_f32(int n, const void* heads, int B, long HW, const float* coef, void* stream_) {
  if (B <= 0 || B > 65535) return CN_ERR_ARG;
  CnLossHeads hs;
  int Cmax;
  const int rc = loss_heads(n, heads, hs, Cmax);
  if (rc != CN_OK) return rc;
  for (int i = 0; i < n; ++i)
    if (hs.h[i].dpred == nullptr) return CN_ERR_ARG;
  dim3 g = loss_grid(B, (long)Cmax * HW);
  g.z = n;
  CN_LAUNCH(cn_tanimoto_multi_bwd_kernel, g, dim3(256), 0, (hipStream_t)stream_, hs, HW, coef, B);
  return cn_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------
// Validation metrics of _shared_eval_step (/root/reference/src/cultionet/models/lightning.py:374-481): one pass over
// the three probability maps builds the masked regression sums and the two 2x2 confusion matrices; a one-thread
// finalize turns them into what the reference gets from torchmetrics:
//   MeanAbsoluteError / MeanSquaredError of distance vs bdist over valid pixels,
//   FBetaScore(task="multiclass", num_classes=2, beta=2) -- micro-averaged, i.e. accuracy -- and
//   MatthewsCorrCoef(task="multiclass", num_classes=2) of thresholded edge / crop labels,
//   score = loss + (1-edge_f) + (1-crop_f) + mae + (1-max(edge_mcc,0)) + (1-max(crop_mcc,0)).
// Valid pixels: labels != -1 (get_true_labels, lightning.py:161-207; all pixels when no -1 is present).
// counts[11] = {n, sum|d|, sum d^2, edge tp, fp, fn, tn, crop tp, fp, fn, tn}
__global__ __launch_bounds__(256) void cn_eval_counts_kernel(const float* __restrict__ dist, const float* __restrict__ edge,
                                                            const float* __restrict__ crop,
                                                            const float* __restrict__ bdist,
                                                            const long long* __restrict__ lab, int klass, float thresh,
                                                            long n, double* __restrict__ counts) {
  __shared__ double scratch[4];
  double a[11];
#pragma unroll
  for (int k = 0; k < 11; ++k) a[k] = 0.0;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long long y = lab[i];
    if (y == -1) continue;
    const float d = dist[i] - bdist[i];
    a[0] += 1.0;
    a[1] += fabsf(d);
    a[2] += (double)d * d;
    const bool te = y == klass, tc = y > 0 && y < klass;
    const bool pe = edge[i] > thresh, pc = crop[i] > thresh;
    a[3 + (pe ? (te ? 0 : 1) : (te ? 2 : 3))] += 1.0;
    a[7 + (pc ? (tc ? 0 : 1) : (tc ? 2 : 3))] += 1.0;
  }
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    const double s = cn_block_sum<double, 256>(a[k], scratch);
    if (threadIdx.x == 0 && s != 0.0) atomicAdd(counts + k, s);
  }
}

// MatthewsCorrCoef(task="multiclass", num_classes=2): torchmetrics >= 1.0 `_matthews_corrcoef_reduce` on the 2x2
// confusion matrix (setup.cfg pins torchmetrics>=1.3; restated, the package is in neither image): all predictions
// right -> 1, all wrong -> -1, and when a marginal is empty (denominator 0) the eps-regularised ratio
// sqrt(eps) * ((tp + tn) - (fp + fn)) / sqrt(prod(marginal + eps)) with eps = float32 machine epsilon -- batches with
// no predicted or no true edge pixels are common early in training and on background chips, and val_score (what the
// reference checkpoints on, callbacks.py:246) contains both MCC terms.
__device__ __forceinline__ double ev_mcc(double tp, double fp, double fn, double tn) {
  if (tp + tn != 0.0 && fp + fn == 0.0) return 1.0;
  if (tp + tn == 0.0 && fp + fn != 0.0) return -1.0;
  const double den = (tp + fp) * (tp + fn) * (tn + fp) * (tn + fn);
  if (den > 0.0) return (tp * tn - fp * fn) / sqrt(den);
  const double eps = 1.1920928955078125e-07;
  const double num = sqrt(eps) * ((tp + tn) - (fp + fn));
  const double d = (tp + fp + eps) * (tp + fn + eps) * (tn + fp + eps) * (tn + fn + eps);
  return num / sqrt(d);
}

// out[7] = {dist_mae, dist_mse, edge_f, crop_f, edge_mcc, crop_mcc, score}
__global__ void cn_eval_finalize_kernel(const double* __restrict__ c, const float* __restrict__ loss,
                                        float* __restrict__ out) {
  const double n = c[0] > 0.0 ? c[0] : 1.0;
  const double mae = c[1] / n, mse = c[2] / n;
  const double ef = (c[3] + c[6]) / n, cf = (c[7] + c[10]) / n;
  const double em = ev_mcc(c[3], c[4], c[5], c[6]), cm = ev_mcc(c[7], c[8], c[9], c[10]);
  out[0] = (float)mae; out[1] = (float)mse; out[2] = (float)ef; out[3] = (float)cf;
  out[4] = (float)em; out[5] = (float)cm;
  out[6] = (float)((double)loss[0] + (1.0 - ef) + (1.0 - cf) + mae + (1.0 - (em > 0.0 ? em : 0.0)) +
                   (1.0 - (cm > 0.0 ? cm : 0.0)));
}

// dist / edge / crop: dense [B][1][H][W] fp32 probabilities; bdist [B][H][W]; labels int64 [B][H][W]; loss: 1 float
// (device); counts: 11 doubles of scratch; out: 7 floats (device).
extern "C" int cn_eval_metrics_f32(const float* dist, const float* edge, const float* crop, const float* bdist,
                                   const long long* labels, int klass, float thresh, long n, const float* loss,
                                   double* counts, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (hipMemsetAsync(counts, 0, 11 * sizeof(double), stream) != hipSuccess) return CN_ERR_LAUNCH;
  if (n > 0) {
    long nb = (n + 255) / 256;
    if (nb > 1024) nb = 1024;
    CN_LAUNCH(cn_eval_counts_kernel, dim3((unsigned)nb), dim3(256), 0, stream, dist, edge, crop, bdist, labels,
                       klass, thresh, n, counts);
  }
  CN_LAUNCH(cn_eval_finalize_kernel, dim3(1), dim3(1), 0, stream, counts, loss, out);
  return cn_check_launch();
}
