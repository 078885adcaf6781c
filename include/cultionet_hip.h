/* cultionet_hip.h -- C ABI of libcultionet_hip.so (hand-written HIP kernels for gfx950 / MI355X).
 *
 * The reference (jgrss/cultionet @ 2024_10_08) has no native code; every op on its TowerUNet hot
 * path runs in ATen/oneDNN/cuDNN or libnatten underneath torch.nn modules. This library is the
 * drop-in for those ops: each entry point names the reference call site it replaces
 * (paths relative to /root/reference/src/cultionet).
 *
 * Conventions
 *   - plain C, `extern "C"`, no torch types; all tensors are dense fp32 NCHW device pointers;
 *     `*bs` arguments are BATCH strides in elements (channel stride is always H*W), so channel
 *     slices of a concat buffer can be read/written in place;
 *   - the CALLER owns every buffer (kernels never allocate); `stream` is a hipStream_t;
 *   - returns 0 on success, <0 on error (CN_ERR_*); no exceptions cross the ABI. Compute entry points keep
 *     no state between calls and may be called from any thread on any stream. The only process state is opt-in
 *     and keyed or locked: the split-K scratch is registered PER STREAM (cn_conv_set_workspace), the autotune
 *     cache is mutex-guarded (cn_conv_set_autotune), and the cn_profile_* diagnostics are single-client
 *     (documented below; never used by the product path);
 *   - precision: entry points ending in _f32 take fp32 tensors (NCHW); entry points ending in _bf16 take
 *     bfloat16 activations in NHWC with fp32 parameters/statistics (see the bf16 section);
 *   - `accumulate != 0` means `out += result` instead of `out = result`;
 *   - entry points documented "zero first" combine partial results with f32 atomics.
 */
#ifndef CULTIONET_HIP_H
#define CULTIONET_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define CN_OK 0
#define CN_ERR_ARG (-1)
#define CN_ERR_LAUNCH (-2)
#define CN_ERR_LDS (-3)

int cn_version(void);

/* ---- packed weights for the implicit-GEMM kernels -------------------------------------------
 * wp[t][k][n] = w[k*sk + n*sn + t*st] zero-padded to [T][cn_conv_kpad(K)][cn_conv_npad(N)].
 *   Conv2d weight [Cout][Cin][KH][KW]:   forward  K=Cin N=Cout sk=KH*KW      sn=Cin*KH*KW st=1
 *                                        bwd-data K=Cout N=Cin sk=Cin*KH*KW  sn=KH*KW     st=1
 *   ConvTranspose2d weight [Cin][Cout][KH][KW]: forward K=Cin N=Cout sk=Cout*KH*KW sn=KH*KW st=1
 *                                        bwd-data K=Cout N=Cin sk=KH*KW      sn=Cout*KH*KW st=1
 *   Linear weight [out][in] == Conv2d 1x1. */
int cn_conv_kpad(int k_in);
int cn_conv_npad(int n_out);
int cn_pack_weights_f32(const float* w, float* wp, int T, int K, int N, long sk, long sn, long st, void* stream);
/* Same for n tensors in one launch. descs: DEVICE array of 72-byte records
 * {const float* w; float* wp; int T, K, N, Kpad, Npad; int pad; long sk, sn, st;}. */
int cn_pack_weights_batched_f32(const void* descs, int n, void* stream);

/* ---- torch.nn.Conv2d (nn/modules/convolution.py:71-120,250-395; unet_parts.py:196-224) ------
 * also nn.Linear of NeighborhoodAttention2D.qkv / .proj (convolution.py:341-350) as 1x1 convs,
 * and the two nn.Conv3d of models/nunet.py:18-57 (kernel (k,1,1)) after a [B,C*T,H,W] view. */
int cn_conv2d_fwd_f32(const float* x, long xbs, const float* wp, const float* bias /*nullable*/, float* y, long ybs,
                      int B, int Cin, int Hin, int Win, int Cout, int KH, int KW, int stride, int pad, int dil,
                      int accumulate, void* stream);
int cn_conv2d_bwd_data_f32(const float* dy, long dybs, const float* wp_t, float* dx, long dxbs, int B, int Cin,
                           int Hin, int Win, int Cout, int KH, int KW, int stride, int pad, int dil, int accumulate,
                           void* stream);
/* dw [Cout][Cin][KH][KW] += ...  (zero first). ws (nullable): ws_floats floats of 16-byte aligned scratch used
 * for aligned / even-width copies when H*W is not a multiple of 4 or W is odd (enables the 16-byte DMA path);
 * B*(Cout*(Hout*(Wout+1)+3) + Cin*(Hin*Win+3)) floats always suffice. */
int cn_conv2d_bwd_weight_f32(const float* x, long xbs, const float* dy, long dybs, float* dw, int B, int Cin, int Hin,
                             int Win, int Cout, int KH, int KW, int stride, int pad, int dil, float* ws,
                             long ws_floats, void* stream);

/* ---- nn.Conv3d(kernel (k,1,1), no bias) of PreTimeReduction (models/nunet.py:18-57) ----------
 * run as a 1x1 conv over the [B, C*T, H, W] view with a banded weight matrix.
 * w [Cout][Cin][k]; transposed=0 packs for forward (K=Cin*Tin, N=Cout*Tout), !=0 for bwd-data.
 * fold: dw[co][ci][dt] += sum_t' dWexp[(co,t')][(ci,t'+dt)], dWexp dense [Cout*Tout][Cin*Tin]. */
int cn_pack_timeconv_f32(const float* w, float* wp, int Cout, int Cin, int Tin, int k, int transposed, void* stream);
int cn_fold_timeconv_grad_f32(const float* dwexp, float* dw, int Cout, int Cin, int Tin, int k, void* stream);

/* ---- torch.nn.ConvTranspose2d(k=3, stride s, padding 1) (convolution.py:45-68) ---------------
 * out_pad = nn.ConvTranspose2d's output_padding (0 <= out_pad < stride): y / dy are [B,Cout,Hout,Wout] with
 * Hout = (Hin-1)*s - 2*pad + KH + out_pad. The reference never sets it; the engine does, to put the (2n-1)^2 result on
 * the 2n x 2n grid of the check_upsample resize that follows (nn/functional.py:72-81): rows / columns [0, 2n-1) ARE the
 * reference's tensor, the planes are 16-byte aligned (99 x 99 is not), and the resize reads them through the stored-grid
 * arguments of cn_bilinear_*_f32. */
int cn_conv_transpose2d_fwd_f32(const float* x, long xbs, const float* wp, const float* bias, float* y, long ybs,
                                int B, int Cin, int Hin, int Win, int Cout, int KH, int KW, int stride, int pad,
                                int out_pad, int accumulate, void* stream);
int cn_conv_transpose2d_bwd_data_f32(const float* dy, long dybs, const float* wp_t, float* dx, long dxbs, int B,
                                     int Cin, int Hin, int Win, int Cout, int KH, int KW, int stride, int pad,
                                     int out_pad, int accumulate, void* stream);
/* dw [Cin][Cout][KH][KW] += ...  (zero first); ws as for cn_conv2d_bwd_weight_f32 */
int cn_conv_transpose2d_bwd_weight_f32(const float* x, long xbs, const float* dy, long dybs, float* dw, int B, int Cin,
                                       int Hin, int Win, int Cout, int KH, int KW, int stride, int pad, int out_pad,
                                       float* ws, long ws_floats, void* stream);

/* Optional scratch for the K-split launches of the cn_conv* entry points (small spatial sizes split the input
 * channels over blocks): with a workspace each split stores its partial output into a private slice and a reduce
 * kernel sums the slices (+ bias) into y; without one (the default) the splits use float atomics on a zero-filled y.
 * The scratch is registered FOR ONE STREAM: every cn_conv* call looks up the buffer of the stream it is given, so two
 * streams (or two models on two streams) never share scratch; launches on one stream are stream-ordered. The caller
 * keeps ws alive until it unregisters it (ws = NULL). ws 16-byte aligned; 64 MB covers every TowerUNet layer at
 * 8 chips of 100x100. */
int cn_conv_set_workspace(void* stream, float* ws, long ws_floats);

/* Opt-in autotuning of the cn_conv* launches (off by default): the first overwriting launch of each distinct shape
 * times the (pixel tile x K split) candidates with HIP events on the launch stream (that call synchronises) and the
 * winner is cached for the process; later launches of the shape, accumulating ones included, reuse it. */
int cn_conv_set_autotune(int on);

/* ---- grouped launches: G (<= 4) convolutions of identical shape in ONE launch -- the dilation branches of
 * ResidualAConv (nn/modules/convolution.py:376-395): same tensor shapes, per-branch padding / dilation.
 * xs/wps/biases/ys (dys/wps_t/dxs), pads, dils: HOST arrays of G entries (device pointers / ints). Inputs may
 * alias; outputs are all distinct, or all the same buffer (the G results are summed into it). */
int cn_conv2d_fwd_grouped_f32(int G, const float* const* xs, long xbs, const float* const* wps,
                              const float* const* biases /*nullable*/, float* const* ys, long ybs, int B, int Cin,
                              int Hin, int Win, int Cout, int KH, int KW, int stride, const int* pads,
                              const int* dils, int accumulate, void* stream);
int cn_conv2d_bwd_data_grouped_f32(int G, const float* const* dys, long dybs, const float* const* wps_t,
                                   float* const* dxs, long dxbs, int B, int Cin, int Hin, int Win, int Cout,
                                   const int* khs, const int* kws /* per group: a 1x1 skip conv may join 3x3 branches */,
                                   int stride, const int* pads, const int* dils, int accumulate, void* stream);

/* Grouped weight gradient (same padding / dilation for every group; xs may repeat one input). ACCUMULATES. */
int cn_conv2d_bwd_weight_grouped_f32(int G, const float* const* xs, long xbs, const float* const* dys, long dybs,
                                     float* const* dws, int B, int Cin, int Hin, int Win, int Cout, int KH, int KW,
                                     int stride, int pad, int dil, float* ws, long ws_floats, void* stream);

/* ---- thin 3x3 "same" convolutions (<= 9 output channels) of the TowerUNetFinal head streams
 * (nn/modules/unet_parts.py:196-224 StreamConv2d; :227-309 TowerUNetFinal): direct VALU kernels on the RAW
 * nn.Conv2d weights [cout_per_set][Cin][3][3] (no packing); HBM-bound, the input is read once for all sets.
 *   grouped == 0: every weight set sees all Cin channels of x; y channel = set*cout_per_set + c
 *   grouped != 0: set g sees channels [g*Cin, (g+1)*Cin) of an nsets*Cin-channel x
 * ws / biases / dws: HOST arrays of nsets device pointers. Supported (nsets, cout_per_set, grouped):
 * (3,3,0) (3,1,1) (1,3,0) (1,1,0); anything else returns CN_ERR_ARG (use cn_conv2d_*). padding == dil.
 * wpack (optional, Cin * 84 floats): scratch for the per-call packed weights of the wide (3,3,0) layers.
 * bwd_weight ACCUMULATES into dws. */
int cn_thin_conv3x3_fwd_f32(const float* x, long xbs, const float* const* ws, const float* const* biases /*nullable*/,
                            float* y, long ybs, int B, int Cin, int H, int W, int nsets, int cout_per_set,
                            int grouped, int dil, float* wpack /*nullable*/, void* stream);
int cn_thin_conv3x3_bwd_data_f32(const float* dy, long dybs, const float* const* ws, float* dx, long dxbs, int B,
                                 int Cin, int H, int W, int nsets, int cout_per_set, int grouped, int dil,
                                 int accumulate, float* wpack /*nullable*/, void* stream);
int cn_thin_conv3x3_bwd_weight_f32(const float* x, long xbs, const float* dy, long dybs, float* const* dws, int B,
                                   int Cin, int H, int W, int nsets, int cout_per_set, int grouped, int dil,
                                   void* stream);

/* bias gradients: out[c] (+)= sum_{b,l} x[b][c][l] */
int cn_channel_sum_f32(const float* x, long xbs, int B, int C, int L, float* out, int accumulate, void* stream);

/* ---- nn.BatchNorm2d / BatchNorm3d (+ SiLU, + residual add) (convolution.py:71-120, nunet.py:18-57)
 * tensors viewed as [B][C][L]; act: 0 none, 1 SiLU; y = act(bn(x)) (+ res).
 * training: batch stats saved to mean/rstd[C], running stats updated (momentum, unbiased var).
 * ws: cn_bn_workspace_doubles(C) doubles; coef: 2*C floats scratch. */
int cn_bn_workspace_doubles(int C);
int cn_bn_act_fwd_f32(const float* x, long xbs, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, const float* res /*nullable*/, long rbs, float* y, long ybs, float* mean,
                      float* rstd, double* ws, int B, int C, int L, int training, float momentum, float eps, int act,
                      void* stream);
int cn_bn_act_bwd_f32(const float* x, long xbs, const float* dy, long dybs, const float* mean, const float* rstd,
                      const float* gamma, const float* beta, float* dx /*nullable*/, long dxbs, float* dgamma,
                      float* dbeta, float* coef, double* ws, int B, int C, int L, int training, int act,
                      int accumulate_dx, int accumulate_params, void* stream);

/* Grouped BatchNorm(+SiLU): G (<= 4) BatchNorm layers over G same-shaped tensors in one launch pair (the dilation
 * branches of ResidualAConv, convolution.py:376-395). All pointer arguments ending in `s` are HOST arrays of G device
 * pointers. sum_outputs == 0: ys[g] = act(bn_g(xs[g])), res must be NULL. sum_outputs != 0: the ResUNet-a sum
 * ys[0] = res + sum_g act(bn_g(xs[g])) fused into the normalisation pass. ws: G * cn_bn_workspace_doubles(C) doubles.
 * Backward: dys[g] = gradient of output g (the same pointer G times after a summed forward; the residual's gradient
 * is that tensor itself); dxs[g] nullable; accumulate_dx: HOST array of G flags. */
int cn_bn_act_group_fwd_f32(int G, const float* const* xs, long xbs, const float* const* gammas,
                            const float* const* betas, float* const* running_means, float* const* running_vars,
                            const float* res /*nullable*/, long rbs, float* const* ys, long ybs, float* const* means,
                            float* const* rstds, double* ws, int B, int C, int L, int training, float momentum,
                            float eps, int act, int sum_outputs, void* stream);
int cn_bn_act_group_bwd_f32(int G, const float* const* xs, long xbs, const float* const* dys, long dybs,
                            const float* const* means, const float* const* rstds, const float* const* gammas,
                            const float* const* betas, float* const* dxs, long dxbs, const int* accumulate_dx,
                            float* const* dgammas, float* const* dbetas, double* ws, int B, int C, int L,
                            int training, int act, int accumulate_params, void* stream);

/* ---- nn.LayerNorm over channels, tensors kept NCHW (nunet.py:93-97, convolution.py:338-353) --
 * mu/rstd: [B][L] saved statistics. dw/db are ACCUMULATED: zero first. ws: cn_layernorm_c_workspace_floats(B,C,L)
 * floats of scratch for the per-block partial parameter gradients (0 => not needed). */
int cn_layernorm_c_fwd_f32(const float* x, long xbs, const float* w, const float* b, const float* res /*nullable*/,
                           long rbs, float* y, long ybs, float* mu, float* rstd, int B, int C, int L, float eps,
                           void* stream);
int cn_layernorm_c_bwd_f32(const float* x, long xbs, const float* dy, long dybs, const float* w, const float* mu,
                           const float* rstd, float* dx, long dxbs, float* dw, float* db, int B, int C, int L,
                           int accumulate_dx, float* ws, long ws_floats, void* stream);
int cn_layernorm_c_workspace_floats(int B, int C, int L);

/* ---- SpatialChannelAttention (attention_weights="spatial_channel"; nn/modules/attention.py:12-126, applied as
 * out *= 1 + gamma*0.5*(ca + sa) at nn/modules/convolution.py:388-393). x viewed [B][C][L].
 * pool_fwd: avg/mx [B][C] over L (idx: position of the first maximum), pooled [B][2][L] = mean / max over C
 *   (cidx: channel of the first maximum). The 3x3 conv 2->1 on `pooled` is cn_thin_conv3x3_* (1 set, 1 output).
 * mlp: w1* [C/2][C], w2* [C][C/2] (fc1 = avg branch "a", fc2 = max branch "m"); hpre_* [B][C/2] saved pre-activations;
 *   ca [B][C]. mlp_bwd ACCUMULATES the four weight gradients and overwrites davg/dmx.
 * apply: y = out*(1 + gamma*0.5*(ca[b,c] + sigmoid(sconv[b,l]))); apply_bwd overwrites dca [B][C], dsconv [B][L],
 *   ACCUMULATES dgamma[1]; dout nullable; scratch: B*C floats.
 * pool_bwd: dx (+)= the four pooling adjoints. */
int cn_sca_pool_fwd_f32(const float* x, long xbs, int B, int C, int L, float* avg, float* mx, int* idx, float* pooled,
                        int* cidx, void* stream);
int cn_sca_pool_bwd_f32(const float* davg, const float* dmx, const int* idx, const float* dpooled, const int* cidx,
                        float* dx, long dxbs, int B, int C, int L, int accumulate, void* stream);
int cn_sca_mlp_fwd_f32(const float* avg, const float* mx, const float* w1a, const float* w2a, const float* w1m,
                       const float* w2m, float* hpre_a, float* hpre_m, float* ca, int B, int C, int Ch, void* stream);
int cn_sca_mlp_bwd_f32(const float* avg, const float* mx, const float* w1a, const float* w2a, const float* w1m,
                       const float* w2m, const float* hpre_a, const float* hpre_m, const float* ca, const float* dca,
                       float* dw1a, float* dw2a, float* dw1m, float* dw2m, float* davg, float* dmx, int B, int C,
                       int Ch, void* stream);
int cn_sca_apply_fwd_f32(const float* out, long obs, const float* ca, const float* sconv, const float* gamma, float* y,
                         long ybs, int B, int C, int L, void* stream);
int cn_sca_apply_bwd_f32(const float* dy, long dybs, const float* out, long obs, const float* ca, const float* sconv,
                         const float* gamma, float* dout /*nullable*/, long dobs, int accumulate_dout, float* dca,
                         float* dsconv, float* dgamma, float* scratch, int B, int C, int L, void* stream);

/* ---- natten.NeighborhoodAttention2D core (convolution.py:341-350; natten 0.17.1 na2d_qk ->
 * softmax -> na2d_av, kernel 3, dilation d, no rpb). qkv [B][3C][H][W] with channel
 * (which*C + head*D + d); attn [B][heads][9][H][W] saved probabilities; dattn same-size scratch.
 * attn_drop in [0,1): dropout on the probabilities (mask recomputed from `seed` / `step`, as cn_dropout_f32). */
int cn_na2d_fwd_f32(const float* qkv, long qbs, float* out, long obs, float* attn, int B, int C, int heads, int H,
                    int W, int kernel_size, int dilation, float attn_drop, unsigned long long seed,
                    const unsigned long long* step, void* stream);
int cn_na2d_bwd_f32(const float* qkv, long qbs, const float* dout, long dobs, const float* attn, float* dattn,
                    float* dqkv, long dqbs, int B, int C, int heads, int H, int W, int kernel_size, int dilation,
                    float attn_drop, unsigned long long seed, const unsigned long long* step, void* stream);

/* ---- ConvTranspose2d with stride >= kernel size + check_upsample in one pointwise pass (TowerUNetFinal.up_conv of
 * final_c: k 3, stride 4, padding 1; unet_parts.py:227-309, convolution.py:45-68). With s >= k no two taps of an input
 * pixel meet, so the contraction is a dense 1x1 GEMM on the SMALL grid -- cn_conv_transpose2d_{fwd,bwd_data,bwd_weight}_f32
 * with KH = KW = 1 on the weight tensor viewed as [Cin][Cout*K*K] -- producing P [B][C*K*K][Hc][Wc], and
 *   cn_convt_taps_fwd_f32: z [B][C][Ho][Wo] = resize_to(Ho, Wo)(bias + scatter(P))   (Ho, Wo) == natural size: no resize
 *   cn_convt_taps_bwd_f32: dP = gather(resize^T(dz))                                  the adjoint, in P's layout
 * The (s*n - c)^2 intermediate of the reference never exists; the bias gradient is the per-channel sum of dz. */
int cn_convt_taps_fwd_f32(const float* P, long pbs, const float* bias /*nullable*/, float* z, long zbs, int B, int C,
                          int Hc, int Wc, int K, int stride, int pad, int Ho, int Wo, void* stream);
int cn_convt_taps_bwd_f32(const float* dz, long dzbs, float* dP, long dpbs, int B, int C, int Hc, int Wc, int K,
                          int stride, int pad, int Ho, int Wo, void* stream);

/* ---- F.interpolate(mode="bilinear", align_corners=True) (nn/functional.py:72-81) ------------
 * Hp x Wp (0, 0 = dense): the SOURCE of the forward / the input gradient of the adjoint is stored on a grid
 * Hp x Wp >= Hi x Wi (row pitch Wp, plane Hp*Wp) whose top-left Hi x Wi is the image -- the output_padding grid of
 * cn_conv_transpose2d_*_f32. The adjoint writes the padding as zeros (leaves it alone when accumulating). */
int cn_bilinear_fwd_f32(const float* x, long xbs, float* y, long ybs, int B, int C, int Hi, int Wi, int Ho, int Wo,
                        int Hp, int Wp, void* stream);
int cn_bilinear_bwd_f32(const float* dy, long dybs, float* dx, long dxbs, int B, int C, int Hi, int Wi, int Ho,
                        int Wo, int Hp, int Wp, int accumulate, void* stream);

/* ---- torch.cat / residual adds / fills on channel slices ------------------------------------ */
int cn_copy_f32(const float* src, long sbs, float* dst, long dbs, int B, long n, int accumulate, void* stream);
int cn_add_f32(const float* a, long abs_, const float* c, long cbs, float* dst, long dbs, int B, long n, void* stream);
int cn_fill_f32(float* p, long n, float v, void* stream);

/* ---- nn.Dropout2d / nn.Dropout (convolution.py:495,511; natten attn_drop, proj_drop) -------------
 * y (+)= x * keep / (1-p); keep = splitmix64(seed + index) >= p*2^64 is recomputed, never stored: call
 * again with x := dy (same seed) for the backward pass. channelwise != 0: one draw per (b, c) plane.
 * `step` (nullable): device word added to the seed as seed + *step * 0xD1B54A32D192ED03 -- the per-step part of the
 * seed lives in HBM so that a recorded launch plan (constant arguments) draws fresh masks every step; the reference
 * draws from torch's global generator, which advances per step the same way (nn.Dropout2d, convolution.py:495). */
int cn_dropout_f32(const float* x, long xbs, float* y, long ybs, int B, int C, int L, float p,
                   unsigned long long seed, const unsigned long long* step, int channelwise, int accumulate,
                   void* stream);
/* *word = set ? value : *word + value (one thread): bumps the dropout step word on the launch stream. */
int cn_rng_advance_u64(unsigned long long* word, unsigned long long value, int set, void* stream);

/* ---- F.adaptive_max_pool2d (pool_by_max=True, convolution.py:499-503) -------------------------
 * idx: int32 [B][C][Ho][Wo] flat argmax inside the input plane (kept for backward). */
int cn_adaptive_maxpool_fwd_f32(const float* x, long xbs, float* y, long ybs, int* idx, int B, int C, int Hi, int Wi,
                                int Ho, int Wo, void* stream);
int cn_adaptive_maxpool_bwd_f32(const float* dy, long dybs, const int* idx, float* dx, long dxbs, int B, int C, int Hi,
                                int Wi, int Ho, int Wo, int accumulate, void* stream);

/* ---- TowerUNetFinalCombine + SigmoidCrisp (nn/modules/unet_parts.py:43-193), fused ----------
 * h*: [B][3][HW] fuse_conv outputs of towers a,b,c (channel = dist, edge, crop).
 * params: HOST array of 16 DEVICE pointers to scalars: [3k+t] gamma of task k tower t, [9+k] 1x1
 * weight, [12+k] bias, [15] crisp gamma. dparams: 16 device pointers, atomically accumulated into. */
int cn_final_combine_fwd_f32(const float* ha, const float* hb, const float* hc, const float* const* params,
                             float* dist, float* edge, float* crop, int B, int HW, float smooth, void* stream);
int cn_final_combine_bwd_f32(const float* ha, const float* hb, const float* hc, const float* const* params,
                             const float* dist, const float* edge, const float* crop, const float* ddist,
                             const float* dedge, const float* dcrop, float* dha, float* dhb, float* dhc,
                             float* const* dparams, int B, int HW, float smooth, void* stream);

/* ---- Tanimoto losses + get_true_labels (losses/losses.py:9-340, models/lightning.py:161-354) -
 * target_mode: 0 float target [B][C][HW]; 1 labels==klass; 2 0<labels<klass; 3 one-hot(labels)
 * mask_mode:   0 none; 1 labels != -1; 2 int64 mask [B][HW]; 3 float mask [B][HW]
 * loss_kind:   0 TanimotoComplementLoss; 1 TanimotoDistLoss; 2 TanimotoCombined
 * sums: 5*B doubles scratch; coef: 4*B floats (kept for backward); loss_out: 1 float (batch mean). */
int cn_tanimoto_fwd_f32(const float* pred, long pbs, const float* target_f, const long long* labels, const void* mask,
                        int target_mode, int mask_mode, int klass, int B, int C, long HW, int loss_kind, float smooth,
                        int depth, double* sums, float* coef, float* loss_out, float weight,
                        float* total_out /*nullable: total_out[0] += weight*loss*/, void* stream);
int cn_tanimoto_bwd_f32(const float* pred, long pbs, const float* target_f, const long long* labels, const void* mask,
                        int target_mode, int mask_mode, int klass, int B, int C, long HW, const float* coef,
                        float upstream, float* dpred, long dbs, int accumulate, void* stream);
/* The three main losses of calc_loss (lightning.py:307-339) in ONE launch per pass: n (<= 4) heads over tensors of the same
 * batch size and H*W. heads: HOST array of n 80-byte records {const float* pred; long pbs; const float* target_f;
 * const long long* labels; const void* mask; float* dpred; long dbs; int target_mode, mask_mode, klass, C; float weight;
 * int accumulate}. Forward: sums n*B*5 doubles (scratch), coef n*B*4 floats (kept for backward), loss_out n floats,
 * total_out (nullable) = sum_h weight_h * loss_h (WRITTEN, not accumulated). Backward: dpred_h (+)= weight_h * dL_h/dpred_h. */
int cn_tanimoto_multi_fwd_f32(int n, const void* heads, int B, long HW, int loss_kind, float smooth, int depth,
                              double* sums, float* coef, float* loss_out, float* total_out, void* stream);
int cn_tanimoto_multi_bwd_f32(int n, const void* heads, int B, long HW, const float* coef, void* stream);

/* ---- validation metrics of _shared_eval_step (models/lightning.py:374-481): masked MAE / MSE of the distance map,
 * micro F-beta (= accuracy, torchmetrics FBetaScore(task="multiclass", num_classes=2)) and MatthewsCorrCoef of the
 * thresholded edge / crop maps, and the checkpoint score. dist / edge / crop dense [B][1][H][W] fp32, n = B*H*W;
 * valid pixels: labels != -1; loss: 1 device float. counts: 11 doubles scratch.
 * out[7] = {dist_mae, dist_mse, edge_f, crop_f, edge_mcc, crop_mcc, score}. */
int cn_eval_metrics_f32(const float* dist, const float* edge, const float* crop, const float* bdist,
                        const long long* labels, int klass, float thresh, long n, const float* loss, double* counts,
                        float* out, void* stream);

/* ---- torch.optim.AdamW + clip_grad_norm_ (models/lightning.py:622-629, model.py:84,173) ------
 * one flat buffer each for params, grads, exp_avg, exp_avg_sq. sumsq nullable (no clipping). */
int cn_grad_sumsq_f32(const float* g, long n, double* out, void* stream);
int cn_adamw_step_f32(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                      float eps, float weight_decay, int step, float grad_scale, const double* sumsq, float max_norm,
                      void* stream);

/* ---- input / output edges ------------------------------------------------
 * prepare: EdgeDataset.get scale + clip (data/datasets.py:443-446) + NormValues z-score
 *   (utils/normalize.py:63-82): y = (clip(x*scale, lo, hi) - mean[c]) / std[c]; x dtype 0 f32, 1 i32, 2 i16, 3 u16.
 * predictions: LightningGTiffWriter (callbacks.py:176-227): drop padding, x scale, clip, cast to uint16. */
int cn_prepare_chips_f32(const void* x, int dtype, float* y, const float* mean, const float* stdv, int B, int C, long L,
                         float scale, float lo, float hi, void* stream);
int cn_predictions_to_u16(const float* dist, const float* edge, const float* crop, unsigned short* out, int B, int H,
                          int W, int pad_top, int pad_left, int h, int w, float scale, void* stream);
/* sliding-window predict (BASELINE configs[4]; data/create.py:176-212, data/store.py:69-100, callbacks.py:176-227):
 * window_chips: window n = crop [r0-pad, r0-pad+S) x [c0-pad, c0-pad+S) of the zero-extended scene [C*T][H][W]
 *   (stored dtype as prepare), scaled / clipped / z-scored into fp32 [nwin][C*T][S][S]; win_rc: DEVICE int [nwin][2].
 * stitch: out u16 [3][H][W] <- window interiors (padding dropped, x scale, clip, cast), clipped at the scene edge. */
int cn_window_chips_f32(const void* scene, int dtype, float* out, const int* win_rc, int nwin, int C, int T, int H, int W,
                        int S, int pad, const float* mean, const float* stdv, float scale, float lo, float hi,
                        void* stream);
int cn_stitch_predictions_u16(const float* dist, const float* edge, const float* crop, unsigned short* out,
                              const int* win_rc, int nwin, int S, int pad, int ws, int H, int W, float scale,
                              void* stream);


/* ================================================================================================================
 * bf16 mixed-precision path (BASELINE configs[2]; the reference's default precision="16-mixed", model.py:168-186).
 * Activations and activation gradients: bfloat16, NHWC ([B][H][W][C] rows; `ld*` = elements between consecutive
 * pixels, a multiple of 8, base pointers 16-byte aligned, so channel slices of a concat buffer are used in place).
 * Parameters, statistics and parameter gradients: fp32. MFMA: v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
 * void* tensor arguments are bf16 device pointers.
 * ================================================================================================================ */

/* packed weights, MFMA-fragment order [tap][ceil(K/16)][ceil(N/32)][64 lanes][8]: element j of lane l is
 * w[k = kstep*16 + 8*(l>>5) + j][n = ntile*32 + (l&31)] of tap t, zero beyond K / N. Strides as cn_pack_weights_f32. */
long cn_bconv_packed_elems(int T, int K, int N);
int cn_pack_weights_bf16(const float* w, void* wp, int T, int K, int N, long sk, long sn, long st, void* stream);
/* descs: DEVICE array of 72-byte records {const float* w; void* wp; int T, K, N, KS, NT, pad; long sk, sn, st;
 * const float* nscale (nullable per-cout factor, see cn_pack_weights_scaled_bf16);} */
int cn_pack_weights_batched_bf16(const void* descs, int n, void* stream);

/* Eval-mode ConvBlock2d (convolution.py:71-120, BatchNorm in inference mode) in ONE launch:
 * y = res + act(conv(x, W') + bias) with the running statistics folded in by the caller
 * (cn_bn_fold_f32 -> scale / shift; cn_pack_weights_scaled_bf16 -> W' = W * scale per cout; bias = shift).
 * act: 0 identity, 1 SiLU. res (nullable): bf16 NHWC [B,Hout,Wout,>=Cout] with pixel stride ldres (the ResUNet-a
 * running sum, convolution.py:376-395; may alias y). Cout % 8 == 0. */
int cn_conv2d_fwd_fused_bf16(const void* x, long ldx, const void* wp, const float* bias, const void* res, long ldres,
                             void* y, long ldy, int B, int Cin, int Hin, int Win, int Cout, int KH, int KW, int stride,
                             int pad, int dil, int act, void* stream);
int cn_pack_weights_scaled_bf16(const float* w, const float* nscale, void* wp, int T, int K, int N, long sk, long sn,
                                long st, void* stream);
int cn_bn_fold_f32(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                   const float* conv_bias, float eps, int C, float* scale, float* shift, void* stream);

/* nn.Conv2d forward (convolution.py:71-120). out_kind 0: y bf16 NHWC (ldy); 1: y f32 NCHW with batch stride y_bs
 * (the thin head convolutions hand over to the fp32 head kernels). stats (nullable): cn_conv2d_stats_rows_bf16(...)
 * rows of [2][Cout] floats, one per pixel tile, receive the sum and sum of squares of the fp32 results (plain stores,
 * no zero-fill needed): BatchNorm statistics without a pass over y (cn_bn_act_fwd_bf16 combines the rows). */
int cn_conv2d_stats_rows_bf16(int B, int Hin, int Win, int Cout, int KH, int KW, int stride, int pad, int dil);
int cn_conv2d_fwd_bf16(const void* x, long ldx, const void* wp, const float* bias /*nullable*/, void* y, long ldy,
                       long y_bs, int B, int Cin, int Hin, int Win, int Cout, int KH, int KW, int stride, int pad,
                       int dil, int accumulate, int out_kind, float* stats /*nullable*/, void* stream);
int cn_conv2d_fwd_grouped_bf16(int G, const void* const* xs, long ldx, const void* const* wps,
                               const float* const* biases /*nullable*/, void* const* ys, long ldy, int B, int Cin,
                               int Hin, int Win, int Cout, int KH, int KW, int stride, const int* pads,
                               const int* dils, int accumulate, float* const* stats /*nullable: G row sets*/,
                               void* stream);
/* ConvBlock2d in training mode (convolution.py:71-120: Conv2d(bias=False) -> BatchNorm2d): the grouped launch also
 * FINISHES the BatchNorm batch statistics of its G outputs when it can (a two-level last-block ticket per group and
 * cout block; <= 1008 pixel tiles per convolution). *finalized = 1: means / rstds [Cout] per group (and the running
 * statistics, updated with `momentum` as torch does, when given) were written by the launch -- go straight to
 * cn_bn_act_group_fwd_bf16(..., conv_sums = NULL, conv_rows = -1). *finalized = 0: only the per-tile rows in `stats` were
 * written -- pass them to cn_bn_act_group_fwd_bf16 as for cn_conv2d_fwd_grouped_bf16. bn_ws: the grouped-BatchNorm
 * workspace of this stream (cn_bn_group_workspace_floats_bf16(G, Cout) floats, zero-filled once after allocation). */
int cn_conv2d_fwd_grouped_bnstats_bf16(int G, const void* const* xs, long ldx, const void* const* wps, void* const* ys,
                                       long ldy, int B, int Cin, int Hin, int Win, int Cout, int KH, int KW, int stride,
                                       const int* pads, const int* dils, float* const* stats, float* const* means,
                                       float* const* rstds, float* const* running_means /*nullable*/,
                                       float* const* running_vars /*nullable*/, float momentum, float eps, float* bn_ws,
                                       long bn_ws_floats, int* finalized, void* stream);
int cn_conv2d_bwd_data_bf16(const void* dy, long lddy, const void* wp_t, void* dx, long lddx, int B, int Cin, int Hin,
                            int Win, int Cout, int KH, int KW, int stride, int pad, int dil, int accumulate,
                            void* stream);
int cn_conv2d_bwd_data_grouped_bf16(int G, const void* const* dys, long lddy, const void* const* wps_t,
                                    void* const* dxs, long lddx, int B, int Cin, int Hin, int Win, int Cout, int KH,
                                    int KW, int stride, const int* pads, const int* dils, int accumulate,
                                    void* stream);
/* nn.ConvTranspose2d (convolution.py:45-68) */
int cn_conv_transpose2d_fwd_bf16(const void* x, long ldx, const void* wp, const float* bias, void* y, long ldy, int B,
                                 int Cin, int Hin, int Win, int Cout, int KH, int KW, int stride, int pad,
                                 int accumulate, void* stream);
int cn_conv_transpose2d_bwd_data_bf16(const void* dy, long lddy, const void* wp_t, void* dx, long lddx, int B, int Cin,
                                      int Hin, int Win, int Cout, int KH, int KW, int stride, int pad, int accumulate,
                                      void* stream);
/* weight gradients: dw fp32 in the layer's own layout ([Cout][Cin][KH][KW] / [Cin][Cout][KH][KW]), ACCUMULATED.
 * ws: fp32 scratch for the per-split partial slices; cn_bwgrad_workspace_floats(...) floats always suffice (the
 * split shrinks to fit smaller buffers). */
long cn_bwgrad_workspace_floats(int B, int Cin, int Hin, int Win, int Cout, int KH, int KW, int stride, int pad,
                                int dil, int transposed);
int cn_conv2d_bwd_weight_bf16(const void* x, long ldx, const void* dy, long lddy, float* dw, int B, int Cin, int Hin,
                              int Win, int Cout, int KH, int KW, int stride, int pad, int dil, float* ws,
                              long ws_floats, void* stream);
int cn_conv_transpose2d_bwd_weight_bf16(const void* x, long ldx, const void* dy, long lddy, float* dw, int B, int Cin,
                                        int Hin, int Win, int Cout, int KH, int KW, int stride, int pad, float* ws,
                                        long ws_floats, void* stream);

/* nn.BatchNorm2d (+SiLU, +residual) on [P = B*H*W][C] rows; C % 8 == 0. ws: cn_bn_workspace_floats_bf16(C) floats.
 * conv_sums (nullable) / conv_rows: the `stats` rows of cn_conv2d_fwd_bf16. Backward ACCUMULATES dgamma / dbeta. */
long cn_bn_workspace_floats_bf16(int C);
int cn_bn_act_fwd_bf16(const void* x, long ldx, const float* gamma, const float* beta, float* running_mean,
                       float* running_var, const void* res /*nullable*/, long ldr, void* y, long ldy, float* mean,
                       float* rstd, float* ws, long P, int C, int training, float momentum, float eps, int act,
                       const float* conv_sums /*nullable*/, int conv_rows, void* stream);
int cn_bn_act_bwd_bf16(const void* x, long ldx, const void* dy, long lddy, const float* mean, const float* rstd,
                       const float* gamma, const float* beta, void* dx /*nullable*/, long lddx, float* dgamma,
                       float* dbeta, float* ws, long P, int C, int training, int act, int accumulate_dx, void* stream);

/* Grouped BatchNorm2d(+SiLU) on the mixed-precision path: the G (<= 4) BatchNorm layers of one ResidualAConv level
 * (convolution.py:376-395) over G same-shaped bf16 NHWC tensors, one launch per pass; pointer arguments ending in `s`
 * are HOST arrays of G device pointers. sum_outputs != 0: ys[0] = res + sum_g act(bn_g(xs[g])) in ONE pass (fp32
 * accumulation, one rounding); else ys[g] = act(bn_g(xs[g])), res must be NULL. conv_sums (nullable, training): per
 * group the `stats` rows of cn_conv2d_fwd[_grouped]_bf16 (conv_rows rows each). The batch statistics are finalized by
 * ONE launch: coalesced column sums of the partial rows whose per-block slices are combined by the last-arriving
 * block (device ticket; fixed summation order, bit-reproducible); statistics passes over x / dy (backward, or a forward
 * without conv_sums) finish themselves the same way (two-level last-block reduction): no finalize launch.
 * conv_rows = -1 (training): means / rstds and the running statistics were already written by the convolution launch
 * (cn_conv2d_fwd_grouped_bnstats_bf16 with *finalized = 1): apply only.
 * ws: cn_bn_group_workspace_floats_bf16(G, C) floats holding ticket counters: ZERO-FILLED before its first use (every
 * call leaves the counters zero), not shared with calls in flight on another stream.
 * Backward: dys[g] = gradient of output g (after a summed forward the same pointer G times: read once per pass);
 * dxs[g] nullable; accumulate_dx: HOST array of G flags; dgammas / dbetas ACCUMULATED. */
long cn_bn_group_workspace_floats_bf16(int G, int C);
int cn_bn_act_group_fwd_bf16(int G, const void* const* xs, long ldx, const float* const* gammas,
                             const float* const* betas, float* const* running_means, float* const* running_vars,
                             const void* res /*nullable*/, long ldr, void* const* ys, long ldy, float* const* means,
                             float* const* rstds, float* ws, long P, int C, int training, float momentum, float eps,
                             int act, int sum_outputs, const float* const* conv_sums /*nullable*/, int conv_rows,
                             void* stream);
int cn_bn_act_group_bwd_bf16(int G, const void* const* xs, long ldx, const void* const* dys, long lddy,
                             const float* const* means, const float* const* rstds, const float* const* gammas,
                             const float* const* betas, void* const* dxs, long lddx, const int* accumulate_dx,
                             float* const* dgammas, float* const* dbetas, float* ws, long P, int C, int training,
                             int act, void* stream);

/* bias gradients: out[c] (+)= sum_p x[p][c] in ONE launch (two-level last-block reduction); ws:
 * cn_bn_workspace_floats_bf16(C) floats, ZERO-FILLED before its first use (ticket counters; left zero by every call), not
 * the buffer handed to cn_bn_act_*_bf16 (those write partial rows over its head) */
int cn_channel_sum_bf16(const void* x, long ldx, long P, int C, float* out, int accumulate, float* ws, void* stream);

/* nn.LayerNorm over channels (rows of the NHWC image), + residual; statistics recomputed in backward.
 * dw / db ACCUMULATED (fp32 atomics). */
int cn_layernorm_c_fwd_bf16(const void* x, long ldx, const float* w, const float* b, const void* res /*nullable*/,
                            long ldr, void* y, long ldy, long P, int C, float eps, void* stream);
int cn_layernorm_c_bwd_bf16(const void* x, long ldx, const void* dy, long lddy, const float* w, void* dx, long lddx,
                            float* dw, float* db, long P, int C, float eps, int accumulate_dx, void* stream);

/* edges of the bf16 region: fp32 NCHW <-> bf16 NHWC (channels C..Cpad-1 zero-filled) */
int cn_convert_f32nchw_to_bf16nhwc(const float* src, long sbs, void* dst, long ld, int B, int C, int Cpad, int HW,
                                   void* stream);
int cn_convert_bf16nhwc_to_f32nchw(const void* src, long ld, float* dst, long dbs, int B, int C, int HW,
                                   int accumulate, void* stream);

/* torch.cat / residual adds / fills on [P][C] row slices */
int cn_copy_bf16(const void* src, long lds, void* dst, long ldd, long P, int C, int accumulate, void* stream);
int cn_add_bf16(const void* a, long lda, const void* c, long ldc, void* dst, long ldd, long P, int C, void* stream);
int cn_zero_bf16(void* dst, long ldd, long P, int C, void* stream);

/* F.interpolate(bilinear, align_corners=True) */
int cn_bilinear_fwd_bf16(const void* x, long ldx, void* y, long ldy, int B, int C, int Hi, int Wi, int Ho, int Wo,
                         void* stream);
int cn_bilinear_bwd_bf16(const void* dy, long lddy, void* dx, long lddx, int B, int C, int Hi, int Wi, int Ho, int Wo,
                         int accumulate, void* stream);

/* NeighborhoodAttention2D core: qkv [B][H][W][3C], out [B][H][W][C]; attn / dattn fp32 [B][heads][9][H][W]. */
int cn_na2d_fwd_bf16(const void* qkv, long ldq, void* out, long ldo, float* attn, int B, int C, int heads, int H, int W,
                     int kernel_size, int dilation, float attn_drop, unsigned long long seed,
                     const unsigned long long* step, void* stream);
int cn_na2d_bwd_bf16(const void* qkv, long ldq, const void* dout, long ldo, const float* attn, float* dattn, void* dqkv,
                     long lddq, int B, int C, int heads, int H, int W, int kernel_size, int dilation, float attn_drop,
                     unsigned long long seed, const unsigned long long* step, void* stream);

/* nn.Dropout2d (channelwise) / nn.Dropout on bf16 NHWC activations [B*HW rows][C], the same counter-based masks as
 * cn_dropout_f32 (convolution.py:495,511; natten proj_drop); backward = the same call on dy with accumulate. */
int cn_dropout_bf16(const void* x, long ldx, void* y, long ldy, int B, int C, int HW, float p, unsigned long long seed,
                    const unsigned long long* step, int channelwise, int accumulate, void* stream);

/* ---- PreTimeReduction (models/nunet.py:18-105) fused: both Conv3d stacks (k = 3, 5: Conv3d(C->C,(k,1,1)) -> BatchNorm3d
 * -> SiLU -> Conv3d(C->Cout,(T-k+1,1,1)) -> BatchNorm2d -> SiLU), their sum and the LayerNorm over Cout, recomputed from
 * x [B][C*T][HW] (batch stride xbs) in every pass: training forward = 3 launches (+ 1 weight transpose), backward = 3,
 * inference = 1; every pass finishes its own cross-block reduction (last-block tickets, fixed summation order).
 * params: HOST array of 22 device pointers: per branch (k = 3, then k = 5) {wa [C][C][k], wb [Cout][C][T-k+1], gamma3,
 * beta3, running_mean3, running_var3 (nullable in training), gamma2, beta2, running_mean2, running_var2}, then
 * {ln_gamma, ln_beta}. stats: HOST array of 8 device pointers: per branch {mean3 [C], rstd3 [C], mean2 [Cout], rstd2
 * [Cout]}: written by a training forward, read by backward. bn: HOST {eps3, momentum3, eps2, momentum2}.
 * y / dy: out_kind 0 = fp32 NCHW (stride = batch stride), 1 = bf16 NHWC (stride = pixel stride, Cout innermost).
 * grads: HOST array of 14 device pointers: per branch {dwa, dwb, dgamma3, dbeta3, dgamma2, dbeta2}, then {d ln_gamma,
 * d ln_beta}; ACCUMULATED. ws: cn_pretime_workspace_floats(...) floats, ZERO-FILLED before its first use (ticket
 * counters at its head; every call leaves them zero); -1 = shape outside the fused kernel (C <= 8, 8 <= Cout <= 64,
 * Cout % 8 == 0, T >= 5, every pass's LDS image within 160 KiB): the caller keeps its generic path. */
long cn_pretime_workspace_floats(int B, int C, int T, int HW, int Cout, int with_backward);
int cn_pretime_fwd_f32(const float* x, long xbs, const void* const* params, float* const* stats, void* y, long y_stride,
                       int out_kind, int B, int C, int T, int HW, int Cout, int training, const float* bn, float eps_ln,
                       float* ws, long ws_floats, void* stream);
int cn_pretime_bwd_f32(const float* x, long xbs, const void* const* params, float* const* stats, const void* dy,
                       long dy_stride, int out_kind, float* const* grads, int B, int C, int T, int HW, int Cout,
                       int training, const float* bn, float eps_ln, float* ws, long ws_floats, void* stream);

/* ---- diagnostics: per-launch HIP-event timing of the contraction kernels (bench.py roofline) ---
 * begin() starts recording event pairs around every implicit-GEMM / weight-gradient launch on the
 * launch stream; end() synchronises them and fills out[8][3] = {milliseconds, algorithmic flops,
 * launches} for kinds {igemm NT=128, igemm NT<=64, wgrad 3x3, wgrad 1x1, bf16 conv, bf16 wgrad, -, -}.
 * Process-global, off by default, single client, not thread-safe. */
int cn_profile_begin(void);
int cn_profile_end(double* out);
/* after cn_profile_end: the rank-th recorded kernel by total time (rocprof-style name, e.g.
 * "cn_conv_igemm_vec_kernel<4, 1, 5, 1, 1>") with out[3] = {milliseconds, algorithmic flops, launches};
 * returns the number of distinct kernels recorded. */
int cn_profile_top(int rank, char* name_out, int cap, double* out);
/* after cn_profile_end: ALGORITHMIC HBM bytes of the rank-th recorded kernel, summed over its recorded launches --
 * every operand read once and every result written once at the true (unpadded) dims, stated at the launch sites of the
 * contraction kernels (the figure bench.py's roofline.traffic, taken from PMC counters, is held against).
 * 0.0 for kernels whose launch site states none. */
double cn_profile_top_bytes(int rank);
/* record only launches of the kernel with exactly this name in the windows that follow ("" / NULL: every
 * contraction launch): bench.py brackets just the dominant kernel inside the timed region, so that the event
 * markers of the other ~250 launches per step do not perturb the time being measured. */
int cn_profile_set_filter(const char* name);
/* kernel launches issued by the library so far (every launch goes through one counting macro);
 * reset != 0 returns the count and zeroes it. bench.py reports it as kernel launches per step. */
long cn_launch_count(int reset);

/* ---- deferred weight-gradient slice sums (cn_slicesum.h) ------------------------------------------------
 * A many-split weight gradient leaves one partial dW slice per pixel split in the caller's scratch and must add
 * them into dW (plain autograd accumulation: /root/reference/src/cultionet/nn/modules/convolution.py:71-120)
 * before the optimizer / the bucket all-reduce reads it -- not before the next backward node. Between
 * cn_slice_sums_begin and cn_slice_sums_end the *_bwd_weight_* entry points called on THIS thread with a dw
 * inside [dw_lo, dw_lo + dw_floats) (the flat gradient buffer; a temporary dw is summed at once) append a
 * 64-byte record to host_table (caller-owned, `capacity` records; pinned if cn_slice_sums_run uploads it)
 * instead of launching their own reduction -- the caller must then keep every such call's scratch intact until
 * the sums have run. cn_slice_sums_count: records so far (-1: no sink).
 * cn_slice_sums_run: ONE launch summing records [first, first+n) (its blocks are dealt to the records on the
 * host; that is written into host_table); dev_table = device copy of the table, refreshed from host_table for
 * that range when upload != 0 (async on `stream`; upload == 0 asserts it already holds this range, dealt alike). Deterministic summation order; a record whose
 * dW already has a pending record is refused by the sink (that call reduces immediately, as without a sink). */
int cn_slice_sums_begin(void* host_table, int capacity, const float* dw_lo, long dw_floats);
int cn_slice_sums_count(void);
int cn_slice_sums_end(void);
int cn_slice_sums_run(void* host_table, void* dev_table, int first, int n, int upload, void* stream);

/* ---- runtime plumbing: streams torch cannot create ---------------------------------------------------
 * The training step issues weight gradients on a second stream (engine.py side_stream; the reference has no
 * counterpart: torch.autograd runs one stream). That stream must not starve the data-gradient chain:
 * cn_stream_create(priority) = hipStreamCreateWithPriority (out[0] = least, out[1] = greatest of
 * cn_stream_priority_range), or with cu_mask != NULL hipExtStreamCreateWithCUMask (bit i = CU i). */
int cn_stream_priority_range(int* out);
int cn_stream_create(int priority, const unsigned* cu_mask, int cu_mask_words, void** stream_out);
int cn_stream_destroy(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CULTIONET_HIP_H */
