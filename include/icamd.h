/* icamd.h -- C ABI of the MI355X (gfx950) training-step kernels.
 *
 * Drop-in boundary for ONE hot path of abelxiaoxing/ImageClassification: the per-step compute of
 * engine.py:train_one_epoch / evaluate (reference: /root/reference/engine.py:10-225).  The reference has no
 * FFI of its own (it is pure Python on torch); these entry points are what a binding for that path would
 * bind: every device-side operation the step performs, as plain C functions over raw device pointers.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'd / torch data_ptr()), 16-byte aligned;
 *   - activations are NHWC bf16 (raw 16-bit patterns); parameters, gradients and optimizer state are fp32;
 *     filters handed to the conv kernels are bf16 [Cout][KH][KW][Cin] ("shadow" copies the optimizer emits);
 *   - `stream` is a hipStream_t passed as void*; functions only enqueue work: they never allocate,
 *     never synchronise, never throw.  Return 0 on success, ICAMD_ERR_* otherwise;
 *   - workspaces are caller-owned; sizes come from the *_workspace_bytes queries; the BatchNorm workspaces hold
 *     arrival counters of a one-launch reduce+finalize and must be ZERO-FILLED once after allocation (the kernels
 *     leave them zeroed again);
 *   - all floating-point reductions are order-fixed (no float atomics): results are bitwise reproducible.
 */
#ifndef ICAMD_H
#define ICAMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICAMD_OK 0
#define ICAMD_ERR_BAD_ARG 1
#define ICAMD_ERR_UNSUPPORTED 2
#define ICAMD_ERR_WORKSPACE 3
#define ICAMD_ERR_LAUNCH 4

int icamd_abi_version(void);

/* ---- convolution (replaces ATen conv2d fwd / dgrad / wgrad under model(samples), loss.backward():
 *      /root/reference/engine.py:48,51,64,72; model built at train.py:194) ---------------------------- */
typedef struct icamd_conv_desc {
  int N, IH, IW, Cin;   /* input  [N, IH, IW, Cin]  (Cin == 8 for the zero-padded RGB stem, else Cin % 64 == 0) */
  int OH, OW, Cout;     /* output [N, OH, OW, Cout] (Cout % 8 == 0)                                             */
  int KH, KW, stride, pad;
} icamd_conv_desc;

/* number of rows of the per-channel statistics partials a forward call writes: ceil(N*OH*OW / 128) */
int icamd_conv2d_stats_rows(const icamd_conv_desc* d);

/* y = conv(x, w) (+ bias[co]) (+ addend, same shape as y), rounded once to bf16.
 * stats (optional): float [stats_rows][2][Cout] <- partial sums and sums of squares of the ROUNDED y over disjoint pixel
 * tiles (rows a kernel's tiling does not need are written as zeros): BatchNorm batch statistics, consumed -- summed over
 * all stats_rows rows -- by icamd_bn_train_finalize. */
int icamd_conv2d_fwd(const icamd_conv_desc* d, const void* x, const void* w, void* y, const float* bias,
                     const void* addend, float* stats, void* stream);
/* Inference form of icamd_conv2d_fwd (evaluate(), engine.py:145-225, on a model whose BatchNorms were folded with
 * icamd_bn_fold_filters): y = [relu](conv(x, w) + bias + addend), one rounding; no statistics. */
int icamd_conv2d_fwd_act(const icamd_conv_desc* d, const void* x, const void* w, void* y, const float* bias,
                         const void* addend, int relu, void* stream);
/* Mlp fc1 forward with the activation fused into the store pass: z = conv(x, w) + bias rounded to bf16, a = gelu(z)
 * (exact erf GELU of the ROUNDED z: bit-identical to icamd_conv2d_fwd followed by icamd_gelu_fwd, one read of z less).
 * z may be NULL for a forward pass that keeps nothing for backward (the reference's second, accuracy-only forward under
 * mixup, engine.py:89-97): only a is written. */
int icamd_conv2d_fwd_gelu(const icamd_conv_desc* d, const void* x, const void* w, void* z, void* a, const float* bias,
                          void* stream);
/* Eval-mode BatchNorm (running statistics) folded into the [Cout][K] fp32 filters in front of it:
 * w_folded = bf16(w * gamma/sqrt(running_var+eps)) per output channel, shift = beta - running_mean * that scale. */
int icamd_bn_fold_filters(const float* w, const float* gamma, const float* beta, const float* running_mean,
                          const float* running_var, float eps, int Cout, int K, void* w_folded, float* shift,
                          void* stream);

/* dx = conv_transpose(dy, w) (+ addend shaped like dx).  w_t is the [Cin][KH][KW][Cout] transposed bf16 filter
 * (icamd_filter_transpose).  Requires Cout % 64 == 0.  addend_maskbits (optional, 1 bit per addend element, as
 * written by icamd_bn_apply): the addend is counted only where its bit is set -- lets the residual branch add
 * "output gradient x ReLU mask" without that product ever being materialised. */
int icamd_conv2d_dgrad(const icamd_conv_desc* d, const void* dy, const void* w_t, void* dx, const void* addend,
                       const uint8_t* addend_maskbits, void* stream);
/* Same, with an addend that exists on the EVEN pixel grid only: addend_sub2 is [N][ceil(IH/2)][ceil(IW/2)][Cin] and is
 * added to dx[n][2p][2q][:]; every other pixel gets no addend.  This is the gradient a 1x1 stride-2 shortcut convolution
 * (ResNet downsample, /root/reference/models/ -> torchvision resnet `downsample`) sends back to the block input:
 * compute it as the stride-1 data gradient on the [N][OH][OW] grid and hand it over here; the 3/4 of the full-size
 * tensor that would be zeros is never written or read. */
int icamd_conv2d_dgrad_sub2(const icamd_conv_desc* d, const void* dy, const void* w_t, void* dx, const void* addend_sub2,
                            void* stream);

/* Mlp fc2 data gradient with the GELU backward fused into the store pass: dz = bf16(conv_transpose(dy, w)) * gelu'(z),
 * z shaped like dz (bit-identical to icamd_conv2d_dgrad followed by icamd_gelu_bwd; the gradient of the GELU output is
 * never written). */
int icamd_conv2d_dgrad_gelu(const icamd_conv_desc* d, const void* dy, const void* w_t, const void* z, void* dz,
                            void* stream);

/* Data gradient with the NEXT BatchNorm backward's first pass fused into the epilogue: the tensor this call produces
 * is the output-gradient of a BatchNorm(+residual)+ReLU layer, so the kernel applies that layer's ReLU mask
 * (mask_src > 0, or y*scale+shift > 0 when mask_src is NULL), stores g = masked gradient, and writes per-tile partial
 * sums of g and g*xhat (xhat = (y-mean)*invstd) -- consumed by icamd_bn_bwd_from_partials.
 * partials: float [icamd_conv2d_dgrad_stats_rows(d)][2][Cin]. */
typedef struct icamd_bn_bwd_fuse {
  const void* y;         /* that BatchNorm's input (a conv output), shaped like dx                                 */
  const void* mask_src;  /* post-activation tensor shaped like dx, or NULL                                         */
  const float *mean, *invstd, *scale, *shift;
  float* partials;
  int relu;
} icamd_bn_bwd_fuse;
int icamd_conv2d_dgrad_stats_rows(const icamd_conv_desc* d);
int icamd_conv2d_dgrad_bnbwd(const icamd_conv_desc* d, const void* dy, const void* w_t, void* g, const void* addend,
                             const icamd_bn_bwd_fuse* f, void* stream);

size_t icamd_conv2d_wgrad_workspace_bytes(const icamd_conv_desc* d);
/* dw (fp32 [Cout][KH][KW][Cin]) = (accumulate ? dw : 0) + sum over pixels of dy (x) x */
int icamd_conv2d_wgrad(const icamd_conv_desc* d, const void* x, const void* dy, float* dw, int accumulate,
                       void* workspace, size_t workspace_bytes, void* stream);
/* Same, and the bias gradient dbias[co] = sum_m dy[m][co] (Cout floats, (+)= per `accumulate`) computed inside the same
 * kernel from the dy fragments it already holds (replaces a separate column-sum pass over dy; nn.Linear / conv biases of
 * ViT and ConvNeXt).  Same workspace. */
int icamd_conv2d_wgrad_bias(const icamd_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias,
                            int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* table-driven batched filter transpose [Cout][T][Cin] -> [Cin][T][Cout] (bf16).
 * descs: int64[nlayers][8] = {src_off, dst_off, Cout, T, Cin, 0,0,0} (element offsets);
 * jobs: int32[njobs][2] = {layer, first destination element}, each job covers 4096 destination elements. */
int icamd_filter_transpose(const void* src_base, void* dst_base, const int64_t* descs, const int32_t* jobs, int njobs,
                           void* stream);
/* same, for layers with Cout % 64 == 0 and Cin % 64 == 0, 64x64 tiles through LDS (coalesced both ways);
 * jobs: int32[njobs][4] = {layer, tap, co0, ci0}, co0 and ci0 multiples of 64. */
int icamd_filter_transpose_tiled(const void* src_base, void* dst_base, const int64_t* descs, const int32_t* jobs,
                                 int njobs, void* stream);

/* ---- BatchNorm / ReLU / residual (timm BatchNorm2d + ReLU layers under the same reference calls) -------- */
size_t icamd_bn_workspace_bytes(int C);
/* training-mode finalize from the conv epilogue partials: batch mean / biased var -> mean, invstd, and the
 * folded scale = gamma*invstd, shift = beta - mean*scale; running stats updated with `momentum`
 * (unbiased variance), as torch.nn.functional.batch_norm(training=True). */
int icamd_bn_train_finalize(const float* partials, int nrows, int C, double count, const float* gamma,
                            const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                            float* mean, float* invstd, float* scale, float* shift, void* workspace, void* stream);
int icamd_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, float eps, float* scale, float* shift, void* stream);
/* out = act(y*scale[c] + shift[c] (+ residual)), act = ReLU if relu else identity.
 * maskbits (optional, numel/8 bytes): bit k of byte i <- [out[8i+k] > 0], the ReLU mask for the backward pass. */
int icamd_bn_apply(const void* y, const float* scale, const float* shift, const void* residual, void* out,
                   uint8_t* maskbits, long long numel, int C, int relu, void* stream);
/* Same with the residual given as the RAW conv output of the shortcut (downsample) branch: its BatchNorm (no ReLU) is
 * applied on the fly, residual = bf16(res_y * res_scale[c] + res_shift[c]) -- bit-identical to running icamd_bn_apply on
 * the shortcut first, without writing and re-reading that activation. */
int icamd_bn_apply_res_bn(const void* y, const float* scale, const float* shift, const void* res_y, const float* res_scale,
                          const float* res_shift, void* out, uint8_t* maskbits, long long numel, int C, int relu,
                          void* stream);
size_t icamd_bn_bwd_workspace_bytes(long long rows, int C);
/* g = dout * mask, mask (when relu) = maskbits if given, else [act > 0] if act given, else [y*scale+shift > 0];
 * dgamma = sum g*xhat, dbeta = sum g, dy = scale*(g - mean(g) - xhat*mean(g*xhat)); gout (optional) <- g */
int icamd_bn_bwd(const void* dout, const void* act, const void* y, const float* mean, const float* invstd,
                 const float* scale, const float* shift, float* dgamma, float* dbeta, void* dy, void* gout,
                 const uint8_t* maskbits, long long rows, int C, int relu, int accumulate, void* workspace,
                 size_t workspace_bytes, void* stream);
/* The same for the BatchNorm + ReLU in front of a 3x3 / stride 2 / pad 1 max-pool (ResNet stem, fused forward:
 * icamd_bn_relu_maxpool3x3s2_fwd): dout_pooled is the gradient of the POOLED map [N][OH][OW][C], idx its recorded argmax;
 * the max-pool backward is folded into both passes (the full-resolution gradient [N][IH][IW][C] is never materialised;
 * bit-identical to icamd_maxpool3x3s2_bwd followed by icamd_bn_bwd with relu = 1 and the mask recomputed from y).
 * Workspace: icamd_bn_bwd_workspace_bytes(N*IH*IW, C). */
int icamd_bn_bwd_maxpool3x3s2(const void* dout_pooled, const uint8_t* idx, const void* y, const float* mean,
                              const float* invstd, const float* scale, const float* shift, float* dgamma, float* dbeta,
                              void* dy, int N, int IH, int IW, int C, int accumulate, void* workspace, size_t workspace_bytes,
                              void* stream);

/* Two BatchNorm backward passes over ONE masked output gradient g = dout * maskbits: the last BatchNorm of a residual block
 * (conv output yA) and the BatchNorm of its projection shortcut (yB), /root/reference's timm Bottleneck/BasicBlock with
 * `downsample`.  Same results as two icamd_bn_bwd calls (relu = 1, maskbits given); dout and the bits are read twice instead
 * of four times.  Two distinct zero-initialised workspaces of icamd_bn_bwd_workspace_bytes(rows, C) each. */
int icamd_bn_bwd_dual(const void* dout, const uint8_t* maskbits, const void* yA, const float* meanA, const float* invstdA,
                      const float* scaleA, float* dgammaA, float* dbetaA, void* dyA, const void* yB, const float* meanB,
                      const float* invstdB, const float* scaleB, float* dgammaB, float* dbetaB, void* dyB, long long rows, int C,
                      int accumulate, void* workspaceA, void* workspaceB, size_t workspace_bytes, void* stream);

/* second half of the fused form: finalize (dgamma, dbeta, means) + dy = scale*(g - mean(g) - xhat*mean(g*xhat)) */
size_t icamd_bn_bwd_apply_workspace_bytes(int C);
int icamd_bn_bwd_from_partials(const float* partials, int nrows, const void* g, const void* y, const float* mean,
                               const float* invstd, const float* scale, float* dgamma, float* dbeta, void* dy,
                               long long rows, int C, int accumulate, void* workspace, size_t workspace_bytes,
                               void* stream);

/* The residual data gradient of a bottleneck's first 1x1 convolution produces d(previous block's output): fused form that
 * also does pass 1 of THAT block's last BatchNorm backward (replaces the bn_bwd reduce pass: one full read of dout and y).
 *   g = (dgrad(dy, w_t) + addend [gated by addend_bits]) gated by bn_bits (the previous block's ReLU mask, 1 bit / element);
 *   addend_sub2 != 0: addend is [N][ceil(IH/2)][ceil(IW/2)][Cin], added at the even pixels (as icamd_conv2d_dgrad_sub2;
 *   addend_bits must then be NULL)
 *   partials float [icamd_conv2d_dgrad_stats_rows(d)][2][Cin]: per-workgroup sums of g and of g * bn_y (bn_y = the raw
 *   convolution output that BatchNorm normalised); icamd_bn_bwd_from_gy_partials turns them into the BatchNorm gradients.
 * _supported: 1 where the register-resident pointwise kernel has this form (1x1 / stride 1, Cout in {64, 128, 256},
 * Cin % 256 == 0, >= 8192 pixels); elsewhere the caller keeps icamd_conv2d_dgrad + icamd_bn_bwd. */
int icamd_conv2d_dgrad_bnred_supported(const icamd_conv_desc* d);
int icamd_conv2d_dgrad_bnred(const icamd_conv_desc* d, const void* dy, const void* w_t, void* g, const void* addend,
                             const uint8_t* addend_bits, int addend_sub2, const void* bn_y, const uint8_t* bn_bits,
                             float* partials, void* stream);
/* as icamd_bn_bwd_from_partials, the second partial sum being sum g*y: sum g*xhat = invstd * (sum g*y - mean * sum g) (fp64) */
int icamd_bn_bwd_from_gy_partials(const float* partials, int nrows, const void* g, const void* y, const float* mean,
                                  const float* invstd, const float* scale, float* dgamma, float* dbeta, void* dy,
                                  long long rows, int C, int accumulate, void* workspace, size_t workspace_bytes,
                                  void* stream);

/* Round 5: the END of one bottleneck block and the START of the next in one pass (timm Bottleneck.forward: bn3, shortcut add, act3 of
 * block b; conv1 of block b + 1 and the statistics of its bn1 -- under model(samples), /root/reference/engine.py:48,51):
 *   out = relu(y * scale[c] + shift[c] + residual)  (bf16 [N,IH,IW,Cin]; exactly what icamd_bn_apply / icamd_bn_apply_res_bn store,
 *         incl. the mask bits; res_scale / res_shift != NULL: residual is the RAW shortcut convolution output, normalised on the fly),
 *   y1  = out * w^T  (bf16 [N,OH,OW,Cout], w = [Cout][Cin] filter of the 1x1 / stride-1 convolution d), stats as icamd_conv2d_fwd's.
 * _supported: 1 for (Cin, Cout) in {(256, 64), (256, 128), (512, 128)} with >= 16384 pixels; elsewhere the caller keeps
 * icamd_bn_apply(_res_bn) + icamd_conv2d_fwd, which store the same bytes. */
int icamd_bn_apply_conv1x1_fused_supported(const icamd_conv_desc* d);
int icamd_bn_apply_conv1x1_fused(const icamd_conv_desc* d, const void* y, const float* scale, const float* shift,
                                 const void* residual, const float* res_scale, const float* res_shift, void* out,
                                 uint8_t* maskbits, const void* w, void* y1, float* stats, void* stream);

/* Round 5: the whole backward of "1x1 convolution -> BatchNorm" at the end of a bottleneck block (timm Bottleneck.conv3 + bn3 under
 * loss.backward(), /root/reference/engine.py:64,72) in ONE pass over the 4*planes-wide tensors:
 *   c1, c2, dgamma, dbeta from the partial rows (sum g, sum g*y) an icamd_conv2d_dgrad_bnred call left (as icamd_bn_bwd_from_gy_partials),
 *   dy = scale * (g - c1 - xhat * c2)  -- never stored --,  dx = dy * w  (bf16 [N,IH,IW,Cin]),  dw (+)= dy^T x  (fp32 [Cout][Cin]).
 * d describes the convolution (1x1, stride 1, no padding); x is its input, w_t its transposed filter [Cin][Cout]; g, y are
 * [N,OH,OW,Cout] (g already masked by the block's ReLU bits).  bn_workspace as icamd_bn_bwd_from_gy_partials
 * (icamd_bn_bwd_apply_workspace_bytes(Cout), zero-filled once); wgrad_workspace >= icamd_conv1x1_bn_bwd_fused_workspace_bytes(d).
 * partials == NULL: no sums exist yet -- the call first runs icamd_bn_bwd's reduce pass over g and y (bn_workspace then as
 * icamd_bn_bwd's: icamd_bn_bwd_workspace_bytes(N*OH*OW, Cout), zero-filled once); used for a projection shortcut's 1x1 / stride-1
 * convolution + BatchNorm, whose output gradient is the block's g as well.
 * _supported: 1 for (Cin, Cout) = (64, 256) and (128, 512) with >= 16384 pixels; elsewhere the caller keeps
 * icamd_bn_bwd_from_gy_partials + icamd_conv2d_dgrad + icamd_conv2d_wgrad, which compute the same three results. */
int icamd_conv1x1_bn_bwd_fused_supported(const icamd_conv_desc* d);
size_t icamd_conv1x1_bn_bwd_fused_workspace_bytes(const icamd_conv_desc* d);
int icamd_conv1x1_bn_bwd_fused(const icamd_conv_desc* d, const float* partials, int nrows, const void* g, const void* y,
                               const float* mean, const float* invstd, const float* scale, float* dgamma, float* dbeta,
                               const void* x, const void* w_t, void* dx, float* dw, int accumulate, void* bn_workspace,
                               size_t bn_workspace_bytes, void* wgrad_workspace, size_t wgrad_workspace_bytes, void* stream);

/* ---- LayerNorm / GELU / long column sums (ViT and ConvNeXt layers of the same reference calls; LayerNorm spec
 *      /root/reference/semantic_segmentation/backbone/convnext.py:158-182, exact-erf GELU :37) ------------------- */
/* y = (x - mean_C) * rstd * gamma + beta over the last dimension of x [rows][C] (C % 4 == 0, C <= 1024);
 * mean / rstd (float [rows]) are saved for the backward pass */
int icamd_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                        long long rows, int C, float eps, void* stream);
size_t icamd_layernorm_bwd_workspace_bytes(long long rows, int C);   /* zero-fill once (arrival counters) */
/* dx = LayerNorm-backward(dy) (+ addend: the skip connection's gradient, same shape) */
int icamd_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                        const void* addend, void* dx, float* dgamma, float* dbeta, long long rows, int C, int accumulate,
                        void* workspace, size_t workspace_bytes, void* stream);
int icamd_gelu_fwd(const void* z, void* a, long long numel, void* stream);              /* a = z * Phi(z) (erf form) */
int icamd_gelu_bwd(const void* da, const void* z, void* dz, long long numel, void* stream);
size_t icamd_colsum_rows_workspace_bytes(long long rows, int cols);   /* zero-fill once */
int icamd_colsum_rows(const void* x, long long rows, int ld, int cols, float* out, int accumulate, void* workspace,
                      size_t workspace_bytes, void* stream);

/* ---- ConvNeXt block pieces (/root/reference/semantic_segmentation/backbone/convnext.py:21-56; timm convnext_tiny
 *      under train.py:194).  Depthwise 7x7 pad 3: x, y NHWC bf16 [N,H,W,C] (C % 32 == 0), w bf16 [7][7][C] (tap-major
 *      shadow of the [C,1,7,7] parameter), bias fp32 [C]. ---------------------------------------------------------- */
int icamd_dwconv7_fwd(const void* x, const void* w, const float* bias, void* y, int N, int H, int W, int C, void* stream);
/* dx = depthwise-conv-transpose(dy, w) (+ addend: the residual branch's gradient) */
int icamd_dwconv7_dgrad(const void* dy, const void* w, const void* addend, void* dx, int N, int H, int W, int C, void* stream);
size_t icamd_dwconv7_wgrad_workspace_bytes(int N, int H, int W, int C);
/* dw fp32 [7][7][C] (+)= sum over pixels dy * shifted x */
int icamd_dwconv7_wgrad(const void* x, const void* dy, float* dw, int accumulate, void* workspace, size_t workspace_bytes,
                        int N, int H, int W, int C, void* stream);
/* Round 5: the same pass also leaves the bias gradient dbias fp32 [C] (+)= sum over pixels dy (every dy row passes through the
 * kernel's registers once): replaces a separate icamd_colsum_rows pass over dy.  _supported: 1 where the register-sliding-window
 * kernel runs (the shapes of ConvNeXt-T at any batch); elsewhere the call returns ICAMD_ERR_UNSUPPORTED and the two calls remain. */
int icamd_dwconv7_wgrad_bias_supported(int N, int H, int W, int C);
int icamd_dwconv7_wgrad_bias(const void* x, const void* dy, float* dw, float* dbias, int accumulate, void* workspace,
                             size_t workspace_bytes, int N, int H, int W, int C, void* stream);
/* out = inp + keep[sample] * gamma[c] * z  (layer scale + stochastic depth + residual; keep NULL = 1; rows_per_image
 * rows of C channels per sample).  bwd: dz = dout*keep*gamma, dgamma (+)= sum dout*z*keep (workspace zero-filled once). */
int icamd_layerscale_fwd(const void* z, const void* inp, const float* gamma, const float* keep, void* out, long long rows, int C,
                         long long rows_per_image, void* stream);
size_t icamd_layerscale_bwd_workspace_bytes(long long rows, int C);
int icamd_layerscale_bwd(const void* dout, const void* z, const float* gamma, const float* keep, void* dz, float* dgamma,
                         long long rows, int C, long long rows_per_image, int accumulate, void* workspace,
                         size_t workspace_bytes, void* stream);
/* Round 5: the layer scale folded into the Mlp's second Linear layer, so that `out = x + drop_path(gamma * fc2(a))`
 * (ConvNeXt Block.forward of the reference tree's timm model, under model(samples) at /root/reference/engine.py:47) is ONE GEMM with a
 * residual addend: with keep[n] in {0, cb} (timm's drop_path: cb = 1 / keep_prob; cb = 1 without stochastic depth)
 *   out[m][c] = x[m][c] + (a W2'^T + b2')[m][c],   W2'[c][:] = cb * gamma[c] * W2[c][:],   b2'[c] = cb * gamma[c] * b2[c]
 * for every kept sample, and out = x for the dropped ones.
 * icamd_layerscale_fold: jobs = device array of njobs rows of 8 x int64 {filter offset (elements: the same in the fp32 arena
 *   `params` and in its bf16 `shadow`), gamma offset, bias offset (both in `params`), folded-bias offset (in fold_bias), C = filter
 *   rows, K = row length (% 4 == 0), first grid row of the job (rows of all jobs are numbered consecutively, total_rows in all), the
 *   float bits of cb}: shadow[filter] <- bf16(cb * gamma[c] * params[filter]), fold_bias <- cb * gamma[c] * bias[c].
 * icamd_rows_fix: for every sample n with keep[n] == 0: dst1[n] <- src1 ? src1[n] : 0 (bytes1 per sample) and dst2[n] <- 0 (bytes2
 *   per sample; either pair may be NULL; byte counts % 16 == 0).  Forward: out[n] <- x[n] and a[n] <- 0 (so that the dropped
 *   samples vanish from the weight gradient); backward: the dropped samples' rows of d(fc1 output) <- 0.
 * icamd_dropped_colsum: partial[n][c] <- sum of the rows of sample n of dy (bf16 [n_images * rows_per_image][C]) where keep[n] == 0,
 *   zeros for the kept samples.
 * icamd_layerscale_param_grads: G = dy^T a (fp32 [C][K], the plain weight gradient of the folded layer over ALL rows -- the dropped
 *   samples' rows of `a` are zero), colsum_all[c] = sum over all rows of dy; S = colsum_all - sum_n dropped[n] (dropped may be NULL):
 *   dw (+)= cb * gamma[c] * G,  dbias (+)= cb * gamma[c] * S,  dgamma[c] (+)= cb * (<G[c], w[c]> + bias[c] * S[c])   (w, bias: fp32
 *   parameters of the UNFOLDED layer). */
int icamd_layerscale_fold(const float* params, void* shadow, float* fold_bias, const long long* jobs, int njobs, int total_rows,
                          long long total_elements, void* stream);
int icamd_rows_fix(const float* keep, int n_images, void* dst1, const void* src1, long long bytes1, void* dst2, long long bytes2,
                   void* stream);
int icamd_dropped_colsum(const void* dy, const float* keep, int n_images, long long rows_per_image, int C, float* partial,
                         void* stream);
int icamd_layerscale_param_grads(const float* G, const float* w, const float* bias, const float* gamma, const float* colsum_all,
                                 const float* dropped, int n_images, float cb, int C, int K, float* dw, float* dbias,
                                 float* dgamma, int accumulate, void* stream);

/* ViT token plumbing: tokens[b][0] = cls + pos[0], tokens[b][1+i] = patches[b][i] + pos[1+i] (bf16 out, fp32 parameters);
 * batch_sum: out[j] (+)= sum_b x[b*stride + j] (cls_token / pos_embed gradients); strided row copies; zero fill. */
int icamd_vit_tokens_fwd(const void* patches, const float* cls_token, const float* pos_embed, void* tokens, int B, int T, int C,
                         void* stream);
int icamd_batch_sum(const void* x, long long stride, int B, long long n, float* out, int accumulate, void* stream);
int icamd_strided_rows_copy(const void* src, long long src_stride, void* dst, long long dst_stride, long long rows, long long C,
                            void* stream);
int icamd_fill_zero(void* ptr, size_t bytes, void* stream);

/* ---- multi-head self-attention for short sequences (timm Attention under the same reference calls, ViT-B/16:
 *      T = 197 tokens, 12 heads of 64).  qkv: bf16 [B*T][3*H*D] (q | k | v, each [head][D]); out: bf16 [B*T][H*D];
 *      lse / delta: float [B][H][T] (log-sum-exp of the scaled scores; rowsum(dout*out), scratch for the backward).
 *      D must be 64, T <= 208.  One workgroup per (image, head); no atomics, fixed summation order. --------------- */
int icamd_attention_fwd(const void* qkv, void* out, float* lse, int B, int T, int H, int D, float scale, void* stream);
int icamd_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                        int B, int T, int H, int D, float scale, void* stream);

/* ---- pooling ---------------------------------------------------------------------------------------- */
int icamd_maxpool3x3s2_fwd(const void* x, void* out, uint8_t* argmax, int N, int IH, int IW, int C, void* stream);
/* ResNet stem in training: BatchNorm-apply + ReLU + max-pool 3x3/2 in one pass over the conv output y.  Bit-identical
 * to icamd_bn_apply(relu) followed by icamd_maxpool3x3s2_fwd (every window element is rounded to bf16 as the stored
 * activation would have been); the full-resolution activation is never written. */
int icamd_bn_relu_maxpool3x3s2_fwd(const void* y, const float* scale, const float* shift, void* out, uint8_t* argmax, int N,
                                   int IH, int IW, int C, void* stream);
int icamd_maxpool3x3s2_bwd(const void* dout, const uint8_t* argmax, void* dx, int N, int IH, int IW, int C, void* stream);
int icamd_avgpool_fwd(const void* x, void* out, int N, int HW, int C, void* stream);
int icamd_avgpool_bwd(const void* dout, void* dx, int N, int HW, int C, void* stream);

/* ---- input packing + mixup/cutmix (replaces samples.to(device) + timm Mixup.__call__, engine.py:40-44) -- */
/* x: fp32 NCHW [B,Cin,H,W] (device) -> out: bf16 NHWC [B,H,W,8]; mode 0 none, 1 mixup(lam), 2 cutmix(box) */
int icamd_pack_input(const float* x, void* out, int B, int Cin, int H, int W, int mode, float lam, int yl, int yh,
                     int xl, int xh, void* stream);

/* ResNet stem layout: bf16 [B][H][We+8][4], We = W rounded up to even -- RGB + one zero channel, 3 zero columns left and 5 (6
 * for an odd W) right of every row, so that the 7x7/2 window row of output column q is 64 contiguous 16 B-aligned bytes.  The
 * stem kernels are then called with IW = We: the extra zero column is part of the convolution's own zero padding, the
 * output is that of the true width.  Same mixing modes as above. */
int icamd_pack_input_rgb4(const float* x, void* out, int B, int Cin, int H, int W, int mode, float lam, int yl, int yh,
                          int xl, int xh, void* stream);

/* ---- ResNet stem convolution (timm resnet conv1: 7x7, stride 2, padding 3; /root/reference/train.py:194) on that layout.
 * w: bf16 [Cout][8][8][4] = filter[co][row][column][channel] with row 7, column 7 and channel 3 ZERO (the reduction runs
 * over 8 x 8 x 4 = 256 entries: 1.74x the algorithmic work, against 2.67x for icamd_conv2d_fwd on 8 zero-padded channels).
 * y / stats / bias / relu as icamd_conv2d_fwd / _fwd_act; stats rows = icamd_stem7x7s2_stats_rows.
 * wgrad: dw fp32 [Cout][8][8][4], the padding entries come out exactly zero. */
int icamd_stem7x7s2_stats_rows(int N, int H, int W);
int icamd_stem7x7s2_fwd(const void* x4, const void* w, void* y, const float* bias, float* stats, int relu, int N, int H,
                        int W, int Cout, void* stream);
size_t icamd_stem7x7s2_wgrad_workspace_bytes(int N, int H, int W, int Cout);
int icamd_stem7x7s2_wgrad(const void* x4, const void* dy, float* dw, int accumulate, void* workspace, size_t workspace_bytes,
                          int N, int H, int W, int Cout, void* stream);

/* ---- input pipeline on the GPU (replaces the per-sample CPU work of the reference's transforms AFTER JPEG decoding:
 *      /root/reference/datasets.py:121-144 -- timm.create_transform(scale=(1,1), ratio=(1,1), vflip=0.5, color_jitter=0.3,
 *      interpolation='bicubic', re_prob=0.25, re_mode='pixel') for training, Resize -> ToTensor -> Normalize for eval;
 *      loaders train.py:152-170).  Arithmetic = Pillow's (two-pass 8-bit resampling with 22-bit weights, ImageEnhance
 *      blends, integer luma): bit-exact, tests/test_image_gpu.py.  Random decisions arrive in the descriptors. */
typedef struct icamd_image_desc {
  int64_t src_offset;                  /* byte offset of this image's uint8 [src_h][src_w][3] pixels in `src`        */
  int32_t src_h, src_w;
  int32_t crop_top, crop_left, crop_h, crop_w;   /* window that is resized to out_h x out_w                          */
  int32_t hflip, vflip;
  int32_t jitter_order[3];             /* ColorJitter: operations in application order, 0 brightness / 1 contrast /
                                          2 saturation, -1 = none                                                     */
  float jitter_factor[3];              /* factor of operation 0, 1, 2                                                 */
  int32_t erase_top, erase_left, erase_h, erase_w;   /* RandomErasing box in the output, erase_h = 0: none            */
  uint32_t erase_seed;                 /* seed of the N(0,1) fill                                                     */
  int32_t reserved;
} icamd_image_desc;
/* kmax: the largest filter window of the batch, ceil(support * max(1, crop / out)) * 2 + 1 with support 2 (bicubic) / 1 */
size_t icamd_image_pipeline_workspace_bytes(int B, int max_crop_h, int out_h, int out_w, int kmax);
/* src, descs, out_nchw (fp32 [B][3][out_h][out_w]) and workspace are device pointers; mean3 / std3 are HOST float[3]
 * (passed on by value).  filter: 0 bilinear, 1 bicubic. */
int icamd_image_pipeline(const uint8_t* src, const icamd_image_desc* descs, int B, int max_crop_h, int out_h, int out_w,
                         int filter, int kmax, const float* mean3, const float* std3, float* out_nchw, void* workspace,
                         size_t workspace_bytes, void* stream);
/* device pointer of the uint8 [B][out_h][out_w][3] image the last call left in `workspace` (after resize, flips and colour
 * jitter, before ToTensor): what Pillow would hold at that point */
int icamd_image_pipeline_u8(const void* workspace, int B, int max_crop_h, int out_h, int out_w, int kmax, const uint8_t** img);

/* ---- loss + metrics (criterion engine.py:49,52,178,181; accuracy / TP-FP-FN engine.py:82-97,184-196) ---- */
/* logits bf16 [B][ld]; targets int64; target distribution lam*onehot_s(y1) + (1-lam)*onehot_s(y2).
 * loss_rows float[B]; pred int32[B] (optional argmax); dlogits bf16 [B][ld] (optional) = (softmax - t)*gscale */
int icamd_softmax_xent(const void* logits, int ld, int B, int C, const int64_t* y1, const int64_t* y2, float lam,
                       float smoothing, float gscale, float* loss_rows, int32_t* pred, void* dlogits, void* stream);
/* mean loss (fixed order) -> loss_out, finite flag, loss_log[log_slot]; unless skipped for a non-finite loss:
 * acc_f64[0]+=loss, [1]+=1, [2]+=correct/B, [3]+=correct, [4]+=B ; counts int32[3][C] = TP, FP, FN.
 * loss_rows == NULL: metrics-only call (uses the finite flag already in *finite_out; acc[0], acc[1] untouched);
 * pred == NULL: loss-only call.  loss_log[log_slot] <- loss; loss_log[log_stride + log_slot] <- correct/B
 * (log_stride > 0).
 * respect_skip is a bit set: 1 = skip the accumulation when the flag is down; 2 = FLAG CALL: write loss_out, finite_out
 * and the log slot, accumulate nothing (data-parallel steps MIN-reduce the flag over the ranks between this call and the
 * accumulating one, so that every rank counts or drops the same steps); 4 = on a metrics-only call (loss_rows == NULL)
 * also add *loss_out to acc[0] and 1 to acc[1] -- the accumulation a flag call left out (pred may then be NULL). */
int icamd_step_metrics(const float* loss_rows, const int32_t* pred, const int64_t* target, int B, int C,
                       float* loss_out, int32_t* finite_out, double* acc_f64, int32_t* counts, float* loss_log,
                       int log_slot, int log_stride, int respect_skip, void* stream);

/* ---- optimizer (optimizer.step / zero_grad / model_ema.update, engine.py:74-77; utils.py:433-468) ------- */
size_t icamd_grad_norm_workspace_bytes(void);
/* out[0] = ||g||_2 * inv_scale ; out[1] = min(1, max_norm/(norm+1e-6)) (1 when max_norm <= 0) */
int icamd_grad_norm(const float* g, long long n, float inv_scale, float max_norm, void* workspace, float* out,
                    void* stream);
/* torch.optim.AdamW step (decoupled wd, bias correction) on flat arenas of n (% 4 == 0) floats, fused with the
 * ModelEmaV3 lerp (ema may be NULL) and the bf16 shadow write (shadow may be NULL).
 * Gradient is scaled by gscale * clip[1] (clip may be NULL).  Skipped when *finite_flag == 0 (reference: non-finite loss
 * -> `continue` before optimizer.step(), engine.py:56-59).
 * `step` >= 1 counts the steps ATTEMPTED (this one included); skipped_steps (device int32, may be NULL) counts the ones
 * the device dropped: the kernel adds 1 to it when it skips, and the bias correction uses t = step - *skipped_steps, the
 * number of steps really taken -- what torch.optim's `step` state would hold in the reference.
 * zero_grad is a bit set: ICAMD_OPT_ZERO_GRAD (1) clears the gradient behind the update; ICAMD_OPT_NO_SKIP_COUNT (2) marks
 * a launch that applies ONE RANGE of a step several launches share (one per gradient bucket, each behind that bucket's
 * all-reduce): p/g/m/v/ema/shadow then point at the range, and only the one launch of the step without the bit adds the
 * dropped step to *skipped_steps (a skipped step is skipped by all of them: they read the same flag). */
enum { ICAMD_OPT_ZERO_GRAD = 1, ICAMD_OPT_NO_SKIP_COUNT = 2 };
int icamd_adamw_ema(float* p, float* g, float* m, float* v, float* ema, void* shadow, long long n, float lr, float wd,
                    float beta1, float beta2, float eps, int step, float gscale, float ema_decay, const float* clip,
                    const int32_t* finite_flag, int32_t* skipped_steps, int zero_grad, void* stream);
/* optimizer.zero_grad() of the reference's non-finite-loss branch (engine.py:56-59) decided on the device: clears the n
 * (% 4 == 0) gradient floats iff *finite_flag == 0.  Called after the backward of every micro-step when update_freq > 1,
 * so a non-finite micro-batch cannot poison the gradients accumulated for the window. */
int icamd_grad_guard(float* g, long long n, const int32_t* finite_flag, void* stream);
/* The reference's other optimizers with a one-pass fused form (optim_factory.py:66-77), same fusion contract as
 * icamd_adamw_ema.  kind: ICAMD_OPT_ADAMW (identical to icamd_adamw_ema), ICAMD_OPT_ADAM (torch.optim.Adam, wd joins
 * the gradient), ICAMD_OPT_SGD_MOMENTUM / ICAMD_OPT_SGD_NESTEROV (torch.optim.SGD, momentum = beta1, dampening 0, wd
 * joins the gradient; m is the momentum buffer, zero-initialised), ICAMD_OPT_LION (timm Lion: p *= 1 - lr*wd;
 * p -= lr*sign(beta1*m + (1-beta1)*g); m = beta2*m + (1-beta2)*g).  v is only used by the Adam kinds (else may be NULL). */
enum { ICAMD_OPT_ADAMW = 0, ICAMD_OPT_ADAM = 1, ICAMD_OPT_SGD_MOMENTUM = 2, ICAMD_OPT_SGD_NESTEROV = 3, ICAMD_OPT_LION = 4 };
int icamd_optim_ema(int kind, float* p, float* g, float* m, float* v, float* ema, void* shadow, long long n, float lr,
                    float wd, float beta1, float beta2, float eps, int step, float gscale, float ema_decay,
                    const float* clip, const int32_t* finite_flag, int32_t* skipped_steps, int zero_grad, void* stream);
int icamd_lerp(float* dst, const float* src, long long n, float w, const int32_t* finite_flag, void* stream);
int icamd_f32_to_bf16(const float* src, void* dst, long long n, void* stream);
int icamd_colsum(const void* x, int rows, int ld, int cols, float* out, int accumulate, void* stream);

/* ---- data-parallel gradient exchange: RCCL over xGMI (replaces DistributedDataParallel's bucketed all-reduce,
 *      /root/reference/train.py:218-222; process group /root/reference/utils.py:339-375, backend 'nccl') ------------
 * One communicator per process (one process per GPU).  RCCL is bound lazily (dlopen): every function returns
 * ICAMD_ERR_UNSUPPORTED where librccl is absent.  unique_id / comm_init / comm_info / comm_destroy are HOST calls
 * (comm_init is the blocking rendezvous of ncclCommInitRank); the *_launch calls only ENQUEUE on `stream`. */
enum { ICAMD_DT_F32 = 0, ICAMD_DT_I32 = 1, ICAMD_DT_F64 = 2, ICAMD_DT_BF16 = 3 };
enum { ICAMD_RED_SUM = 0, ICAMD_RED_MIN = 1, ICAMD_RED_MAX = 2 };
int icamd_rccl_available(void);                 /* 1 when librccl could be bound */
int icamd_rccl_version(void);                   /* ncclGetVersion code, 0 when unavailable */
int icamd_rccl_unique_id(void* id128);          /* rank 0: 128-byte rendezvous blob (HOST memory) for every rank */
int icamd_rccl_comm_init(const void* id128, int nranks, int rank, void** comm_out);
int icamd_rccl_comm_info(void* comm, int* nranks, int* rank);   /* ncclCommCount / ncclCommUserRank of the live communicator */
int icamd_rccl_comm_destroy(void* comm);
/* In-place all-reduce of one bucket (a contiguous slice of the flat gradient arena, or the int32 finite flag with
 * ICAMD_RED_MIN) on the caller's stream: the reducer's side stream, ordered after the backward kernels that produced
 * the slice and before the optimizer kernel by HIP events on the caller's side. */
int icamd_allreduce_bucket_launch(void* comm, void* buf, long long count, int dtype, int op, void* stream);
/* In-place broadcast from `root` (DDP's constructor-time parameter / buffer broadcast; BatchNorm buffers before evaluate) */
int icamd_broadcast_launch(void* comm, void* buf, long long count, int dtype, int root, void* stream);

/* ---- measurement aid (bench.py): HIP-event timing of every entry point on its launch stream ----------------
 * classes: 0 conv fwd, 1 conv dgrad, 2 conv wgrad(+slab reduce), 3 bn finalize, 4 bn apply, 5 bn bwd, 6 pooling,
 * 7 input pack, 8 loss/metrics, 9 optimizer (+filter transpose), 10 misc, 11 attention fwd, 12 attention bwd, 13 LayerNorm fwd,
 * 14 LayerNorm bwd, 15 elementwise (GELU, layer scale), 16 depthwise 7x7.  collect() adds, per class, elapsed ms, calls and
 * (ABI 4; either array may be NULL) the ALGORITHMIC bytes and flops of the calls: every operand tensor read once and every
 * result written once at its stored width (two-pass kernels as their two passes), flops = 2 x multiply-adds. */
int icamd_prof_enable(int on);
int icamd_prof_classes(void);
int icamd_prof_collect(double* ms, long long* calls, double* bytes, double* flops, int n);

#ifdef __cplusplus
}
#endif
#endif /* ICAMD_H */
