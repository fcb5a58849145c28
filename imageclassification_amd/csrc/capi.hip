// C-ABI entry points (include/icamd.h): argument checks + translation into kernel launch parameters.
#include "../../include/icamd.h"
#include "common.h"
#include "icamd_internal.h"
#include <cstdlib>
#include <string.h>

// launchers defined in the kernel translation units
int icamd_bn_finalize_launch(const float* part, int nrows, int C, double count, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, float momentum, float eps, float* mean,
                             float* invstd, float* scale, float* shift, double* chunks, hipStream_t s);
int icamd_bn_eval_coeffs_launch(int C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                float* scale, float* shift, hipStream_t s);
int icamd_bn_apply_launch(const bf16_t* y, const float* scale, const float* shift, const bf16_t* residual, bf16_t* out,
                          unsigned char* maskbits, long long numel, int C, int relu, hipStream_t s,
                          const float* res_scale = nullptr, const float* res_shift = nullptr);
int icamd_bn_bwd_rows_per_block(long long rows, int C);
int icamd_bn_bwd_launch(const bf16_t* dout, const bf16_t* act, const bf16_t* y, const float* mean, const float* invstd,
                        const float* scale, const float* shift, float* dgamma, float* dbeta, bf16_t* dy, bf16_t* gout,
                        const unsigned char* maskbits, long long rows, int C, int relu, int accumulate, float* part,
                        double* chunks, float* c1c2, hipStream_t s, const unsigned char* pool_idx = nullptr,
                        int pool_ih = 0, int pool_iw = 0);
int icamd_bn_bwd_dual_launch(const bf16_t* dout, const unsigned char* maskbits, const bf16_t* yA, const float* meanA,
                             const float* invstdA, const float* scaleA, float* dgammaA, float* dbetaA, bf16_t* dyA,
                             const bf16_t* yB, const float* meanB, const float* invstdB, const float* scaleB, float* dgammaB,
                             float* dbetaB, bf16_t* dyB, long long rows, int C, int accumulate, float* partA, double* chunksA,
                             float* cA, float* partB, double* chunksB, float* cB, hipStream_t s);
int icamd_bn_bwd_apply_launch(const float* part, int nrows, const bf16_t* g, const bf16_t* y, const float* mean,
                              const float* invstd, const float* scale, float* dgamma, float* dbeta, bf16_t* dy,
                              long long rows, int C, int accumulate, double* chunks, float* c1c2, hipStream_t s, int sums_are_gy = 0);
int icamd_bn_bwd_finalize_launch(const float* part, int nrows, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                 long long rows, int C, int accumulate, double* chunks, float* c1c2, hipStream_t s, int sums_are_gy);
int icamd_bn_bwd_reduce_launch(const bf16_t* g, const bf16_t* y, const float* mean, const float* invstd, float* part, long long rows,
                               int C, int* nblk_out, hipStream_t s);
int icamd_maxpool_fwd_launch(const bf16_t* x, bf16_t* out, unsigned char* idx, int N, int IH, int IW, int C, int OH, int OW,
                             hipStream_t s);
int icamd_bn_relu_maxpool_fwd_launch(const bf16_t* y, const float* scale, const float* shift, bf16_t* out, unsigned char* idx,
                                     int N, int IH, int IW, int C, int OH, int OW, hipStream_t s);
int icamd_maxpool_bwd_launch(const bf16_t* dout, const unsigned char* idx, bf16_t* dx, int N, int IH, int IW, int C, int OH,
                             int OW, hipStream_t s);
int icamd_avgpool_fwd_launch(const bf16_t* x, bf16_t* out, int N, int HW, int C, hipStream_t s);
int icamd_avgpool_bwd_launch(const bf16_t* dout, bf16_t* dx, int N, int HW, int C, hipStream_t s);
int icamd_pack_input_launch(const float* x, bf16_t* out, int B, int Cin, int H, int W, int mode, float lam, int yl, int yh,
                            int xl, int xh, hipStream_t s);
int icamd_pack_input_rgb4_launch(const float* x, bf16_t* out, int B, int Cin, int H, int W, int mode, float lam, int yl,
                                 int yh, int xl, int xh, hipStream_t s);
int icamd_softmax_xent_launch(const bf16_t* logits, int ld, int B, int C, const long long* y1, const long long* y2,
                              float lam, float smoothing, float gscale, float* loss_rows, int* pred, bf16_t* dlogits,
                              hipStream_t s);
int icamd_step_metrics_launch(const float* loss_rows, const int* pred, const long long* target, int B, int C,
                              float* loss_out, int* finite_out, double* acc_f64, int* counts, float* loss_log,
                              int log_slot, int log_stride, int respect_skip, hipStream_t s);
int icamd_grad_norm_launch(const float* g, long long n, float inv_scale, float max_norm, double* partial, float* out,
                           hipStream_t s);
int icamd_adamw_ema_launch(float* p, float* g, float* m, float* v, float* ema, bf16_t* shadow, long long n, float lr,
                           float wd, float beta1, float beta2, float eps, int step, float gscale, float ema_decay,
                           const float* clip, const int* finite_flag, int* skipped, int zero_grad, hipStream_t s);
int icamd_optim_ema_launch(int kind, float* p, float* g, float* m, float* v, float* ema, bf16_t* shadow, long long n,
                           float lr, float wd, float beta1, float beta2, float eps, int step, float gscale,
                           float ema_decay, const float* clip, const int* finite_flag, int* skipped, int zero_grad,
                           hipStream_t s);
int icamd_grad_guard_launch(float* g, long long n, const int* finite_flag, hipStream_t s);
int icamd_lerp_launch(float* dst, const float* src, long long n, float w, const int* finite_flag, hipStream_t s);
int icamd_bn_fold_launch(const float* w, const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                         int Cout, int K, bf16_t* w_folded, float* shift, hipStream_t s);
int icamd_f32_to_bf16_launch(const float* src, bf16_t* dst, long long n, hipStream_t s);
int icamd_filter_transpose_launch(const bf16_t* src_base, bf16_t* dst_base, const long long* descs, const int* jobs,
                                  int njobs, hipStream_t s);
int icamd_colsum_launch(const bf16_t* x, int rows, int ld, int cols, float* out, int accumulate, hipStream_t s);
int icamd_sum_partials_launch(const float* part, int nrows, int C, float* out1, float* out2, int accumulate, double* chunks,
                              float* c1c2, hipStream_t s);
int icamd_layernorm_fwd_launch(const bf16_t* x, const float* gamma, const float* beta, bf16_t* y, float* mean, float* rstd,
                               long long rows, int C, float eps, hipStream_t s);
int icamd_layernorm_bwd_blocks(long long rows);
int icamd_layernorm_bwd_launch(const bf16_t* dy, const bf16_t* x, const float* mean, const float* rstd, const float* gamma,
                               const bf16_t* addend, bf16_t* dx, float* part, long long rows, int C, hipStream_t s);
int icamd_vit_tokens_fwd_launch(const bf16_t* patches, const float* cls, const float* pos, bf16_t* tok, int B, int T, int C,
                                hipStream_t s);
int icamd_batch_sum_launch(const bf16_t* x, long long stride, int B, long long n, float* out, int accumulate, hipStream_t s);
int icamd_strided_rows_copy_launch(const bf16_t* src, long long sstride, bf16_t* dst, long long dstride, long long rows,
                                   long long C, hipStream_t s);
int icamd_gelu_fwd_launch(const bf16_t* z, bf16_t* a, long long numel, hipStream_t s);
int icamd_gelu_bwd_launch(const bf16_t* da, const bf16_t* z, bf16_t* dz, long long numel, hipStream_t s);
int icamd_colsum_blocks(long long rows);
int icamd_colsum_partial_launch(const bf16_t* x, float* part, long long rows, int ld, int cols, hipStream_t s);
int icamd_attention_fwd_launch(const bf16_t* qkv, bf16_t* out, float* lse, int B, int T, int H, float scale, hipStream_t s);
int icamd_attention_bwd_launch(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* lse, float* delta,
                               bf16_t* dqkv, int B, int T, int H, float scale, hipStream_t s);
int icamd_dwconv7_launch(const bf16_t* x, const bf16_t* w, const float* bias, const bf16_t* addend, bf16_t* y, int N, int H,
                         int W, int C, int flip, hipStream_t s);
int icamd_dwconv7_wgrad_blocks(int N, int H, int W, int C);
int icamd_dwconv7_wgrad_launch(const bf16_t* x, const bf16_t* dy, float* part, float* dw, float* dbias, int N, int H, int W, int C,
                               int accumulate, hipStream_t s);
bool icamd_dwconv7_wgrad_bias_supported_cxx(int N, int H, int W, int C);
int icamd_layerscale_fwd_launch(const bf16_t* z, const bf16_t* inp, const float* gamma, const float* keep, bf16_t* out,
                                long long rows, int C, long long rows_per_image, hipStream_t s);
int icamd_layerscale_bwd_blocks(long long rows);
int icamd_layerscale_fold_launch(const float* params, bf16_t* shadow, float* fold_bias, const long long* jobs, int njobs,
                                 int total_rows, hipStream_t s);
int icamd_rows_fix_launch(const float* keep, int n_images, void* dst1, const void* src1, long long bytes1, void* dst2,
                          long long bytes2, hipStream_t s);
int icamd_dropped_colsum_launch(const bf16_t* dy, const float* keep, int n_images, long long rows_per_image, int C, float* partial,
                                hipStream_t s);
int icamd_layerscale_param_grads_launch(const float* G, const float* w, const float* bias, const float* gamma,
                                        const float* colsum_all, const float* dropped, int n_images, float cb, int C, int K,
                                        float* dw, float* dbias, float* dgamma, int accumulate, hipStream_t s);
int icamd_layerscale_bwd_launch(const bf16_t* dout, const bf16_t* z, const float* gamma, const float* keep, bf16_t* dz,
                                float* part, long long rows, int C, long long rows_per_image, hipStream_t s);
int icamd_filter_transpose_tiled_launch(const bf16_t* src_base, bf16_t* dst_base, const long long* descs, const int* jobs,
                                        int njobs, hipStream_t s);


// ---- optional in-process kernel timing (HIP events on the launch stream), used by bench.py ------------------
#include <vector>
namespace {
enum ProfClass { PC_IGEMM_FWD = 0, PC_IGEMM_DGRAD, PC_WGRAD, PC_BN_FINALIZE, PC_BN_APPLY, PC_BN_BWD, PC_POOL, PC_PACK,
                 PC_LOSS, PC_OPTIM, PC_MISC, PC_ATTN_FWD, PC_ATTN_BWD, PC_LN_FWD, PC_LN_BWD, PC_ELEMWISE, PC_DWCONV, PC_FUSED_BWD, PC_FUSED_FWD, PC_COUNT };
// Besides the elapsed time every call books its ALGORITHMIC work (round 4): bytes = each operand tensor of the call
// read once and each result written once at the stored width (bf16 activations, fp32 parameters / gradients), two-pass
// kernels counted as the two passes they are; flops = 2 x multiply-adds of the contraction.  bench.py divides by the time.
struct ProfRec { int cls; hipEvent_t a, b; double bytes, flops; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;
hipEvent_t prof_event() {
  if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
  hipEvent_t e; (void)hipEventCreate(&e); return e;
}
struct ProfScope {
  int cls; hipStream_t s; hipEvent_t a; bool on; double bytes = 0.0, flops = 0.0;
  ProfScope(int c, void* stream) : cls(c), s((hipStream_t)stream), on(g_prof_on) {
    if (on) { a = prof_event(); (void)hipEventRecord(a, s); }
  }
  void work(double b, double f = 0.0) { bytes = b; flops = f; }
  ~ProfScope() {
    if (on) { hipEvent_t b = prof_event(); (void)hipEventRecord(b, s); g_prof_recs.push_back({cls, a, b, bytes, flops}); }
  }
};
// operand sizes of a convolution call: input / output activations (bf16), filter elements, multiply-adds x 2
struct ConvWork { double in, out, w, flops; };
static ConvWork conv_work(const icamd_conv_desc* d) {
  ConvWork c = {0, 0, 0, 0};
  if (d == nullptr) return c;
  c.in = 2.0 * d->N * d->IH * d->IW * d->Cin;
  c.out = 2.0 * d->N * d->OH * d->OW * d->Cout;
  c.w = (double)d->Cout * d->KH * d->KW * d->Cin;
  c.flops = 2.0 * d->N * d->OH * d->OW * c.w;
  return c;
}
}  // namespace

namespace {

bool conv_desc_ok(const icamd_conv_desc* d) {
  if (d == nullptr) return false;
  if (d->N <= 0 || d->IH <= 0 || d->IW <= 0 || d->Cin <= 0 || d->OH <= 0 || d->OW <= 0 || d->Cout <= 0) return false;
  if (d->KH <= 0 || d->KW <= 0 || d->stride <= 0 || d->pad < 0) return false;
  if (d->Cin % 64 == 0 && d->KH * d->KW > ICAMD_MAX_TAPS) return false;   // the general path derives taps arithmetically
  if (d->KH * d->KW > 1024) return false;
  if ((d->IH + 2 * d->pad - d->KH) / d->stride + 1 != d->OH) return false;
  if ((d->IW + 2 * d->pad - d->KW) / d->stride + 1 != d->OW) return false;
  return true;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

extern "C" {

int icamd_abi_version(void) { return 5; }

int icamd_prof_enable(int on) { g_prof_on = on != 0; return ICAMD_OK; }
int icamd_prof_classes(void) { return PC_COUNT; }
// Synchronises the recorded events, adds per-class elapsed ms / call counts into the arrays, clears the log.
int icamd_prof_collect(double* ms, long long* calls, double* bytes, double* flops, int n) {
  if (ms == nullptr || calls == nullptr || n < PC_COUNT) return ICAMD_ERR_BAD_ARG;
  for (auto& r : g_prof_recs) {
    (void)hipEventSynchronize(r.b);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, r.a, r.b);
    ms[r.cls] += (double)t;
    calls[r.cls] += 1;
    if (bytes != nullptr) bytes[r.cls] += r.bytes;
    if (flops != nullptr) flops[r.cls] += r.flops;
    g_prof_pool.push_back(r.a);
    g_prof_pool.push_back(r.b);
  }
  g_prof_recs.clear();
  return ICAMD_OK;
}

int icamd_conv2d_stats_rows(const icamd_conv_desc* d) {
  if (!conv_desc_ok(d)) return 0;
  const long long M = (long long)d->N * d->OH * d->OW;
  return (int)((M + 127) / 128);
}

static int conv_fwd_impl(const icamd_conv_desc* d, const void* x, const void* w, void* y, const float* bias,
                         const void* addend, float* stats, int relu, void* stream, void* gelu_out = nullptr,
                         int gelu_inplace = 0) {
  if (!conv_desc_ok(d) || x == nullptr || w == nullptr || y == nullptr) return ICAMD_ERR_BAD_ARG;
  if ((long long)d->N * d->OH * d->OW >= (1ll << 31)) return ICAMD_ERR_UNSUPPORTED;
  if (d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && addend == nullptr && gelu_out == nullptr &&
      !gelu_inplace && icamd_halo3x3_wanted(d->N, d->IH, d->IW, d->Cin, d->Cout)) {
    Halo3x3Params h;
    memset(&h, 0, sizeof(h));
    h.in = (const bf16_t*)x; h.wt = (const bf16_t*)w; h.out = (bf16_t*)y; h.bias = bias; h.stats = stats; h.relu = relu;
    h.N = d->N; h.H = d->IH; h.W = d->IW; h.C = d->Cin; h.Cout = d->Cout;
    return icamd_halo3x3_launch(h, (hipStream_t)stream);
  }
  // round 5: a 1x1 / stride-2 convolution (ResNet's projection shortcuts) is the same pointwise problem on the rows (n, 2 oh, 2 ow):
  // the register-resident kernel gathers them in its LDS-DMA staging (ICAMD_PW_S2=0: back on conv_igemm)
  static const bool pw_s2 = [] { const char* e = getenv("ICAMD_PW_S2"); return !(e && atoi(e) == 0); }();
  const bool pw_gather = d->KH == 1 && d->KW == 1 && d->stride == 2 && d->pad == 0 && pw_s2 && addend == nullptr;
  if (d->KH == 1 && d->KW == 1 && (d->stride == 1 || pw_gather) && d->pad == 0 && bias == nullptr && addend == nullptr && !relu &&
      gelu_out == nullptr && !gelu_inplace && icamd_pw_resident_wanted((long long)d->N * d->OH * d->OW, d->Cout, d->Cin)) {
    PwResidentParams g;
    memset(&g, 0, sizeof(g));
    g.A = (const bf16_t*)x; g.B = (const bf16_t*)w; g.out = (bf16_t*)y; g.stats = stats;
    g.M = d->N * d->OH * d->OW; g.N = d->Cout; g.K = d->Cin;
    if (pw_gather) { g.gat_oh = d->OH; g.gat_ow = d->OW; g.gat_ih = d->IH; g.gat_iw = d->IW; }
    return icamd_pw_resident_launch(g, (hipStream_t)stream);
  }
  // evaluate()'s BatchNorm-folded forward (bias = the folded shift, optional residual addend, ReLU): the same register-resident
  // kernel with the inference epilogue (round 4; these launches ran on conv_igemm's single-stage tiles before)
  if (d->KH == 1 && d->KW == 1 && (d->stride == 1 || pw_gather) && d->pad == 0 && (bias != nullptr || relu) && stats == nullptr &&
      gelu_out == nullptr && !gelu_inplace && icamd_pw_resident_epi_wanted() &&
      icamd_pw_resident_wanted((long long)d->N * d->OH * d->OW, d->Cout, d->Cin, addend != nullptr)) {
    PwResidentParams g;
    memset(&g, 0, sizeof(g));
    g.A = (const bf16_t*)x; g.B = (const bf16_t*)w; g.out = (bf16_t*)y; g.bias = bias; g.relu = relu;
    g.addend = (const bf16_t*)addend;
    g.M = d->N * d->OH * d->OW; g.N = d->Cout; g.K = d->Cin;
    if (pw_gather) { g.gat_oh = d->OH; g.gat_ow = d->OW; g.gat_ih = d->IH; g.gat_iw = d->IW; }
    return icamd_pw_resident_launch(g, (hipStream_t)stream);
  }
  if (d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && addend == nullptr && !relu && stats == nullptr &&
      icamd_pw_resident_ext_wanted((long long)d->N * d->OH * d->OW, d->Cout, d->Cin)) {
    PwResidentParams g;
    memset(&g, 0, sizeof(g));
    g.A = (const bf16_t*)x; g.B = (const bf16_t*)w; g.out = (bf16_t*)y; g.bias = bias;
    g.gelu_out = (bf16_t*)gelu_out; g.gelu_inplace = gelu_inplace;
    g.M = d->N * d->OH * d->OW; g.N = d->Cout; g.K = d->Cin;
    return icamd_pw_resident_ext_launch(g, (hipStream_t)stream);
  }
  if (d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 &&
      icamd_gemm_nt_wanted((long long)d->N * d->OH * d->OW, d->Cout, d->Cin)) {
    GemmNtParams g;
    memset(&g, 0, sizeof(g));
    g.A = (const bf16_t*)x; g.B = (const bf16_t*)w; g.out = (bf16_t*)y; g.addend = (const bf16_t*)addend; g.bias = bias;
    g.stats = stats;
    g.M = d->N * d->OH * d->OW; g.N = d->Cout; g.K = d->Cin; g.relu = relu; g.gelu_out = (bf16_t*)gelu_out; g.gelu_inplace = gelu_inplace;
    return icamd_gemm_nt_launch(g, (hipStream_t)stream);
  }
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.in = (const bf16_t*)x; p.wt = (const bf16_t*)w; p.out = (bf16_t*)y;
  p.addend = (const bf16_t*)addend; p.bias = bias; p.stats = stats; p.relu = relu; p.gelu_out = (bf16_t*)gelu_out; p.gelu_inplace = gelu_inplace;
  p.N = d->N; p.IH = d->IH; p.IW = d->IW; p.Cin = d->Cin;
  p.OH = d->OH; p.OW = d->OW; p.Cout = d->Cout;
  p.P = d->OH; p.Q = d->OW; p.M = d->N * d->OH * d->OW;
  p.ostr = 1; p.ooff_h = 0; p.ooff_w = 0; p.istr = d->stride;
  p.ntaps = d->KH * d->KW; p.Ktot = p.ntaps * d->Cin;
  p.KW = d->KW; p.pad = d->pad; p.tap_sign = 1; p.regular_taps = 1;
  if (p.ntaps <= ICAMD_MAX_TAPS)
    for (int r = 0; r < d->KH; ++r)
      for (int s = 0; s < d->KW; ++s) {
        const int t = r * d->KW + s;
        p.dh[t] = (short)(r - d->pad); p.dw[t] = (short)(s - d->pad); p.wtap[t] = (short)t;
      }
  return icamd_igemm_launch(p, (hipStream_t)stream);
}

int icamd_conv2d_fwd(const icamd_conv_desc* d, const void* x, const void* w, void* y, const float* bias,
                     const void* addend, float* stats, void* stream) {
  ProfScope _prof(PC_IGEMM_FWD, stream);
  const ConvWork cw = conv_work(d); _prof.work(cw.in + cw.out + 2 * cw.w + (addend ? cw.out : 0), cw.flops);
  return conv_fwd_impl(d, x, w, y, bias, addend, stats, 0, stream);
}

int icamd_conv2d_fwd_act(const icamd_conv_desc* d, const void* x, const void* w, void* y, const float* bias,
                         const void* addend, int relu, void* stream) {
  ProfScope _prof(PC_IGEMM_FWD, stream);
  const ConvWork cw = conv_work(d); _prof.work(cw.in + cw.out + 2 * cw.w + (addend ? cw.out : 0), cw.flops);
  return conv_fwd_impl(d, x, w, y, bias, addend, nullptr, relu ? 1 : 0, stream);
}

int icamd_conv2d_fwd_gelu(const icamd_conv_desc* d, const void* x, const void* w, void* z, void* a, const float* bias,
                          void* stream) {
  ProfScope _prof(PC_IGEMM_FWD, stream);
  const ConvWork cw = conv_work(d); _prof.work(cw.in + (z ? cw.out : 0) + cw.out + 2 * cw.w, cw.flops);
  if (a == nullptr) return ICAMD_ERR_BAD_ARG;
  if (z == nullptr) return conv_fwd_impl(d, x, w, a, bias, nullptr, nullptr, 0, stream, nullptr, /*gelu_inplace=*/1);
  return conv_fwd_impl(d, x, w, z, bias, nullptr, nullptr, 0, stream, a);
}

int icamd_bn_fold_filters(const float* w, const float* gamma, const float* beta, const float* running_mean,
                          const float* running_var, float eps, int Cout, int K, void* w_folded, float* shift,
                          void* stream) {
  ProfScope _prof(PC_BN_FINALIZE, stream);
  _prof.work(6.0 * Cout * K);
  if (w == nullptr || gamma == nullptr || beta == nullptr || running_mean == nullptr || running_var == nullptr ||
      w_folded == nullptr || shift == nullptr || Cout <= 0 || K <= 0)
    return ICAMD_ERR_BAD_ARG;
  return icamd_bn_fold_launch(w, gamma, beta, running_mean, running_var, eps, Cout, K, (bf16_t*)w_folded, shift,
                              (hipStream_t)stream);
}

static int dgrad_impl(const icamd_conv_desc* d, const void* dy, const void* w_t, void* dx, const void* addend,
                      const uint8_t* addend_bits, const icamd_bn_bwd_fuse* f, void* stream, const void* gelu_z = nullptr,
                      int addend_sub2 = 0) {
  if (!conv_desc_ok(d) || dy == nullptr || w_t == nullptr || dx == nullptr) return ICAMD_ERR_BAD_ARG;
  if (d->Cout % 8 != 0 || d->Cin % 8 != 0) return ICAMD_ERR_UNSUPPORTED;
  if ((long long)d->N * d->IH * d->IW >= (1ll << 31)) return ICAMD_ERR_UNSUPPORTED;
  const int st = d->stride;
  // (ICAMD_PW_RESIDENT=5: the residual / even-grid addend launches stay on conv_igemm -- the A/B switch for the
  // LDS-DMA-staged addend of conv1x1_resident.hip)
  static const bool pw_addend = [] { const char* e = getenv("ICAMD_PW_RESIDENT"); return !(e && atoi(e) == 5); }();
  if (d->KH == 1 && d->KW == 1 && st == 1 && d->pad == 0 && f == nullptr && gelu_z == nullptr &&
      (addend == nullptr || pw_addend) && !(addend_bits != nullptr && addend_sub2) &&
      icamd_pw_resident_wanted((long long)d->N * d->IH * d->IW, d->Cin, d->Cout, addend != nullptr)) {
    PwResidentParams g;
    memset(&g, 0, sizeof(g));
    g.A = (const bf16_t*)dy; g.B = (const bf16_t*)w_t; g.out = (bf16_t*)dx; g.addend = (const bf16_t*)addend;
    g.addend_bits = addend_bits;
    g.M = d->N * d->IH * d->IW; g.N = d->Cin; g.K = d->Cout;
    if (addend_sub2) { g.sub2_h = d->IH; g.sub2_w = d->IW; }
    return icamd_pw_resident_launch(g, (hipStream_t)stream);
  }
  if (d->KH == 1 && d->KW == 1 && st == 1 && d->pad == 0 && f == nullptr && addend == nullptr &&
      icamd_pw_resident_ext_wanted((long long)d->N * d->IH * d->IW, d->Cin, d->Cout)) {
    PwResidentParams g;
    memset(&g, 0, sizeof(g));
    g.A = (const bf16_t*)dy; g.B = (const bf16_t*)w_t; g.out = (bf16_t*)dx; g.gelu_z = (const bf16_t*)gelu_z;
    g.M = d->N * d->IH * d->IW; g.N = d->Cin; g.K = d->Cout;
    return icamd_pw_resident_ext_launch(g, (hipStream_t)stream);
  }
  if (d->KH == 1 && d->KW == 1 && st == 1 && d->pad == 0 && f == nullptr && !(addend_bits != nullptr && addend_sub2) &&
      icamd_gemm_nt_wanted((long long)d->N * d->IH * d->IW, d->Cin, d->Cout)) {
    GemmNtParams g;
    memset(&g, 0, sizeof(g));
    g.A = (const bf16_t*)dy; g.B = (const bf16_t*)w_t; g.out = (bf16_t*)dx; g.addend = (const bf16_t*)addend;
    g.addend_bits = addend_bits;
    g.M = d->N * d->IH * d->IW; g.N = d->Cin; g.K = d->Cout; g.gelu_z = (const bf16_t*)gelu_z;
    if (addend_sub2) { g.sub2_h = d->IH; g.sub2_w = d->IW; }
    return icamd_gemm_nt_launch(g, (hipStream_t)stream);
  }
  if (d->KH == 3 && d->KW == 3 && st == 1 && d->pad == 1 && addend == nullptr && addend_bits == nullptr && f == nullptr &&
      gelu_z == nullptr && icamd_halo3x3_wanted(d->N, d->IH, d->IW, d->Cout, d->Cin)) {
    // dX = conv(dY, transposed filter, mirrored taps): same spatial size in and out for stride 1 / pad 1
    Halo3x3Params h;
    memset(&h, 0, sizeof(h));
    h.in = (const bf16_t*)dy; h.wt = (const bf16_t*)w_t; h.out = (bf16_t*)dx; h.flip = 1;
    h.N = d->N; h.H = d->IH; h.W = d->IW; h.C = d->Cout; h.Cout = d->Cin;
    return icamd_halo3x3_launch(h, (hipStream_t)stream);
  }
  float* partials = f ? f->partials : nullptr;
  // one launch per output parity class (ph, pw): pixels h = st*p + ph, w = st*q + pw receive only the taps
  // r with (ph + pad - r) % st == 0, read at dy row p + (ph + pad - r)/st
  for (int ph = 0; ph < st; ++ph)
    for (int pw = 0; pw < st; ++pw) {
      const int P = (d->IH - ph + st - 1) / st, Q = (d->IW - pw + st - 1) / st;
      if (P <= 0 || Q <= 0) continue;
      IgemmParams p;
      memset(&p, 0, sizeof(p));
      p.in = (const bf16_t*)dy; p.wt = (const bf16_t*)w_t; p.out = (bf16_t*)dx;
      p.addend = (const bf16_t*)addend;
      p.addend_bits = addend_bits;
      p.addend_sub2 = addend_sub2;
      p.gelu_z = (const bf16_t*)gelu_z;
      p.N = d->N; p.IH = d->OH; p.IW = d->OW; p.Cin = d->Cout;
      p.OH = d->IH; p.OW = d->IW; p.Cout = d->Cin;
      p.P = P; p.Q = Q; p.M = d->N * P * Q;
      p.ostr = st; p.ooff_h = ph; p.ooff_w = pw; p.istr = 1;
      p.Ktot = d->KH * d->KW * d->Cout;
      p.KW = d->KW; p.pad = d->pad; p.tap_sign = -1; p.regular_taps = (st == 1) ? 1 : 0;
      if (f != nullptr) {
        p.bnb_y = (const bf16_t*)f->y; p.bnb_mask = (const bf16_t*)f->mask_src;
        p.bnb_mean = f->mean; p.bnb_invstd = f->invstd; p.bnb_scale = f->scale; p.bnb_shift = f->shift;
        p.bnb_relu = f->relu; p.stats = partials;
        partials += (size_t)((p.M + 127) / 128) * 2 * d->Cin;   // each parity class writes its own partial rows
      }
      int nt = 0;
      for (int r = 0; r < d->KH; ++r) {
        const int eh = ph + d->pad - r;
        if (((eh % st) + st) % st != 0) continue;
        for (int s = 0; s < d->KW; ++s) {
          const int ew = pw + d->pad - s;
          if (((ew % st) + st) % st != 0) continue;
          // exact division (eh, ew are multiples of st, possibly negative)
          if (nt < ICAMD_MAX_TAPS) { p.dh[nt] = (short)(eh / st); p.dw[nt] = (short)(ew / st); p.wtap[nt] = (short)(r * d->KW + s); }
          ++nt;
        }
      }
      p.ntaps = nt;
      const int rc = icamd_igemm_launch(p, (hipStream_t)stream);
      if (rc) return rc;
    }
  return ICAMD_OK;
}

int icamd_conv2d_dgrad(const icamd_conv_desc* d, const void* dy, const void* w_t, void* dx, const void* addend,
                       const uint8_t* addend_maskbits, void* stream) {
  ProfScope _prof(PC_IGEMM_DGRAD, stream);
  const ConvWork cw = conv_work(d); _prof.work(cw.out + cw.in + 2 * cw.w + (addend ? cw.in : 0) + (addend_maskbits ? cw.in / 16 : 0), cw.flops);
  if (addend_maskbits != nullptr && (addend == nullptr || d == nullptr || d->Cin % 64 != 0)) return ICAMD_ERR_BAD_ARG;
  return dgrad_impl(d, dy, w_t, dx, addend, addend_maskbits, nullptr, stream);
}

int icamd_conv2d_dgrad_sub2(const icamd_conv_desc* d, const void* dy, const void* w_t, void* dx, const void* addend_sub2,
                            void* stream) {
  ProfScope _prof(PC_IGEMM_DGRAD, stream);
  const ConvWork cw = conv_work(d); _prof.work(cw.out + cw.in + 2 * cw.w + cw.in / 4, cw.flops);
  if (addend_sub2 == nullptr) return ICAMD_ERR_BAD_ARG;
  return dgrad_impl(d, dy, w_t, dx, addend_sub2, nullptr, nullptr, stream, nullptr, 1);
}

int icamd_conv2d_dgrad_gelu(const icamd_conv_desc* d, const void* dy, const void* w_t, const void* z, void* dz,
                            void* stream) {
  ProfScope _prof(PC_IGEMM_DGRAD, stream);
  const ConvWork cw = conv_work(d); _prof.work(cw.out + 2 * cw.in + 2 * cw.w, cw.flops);
  if (z == nullptr) return ICAMD_ERR_BAD_ARG;
  return dgrad_impl(d, dy, w_t, dz, nullptr, nullptr, nullptr, stream, z);
}

int icamd_conv2d_dgrad_stats_rows(const icamd_conv_desc* d) {
  if (!conv_desc_ok(d)) return 0;
  int rows = 0;
  for (int ph = 0; ph < d->stride; ++ph)
    for (int pw = 0; pw < d->stride; ++pw) {
      const long long P = (d->IH - ph + d->stride - 1) / d->stride, Q = (d->IW - pw + d->stride - 1) / d->stride;
      if (P > 0 && Q > 0) rows += (int)((d->N * P * Q + 127) / 128);
    }
  return rows;
}

int icamd_conv2d_dgrad_bnbwd(const icamd_conv_desc* d, const void* dy, const void* w_t, void* g, const void* addend,
                             const icamd_bn_bwd_fuse* f, void* stream) {
  ProfScope _prof(PC_IGEMM_DGRAD, stream);
  const ConvWork cw = conv_work(d); _prof.work(cw.out + 2 * cw.in + 2 * cw.w + (addend ? cw.in : 0), cw.flops);
  if (f == nullptr || f->y == nullptr || f->mean == nullptr || f->invstd == nullptr || f->partials == nullptr)
    return ICAMD_ERR_BAD_ARG;
  return dgrad_impl(d, dy, w_t, g, addend, nullptr, f, stream);
}

int icamd_conv2d_dgrad_bnred_supported(const icamd_conv_desc* d) {
  if (!conv_desc_ok(d) || d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad != 0) return 0;
  return icamd_pw_resident_bnred_wanted((long long)d->N * d->IH * d->IW, d->Cin, d->Cout) ? 1 : 0;
}

int icamd_conv2d_dgrad_bnred(const icamd_conv_desc* d, const void* dy, const void* w_t, void* g, const void* addend,
                             const uint8_t* addend_bits, int addend_sub2, const void* bn_y, const uint8_t* bn_bits,
                             float* partials, void* stream) {
  ProfScope _prof(PC_IGEMM_DGRAD, stream);
  const ConvWork cw = conv_work(d); _prof.work(cw.out + 3 * cw.in + 2 * cw.w + cw.in / 16 + (addend_bits ? cw.in / 16 : 0), cw.flops);
  if (!conv_desc_ok(d) || dy == nullptr || w_t == nullptr || g == nullptr || addend == nullptr || bn_y == nullptr ||
      bn_bits == nullptr || partials == nullptr || (addend_sub2 && addend_bits != nullptr))
    return ICAMD_ERR_BAD_ARG;
  if (!icamd_conv2d_dgrad_bnred_supported(d)) return ICAMD_ERR_UNSUPPORTED;
  PwResidentParams p;
  memset(&p, 0, sizeof(p));
  p.A = (const bf16_t*)dy; p.B = (const bf16_t*)w_t; p.out = (bf16_t*)g; p.addend = (const bf16_t*)addend;
  p.addend_bits = addend_bits;
  p.bn_y = (const bf16_t*)bn_y; p.bn_bits = bn_bits; p.bn_part = partials;
  p.M = d->N * d->IH * d->IW; p.N = d->Cin; p.K = d->Cout;
  if (addend_sub2) { p.sub2_h = d->IH; p.sub2_w = d->IW; }
  return icamd_pw_resident_bnred_launch(p, (hipStream_t)stream);
}

size_t icamd_conv2d_wgrad_workspace_bytes(const icamd_conv_desc* d) {
  if (!conv_desc_ok(d)) return 0;
  int S = 1, rows = 0;
  const long long M = (long long)d->N * d->OH * d->OW;
  if (M >= (1ll << 30)) return 0;
  const int Ktot = d->KH * d->KW * d->Cin;
  icamd_wgrad_plan((int)M, d->Cout, Ktot, &S, &rows);
  if (icamd_wgrad_halo_wanted(d->KH, d->KW, d->stride, d->pad, d->OH, d->OW, d->Cin, d->Cout, M)) {
    int S2 = 1;   // the halo kernel has its own split; the bias-gradient form of the same layer stays on the general one
    icamd_wgrad_halo_plan((int)M, d->Cin, d->Cout, &S2, &rows);
    if (S2 > S) S = S2;
  }
  return (size_t)S * d->Cout * ((size_t)Ktot + 1) * sizeof(float);   // filter slabs + one bias row per split
}

static int wgrad_impl(const icamd_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias, int accumulate,
                      void* workspace, size_t workspace_bytes, void* stream);

int icamd_conv2d_wgrad(const icamd_conv_desc* d, const void* x, const void* dy, float* dw, int accumulate,
                       void* workspace, size_t workspace_bytes, void* stream) {
  ProfScope _prof(PC_WGRAD, stream);
  const ConvWork cw = conv_work(d); _prof.work(cw.in + cw.out + 4 * cw.w, cw.flops);
  return wgrad_impl(d, x, dy, dw, nullptr, accumulate, workspace, workspace_bytes, stream);
}

int icamd_conv2d_wgrad_bias(const icamd_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias,
                            int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  ProfScope _prof(PC_WGRAD, stream);
  const ConvWork cw = conv_work(d); _prof.work(cw.in + cw.out + 4 * cw.w, cw.flops);
  if (dbias == nullptr) return ICAMD_ERR_BAD_ARG;
  return wgrad_impl(d, x, dy, dw, dbias, accumulate, workspace, workspace_bytes, stream);
}

static int wgrad_impl(const icamd_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias, int accumulate,
                      void* workspace, size_t workspace_bytes, void* stream) {
  if (!conv_desc_ok(d) || x == nullptr || dy == nullptr || dw == nullptr || workspace == nullptr) return ICAMD_ERR_BAD_ARG;
  const size_t need = icamd_conv2d_wgrad_workspace_bytes(d);
  if (need == 0 || workspace_bytes < need) return ICAMD_ERR_WORKSPACE;
  WgradParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.slab = (float*)workspace;
  p.N = d->N; p.IH = d->IH; p.IW = d->IW; p.Cin = d->Cin; p.OH = d->OH; p.OW = d->OW; p.Cout = d->Cout;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
  p.M = d->N * d->OH * d->OW; p.Ktot = d->KH * d->KW * d->Cin;
  int rc;
  if (dbias == nullptr && icamd_wgrad_halo_wanted(d->KH, d->KW, d->stride, d->pad, d->OH, d->OW, d->Cin, d->Cout, p.M)) {
    icamd_wgrad_halo_plan(p.M, p.Cin, p.Cout, &p.S, &p.rows_per_split);
    rc = icamd_wgrad_halo_launch(p, (hipStream_t)stream);
  } else {
    icamd_wgrad_plan(p.M, p.Cout, p.Ktot, &p.S, &p.rows_per_split);
    if (dbias != nullptr) p.bias_slab = p.slab + (size_t)p.S * p.Cout * p.Ktot;
    rc = icamd_wgrad_launch(p, (hipStream_t)stream);
  }
  if (rc) return rc;
  rc = icamd_slab_reduce_launch(p.slab, dw, (long long)p.Cout * p.Ktot, p.S, accumulate, (hipStream_t)stream);
  if (rc || dbias == nullptr) return rc;
  return icamd_slab_reduce_launch(p.bias_slab, dbias, (long long)p.Cout, p.S, accumulate, (hipStream_t)stream);
}

int icamd_filter_transpose(const void* src_base, void* dst_base, const int64_t* descs, const int32_t* jobs, int njobs,
                           void* stream) {
  ProfScope _prof(PC_OPTIM, stream);
  _prof.work(4.0 * 4096 * njobs);
  if (src_base == nullptr || dst_base == nullptr || descs == nullptr || jobs == nullptr || njobs < 0) return ICAMD_ERR_BAD_ARG;
  return icamd_filter_transpose_launch((const bf16_t*)src_base, (bf16_t*)dst_base, (const long long*)descs, jobs, njobs,
                                       (hipStream_t)stream);
}

int icamd_filter_transpose_tiled(const void* src_base, void* dst_base, const int64_t* descs, const int32_t* jobs, int njobs,
                                 void* stream) {
  ProfScope _prof(PC_OPTIM, stream);
  _prof.work(4.0 * 4096 * njobs);
  if (src_base == nullptr || dst_base == nullptr || descs == nullptr || jobs == nullptr || njobs < 0) return ICAMD_ERR_BAD_ARG;
  return icamd_filter_transpose_tiled_launch((const bf16_t*)src_base, (bf16_t*)dst_base, (const long long*)descs, jobs, njobs,
                                             (hipStream_t)stream);
}

// BN workspace: [64 chunks][2][C] doubles
// [64 chunks][2][C] doubles + ceil(C/64) arrival counters (uint32, must be zero before first use; self-resetting)
static size_t bn_chunk_bytes(int C) { return 256 + align_up((size_t)64 * 2 * C * sizeof(double), 256); }
size_t icamd_bn_workspace_bytes(int C) { return C > 0 ? bn_chunk_bytes(C) : 0; }

int icamd_bn_train_finalize(const float* partials, int nrows, int C, double count, const float* gamma,
                            const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                            float* mean, float* invstd, float* scale, float* shift, void* workspace, void* stream) {
  ProfScope _prof(PC_BN_FINALIZE, stream);
  _prof.work(8.0 * nrows * C);
  if (partials == nullptr || nrows <= 0 || C <= 0 || count <= 0 || gamma == nullptr || beta == nullptr ||
      mean == nullptr || invstd == nullptr || scale == nullptr || shift == nullptr || workspace == nullptr || C > 4096)
    return ICAMD_ERR_BAD_ARG;
  return icamd_bn_finalize_launch(partials, nrows, C, count, gamma, beta, running_mean, running_var, momentum, eps, mean,
                                  invstd, scale, shift, (double*)((char*)workspace + 256), (hipStream_t)stream);
}

int icamd_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, float eps, float* scale, float* shift, void* stream) {
  ProfScope _prof(PC_BN_FINALIZE, stream);
  _prof.work(24.0 * C);
  if (C <= 0 || gamma == nullptr || beta == nullptr || running_mean == nullptr || running_var == nullptr ||
      scale == nullptr || shift == nullptr)
    return ICAMD_ERR_BAD_ARG;
  return icamd_bn_eval_coeffs_launch(C, gamma, beta, running_mean, running_var, eps, scale, shift, (hipStream_t)stream);
}

int icamd_bn_apply(const void* y, const float* scale, const float* shift, const void* residual, void* out,
                   uint8_t* maskbits, long long numel, int C, int relu, void* stream) {
  ProfScope _prof(PC_BN_APPLY, stream);
  _prof.work((double)numel * (4 + (residual ? 2 : 0)) + (maskbits ? numel / 8.0 : 0));
  if (y == nullptr || scale == nullptr || shift == nullptr || out == nullptr || numel <= 0 || C <= 0) return ICAMD_ERR_BAD_ARG;
  return icamd_bn_apply_launch((const bf16_t*)y, scale, shift, (const bf16_t*)residual, (bf16_t*)out, maskbits, numel, C,
                               relu, (hipStream_t)stream);
}

int icamd_bn_apply_res_bn(const void* y, const float* scale, const float* shift, const void* res_y, const float* res_scale,
                          const float* res_shift, void* out, uint8_t* maskbits, long long numel, int C, int relu,
                          void* stream) {
  ProfScope _prof(PC_BN_APPLY, stream);
  _prof.work((double)numel * 6 + (maskbits ? numel / 8.0 : 0));
  if (y == nullptr || scale == nullptr || shift == nullptr || res_y == nullptr || res_scale == nullptr ||
      res_shift == nullptr || out == nullptr || numel <= 0 || C <= 0)
    return ICAMD_ERR_BAD_ARG;
  return icamd_bn_apply_launch((const bf16_t*)y, scale, shift, (const bf16_t*)res_y, (bf16_t*)out, maskbits, numel, C, relu,
                               (hipStream_t)stream, res_scale, res_shift);
}

// bwd workspace: partial rows [nblk][2][C] floats | chunks [64][2][C] doubles | c1,c2 [2][C] floats
size_t icamd_bn_bwd_workspace_bytes(long long rows, int C) {
  if (rows <= 0 || C <= 0) return 0;
  const int rpb = icamd_bn_bwd_rows_per_block(rows, C);
  const long long nblk = (rows + rpb - 1) / rpb;
  return align_up((size_t)nblk * 2 * C * sizeof(float), 256) + bn_chunk_bytes(C) + align_up((size_t)2 * C * sizeof(float), 256);
}

int icamd_bn_bwd(const void* dout, const void* act, const void* y, const float* mean, const float* invstd,
                 const float* scale, const float* shift, float* dgamma, float* dbeta, void* dy, void* gout,
                 const uint8_t* maskbits, long long rows, int C, int relu, int accumulate, void* workspace,
                 size_t workspace_bytes, void* stream) {
  ProfScope _prof(PC_BN_BWD, stream);
  _prof.work((double)rows * C * (2 * (4 + (act ? 2 : 0)) + 2 + (gout ? 2 : 0)) + (maskbits ? rows * C / 4.0 : 0));
  if (dout == nullptr || y == nullptr || mean == nullptr || invstd == nullptr || scale == nullptr || shift == nullptr ||
      dgamma == nullptr || dbeta == nullptr || dy == nullptr || workspace == nullptr || rows <= 0 || C <= 0)
    return ICAMD_ERR_BAD_ARG;
  const size_t need = icamd_bn_bwd_workspace_bytes(rows, C);
  if (workspace_bytes < need) return ICAMD_ERR_WORKSPACE;
  const int rpb = icamd_bn_bwd_rows_per_block(rows, C);
  const long long nblk = (rows + rpb - 1) / rpb;
  if (C > 4096) return ICAMD_ERR_UNSUPPORTED;
  char* ws = (char*)workspace;
  double* chunks = (double*)(ws + 256);     // arrival counters live in the first 256 B
  ws += bn_chunk_bytes(C);
  float* part = (float*)ws;
  ws += align_up((size_t)nblk * 2 * C * sizeof(float), 256);
  float* c1c2 = (float*)ws;
  return icamd_bn_bwd_launch((const bf16_t*)dout, (const bf16_t*)act, (const bf16_t*)y, mean, invstd, scale, shift, dgamma,
                             dbeta, (bf16_t*)dy, (bf16_t*)gout, maskbits, rows, C, relu, accumulate, part, chunks, c1c2,
                             (hipStream_t)stream);
}

int icamd_bn_bwd_maxpool3x3s2(const void* dout_pooled, const uint8_t* idx, const void* y, const float* mean,
                              const float* invstd, const float* scale, const float* shift, float* dgamma, float* dbeta,
                              void* dy, int N, int IH, int IW, int C, int accumulate, void* workspace, size_t workspace_bytes,
                              void* stream) {
  ProfScope _prof(PC_BN_BWD, stream);
  _prof.work((double)N * IH * IW * C * (2 * 2 + 2) + 2.0 * N * IH * IW * C / 4 * 3);
  if (dout_pooled == nullptr || idx == nullptr || y == nullptr || mean == nullptr || invstd == nullptr || scale == nullptr ||
      shift == nullptr || dgamma == nullptr || dbeta == nullptr || dy == nullptr || workspace == nullptr || N <= 0 || IH <= 0 ||
      IW <= 0 || C <= 0 || C % 8 != 0)
    return ICAMD_ERR_BAD_ARG;
  const long long rows = (long long)N * IH * IW;
  const size_t need = icamd_bn_bwd_workspace_bytes(rows, C);
  if (workspace_bytes < need) return ICAMD_ERR_WORKSPACE;
  if (C > 4096) return ICAMD_ERR_UNSUPPORTED;
  const int rpb = icamd_bn_bwd_rows_per_block(rows, C);
  const long long nblk = (rows + rpb - 1) / rpb;
  char* ws = (char*)workspace;
  double* chunks = (double*)(ws + 256);
  ws += bn_chunk_bytes(C);
  float* part = (float*)ws;
  ws += align_up((size_t)nblk * 2 * C * sizeof(float), 256);
  float* c1c2 = (float*)ws;
  return icamd_bn_bwd_launch((const bf16_t*)dout_pooled, nullptr, (const bf16_t*)y, mean, invstd, scale, shift, dgamma, dbeta,
                             (bf16_t*)dy, nullptr, nullptr, rows, C, /*relu=*/1, accumulate, part, chunks, c1c2,
                             (hipStream_t)stream, idx, IH, IW);
}

int icamd_bn_bwd_dual(const void* dout, const uint8_t* maskbits, const void* yA, const float* meanA, const float* invstdA,
                      const float* scaleA, float* dgammaA, float* dbetaA, void* dyA, const void* yB, const float* meanB,
                      const float* invstdB, const float* scaleB, float* dgammaB, float* dbetaB, void* dyB, long long rows, int C,
                      int accumulate, void* workspaceA, void* workspaceB, size_t workspace_bytes, void* stream) {
  ProfScope _prof(PC_BN_BWD, stream);
  _prof.work((double)rows * C * (2 * 6 + 4) + rows * C / 4.0);
  if (dout == nullptr || maskbits == nullptr || yA == nullptr || yB == nullptr || meanA == nullptr || meanB == nullptr ||
      invstdA == nullptr || invstdB == nullptr || scaleA == nullptr || scaleB == nullptr || dgammaA == nullptr ||
      dgammaB == nullptr || dbetaA == nullptr || dbetaB == nullptr || dyA == nullptr || dyB == nullptr ||
      workspaceA == nullptr || workspaceB == nullptr || workspaceA == workspaceB || rows <= 0 || C <= 0)
    return ICAMD_ERR_BAD_ARG;
  if (workspace_bytes < icamd_bn_bwd_workspace_bytes(rows, C)) return ICAMD_ERR_WORKSPACE;
  if (C > 4096) return ICAMD_ERR_UNSUPPORTED;
  const int rpb = icamd_bn_bwd_rows_per_block(rows, C);
  const long long nblk = (rows + rpb - 1) / rpb;
  float* part[2]; double* chunks[2]; float* cc[2];
  void* wsv[2] = {workspaceA, workspaceB};
  for (int i = 0; i < 2; ++i) {
    char* ws = (char*)wsv[i];
    chunks[i] = (double*)(ws + 256);
    ws += bn_chunk_bytes(C);
    part[i] = (float*)ws;
    ws += align_up((size_t)nblk * 2 * C * sizeof(float), 256);
    cc[i] = (float*)ws;
  }
  return icamd_bn_bwd_dual_launch((const bf16_t*)dout, maskbits, (const bf16_t*)yA, meanA, invstdA, scaleA, dgammaA, dbetaA,
                                  (bf16_t*)dyA, (const bf16_t*)yB, meanB, invstdB, scaleB, dgammaB, dbetaB, (bf16_t*)dyB, rows,
                                  C, accumulate, part[0], chunks[0], cc[0], part[1], chunks[1], cc[1], (hipStream_t)stream);
}

// workspace: chunks [64][2][C] doubles | c1,c2 [2][C] floats
size_t icamd_bn_bwd_apply_workspace_bytes(int C) {
  return C > 0 ? bn_chunk_bytes(C) + align_up((size_t)2 * C * sizeof(float), 256) : 0;
}

int icamd_bn_bwd_from_partials(const float* partials, int nrows, const void* g, const void* y, const float* mean,
                               const float* invstd, const float* scale, float* dgamma, float* dbeta, void* dy,
                               long long rows, int C, int accumulate, void* workspace, size_t workspace_bytes,
                               void* stream) {
  ProfScope _prof(PC_BN_BWD, stream);
  _prof.work((double)rows * C * 6 + 8.0 * nrows * C);
  if (partials == nullptr || nrows <= 0 || g == nullptr || y == nullptr || mean == nullptr || invstd == nullptr ||
      scale == nullptr || dgamma == nullptr || dbeta == nullptr || dy == nullptr || workspace == nullptr || rows <= 0 ||
      C <= 0 || C % 8 != 0)
    return ICAMD_ERR_BAD_ARG;
  if (workspace_bytes < icamd_bn_bwd_apply_workspace_bytes(C)) return ICAMD_ERR_WORKSPACE;
  if (C > 4096) return ICAMD_ERR_UNSUPPORTED;
  char* ws = (char*)workspace;
  double* chunks = (double*)(ws + 256);
  ws += bn_chunk_bytes(C);
  return icamd_bn_bwd_apply_launch(partials, nrows, (const bf16_t*)g, (const bf16_t*)y, mean, invstd, scale, dgamma, dbeta,
                                   (bf16_t*)dy, rows, C, accumulate, chunks, (float*)ws, (hipStream_t)stream);
}

int icamd_bn_bwd_from_gy_partials(const float* partials, int nrows, const void* g, const void* y, const float* mean,
                                  const float* invstd, const float* scale, float* dgamma, float* dbeta, void* dy,
                                  long long rows, int C, int accumulate, void* workspace, size_t workspace_bytes,
                                  void* stream) {
  ProfScope _prof(PC_BN_BWD, stream);
  _prof.work((double)rows * C * 6 + 8.0 * nrows * C);
  if (partials == nullptr || nrows <= 0 || g == nullptr || y == nullptr || mean == nullptr || invstd == nullptr ||
      scale == nullptr || dgamma == nullptr || dbeta == nullptr || dy == nullptr || workspace == nullptr || rows <= 0 ||
      C <= 0 || C % 8 != 0)
    return ICAMD_ERR_BAD_ARG;
  if (workspace_bytes < icamd_bn_bwd_apply_workspace_bytes(C)) return ICAMD_ERR_WORKSPACE;
  if (C > 4096) return ICAMD_ERR_UNSUPPORTED;
  char* ws = (char*)workspace;
  double* chunks = (double*)(ws + 256);
  ws += bn_chunk_bytes(C);
  return icamd_bn_bwd_apply_launch(partials, nrows, (const bf16_t*)g, (const bf16_t*)y, mean, invstd, scale, dgamma, dbeta,
                                   (bf16_t*)dy, rows, C, accumulate, chunks, (float*)ws, (hipStream_t)stream, 1);
}

// ---- fused forward across a bottleneck boundary (conv_fused_fwd.hip) --------------------------------------------------
int icamd_bn_apply_conv1x1_fused_supported(const icamd_conv_desc* d) {
  if (!conv_desc_ok(d) || d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad != 0) return 0;
  return icamd_bn_apply_conv1x1_fused_wanted((long long)d->N * d->OH * d->OW, d->Cin, d->Cout) ? 1 : 0;
}

int icamd_bn_apply_conv1x1_fused(const icamd_conv_desc* d, const void* y, const float* scale, const float* shift,
                                 const void* residual, const float* res_scale, const float* res_shift, void* out,
                                 uint8_t* maskbits, const void* w, void* y1, float* stats, void* stream) {
  ProfScope _prof(PC_FUSED_FWD, stream);
  if (d != nullptr) {
    const ConvWork cw = conv_work(d);
    _prof.work(3.0 * cw.in + cw.in / 16 + cw.out + 2 * cw.w, cw.flops);   // y, residual read, out + mask written; y1 written
  }
  if (y == nullptr || scale == nullptr || shift == nullptr || residual == nullptr || out == nullptr || maskbits == nullptr ||
      w == nullptr || y1 == nullptr || (res_scale == nullptr) != (res_shift == nullptr))
    return ICAMD_ERR_BAD_ARG;
  if (!icamd_bn_apply_conv1x1_fused_supported(d)) return ICAMD_ERR_UNSUPPORTED;
  FusedFwdParams p;
  memset(&p, 0, sizeof(p));
  p.y = (const bf16_t*)y; p.res = (const bf16_t*)residual; p.scale = scale; p.shift = shift; p.res_scale = res_scale;
  p.res_shift = res_shift; p.out = (bf16_t*)out; p.maskbits = maskbits; p.w = (const bf16_t*)w; p.y1 = (bf16_t*)y1; p.stats = stats;
  p.M = d->N * d->OH * d->OW; p.K = d->Cin; p.N = d->Cout;
  return icamd_bn_apply_conv1x1_fused_launch(p, (hipStream_t)stream);
}

// ---- fused backward of "pointwise convolution -> BatchNorm" (conv_fused_bwd.hip) -----------------------------------------
int icamd_conv1x1_bn_bwd_fused_supported(const icamd_conv_desc* d) {
  if (!conv_desc_ok(d) || d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad != 0) return 0;
  return icamd_conv1x1_bn_bwd_fused_wanted((long long)d->N * d->OH * d->OW, d->Cin, d->Cout) ? 1 : 0;
}

size_t icamd_conv1x1_bn_bwd_fused_workspace_bytes(const icamd_conv_desc* d) {
  if (!icamd_conv1x1_bn_bwd_fused_supported(d)) return 0;
  int S = 1, rows = 0;
  icamd_conv1x1_bn_bwd_fused_plan(d->N * d->OH * d->OW, d->Cin, &S, &rows);
  return (size_t)S * d->Cout * d->Cin * sizeof(float);
}

int icamd_conv1x1_bn_bwd_fused(const icamd_conv_desc* d, const float* partials, int nrows, const void* g, const void* y,
                               const float* mean, const float* invstd, const float* scale, float* dgamma, float* dbeta,
                               const void* x, const void* w_t, void* dx, float* dw, int accumulate, void* bn_workspace,
                               size_t bn_workspace_bytes, void* wgrad_workspace, size_t wgrad_workspace_bytes, void* stream) {
  ProfScope _prof(PC_FUSED_BWD, stream);
  if (d != nullptr) {
    const ConvWork cw = conv_work(d);
    // g, y read (twice when the sums are formed here); x read, dx written; dw
    _prof.work((partials ? 2.0 : 4.0) * cw.out + 2.0 * cw.in + 4.0 * cw.w + (partials ? 8.0 * nrows * d->Cout : 0.0), 2.0 * cw.flops);
  }
  if ((partials != nullptr && nrows <= 0) || g == nullptr || y == nullptr || mean == nullptr || invstd == nullptr || scale == nullptr ||
      dgamma == nullptr || dbeta == nullptr || x == nullptr || w_t == nullptr || dx == nullptr || dw == nullptr ||
      bn_workspace == nullptr || wgrad_workspace == nullptr)
    return ICAMD_ERR_BAD_ARG;
  if (!icamd_conv1x1_bn_bwd_fused_supported(d)) return ICAMD_ERR_UNSUPPORTED;
  const int C = d->Cout;
  const long long M = (long long)d->N * d->OH * d->OW;
  if (bn_workspace_bytes < (partials ? icamd_bn_bwd_apply_workspace_bytes(C) : icamd_bn_bwd_workspace_bytes(M, C)) ||
      wgrad_workspace_bytes < icamd_conv1x1_bn_bwd_fused_workspace_bytes(d))
    return ICAMD_ERR_WORKSPACE;
  char* ws = (char*)bn_workspace;
  double* chunks = (double*)(ws + 256);
  ws += bn_chunk_bytes(C);
  int rc;
  float* c1c2;
  if (partials != nullptr) {
    // (sum g, sum g * y) rows left by icamd_conv2d_dgrad_bnred
    c1c2 = (float*)ws;
    rc = icamd_bn_bwd_finalize_launch(partials, nrows, mean, invstd, dgamma, dbeta, M, C, accumulate, chunks, c1c2, (hipStream_t)stream, 1);
  } else {
    // no sums yet: the reduce pass of icamd_bn_bwd over the (already masked) g and y first; workspace laid out as icamd_bn_bwd's
    const int rpb = icamd_bn_bwd_rows_per_block(M, C);
    const long long nblk = (M + rpb - 1) / rpb;
    float* part = (float*)ws;
    c1c2 = (float*)(ws + align_up((size_t)nblk * 2 * C * sizeof(float), 256));
    int nb = 0;
    rc = icamd_bn_bwd_reduce_launch((const bf16_t*)g, (const bf16_t*)y, mean, invstd, part, M, C, &nb, (hipStream_t)stream);
    if (rc) return rc;
    rc = icamd_bn_bwd_finalize_launch(part, nb, mean, invstd, dgamma, dbeta, M, C, accumulate, chunks, c1c2, (hipStream_t)stream, 0);
  }
  if (rc) return rc;
  FusedBwdParams p;
  memset(&p, 0, sizeof(p));
  p.g = (const bf16_t*)g; p.y = (const bf16_t*)y; p.x = (const bf16_t*)x; p.wt = (const bf16_t*)w_t; p.dx = (bf16_t*)dx;
  p.slab = (float*)wgrad_workspace;
  p.mean = mean; p.invstd = invstd; p.scale = scale; p.c1 = c1c2; p.c2 = c1c2 + C;
  p.M = (int)M; p.CI = d->Cin; p.CO = C;
  rc = icamd_conv1x1_bn_bwd_fused_launch(p, (hipStream_t)stream);
  if (rc) return rc;
  return icamd_slab_reduce_launch(p.slab, dw, (long long)C * d->Cin, p.S, accumulate, (hipStream_t)stream);
}

// ---- LayerNorm / GELU / column sums (ViT, ConvNeXt) -------------------------------------------------------------
int icamd_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                        long long rows, int C, float eps, void* stream) {
  ProfScope _prof(PC_LN_FWD, stream);
  _prof.work((double)rows * C * 4 + 8.0 * rows);
  if (x == nullptr || gamma == nullptr || beta == nullptr || y == nullptr || mean == nullptr || rstd == nullptr || rows <= 0 ||
      C <= 0)
    return ICAMD_ERR_BAD_ARG;
  return icamd_layernorm_fwd_launch((const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, rows, C, eps, (hipStream_t)stream);
}

// workspace: [counters|chunks] | partial rows [blocks][2][C] | scratch [2][C]
size_t icamd_layernorm_bwd_workspace_bytes(long long rows, int C) {
  if (rows <= 0 || C <= 0) return 0;
  return bn_chunk_bytes(C) + align_up((size_t)icamd_layernorm_bwd_blocks(rows) * 2 * C * sizeof(float), 256) +
         align_up((size_t)2 * C * sizeof(float), 256);
}

int icamd_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                        const void* addend, void* dx, float* dgamma, float* dbeta, long long rows, int C, int accumulate,
                        void* workspace, size_t workspace_bytes, void* stream) {
  ProfScope _prof(PC_LN_BWD, stream);
  _prof.work((double)rows * C * (6 + (addend ? 2 : 0)) + 8.0 * rows);
  if (dy == nullptr || x == nullptr || mean == nullptr || rstd == nullptr || gamma == nullptr || dx == nullptr ||
      dgamma == nullptr || dbeta == nullptr || workspace == nullptr || rows <= 0 || C <= 0 || C > 4096)
    return ICAMD_ERR_BAD_ARG;
  if (workspace_bytes < icamd_layernorm_bwd_workspace_bytes(rows, C)) return ICAMD_ERR_WORKSPACE;
  char* ws = (char*)workspace;
  double* chunks = (double*)(ws + 256);
  ws += bn_chunk_bytes(C);
  float* part = (float*)ws;
  const int nblk = icamd_layernorm_bwd_blocks(rows);
  ws += align_up((size_t)nblk * 2 * C * sizeof(float), 256);
  int rc = icamd_layernorm_bwd_launch((const bf16_t*)dy, (const bf16_t*)x, mean, rstd, gamma, (const bf16_t*)addend,
                                      (bf16_t*)dx, part, rows, C, (hipStream_t)stream);
  if (rc) return rc;
  return icamd_sum_partials_launch(part, nblk, C, dbeta, dgamma, accumulate, chunks, (float*)ws, (hipStream_t)stream);
}

int icamd_gelu_fwd(const void* z, void* a, long long numel, void* stream) {
  ProfScope _prof(PC_ELEMWISE, stream);
  _prof.work(4.0 * numel);
  if (z == nullptr || a == nullptr || numel <= 0) return ICAMD_ERR_BAD_ARG;
  return icamd_gelu_fwd_launch((const bf16_t*)z, (bf16_t*)a, numel, (hipStream_t)stream);
}

int icamd_gelu_bwd(const void* da, const void* z, void* dz, long long numel, void* stream) {
  ProfScope _prof(PC_ELEMWISE, stream);
  _prof.work(6.0 * numel);
  if (da == nullptr || z == nullptr || dz == nullptr || numel <= 0) return ICAMD_ERR_BAD_ARG;
  return icamd_gelu_bwd_launch((const bf16_t*)da, (const bf16_t*)z, (bf16_t*)dz, numel, (hipStream_t)stream);
}

size_t icamd_colsum_rows_workspace_bytes(long long rows, int cols) {
  if (rows <= 0 || cols <= 0) return 0;
  return bn_chunk_bytes(cols) + align_up((size_t)icamd_colsum_blocks(rows) * 2 * cols * sizeof(float), 256) +
         align_up((size_t)3 * cols * sizeof(float), 256);
}

// out[c] = (accumulate ? out[c] : 0) + sum_r x[r][c], two-level, fixed order (bias gradients of long token matrices)
int icamd_colsum_rows(const void* x, long long rows, int ld, int cols, float* out, int accumulate, void* workspace,
                      size_t workspace_bytes, void* stream) {
  ProfScope _prof(PC_MISC, stream);
  _prof.work(2.0 * rows * cols);
  if (x == nullptr || out == nullptr || workspace == nullptr || rows <= 0 || cols <= 0 || cols > 4096 || ld < cols)
    return ICAMD_ERR_BAD_ARG;
  if (workspace_bytes < icamd_colsum_rows_workspace_bytes(rows, cols)) return ICAMD_ERR_WORKSPACE;
  char* ws = (char*)workspace;
  double* chunks = (double*)(ws + 256);
  ws += bn_chunk_bytes(cols);
  float* part = (float*)ws;
  const int nblk = icamd_colsum_blocks(rows);
  ws += align_up((size_t)nblk * 2 * cols * sizeof(float), 256);
  float* scratch = (float*)ws;   // [3][cols]: discarded second sum + c1/c2
  int rc = icamd_colsum_partial_launch((const bf16_t*)x, part, rows, ld, cols, (hipStream_t)stream);
  if (rc) return rc;
  return icamd_sum_partials_launch(part, nblk, cols, out, scratch, accumulate, chunks, scratch + cols, (hipStream_t)stream);
}

int icamd_vit_tokens_fwd(const void* patches, const float* cls_token, const float* pos_embed, void* tokens, int B, int T, int C,
                         void* stream) {
  ProfScope _prof(PC_MISC, stream);
  _prof.work(4.0 * B * T * C);
  if (patches == nullptr || cls_token == nullptr || pos_embed == nullptr || tokens == nullptr || B <= 0 || T <= 1 || C <= 0)
    return ICAMD_ERR_BAD_ARG;
  return icamd_vit_tokens_fwd_launch((const bf16_t*)patches, cls_token, pos_embed, (bf16_t*)tokens, B, T, C, (hipStream_t)stream);
}

int icamd_batch_sum(const void* x, long long stride, int B, long long n, float* out, int accumulate, void* stream) {
  ProfScope _prof(PC_MISC, stream);
  _prof.work(2.0 * B * n);
  if (x == nullptr || out == nullptr || B <= 0 || n <= 0) return ICAMD_ERR_BAD_ARG;
  return icamd_batch_sum_launch((const bf16_t*)x, stride, B, n, out, accumulate, (hipStream_t)stream);
}

int icamd_strided_rows_copy(const void* src, long long src_stride, void* dst, long long dst_stride, long long rows, long long C,
                            void* stream) {
  ProfScope _prof(PC_MISC, stream);
  _prof.work(4.0 * rows * C);
  if (src == nullptr || dst == nullptr || rows <= 0 || C <= 0) return ICAMD_ERR_BAD_ARG;
  return icamd_strided_rows_copy_launch((const bf16_t*)src, src_stride, (bf16_t*)dst, dst_stride, rows, C, (hipStream_t)stream);
}

int icamd_fill_zero(void* ptr, size_t bytes, void* stream) {
  if (ptr == nullptr) return ICAMD_ERR_BAD_ARG;
  return hipMemsetAsync(ptr, 0, bytes, (hipStream_t)stream) == hipSuccess ? ICAMD_OK : ICAMD_ERR_LAUNCH;
}

// ---- ConvNeXt: depthwise 7x7 + layer scale / stochastic depth / residual ------------------------------------------
int icamd_dwconv7_fwd(const void* x, const void* w, const float* bias, void* y, int N, int H, int W, int C, void* stream) {
  ProfScope _prof(PC_DWCONV, stream);
  _prof.work(4.0 * N * H * W * C, 98.0 * N * H * W * C);
  if (x == nullptr || w == nullptr || y == nullptr || N <= 0 || H <= 0 || W <= 0 || C <= 0) return ICAMD_ERR_BAD_ARG;
  return icamd_dwconv7_launch((const bf16_t*)x, (const bf16_t*)w, bias, nullptr, (bf16_t*)y, N, H, W, C, 0, (hipStream_t)stream);
}

int icamd_dwconv7_dgrad(const void* dy, const void* w, const void* addend, void* dx, int N, int H, int W, int C, void* stream) {
  ProfScope _prof(PC_DWCONV, stream);
  _prof.work((4.0 + (addend ? 2 : 0)) * N * H * W * C, 98.0 * N * H * W * C);
  if (dy == nullptr || w == nullptr || dx == nullptr || N <= 0 || H <= 0 || W <= 0 || C <= 0) return ICAMD_ERR_BAD_ARG;
  return icamd_dwconv7_launch((const bf16_t*)dy, (const bf16_t*)w, nullptr, (const bf16_t*)addend, (bf16_t*)dx, N, H, W, C, 1,
                              (hipStream_t)stream);
}

size_t icamd_dwconv7_wgrad_workspace_bytes(int N, int H, int W, int C) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 32 != 0) return 0;
  return (size_t)icamd_dwconv7_wgrad_blocks(N, H, W, C) * 50 * C * sizeof(float);   // [blocks][49][C] + the bias rows [blocks][C]
}

int icamd_dwconv7_wgrad(const void* x, const void* dy, float* dw, int accumulate, void* workspace, size_t workspace_bytes,
                        int N, int H, int W, int C, void* stream) {
  ProfScope _prof(PC_DWCONV, stream);
  _prof.work(4.0 * N * H * W * C, 98.0 * N * H * W * C);
  if (x == nullptr || dy == nullptr || dw == nullptr || workspace == nullptr) return ICAMD_ERR_BAD_ARG;
  const size_t need = icamd_dwconv7_wgrad_workspace_bytes(N, H, W, C);
  if (need == 0 || workspace_bytes < need) return ICAMD_ERR_WORKSPACE;
  return icamd_dwconv7_wgrad_launch((const bf16_t*)x, (const bf16_t*)dy, (float*)workspace, dw, nullptr, N, H, W, C, accumulate,
                                    (hipStream_t)stream);
}

int icamd_dwconv7_wgrad_bias_supported(int N, int H, int W, int C) {
  return (N > 0 && H > 0 && W > 0 && C > 0 && C % 32 == 0 && ::icamd_dwconv7_wgrad_bias_supported_cxx(N, H, W, C)) ? 1 : 0;
}

int icamd_dwconv7_wgrad_bias(const void* x, const void* dy, float* dw, float* dbias, int accumulate, void* workspace,
                             size_t workspace_bytes, int N, int H, int W, int C, void* stream) {
  ProfScope _prof(PC_DWCONV, stream);
  _prof.work(4.0 * N * H * W * C, 100.0 * N * H * W * C);
  if (x == nullptr || dy == nullptr || dw == nullptr || dbias == nullptr || workspace == nullptr) return ICAMD_ERR_BAD_ARG;
  const size_t need = icamd_dwconv7_wgrad_workspace_bytes(N, H, W, C);
  if (need == 0 || workspace_bytes < need) return ICAMD_ERR_WORKSPACE;
  return icamd_dwconv7_wgrad_launch((const bf16_t*)x, (const bf16_t*)dy, (float*)workspace, dw, dbias, N, H, W, C, accumulate,
                                    (hipStream_t)stream);
}

int icamd_layerscale_fwd(const void* z, const void* inp, const float* gamma, const float* keep, void* out, long long rows, int C,
                         long long rows_per_image, void* stream) {
  ProfScope _prof(PC_ELEMWISE, stream);
  _prof.work(6.0 * rows * C);
  if (z == nullptr || inp == nullptr || gamma == nullptr || out == nullptr || rows <= 0 || C <= 0 || rows_per_image <= 0)
    return ICAMD_ERR_BAD_ARG;
  return icamd_layerscale_fwd_launch((const bf16_t*)z, (const bf16_t*)inp, gamma, keep, (bf16_t*)out, rows, C, rows_per_image,
                                     (hipStream_t)stream);
}

size_t icamd_layerscale_bwd_workspace_bytes(long long rows, int C) {
  if (rows <= 0 || C <= 0) return 0;
  return bn_chunk_bytes(C) + align_up((size_t)icamd_layerscale_bwd_blocks(rows) * 2 * C * sizeof(float), 256) +
         align_up((size_t)3 * C * sizeof(float), 256);
}

int icamd_layerscale_bwd(const void* dout, const void* z, const float* gamma, const float* keep, void* dz, float* dgamma,
                         long long rows, int C, long long rows_per_image, int accumulate, void* workspace,
                         size_t workspace_bytes, void* stream) {
  ProfScope _prof(PC_ELEMWISE, stream);
  _prof.work(6.0 * rows * C);
  if (dout == nullptr || z == nullptr || gamma == nullptr || dz == nullptr || dgamma == nullptr || workspace == nullptr ||
      rows <= 0 || C <= 0 || C > 4096 || rows_per_image <= 0)
    return ICAMD_ERR_BAD_ARG;
  if (workspace_bytes < icamd_layerscale_bwd_workspace_bytes(rows, C)) return ICAMD_ERR_WORKSPACE;
  char* ws = (char*)workspace;
  double* chunks = (double*)(ws + 256);
  ws += bn_chunk_bytes(C);
  float* part = (float*)ws;
  const int nblk = icamd_layerscale_bwd_blocks(rows);
  ws += align_up((size_t)nblk * 2 * C * sizeof(float), 256);
  float* scratch = (float*)ws;
  int rc = icamd_layerscale_bwd_launch((const bf16_t*)dout, (const bf16_t*)z, gamma, keep, (bf16_t*)dz, part, rows, C,
                                       rows_per_image, (hipStream_t)stream);
  if (rc) return rc;
  return icamd_sum_partials_launch(part, nblk, C, dgamma, scratch, accumulate, chunks, scratch + C, (hipStream_t)stream);
}

// Layer scale folded into the Mlp's second Linear layer (round 5): see include/icamd.h
int icamd_layerscale_fold(const float* params, void* shadow, float* fold_bias, const long long* jobs, int njobs, int total_rows,
                          long long total_elements, void* stream) {
  ProfScope _prof(PC_ELEMWISE, stream);
  _prof.work(6.0 * (double)total_elements);
  if (params == nullptr || shadow == nullptr || fold_bias == nullptr || jobs == nullptr || njobs <= 0 || total_rows <= 0)
    return ICAMD_ERR_BAD_ARG;
  return icamd_layerscale_fold_launch(params, (bf16_t*)shadow, fold_bias, jobs, njobs, total_rows, (hipStream_t)stream);
}

int icamd_rows_fix(const float* keep, int n_images, void* dst1, const void* src1, long long bytes1, void* dst2, long long bytes2,
                   void* stream) {
  ProfScope _prof(PC_ELEMWISE, stream);
  _prof.work(4.0 * n_images);   // (the bytes of the dropped samples are data-dependent: not booked)
  if (keep == nullptr || n_images <= 0 || n_images > 65535 || (dst1 == nullptr && dst2 == nullptr) || bytes1 < 0 || bytes2 < 0 ||
      bytes1 % 16 != 0 || bytes2 % 16 != 0 || (dst1 == nullptr && src1 != nullptr))
    return ICAMD_ERR_BAD_ARG;
  return icamd_rows_fix_launch(keep, n_images, dst1, src1, bytes1, dst2, bytes2, (hipStream_t)stream);
}

int icamd_dropped_colsum(const void* dy, const float* keep, int n_images, long long rows_per_image, int C, float* partial,
                         void* stream) {
  ProfScope _prof(PC_ELEMWISE, stream);
  _prof.work(4.0 * n_images * C);
  if (dy == nullptr || keep == nullptr || partial == nullptr || n_images <= 0 || rows_per_image <= 0 || C <= 0 || C % 8 != 0)
    return ICAMD_ERR_BAD_ARG;
  return icamd_dropped_colsum_launch((const bf16_t*)dy, keep, n_images, rows_per_image, C, partial, (hipStream_t)stream);
}

int icamd_layerscale_param_grads(const float* G, const float* w, const float* bias, const float* gamma, const float* colsum_all,
                                 const float* dropped, int n_images, float cb, int C, int K, float* dw, float* dbias,
                                 float* dgamma, int accumulate, void* stream) {
  ProfScope _prof(PC_ELEMWISE, stream);
  _prof.work((accumulate ? 16.0 : 12.0) * C * K);
  if (G == nullptr || w == nullptr || bias == nullptr || gamma == nullptr || colsum_all == nullptr || dw == nullptr ||
      dbias == nullptr || dgamma == nullptr || C <= 0 || K <= 0 || K % 4 != 0 || (dropped != nullptr && n_images <= 0))
    return ICAMD_ERR_BAD_ARG;
  return icamd_layerscale_param_grads_launch(G, w, bias, gamma, colsum_all, dropped, n_images, cb, C, K, dw, dbias, dgamma,
                                             accumulate, (hipStream_t)stream);
}

// ---- attention (ViT) --------------------------------------------------------------------------------------------
int icamd_attention_fwd(const void* qkv, void* out, float* lse, int B, int T, int H, int D, float scale, void* stream) {
  ProfScope _prof(PC_ATTN_FWD, stream);
  _prof.work(8.0 * B * T * H * D, 4.0 * B * H * (double)T * T * D);
  if (qkv == nullptr || out == nullptr || lse == nullptr || B <= 0 || T <= 0 || H <= 0) return ICAMD_ERR_BAD_ARG;
  if (D != 64) return ICAMD_ERR_UNSUPPORTED;
  return icamd_attention_fwd_launch((const bf16_t*)qkv, (bf16_t*)out, lse, B, T, H, scale, (hipStream_t)stream);
}

int icamd_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                        int B, int T, int H, int D, float scale, void* stream) {
  ProfScope _prof(PC_ATTN_BWD, stream);
  _prof.work(16.0 * B * T * H * D, 10.0 * B * H * (double)T * T * D);
  if (qkv == nullptr || out == nullptr || dout == nullptr || lse == nullptr || delta == nullptr || dqkv == nullptr || B <= 0 ||
      T <= 0 || H <= 0)
    return ICAMD_ERR_BAD_ARG;
  if (D != 64) return ICAMD_ERR_UNSUPPORTED;
  return icamd_attention_bwd_launch((const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, B,
                                    T, H, scale, (hipStream_t)stream);
}

int icamd_maxpool3x3s2_fwd(const void* x, void* out, uint8_t* argmax, int N, int IH, int IW, int C, void* stream) {
  ProfScope _prof(PC_POOL, stream);
  _prof.work((double)N * IH * IW * C * (2 + 0.75));
  if (x == nullptr || out == nullptr || N <= 0 || IH <= 0 || IW <= 0 || C <= 0) return ICAMD_ERR_BAD_ARG;
  const int OH = (IH + 2 - 3) / 2 + 1, OW = (IW + 2 - 3) / 2 + 1;
  return icamd_maxpool_fwd_launch((const bf16_t*)x, (bf16_t*)out, argmax, N, IH, IW, C, OH, OW, (hipStream_t)stream);
}

int icamd_bn_relu_maxpool3x3s2_fwd(const void* y, const float* scale, const float* shift, void* out, uint8_t* argmax, int N,
                                   int IH, int IW, int C, void* stream) {
  ProfScope _prof(PC_BN_APPLY, stream);
  _prof.work((double)N * IH * IW * C * (2 + 0.75));
  if (y == nullptr || scale == nullptr || shift == nullptr || out == nullptr || N <= 0 || IH <= 0 || IW <= 0 || C <= 0)
    return ICAMD_ERR_BAD_ARG;
  const int OH = (IH + 2 - 3) / 2 + 1, OW = (IW + 2 - 3) / 2 + 1;
  return icamd_bn_relu_maxpool_fwd_launch((const bf16_t*)y, scale, shift, (bf16_t*)out, argmax, N, IH, IW, C, OH, OW,
                                          (hipStream_t)stream);
}

int icamd_maxpool3x3s2_bwd(const void* dout, const uint8_t* argmax, void* dx, int N, int IH, int IW, int C, void* stream) {
  ProfScope _prof(PC_POOL, stream);
  _prof.work((double)N * IH * IW * C * (2 + 0.75));
  if (dout == nullptr || argmax == nullptr || dx == nullptr || N <= 0 || IH <= 0 || IW <= 0 || C <= 0) return ICAMD_ERR_BAD_ARG;
  const int OH = (IH + 2 - 3) / 2 + 1, OW = (IW + 2 - 3) / 2 + 1;
  return icamd_maxpool_bwd_launch((const bf16_t*)dout, argmax, (bf16_t*)dx, N, IH, IW, C, OH, OW, (hipStream_t)stream);
}

int icamd_avgpool_fwd(const void* x, void* out, int N, int HW, int C, void* stream) {
  ProfScope _prof(PC_POOL, stream);
  _prof.work(2.0 * N * HW * C + 2.0 * N * C);
  if (x == nullptr || out == nullptr || N <= 0 || HW <= 0 || C <= 0) return ICAMD_ERR_BAD_ARG;
  return icamd_avgpool_fwd_launch((const bf16_t*)x, (bf16_t*)out, N, HW, C, (hipStream_t)stream);
}

int icamd_avgpool_bwd(const void* dout, void* dx, int N, int HW, int C, void* stream) {
  ProfScope _prof(PC_POOL, stream);
  _prof.work(2.0 * N * HW * C + 2.0 * N * C);
  if (dout == nullptr || dx == nullptr || N <= 0 || HW <= 0 || C <= 0) return ICAMD_ERR_BAD_ARG;
  return icamd_avgpool_bwd_launch((const bf16_t*)dout, (bf16_t*)dx, N, HW, C, (hipStream_t)stream);
}

int icamd_pack_input(const float* x, void* out, int B, int Cin, int H, int W, int mode, float lam, int yl, int yh,
                     int xl, int xh, void* stream) {
  ProfScope _prof(PC_PACK, stream);
  _prof.work((mode ? 8.0 : 4.0) * B * Cin * H * W + 16.0 * B * H * W);
  if (x == nullptr || out == nullptr || B <= 0 || H <= 0 || W <= 0 || mode < 0 || mode > 2) return ICAMD_ERR_BAD_ARG;
  if (mode != 0 && (B % 2) != 0) return ICAMD_ERR_BAD_ARG;  // timm Mixup asserts an even batch
  return icamd_pack_input_launch(x, (bf16_t*)out, B, Cin, H, W, mode, lam, yl, yh, xl, xh, (hipStream_t)stream);
}

int icamd_pack_input_rgb4(const float* x, void* out, int B, int Cin, int H, int W, int mode, float lam, int yl, int yh,
                          int xl, int xh, void* stream) {
  ProfScope _prof(PC_PACK, stream);
  _prof.work((mode ? 8.0 : 4.0) * B * Cin * H * W + 8.0 * B * H * (W + 8));
  if (x == nullptr || out == nullptr || B <= 0 || H <= 0 || W <= 0 || mode < 0 || mode > 2) return ICAMD_ERR_BAD_ARG;
  return icamd_pack_input_rgb4_launch(x, (bf16_t*)out, B, Cin, H, W, mode, lam, yl, yh, xl, xh, (hipStream_t)stream);
}

// ---- ResNet stem: 7x7 stride 2 pad 3 convolution on the rgb4 layout ------------------------------------------------
static bool stem_shape_ok(int N, int H, int W, int Cout) {
  return N > 0 && H >= 7 && W >= 8 && W % 2 == 0 && Cout > 0 && Cout % 8 == 0 && (long long)N * H * (W + 8) * 4 < (1ll << 31);
}

int icamd_stem7x7s2_stats_rows(int N, int H, int W) {
  const long long M = (long long)N * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1);
  return (int)((M + 127) / 128);
}

int icamd_stem7x7s2_fwd(const void* x4, const void* w, void* y, const float* bias, float* stats, int relu, int N, int H,
                        int W, int Cout, void* stream) {
  ProfScope _prof(PC_IGEMM_FWD, stream);
  {   // rgb4 layout in, [N][OH][OW][Cout] out, [Cout][8][8][4] filters; 147 real taps per output
    const double oh = (H - 1) / 2 + 1, ow = (W - 1) / 2 + 1;
    _prof.work(8.0 * N * H * (W + 8) + 2.0 * N * oh * ow * Cout + 512.0 * Cout, 2.0 * N * oh * ow * Cout * 147);
  }
  if (x4 == nullptr || w == nullptr || y == nullptr || !stem_shape_ok(N, H, W, Cout)) return ICAMD_ERR_BAD_ARG;
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;   // (H + 6 - 7) / 2 + 1
  if ((long long)N * OH * OW >= (1ll << 31)) return ICAMD_ERR_UNSUPPORTED;
  // conv_stem.hip: the training form (statistics) and, round 4, the inference form (bias + ReLU, no statistics)
  if ((stats == nullptr || (bias == nullptr && !relu)) && icamd_stem_resident_wanted(N, H, W, Cout)) {
    StemParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.x = (const bf16_t*)x4; sp.w = (const bf16_t*)w; sp.y = (bf16_t*)y; sp.stats = stats; sp.N = N; sp.H = H; sp.W = W;
    sp.bias = bias; sp.relu = relu;
    return icamd_stem_resident_launch(sp, (hipStream_t)stream);
  }
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.in = (const bf16_t*)x4; p.wt = (const bf16_t*)w; p.out = (bf16_t*)y; p.bias = bias; p.stats = stats; p.relu = relu;
  p.N = N; p.IH = H; p.IW = W + 8; p.Cin = 4;              // IW: padded row pitch in pixels; Cin: elements per pixel
  p.OH = OH; p.OW = OW; p.Cout = Cout;
  p.P = OH; p.Q = OW; p.M = N * OH * OW;
  p.ostr = 1; p.istr = 2;
  p.ntaps = 1; p.Ktot = 256; p.KW = 1; p.tap_sign = 1; p.regular_taps = 1;
  p.stem7 = 1;
  return icamd_igemm_launch(p, (hipStream_t)stream);
}

size_t icamd_stem7x7s2_wgrad_workspace_bytes(int N, int H, int W, int Cout) {
  if (!stem_shape_ok(N, H, W, Cout)) return 0;
  const long long M = (long long)N * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1);
  if (M >= (1ll << 30)) return 0;
  int S = 1, rows = 0;
  icamd_wgrad_plan((int)M, Cout, 256, &S, &rows);
  if (icamd_stem_resident_wanted(N, H, W, Cout)) {
    const int S2 = icamd_stem_wgrad_resident_splits(N, H);
    if (S2 > S) S = S2;
  }
  return (size_t)S * Cout * (256 + 1) * sizeof(float);
}

int icamd_stem7x7s2_wgrad(const void* x4, const void* dy, float* dw, int accumulate, void* workspace, size_t workspace_bytes,
                          int N, int H, int W, int Cout, void* stream) {
  ProfScope _prof(PC_WGRAD, stream);
  {
    const double oh = (H - 1) / 2 + 1, ow = (W - 1) / 2 + 1;
    _prof.work(8.0 * N * H * (W + 8) + 2.0 * N * oh * ow * Cout + 1024.0 * Cout, 2.0 * N * oh * ow * Cout * 147);
  }
  if (x4 == nullptr || dy == nullptr || dw == nullptr || workspace == nullptr) return ICAMD_ERR_BAD_ARG;
  const size_t need = icamd_stem7x7s2_wgrad_workspace_bytes(N, H, W, Cout);
  if (need == 0) return ICAMD_ERR_BAD_ARG;
  if (workspace_bytes < need) return ICAMD_ERR_WORKSPACE;
  if (icamd_stem_resident_wanted(N, H, W, Cout)) {   // conv_stem.hip
    StemWgradParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.x = (const bf16_t*)x4; sp.dy = (const bf16_t*)dy; sp.slab = (float*)workspace; sp.N = N; sp.H = H; sp.W = W;
    const int S = icamd_stem_wgrad_resident_splits(N, H);
    const int rc = icamd_stem_wgrad_resident_launch(sp, S, (hipStream_t)stream);
    if (rc) return rc;
    return icamd_slab_reduce_launch(sp.slab, dw, (long long)Cout * 256, S, accumulate, (hipStream_t)stream, 1);
  }
  WgradParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const bf16_t*)x4; p.dy = (const bf16_t*)dy; p.slab = (float*)workspace;
  p.N = N; p.IH = H; p.IW = W + 8; p.Cin = 4; p.OH = (H - 1) / 2 + 1; p.OW = (W - 1) / 2 + 1; p.Cout = Cout;
  p.KH = 8; p.KW = 8; p.stride = 2; p.pad = 3;
  p.M = N * p.OH * p.OW; p.Ktot = 256;
  p.stem7 = 1;
  icamd_wgrad_plan(p.M, p.Cout, p.Ktot, &p.S, &p.rows_per_split);
  int rc = icamd_wgrad_launch(p, (hipStream_t)stream);
  if (rc) return rc;
  return icamd_slab_reduce_launch(p.slab, dw, (long long)p.Cout * p.Ktot, p.S, accumulate, (hipStream_t)stream, 1);
}

int icamd_softmax_xent(const void* logits, int ld, int B, int C, const int64_t* y1, const int64_t* y2, float lam,
                       float smoothing, float gscale, float* loss_rows, int32_t* pred, void* dlogits, void* stream) {
  ProfScope _prof(PC_LOSS, stream);
  _prof.work(4.0 * B * ld);
  if (logits == nullptr || y1 == nullptr || loss_rows == nullptr) return ICAMD_ERR_BAD_ARG;
  return icamd_softmax_xent_launch((const bf16_t*)logits, ld, B, C, (const long long*)y1, (const long long*)y2, lam,
                                   smoothing, gscale, loss_rows, pred, (bf16_t*)dlogits, (hipStream_t)stream);
}

int icamd_step_metrics(const float* loss_rows, const int32_t* pred, const int64_t* target, int B, int C,
                       float* loss_out, int32_t* finite_out, double* acc_f64, int32_t* counts, float* loss_log,
                       int log_slot, int log_stride, int respect_skip, void* stream) {
  ProfScope _prof(PC_LOSS, stream);
  _prof.work(16.0 * B);
  if (loss_out == nullptr || finite_out == nullptr || acc_f64 == nullptr || B <= 0) return ICAMD_ERR_BAD_ARG;
  if (pred != nullptr && target == nullptr) return ICAMD_ERR_BAD_ARG;
  if (loss_rows == nullptr && pred == nullptr && !(respect_skip & 4)) return ICAMD_ERR_BAD_ARG;
  return icamd_step_metrics_launch(loss_rows, pred, (const long long*)target, B, C, loss_out, finite_out, acc_f64, counts,
                                   loss_log, log_slot, log_stride, respect_skip, (hipStream_t)stream);
}

size_t icamd_grad_norm_workspace_bytes(void) { return 512 * sizeof(double); }

int icamd_grad_norm(const float* g, long long n, float inv_scale, float max_norm, void* workspace, float* out,
                    void* stream) {
  ProfScope _prof(PC_OPTIM, stream);
  _prof.work(4.0 * n);
  if (g == nullptr || n <= 0 || workspace == nullptr || out == nullptr) return ICAMD_ERR_BAD_ARG;
  return icamd_grad_norm_launch(g, n, inv_scale, max_norm, (double*)workspace, out, (hipStream_t)stream);
}

int icamd_adamw_ema(float* p, float* g, float* m, float* v, float* ema, void* shadow, long long n, float lr, float wd,
                    float beta1, float beta2, float eps, int step, float gscale, float ema_decay, const float* clip,
                    const int32_t* finite_flag, int32_t* skipped_steps, int zero_grad, void* stream) {
  ProfScope _prof(PC_OPTIM, stream);
  _prof.work((30.0 + (ema ? 8 : 0)) * n);
  if (p == nullptr || g == nullptr || m == nullptr || v == nullptr || n <= 0) return ICAMD_ERR_BAD_ARG;
  return icamd_adamw_ema_launch(p, g, m, v, ema, (bf16_t*)shadow, n, lr, wd, beta1, beta2, eps, step, gscale, ema_decay,
                                clip, finite_flag, skipped_steps, zero_grad, (hipStream_t)stream);
}

int icamd_grad_guard(float* g, long long n, const int32_t* finite_flag, void* stream) {
  ProfScope _prof(PC_OPTIM, stream);
  _prof.work(0.0);
  if (g == nullptr || finite_flag == nullptr || n <= 0) return ICAMD_ERR_BAD_ARG;
  return icamd_grad_guard_launch(g, n, finite_flag, (hipStream_t)stream);
}

int icamd_optim_ema(int kind, float* p, float* g, float* m, float* v, float* ema, void* shadow, long long n, float lr,
                    float wd, float beta1, float beta2, float eps, int step, float gscale, float ema_decay,
                    const float* clip, const int32_t* finite_flag, int32_t* skipped_steps, int zero_grad, void* stream) {
  ProfScope _prof(PC_OPTIM, stream);
  _prof.work((22.0 + (v ? 8 : 0) + (ema ? 8 : 0)) * n);
  if (p == nullptr || g == nullptr || m == nullptr || n <= 0) return ICAMD_ERR_BAD_ARG;
  if ((kind == ICAMD_OPT_ADAMW || kind == ICAMD_OPT_ADAM) && v == nullptr) return ICAMD_ERR_BAD_ARG;
  return icamd_optim_ema_launch(kind, p, g, m, v, ema, (bf16_t*)shadow, n, lr, wd, beta1, beta2, eps, step, gscale,
                                ema_decay, clip, finite_flag, skipped_steps, zero_grad, (hipStream_t)stream);
}

int icamd_lerp(float* dst, const float* src, long long n, float w, const int32_t* finite_flag, void* stream) {
  ProfScope _prof(PC_OPTIM, stream);
  _prof.work(12.0 * n);
  if (dst == nullptr || src == nullptr || n <= 0) return ICAMD_ERR_BAD_ARG;
  return icamd_lerp_launch(dst, src, n, w, finite_flag, (hipStream_t)stream);
}

int icamd_f32_to_bf16(const float* src, void* dst, long long n, void* stream) {
  ProfScope _prof(PC_OPTIM, stream);
  _prof.work(6.0 * n);
  if (src == nullptr || dst == nullptr || n <= 0) return ICAMD_ERR_BAD_ARG;
  return icamd_f32_to_bf16_launch(src, (bf16_t*)dst, n, (hipStream_t)stream);
}

int icamd_colsum(const void* x, int rows, int ld, int cols, float* out, int accumulate, void* stream) {
  ProfScope _prof(PC_MISC, stream);
  _prof.work(2.0 * rows * cols);
  if (x == nullptr || out == nullptr || rows <= 0 || cols <= 0 || ld < cols) return ICAMD_ERR_BAD_ARG;
  return icamd_colsum_launch((const bf16_t*)x, rows, ld, cols, out, accumulate, (hipStream_t)stream);
}

}  // extern "C"
