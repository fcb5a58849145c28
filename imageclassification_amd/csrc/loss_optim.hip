// Loss, metrics and optimizer kernels of the training step for gfx950.
//   * softmax cross-entropy forward+backward in one pass: hard labels, label smoothing, and the soft
//     mixup/cutmix targets of timm.data.Mixup (reference: criterion call /root/reference/engine.py:49,52,
//     criterion choice train.py:256-261, evaluate's plain CE engine.py:147).
//   * device-side metric accumulators: loss sum, top-1 count, per-class TP/FP/FN (reference does 3*C
//     blocking .item() calls per step, engine.py:83-97,184-190); integer atomics only (deterministic).
//   * fused AdamW + EMA over flat fp32 arenas, emitting the bf16 shadow weights the conv kernels read
//     (reference: optimizer.step() engine.py:74 / utils.py:443, model_ema.update engine.py:68,77).
//   * global grad-norm / clip coefficient (reference utils.py:438-442,456-468).
// once-read streams of this translation unit use non-temporal loads (round 5: ViT-B/16 34.9-35.0 -> 34.7 ms, ResNet-50 -0.03..-0.06 ms
// in two A/B pairs each; dwconv.hip measured worse with them and keeps the default)
#define ICAMD_STREAM_NT 1
#include "common.h"
#include "icamd_internal.h"

namespace {

// ------------------------------------------------------------------------------------------------
// softmax cross-entropy; one wave per row. logits bf16 [B][ld] (columns >= C are padding).
// target distribution t = lam*onehot_s(y1) + (1-lam)*onehot_s(y2), onehot_s: on = 1-s+s/C, off = s/C
// (y2/lam unused when lam == 1). loss_b = -sum_c t_c*logp_c ; dlogits = (softmax - t) * gscale
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_xent_kernel(const bf16_t* __restrict__ logits, int ld, int B, int C,
                                                           const long long* __restrict__ y1, const long long* __restrict__ y2,
                                                           float lam, float smoothing, float gscale,
                                                           float* __restrict__ loss_rows, int* __restrict__ pred,
                                                           bf16_t* __restrict__ dlogits) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const bf16_t* x = logits + (long long)row * ld;
  float mx = -INFINITY;
  int amax = lane < C ? lane : 0;
  for (int c = lane; c < C; c += 64) {
    const float v = bf16_to_f32(x[c]);
    if (v > mx) { mx = v; amax = c; }
  }
  // wave argmax, lowest index wins on ties (a NaN row gives a NaN loss and the step is skipped)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(mx, o, 64);
    const int oa = __shfl_xor(amax, o, 64);
    if (om > mx || (om == mx && oa < amax)) { mx = om; amax = oa; }
  }
  float se = 0.f, sx = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float v = bf16_to_f32(x[c]);
    se += __expf(v - mx);
    sx += v;
  }
  se = wave_sum(se);
  sx = wave_sum(sx);
  const float lse = mx + __logf(se);
  const int t1 = (int)y1[row];
  const int t2 = (y2 != nullptr) ? (int)y2[row] : t1;
  const float on = 1.f - smoothing, off = smoothing / (float)C;
  // -sum_c t_c logp_c with logp_c = x_c - lse
  const float lp1 = bf16_to_f32(x[t1]) - lse, lp2 = bf16_to_f32(x[t2]) - lse;
  const float sum_logp = sx - (float)C * lse;
  const float loss = -(off * sum_logp + on * (lam * lp1 + (1.f - lam) * lp2));
  if (lane == 0) {
    loss_rows[row] = loss;
    if (pred != nullptr) pred[row] = amax;
  }
  if (dlogits != nullptr) {
    bf16_t* d = dlogits + (long long)row * ld;
    const float inv = 1.f / se;
    for (int c = lane; c < ld; c += 64) {
      float g = 0.f;
      if (c < C) {
        const float pc = __expf(bf16_to_f32(x[c]) - mx) * inv;
        float t = off;
        if (c == t1) t += on * lam;
        if (c == t2) t += on * (1.f - lam);
        g = (pc - t) * gscale;
      }
      d[c] = f32_to_bf16(g);
    }
  }
}

// Per-step bookkeeping (single workgroup): mean loss in fixed order, finiteness flag, and -- unless the
// step is skipped for a non-finite loss -- metric accumulation. state layout (doubles/ints) is in icamd.h.
//   acc_f64[0] += loss (per-step, pre-division), acc_f64[1] += 1 (steps counted)
//   acc_f64[2] += correct/B (class_acc of this step); acc_f64[3] += correct ; acc_f64[4] += B
//   counts[3][C]: TP, FP, FN
__global__ __launch_bounds__(256) void step_metrics_kernel(const float* __restrict__ loss_rows,
                                                           const int* __restrict__ pred,
                                                           const long long* __restrict__ target, int B, int C,
                                                           float* __restrict__ loss_out, int* __restrict__ finite_out,
                                                           double* __restrict__ acc_f64, int* __restrict__ counts,
                                                           float* __restrict__ loss_log, int log_slot, int log_stride,
                                                           int respect_skip) {
  __shared__ double red[256];
  __shared__ int cred[256];
  __shared__ int s_finite;
  const bool have_loss = loss_rows != nullptr;
  if (have_loss) {
    double s = 0.0;
    for (int i = threadIdx.x; i < B; i += 256) s += (double)loss_rows[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const float mean = (float)(red[0] / (double)B);
      const int fin = isfinite(mean) ? 1 : 0;
      *loss_out = mean;
      *finite_out = fin;
      s_finite = fin;
      if (loss_log != nullptr) loss_log[log_slot] = mean;
    }
  } else if (threadIdx.x == 0) {
    s_finite = *finite_out;   // metrics-only call: honour the flag the loss call of this step left behind
  }
  __syncthreads();
  if (respect_skip & 2) return;                    // flag call: loss, flag and log only; accumulation is a later call's
  if ((respect_skip & 1) && !s_finite) return;
  const bool add_loss = have_loss || (respect_skip & 4);   // bit 2: the deferred accumulation of an earlier flag call
  if (pred == nullptr) {
    if (threadIdx.x == 0 && add_loss) { acc_f64[0] += (double)*loss_out; acc_f64[1] += 1.0; }
    return;
  }
  int correct = 0;
  for (int i = threadIdx.x; i < B; i += 256) {
    const int p = pred[i];
    const int t = (int)target[i];
    if (p == t) {
      ++correct;
      if (counts != nullptr && t >= 0 && t < C) atomicAdd(&counts[t], 1);            // TP[t]
    } else if (counts != nullptr) {
      if (p >= 0 && p < C) atomicAdd(&counts[C + p], 1);                              // FP[pred]
      if (t >= 0 && t < C) atomicAdd(&counts[2 * C + t], 1);                          // FN[target]
    }
  }
  cred[threadIdx.x] = correct;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) cred[threadIdx.x] += cred[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (add_loss) { acc_f64[0] += (double)*loss_out; acc_f64[1] += 1.0; }
    acc_f64[2] += (double)((float)cred[0] / (float)B);
    if (loss_log != nullptr && log_stride > 0) loss_log[log_stride + log_slot] = (float)cred[0] / (float)B;
    acc_f64[3] += (double)cred[0];
    acc_f64[4] += (double)B;
  }
}

// ------------------------------------------------------------------------------------------------
// sum of squares of a flat fp32 arena -> partial[blk] (double); then norm + clip coefficient
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long long n4, long long n,
                                                            double* __restrict__ partial) {
  __shared__ double red[4];
  const long long stride = (long long)gridDim.x * blockDim.x;
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 v = ((const f32x4*)g)[i];
    s += (double)(v[0] * v[0] + v[1] * v[1]) + (double)(v[2] * v[2] + v[3] * v[3]);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long long i = n4 * 4; i < n; ++i) s += (double)g[i] * (double)g[i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// out[0] = norm * inv_scale ; out[1] = clip coefficient (torch clip_grad_norm_: min(1, max_norm/(norm+1e-6)))
__global__ void gradnorm_finalize_kernel(const double* __restrict__ partial, int nblk, float inv_scale, float max_norm,
                                         float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0;
  for (int i = 0; i < nblk; ++i) s += partial[i];
  const float norm = (float)sqrt(s) * inv_scale;
  out[0] = norm;
  float coef = 1.f;
  if (max_norm > 0.f) { coef = max_norm / (norm + 1e-6f); if (coef > 1.f) coef = 1.f; }
  out[1] = coef;
}

// ------------------------------------------------------------------------------------------------
// fused AdamW (torch.optim.AdamW single-tensor op order) + EMA lerp + bf16 shadow write
//   g' = g * gscale * (clip ? clip[1] : 1)
//   p *= 1 - lr*wd ; m += (g'-m)*(1-b1) ; v = v*b2 + (1-b2)*g'^2
//   p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
//   ema += (p - ema)*(1-decay)          (timm ModelEmaV3.update: lerp_(model_v, 1-decay))
// skipped entirely when *finite_flag == 0 (reference: non-finite loss -> no step, engine.py:56-59)
// ------------------------------------------------------------------------------------------------
struct AdamArgs {
  float lr, wd, beta1, beta2, eps, gscale, ema_w;
  int step;          // optimizer steps ATTEMPTED so far, this one included (host count)
  int bias_correct;  // Adam family: bc1 = 1 - beta1^t, bc2 = 1 - beta2^t with t = step - *skipped (steps really taken)
};

// Steps dropped for a non-finite loss never happened as far as torch.optim's `step` state is concerned (the reference
// `continue`s before optimizer.step(), engine.py:56-59), so the bias correction uses the device-side count of steps that
// were really taken.  One double pow per THREAD (not per element): invisible next to 36 B/parameter of traffic.
__device__ __forceinline__ void bias_corrections(const AdamArgs& a, const int* __restrict__ skipped, float& bc1,
                                                 float& bc2_sqrt) {
  bc1 = 1.f; bc2_sqrt = 1.f;
  if (a.bias_correct) {
    int t = a.step - (skipped != nullptr ? *skipped : 0);
    if (t < 1) t = 1;
    bc1 = (float)(1.0 - pow((double)a.beta1, (double)t));
    bc2_sqrt = (float)sqrt(1.0 - pow((double)a.beta2, (double)t));
  }
}
// the skip path of every optimizer kernel: one thread counts the dropped step (nobody reads the counter in this launch)
// `flags` bit 1 (ICAMD_OPT_NO_SKIP_COUNT): this launch covers one range of a step that several launches apply (one per
// gradient bucket); only the launch without the bit counts the dropped step.
__device__ __forceinline__ bool step_skipped(const int* __restrict__ finite_flag, int* __restrict__ skipped, int flags) {
  if (finite_flag == nullptr || *finite_flag != 0) return false;
  if (skipped != nullptr && !(flags & 2) && blockIdx.x == 0 && threadIdx.x == 0) *skipped += 1;
  return true;
}

__global__ __launch_bounds__(256) void adamw_ema_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, float* __restrict__ ema,
                                                        bf16_t* __restrict__ shadow, long long n4, AdamArgs a,
                                                        const float* __restrict__ clip, const int* __restrict__ finite_flag,
                                                        int* __restrict__ skipped, int zero_grad) {
  if (step_skipped(finite_flag, skipped, zero_grad)) return;
  float bc1, bc2_sqrt;
  bias_corrections(a, skipped, bc1, bc2_sqrt);
  const float gs = a.gscale * (clip != nullptr ? clip[1] : 1.f);
  const float decay_mul = 1.f - a.lr * a.wd;
  const float step_size = a.lr / bc1;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 pv = ld_stream((const f32x4*)p + i);
    f32x4 gv = ld_stream((const f32x4*)g + i);
    f32x4 mv = ld_stream((const f32x4*)m + i);
    f32x4 vv = ld_stream((const f32x4*)v + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gg = gv[e] * gs;
      float pp = pv[e] * decay_mul;
      const float mm = mv[e] + (gg - mv[e]) * (1.f - a.beta1);
      const float v2 = vv[e] * a.beta2 + (1.f - a.beta2) * gg * gg;
      const float denom = sqrtf(v2) / bc2_sqrt + a.eps;
      pp = pp - step_size * (mm / denom);
      pv[e] = pp; mv[e] = mm; vv[e] = v2;
    }
    ((f32x4*)p)[i] = pv;
    ((f32x4*)m)[i] = mv;
    ((f32x4*)v)[i] = vv;
    if (zero_grad & 1) ((f32x4*)g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ema != nullptr) {
      f32x4 ev = ld_stream((const f32x4*)ema + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) ev[e] = ev[e] + a.ema_w * (pv[e] - ev[e]);
      ((f32x4*)ema)[i] = ev;
    }
    if (shadow != nullptr) {
      u32x2 sv;
      sv[0] = pack_bf16x2(pv[0], pv[1]);
      sv[1] = pack_bf16x2(pv[2], pv[3]);
      ((u32x2*)shadow)[i] = sv;
    }
  }
}

// optimizer.zero_grad() of the reference's non-finite branch (engine.py:56-59), decided on the device: the gradient arena
// is cleared iff *finite_flag == 0 (the micro-step that just accumulated into it had a non-finite loss).  With the flag set
// every workgroup leaves after one scalar load.
__global__ __launch_bounds__(256) void grad_guard_kernel(float* __restrict__ g, long long n4, const int* __restrict__ finite_flag) {
  if (*finite_flag != 0) return;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
    ((f32x4*)g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// dst += w*(src-dst)  (EMA of float buffers) ; with w == 1 a plain copy
__global__ __launch_bounds__(256) void lerp_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n,
                                                   float w, const int* __restrict__ finite_flag) {
  if (finite_flag != nullptr && *finite_flag == 0) return;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dst[i] = dst[i] + w * (src[i] - dst[i]);
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                          long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = f32_to_bf16(src[i]);
}

// ------------------------------------------------------------------------------------------------
// batched filter transpose for the data-gradient kernels: [Cout][T][Cin] -> [Cin][T][Cout], table driven.
// job j: layer jobs[j].x, destination elements [jobs[j].y, +4096) of that layer
// desc per layer (8 int64): src_off, dst_off, Cout, T, Cin, (unused x3); offsets in elements of `shadow`/`dst`
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void filter_transpose_kernel(const bf16_t* __restrict__ src_base,
                                                               bf16_t* __restrict__ dst_base,
                                                               const long long* __restrict__ descs,
                                                               const int* __restrict__ jobs) {
  const int layer = jobs[blockIdx.x * 2 + 0];
  const int start = jobs[blockIdx.x * 2 + 1];
  const long long* d = descs + (long long)layer * 8;
  const bf16_t* src = src_base + d[0];
  bf16_t* dst = dst_base + d[1];
  const int Cout = (int)d[2], T = (int)d[3], Cin = (int)d[4];
  const int total = Cout * T * Cin;
  for (int k = 0; k < 16; ++k) {
    const int o = start + k * 256 + threadIdx.x;   // dst index = (ci*T + t)*Cout + co
    if (o < total) {
      const int co = o % Cout;
      const int rest = o / Cout;
      const int t = rest % T;
      const int ci = rest / T;
      dst[o] = src[((long long)co * T + t) * Cin + ci];
    }
  }
}

// Tiled variant for layers with Cout % 64 == 0 and Cin % 64 == 0 (every ResNet filter but the stem): one workgroup
// moves a 64(co) x 64(ci) tile of one tap through LDS so that both the reads and the writes are 128 B row segments.
// jobs: int32[njobs][4] = {layer, tap, co0, ci0}
__global__ __launch_bounds__(256) void filter_transpose_tiled_kernel(const bf16_t* __restrict__ src_base,
                                                                     bf16_t* __restrict__ dst_base,
                                                                     const long long* __restrict__ descs,
                                                                     const int* __restrict__ jobs) {
  __shared__ bf16_t tile[64][72];   // [co][ci], rows padded to 144 B
  const int* job = jobs + blockIdx.x * 4;
  const long long* d = descs + (long long)job[0] * 8;
  const int t = job[1], co0 = job[2], ci0 = job[3];
  const int Cout = (int)d[2], T = (int)d[3], Cin = (int)d[4];
  const bf16_t* src = src_base + d[0];
  bf16_t* dst = dst_base + d[1];
  const int r = threadIdx.x >> 3, c8 = (threadIdx.x & 7) * 8;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int co = r + pass * 32;
    const u32x4 v = *(const u32x4*)(src + ((long long)(co0 + co) * T + t) * Cin + ci0 + c8);
    *(u32x4*)&tile[co][c8] = v;
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int ci = r + pass * 32;
    unsigned int w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      w[e] = (unsigned int)tile[c8 + 2 * e][ci] | ((unsigned int)tile[c8 + 2 * e + 1][ci] << 16);
    *(u32x4*)(dst + ((long long)(ci0 + ci) * T + t) * Cout + co0 + c8) = u32x4{w[0], w[1], w[2], w[3]};
  }
}

// column sums of a bf16 matrix [rows][ld] -> fp32 [cols] (FC bias gradient); one thread per column
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, int rows, int ld, int cols,
                                                     float* __restrict__ out, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  for (int r = 0; r < rows; ++r) s += bf16_to_f32(x[(long long)r * ld + c]);
  out[c] = accumulate ? out[c] + s : s;
}

inline unsigned int grid_for(long long work_items, int threads) {
  long long blocks = (work_items + threads - 1) / threads;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  return (unsigned int)blocks;
}

}  // namespace

int icamd_softmax_xent_launch(const bf16_t* logits, int ld, int B, int C, const long long* y1, const long long* y2,
                              float lam, float smoothing, float gscale, float* loss_rows, int* pred, bf16_t* dlogits,
                              hipStream_t s) {
  if (B <= 0 || C <= 0 || ld < C) return ICAMD_ERR_BAD_ARG;
  hipLaunchKernelGGL(softmax_xent_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s, logits, ld, B, C, y1, y2, lam,
                     smoothing, gscale, loss_rows, pred, dlogits);
  return icamd_launch_status();
}

int icamd_step_metrics_launch(const float* loss_rows, const int* pred, const long long* target, int B, int C,
                              float* loss_out, int* finite_out, double* acc_f64, int* counts, float* loss_log,
                              int log_slot, int log_stride, int respect_skip, hipStream_t s) {
  hipLaunchKernelGGL(step_metrics_kernel, dim3(1), dim3(256), 0, s, loss_rows, pred, target, B, C, loss_out, finite_out,
                     acc_f64, counts, loss_log, log_slot, log_stride, respect_skip);
  return icamd_launch_status();
}

#define ICAMD_SUMSQ_BLOCKS 512
int icamd_grad_norm_launch(const float* g, long long n, float inv_scale, float max_norm, double* partial, float* out,
                           hipStream_t s) {
  const long long n4 = n / 4;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(ICAMD_SUMSQ_BLOCKS), dim3(256), 0, s, g, n4, n, partial);
  hipLaunchKernelGGL(gradnorm_finalize_kernel, dim3(1), dim3(64), 0, s, partial, ICAMD_SUMSQ_BLOCKS, inv_scale, max_norm,
                     out);
  return icamd_launch_status();
}

int icamd_adamw_ema_launch(float* p, float* g, float* m, float* v, float* ema, bf16_t* shadow, long long n, float lr,
                           float wd, float beta1, float beta2, float eps, int step, float gscale, float ema_decay,
                           const float* clip, const int* finite_flag, int* skipped, int zero_grad, hipStream_t s) {
  if (n % 4 != 0 || step < 1) return ICAMD_ERR_BAD_ARG;
  AdamArgs a;
  a.lr = lr; a.wd = wd; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.gscale = gscale;
  a.step = step; a.bias_correct = 1;
  a.ema_w = 1.f - ema_decay;
  hipLaunchKernelGGL(adamw_ema_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, s, p, g, m, v, ema, shadow, n / 4, a, clip,
                     finite_flag, skipped, zero_grad);
  return icamd_launch_status();
}

int icamd_grad_guard_launch(float* g, long long n, const int* finite_flag, hipStream_t s) {
  if (n % 4 != 0) return ICAMD_ERR_BAD_ARG;
  hipLaunchKernelGGL(grad_guard_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, s, g, n / 4, finite_flag);
  return icamd_launch_status();
}

// The reference's other optimizers with trivial fused forms (optim_factory.py:66-77): torch.optim.SGD (momentum 0.9,
// Nesterov or not, coupled weight decay), torch.optim.Adam (coupled weight decay) and timm Lion (decoupled decay, sign
// update).  Same fusion as adamw_ema_kernel: gradient scale / clip, NaN-skip flag, EMA lerp, bf16 shadow, zero_grad.
// KIND: 1 = Adam, 2 = SGD with momentum, 3 = SGD with Nesterov momentum, 4 = Lion.
template <int KIND>
__global__ __launch_bounds__(256) void optim_ema_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, float* __restrict__ ema,
                                                        bf16_t* __restrict__ shadow, long long n4, AdamArgs a,
                                                        const float* __restrict__ clip, const int* __restrict__ finite_flag,
                                                        int* __restrict__ skipped, int zero_grad) {
  if (step_skipped(finite_flag, skipped, zero_grad)) return;
  float bc1, bc2_sqrt;
  bias_corrections(a, skipped, bc1, bc2_sqrt);
  const float gs = a.gscale * (clip != nullptr ? clip[1] : 1.f);
  const float step_size = a.lr / bc1;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 pv = ((f32x4*)p)[i];
    const f32x4 gv = ((const f32x4*)g)[i];
    f32x4 mv = ((f32x4*)m)[i];
    f32x4 vv = {0.f, 0.f, 0.f, 0.f};
    if (KIND == 1) vv = ((f32x4*)v)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gg = gv[e] * gs;
      if (KIND == 1) {          // torch.optim.Adam: L2 term joins the gradient
        gg += a.wd * pv[e];
        const float mm = mv[e] + (gg - mv[e]) * (1.f - a.beta1);
        const float v2 = vv[e] * a.beta2 + (1.f - a.beta2) * gg * gg;
        pv[e] -= step_size * (mm / (sqrtf(v2) / bc2_sqrt + a.eps));
        mv[e] = mm; vv[e] = v2;
      } else if (KIND == 2 || KIND == 3) {   // torch.optim.SGD, dampening 0; a zero buffer reproduces buf = g at step 1
        gg += a.wd * pv[e];
        const float buf = a.beta1 * mv[e] + gg;
        pv[e] -= a.lr * (KIND == 3 ? gg + a.beta1 * buf : buf);
        mv[e] = buf;
      } else {                  // Lion: p *= 1 - lr wd; p -= lr sign(b1 m + (1-b1) g); m = b2 m + (1-b2) g
        const float u = mv[e] * a.beta1 + gg * (1.f - a.beta1);
        const float sg = u > 0.f ? 1.f : (u < 0.f ? -1.f : 0.f);
        pv[e] = pv[e] * (1.f - a.lr * a.wd) - a.lr * sg;
        mv[e] = mv[e] + (gg - mv[e]) * (1.f - a.beta2);
      }
    }
    ((f32x4*)p)[i] = pv;
    ((f32x4*)m)[i] = mv;
    if (KIND == 1) ((f32x4*)v)[i] = vv;
    if (zero_grad & 1) ((f32x4*)g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ema != nullptr) {
      f32x4 ev = ld_stream((const f32x4*)ema + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) ev[e] = ev[e] + a.ema_w * (pv[e] - ev[e]);
      ((f32x4*)ema)[i] = ev;
    }
    if (shadow != nullptr) {
      u32x2 sv;
      sv[0] = pack_bf16x2(pv[0], pv[1]);
      sv[1] = pack_bf16x2(pv[2], pv[3]);
      ((u32x2*)shadow)[i] = sv;
    }
  }
}

int icamd_optim_ema_launch(int kind, float* p, float* g, float* m, float* v, float* ema, bf16_t* shadow, long long n,
                           float lr, float wd, float beta1, float beta2, float eps, int step, float gscale,
                           float ema_decay, const float* clip, const int* finite_flag, int* skipped, int zero_grad,
                           hipStream_t s) {
  if (kind == 0) return icamd_adamw_ema_launch(p, g, m, v, ema, shadow, n, lr, wd, beta1, beta2, eps, step, gscale, ema_decay,
                                               clip, finite_flag, skipped, zero_grad, s);
  if (n % 4 != 0 || step < 1 || kind < 0 || kind > 4 || (kind == 1 && v == nullptr)) return ICAMD_ERR_BAD_ARG;
  AdamArgs a;
  a.lr = lr; a.wd = wd; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.gscale = gscale;
  a.step = step; a.bias_correct = kind == 1 ? 1 : 0;
  a.ema_w = 1.f - ema_decay;
  const dim3 grid(grid_for(n / 4, 256)), block(256);
  if (kind == 1) hipLaunchKernelGGL(optim_ema_kernel<1>, grid, block, 0, s, p, g, m, v, ema, shadow, n / 4, a, clip, finite_flag, skipped, zero_grad);
  else if (kind == 2) hipLaunchKernelGGL(optim_ema_kernel<2>, grid, block, 0, s, p, g, m, v, ema, shadow, n / 4, a, clip, finite_flag, skipped, zero_grad);
  else if (kind == 3) hipLaunchKernelGGL(optim_ema_kernel<3>, grid, block, 0, s, p, g, m, v, ema, shadow, n / 4, a, clip, finite_flag, skipped, zero_grad);
  else hipLaunchKernelGGL(optim_ema_kernel<4>, grid, block, 0, s, p, g, m, v, ema, shadow, n / 4, a, clip, finite_flag, skipped, zero_grad);
  return icamd_launch_status();
}

int icamd_lerp_launch(float* dst, const float* src, long long n, float w, const int* finite_flag, hipStream_t s) {
  hipLaunchKernelGGL(lerp_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, dst, src, n, w, finite_flag);
  return icamd_launch_status();
}

int icamd_f32_to_bf16_launch(const float* src, bf16_t* dst, long long n, hipStream_t s) {
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, src, dst, n);
  return icamd_launch_status();
}

int icamd_filter_transpose_launch(const bf16_t* src_base, bf16_t* dst_base, const long long* descs, const int* jobs,
                                  int njobs, hipStream_t s) {
  if (njobs <= 0) return ICAMD_OK;
  hipLaunchKernelGGL(filter_transpose_kernel, dim3((unsigned)njobs), dim3(256), 0, s, src_base, dst_base, descs, jobs);
  return icamd_launch_status();
}

int icamd_filter_transpose_tiled_launch(const bf16_t* src_base, bf16_t* dst_base, const long long* descs, const int* jobs,
                                        int njobs, hipStream_t s) {
  if (njobs <= 0) return ICAMD_OK;
  hipLaunchKernelGGL(filter_transpose_tiled_kernel, dim3((unsigned)njobs), dim3(256), 0, s, src_base, dst_base, descs, jobs);
  return icamd_launch_status();
}

int icamd_colsum_launch(const bf16_t* x, int rows, int ld, int cols, float* out, int accumulate, hipStream_t s) {
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, s, x, rows, ld, cols, out,
                     accumulate);
  return icamd_launch_status();
}
