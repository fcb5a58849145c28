// Token-wise kernels for the transformer / ConvNeXt branches of the step (gfx950): LayerNorm forward / backward over
// the channel dimension of [rows][C] bf16 tensors, exact (erf) GELU forward / backward, and a two-level column sum
// (bias gradients).  One wavefront per row for LayerNorm: 8 B per lane per 256-channel slice, statistics by
// wave shuffles in fp32 (two-pass variance, as torch.nn.functional.layer_norm), fp32 mean / rstd saved for backward.
// Replaces what ATen runs for timm's LayerNorm / GELU / Linear-bias layers under `model(samples)` and
// `loss.backward()` (/root/reference/engine.py:48,51,64,72; ConvNeXt block spec
// /root/reference/semantic_segmentation/backbone/convnext.py:43-56,158-182).
#include "common.h"
#include "icamd_internal.h"

namespace {

constexpr int LN_MAX_IT = 4;   // C <= 1024

__device__ __forceinline__ void unpack4(u32x2 v, float* f) {
  f[0] = bf16_lo(v[0]); f[1] = bf16_hi(v[0]); f[2] = bf16_lo(v[1]); f[3] = bf16_hi(v[1]);
}

// y = (x - mean) * rstd * gamma + beta ; grid = ceil(rows / 4), block 256 (one wave per row)
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            long long rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + row * C;
  float v[LN_MAX_IT][4];
  const int nit = (C + 255) / 256;
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < LN_MAX_IT; ++it) {
    if (it < nit) {
      const int c = it * 256 + lane * 4;
      if (c < C) { unpack4(*(const u32x2*)(xr + c), v[it]); s += (v[it][0] + v[it][1]) + (v[it][2] + v[it][3]); }
      else { v[it][0] = v[it][1] = v[it][2] = v[it][3] = 0.f; }
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int it = 0; it < LN_MAX_IT; ++it) {
    if (it < nit) {
      const int c = it * 256 + lane * 4;
      if (c < C) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[it][e] - mean; q += d * d; }
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
  bf16_t* yr = y + row * C;
#pragma unroll
  for (int it = 0; it < LN_MAX_IT; ++it) {
    if (it < nit) {
      const int c = it * 256 + lane * 4;
      if (c < C) {
        const f32x4 g = *(const f32x4*)(gamma + c);
        const f32x4 b = *(const f32x4*)(beta + c);
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[it][e] - mean) * rstd * g[e] + b[e];
        u32x2 pk;
        pk[0] = pack_bf16x2(o[0], o[1]);
        pk[1] = pack_bf16x2(o[2], o[3]);
        *(u32x2*)(yr + c) = pk;
      }
    }
  }
}

// dx = rstd * (dy*gamma - mean_C(dy*gamma) - xhat * mean_C(dy*gamma*xhat)); per-workgroup partial column sums of
// dy (-> dbeta) and dy*xhat (-> dgamma) in part[blk][2][C].  Each wave walks `rows_per_wave` rows.
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const bf16_t* __restrict__ addend,
                                                            bf16_t* __restrict__ dx, float* __restrict__ part, long long rows,
                                                            int C, int rows_per_wave) {
  __shared__ float red[4][2][LN_MAX_IT * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nit = (C + 255) / 256;
  float sb[LN_MAX_IT][4], sg[LN_MAX_IT][4];
#pragma unroll
  for (int it = 0; it < LN_MAX_IT; ++it)
#pragma unroll
    for (int e = 0; e < 4; ++e) { sb[it][e] = 0.f; sg[it][e] = 0.f; }
  const long long r0 = ((long long)blockIdx.x * 4 + wave) * rows_per_wave;
  const long long r1 = (rows < r0 + rows_per_wave) ? rows : r0 + rows_per_wave;
  for (long long row = r0; row < r1; ++row) {
    const float mu = mean[row], rs = rstd[row];
    float dyg[LN_MAX_IT][4], xh[LN_MAX_IT][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < LN_MAX_IT; ++it) {
      if (it < nit) {
        const int c = it * 256 + lane * 4;
        if (c < C) {
          float d[4], xv[4];
          unpack4(*(const u32x2*)(dy + row * C + c), d);
          unpack4(*(const u32x2*)(x + row * C + c), xv);
          const f32x4 g = *(const f32x4*)(gamma + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            xh[it][e] = (xv[e] - mu) * rs;
            dyg[it][e] = d[e] * g[e];
            s1 += dyg[it][e];
            s2 += dyg[it][e] * xh[it][e];
            sb[it][e] += d[e];
            sg[it][e] += d[e] * xh[it][e];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) { xh[it][e] = 0.f; dyg[it][e] = 0.f; }
        }
      }
    }
    const float c1 = wave_sum(s1) / (float)C, c2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int it = 0; it < LN_MAX_IT; ++it) {
      if (it < nit) {
        const int c = it * 256 + lane * 4;
        if (c < C) {
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = rs * (dyg[it][e] - c1 - xh[it][e] * c2);
          if (addend != nullptr) {   // residual branch: the skip connection's gradient joins here
            float a4[4];
            unpack4(*(const u32x2*)(addend + row * C + c), a4);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += a4[e];
          }
          u32x2 pk;
          pk[0] = pack_bf16x2(o[0], o[1]);
          pk[1] = pack_bf16x2(o[2], o[3]);
          *(u32x2*)(dx + row * C + c) = pk;
        }
      }
    }
  }
  // fold the 4 waves' column sums in wave order, one partial row per workgroup
#pragma unroll
  for (int it = 0; it < LN_MAX_IT; ++it) {
    if (it < nit) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[wave][0][it * 256 + lane * 4 + e] = sb[it][e];
        red[wave][1][it * 256 + lane * 4 + e] = sg[it][e];
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int which = i / C, c = i - which * C;
    const float s = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
    part[((long long)blockIdx.x * 2 + which) * C + c] = s;
  }
}


__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16_t* __restrict__ z, bf16_t* __restrict__ a, long long nvec) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    const u32x4 v = ((const u32x4*)z)[i];
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(gelu_f(bf16_lo(v[e])), gelu_f(bf16_hi(v[e])));
    ((u32x4*)a)[i] = o;
  }
}

// dz = da * gelu'(z); optional per-workgroup column sums of dz (the preceding Linear's bias gradient):
// part[blk][2][C] (second row zero) -- the launcher makes the grid stride a multiple of C/8
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* __restrict__ da, const bf16_t* __restrict__ z,
                                                       bf16_t* __restrict__ dz, long long nvec) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    const u32x4 g = ((const u32x4*)da)[i];
    const u32x4 v = ((const u32x4*)z)[i];
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      o[e] = pack_bf16x2(bf16_lo(g[e]) * gelu_grad_f(bf16_lo(v[e])), bf16_hi(g[e]) * gelu_grad_f(bf16_hi(v[e])));
    ((u32x4*)dz)[i] = o;
  }
}

// column sums of x [rows][ld] (first `cols` columns, cols % 8 == 0): per-workgroup partial rows part[blk][2][cols]
// (second row zero so the BatchNorm reduce+finalize kernel can fold them)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16_t* __restrict__ x, float* __restrict__ part,
                                                             long long rows, int ld, int cols, int rows_per_block) {
  __shared__ float red[256 * 8];
  const int cpr = cols >> 3;
  const int tid = threadIdx.x;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = (rows < r0 + rows_per_block) ? rows : r0 + rows_per_block;
  for (int cg0 = 0; cg0 < cpr; cg0 += 256) {
    const int tcols = (cpr - cg0 < 256) ? (cpr - cg0) : 256;
    const int rlanes = 256 / tcols;
    const int cgi = tid % tcols, rl = tid / tcols;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    if (rl < rlanes) {
      for (long long r = r0 + rl; r < r1; r += rlanes) {
        const u32x4 v = *(const u32x4*)(x + r * ld + (cg0 + cgi) * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[2 * e] += bf16_lo(v[e]); s[2 * e + 1] += bf16_hi(v[e]); }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[tid * 8 + e] = s[e];
    __syncthreads();
    for (int o = tid; o < tcols * 8; o += 256) {
      const int cgo = o >> 3, e = o & 7;
      float t = 0.f;
      for (int l = 0; l < rlanes; ++l) t += red[(l * tcols + cgo) * 8 + e];
      part[((long long)blockIdx.x * 2 + 0) * cols + (cg0 + cgo) * 8 + e] = t;
      part[((long long)blockIdx.x * 2 + 1) * cols + (cg0 + cgo) * 8 + e] = 0.f;
    }
    __syncthreads();
  }
}

// ViT token assembly: tok[b][0] = cls + pos[0]; tok[b][1+i] = patch[b][i] + pos[1+i]   (C % 8 == 0)
__global__ __launch_bounds__(256) void vit_tokens_fwd_kernel(const bf16_t* __restrict__ patches, const float* __restrict__ cls,
                                                             const float* __restrict__ pos, bf16_t* __restrict__ tok, int B,
                                                             int T, int C) {
  const int cpr = C >> 3;
  const long long total = (long long)B * T * cpr;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int cg = (int)(i % cpr);
    const long long row = i / cpr;
    const int t = (int)(row % T);
    const int b = (int)(row / T);
    float v[8];
    if (t == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = cls[cg * 8 + e];
    } else {
      const u32x4 p = ((const u32x4*)patches)[((long long)b * (T - 1) + (t - 1)) * cpr + cg];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[2 * e] = bf16_lo(p[e]); v[2 * e + 1] = bf16_hi(p[e]); }
    }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      o[e] = pack_bf16x2(v[2 * e] + pos[(long long)t * C + cg * 8 + 2 * e], v[2 * e + 1] + pos[(long long)t * C + cg * 8 + 2 * e + 1]);
    ((u32x4*)tok)[i] = o;
  }
}

// out[j] = (accumulate ? out[j] : 0) + sum_b x[b*stride + j], j < n (n % 8 == 0): gradients of cls_token / pos_embed
__global__ __launch_bounds__(256) void batch_sum_kernel(const bf16_t* __restrict__ x, long long stride, int B, long long n8,
                                                        float* __restrict__ out, int accumulate) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  for (int b = 0; b < B; ++b) {
    const u32x4 v = *(const u32x4*)(x + (long long)b * stride + i * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) { s[2 * e] += bf16_lo(v[e]); s[2 * e + 1] += bf16_hi(v[e]); }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) out[i * 8 + e] = accumulate ? out[i * 8 + e] + s[e] : s[e];
}

// dst[r][0..C) = src[r][0..C) for r < rows, with independent row strides (elements; C % 8 == 0)
__global__ __launch_bounds__(256) void strided_rows_copy_kernel(const bf16_t* __restrict__ src, long long sstride,
                                                                bf16_t* __restrict__ dst, long long dstride, long long rows,
                                                                long long c8) {
  const long long total = rows * c8;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long r = i / c8, c = i - r * c8;
    *(u32x4*)(dst + r * dstride + c * 8) = *(const u32x4*)(src + r * sstride + c * 8);
  }
}

inline unsigned int ew_grid(long long nvec) {
  long long blocks = (nvec + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  return (unsigned int)blocks;
}

}  // namespace

int icamd_layernorm_fwd_launch(const bf16_t* x, const float* gamma, const float* beta, bf16_t* y, float* mean, float* rstd,
                               long long rows, int C, float eps, hipStream_t s) {
  if (C % 4 != 0 || C > LN_MAX_IT * 256) return ICAMD_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, gamma, beta, y, mean, rstd,
                     rows, C, eps);
  return icamd_launch_status();
}

int icamd_layernorm_bwd_blocks(long long rows) {
  // ~1024 workgroups of 4 waves, at least one row per wave
  long long waves = (rows < 4096) ? rows : 4096;
  return (int)((waves + 3) / 4);
}

int icamd_layernorm_bwd_launch(const bf16_t* dy, const bf16_t* x, const float* mean, const float* rstd, const float* gamma,
                               const bf16_t* addend, bf16_t* dx, float* part, long long rows, int C, hipStream_t s) {
  if (C % 4 != 0 || C > LN_MAX_IT * 256) return ICAMD_ERR_UNSUPPORTED;
  const int nblk = icamd_layernorm_bwd_blocks(rows);
  const int rpw = (int)((rows + (long long)nblk * 4 - 1) / ((long long)nblk * 4));
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)nblk), dim3(256), 0, s, dy, x, mean, rstd, gamma, addend, dx, part,
                     rows, C, rpw);
  return icamd_launch_status();
}

int icamd_gelu_fwd_launch(const bf16_t* z, bf16_t* a, long long numel, hipStream_t s) {
  if (numel % 8 != 0) return ICAMD_ERR_BAD_ARG;
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(ew_grid(numel / 8)), dim3(256), 0, s, z, a, numel / 8);
  return icamd_launch_status();
}

int icamd_gelu_bwd_launch(const bf16_t* da, const bf16_t* z, bf16_t* dz, long long numel, hipStream_t s) {
  if (numel % 8 != 0) return ICAMD_ERR_BAD_ARG;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(ew_grid(numel / 8)), dim3(256), 0, s, da, z, dz, numel / 8);
  return icamd_launch_status();
}

int icamd_colsum_blocks(long long rows) {
  long long rpb = (rows + 1023) / 1024;
  if (rpb < 32) rpb = 32;
  return (int)((rows + rpb - 1) / rpb);
}

int icamd_colsum_partial_launch(const bf16_t* x, float* part, long long rows, int ld, int cols, hipStream_t s) {
  if (cols % 8 != 0 || ld % 8 != 0) return ICAMD_ERR_BAD_ARG;
  const int nblk = icamd_colsum_blocks(rows);
  const int rpb = (int)((rows + nblk - 1) / nblk);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)nblk), dim3(256), 0, s, x, part, rows, ld, cols, rpb);
  return icamd_launch_status();
}

int icamd_vit_tokens_fwd_launch(const bf16_t* patches, const float* cls, const float* pos, bf16_t* tok, int B, int T, int C,
                                hipStream_t s) {
  if (C % 8 != 0) return ICAMD_ERR_BAD_ARG;
  hipLaunchKernelGGL(vit_tokens_fwd_kernel, dim3(ew_grid((long long)B * T * (C / 8))), dim3(256), 0, s, patches, cls, pos, tok, B,
                     T, C);
  return icamd_launch_status();
}

int icamd_batch_sum_launch(const bf16_t* x, long long stride, int B, long long n, float* out, int accumulate, hipStream_t s) {
  if (n % 8 != 0 || stride % 8 != 0) return ICAMD_ERR_BAD_ARG;
  const long long n8 = n / 8;
  hipLaunchKernelGGL(batch_sum_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, x, stride, B, n8, out, accumulate);
  return icamd_launch_status();
}

int icamd_strided_rows_copy_launch(const bf16_t* src, long long sstride, bf16_t* dst, long long dstride, long long rows,
                                   long long C, hipStream_t s) {
  if (C % 8 != 0 || sstride % 8 != 0 || dstride % 8 != 0) return ICAMD_ERR_BAD_ARG;
  hipLaunchKernelGGL(strided_rows_copy_kernel, dim3(ew_grid(rows * (C / 8))), dim3(256), 0, s, src, sstride, dst, dstride, rows,
                     C / 8);
  return icamd_launch_status();
}
