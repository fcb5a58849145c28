// HBM-bound kernels of the ResNet step for gfx950: BatchNorm (training statistics, apply, backward),
// ReLU / residual add fused into the BN apply, 3x3/s2 max-pool, global average pool, input packing.
// All activations are NHWC bf16 moved as 16 B (8-channel) vectors; statistics are fp32 per thread,
// fp64 across workgroups, combined in a fixed order (bitwise reproducible, no float atomics).
//
// Replaces what ATen runs for timm's BatchNorm2d / ReLU / MaxPool2d / pooling layers under
// `model(samples)` and `loss.backward()` in /root/reference/engine.py:48,51,64,72.
#include "common.h"
#include "icamd_internal.h"
#include <stdlib.h>
#include <string.h>

// Cache policy of the streaming 16 B loads / stores of the BatchNorm passes: bit 0 = non-temporal loads, bit 1 = non-temporal stores
// (compile-time: the product value is set below; tools/r5_bn_nt_variants.sh builds the others for A/B runs).
#ifndef ICAMD_BN_NT
#define ICAMD_BN_NT 1   // round 5: non-temporal LOADS (ResNet-50 17.94-17.99 -> 17.68-17.76 ms, two A/B pairs on one box); non-temporal stores lost (18.0-18.1)
#endif

namespace {

__device__ __forceinline__ u32x4 bn_ld(const void* base, long long i) {
  if constexpr (ICAMD_BN_NT & 1) return __builtin_nontemporal_load((const u32x4*)base + i);
  else return ((const u32x4*)base)[i];
}
__device__ __forceinline__ void bn_st(void* base, long long i, const u32x4 v) {
  if constexpr (ICAMD_BN_NT & 2) __builtin_nontemporal_store(v, (u32x4*)base + i);
  else ((u32x4*)base)[i] = v;
}

// eval-mode BN: scale/shift from the running estimates
__global__ void bn_eval_coeffs_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                      float eps, float* __restrict__ scale_out, float* __restrict__ shift_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.0f / sqrtf(running_var[c] + eps);
  const float sc = gamma[c] * invstd;
  scale_out[c] = sc;
  shift_out[c] = beta[c] - running_mean[c] * sc;
}

// ------------------------------------------------------------------------------------------------
// One-launch reduce + finalize: grid (ceil(C/64), nchunks) blocks reduce their chunk of partial rows to fp64,
// publish it (agent-scope release), and the LAST block to arrive for a channel group (agent-scope acquire)
// sums the chunk rows in index order and runs the finalize -- bitwise reproducible, one kernel boundary instead
// of two.  counters[blockIdx.x] must be zero on entry; the last arriver resets it.
// ------------------------------------------------------------------------------------------------
struct BnFinalizeArgs {
  // MODE 0 (forward statistics)
  const float* gamma; const float* beta; float* running_mean; float* running_var;
  float momentum, eps;
  float* mean_out; float* invstd_out; float* scale_out; float* shift_out;
  // MODE 1 (backward sums)
  float* dgamma; float* dbeta; float* c1_out; float* c2_out; int accumulate;
  const float* gy_mean; const float* gy_invstd;   // set: the second sum is sum g*y, turned into sum g*xhat here
  double count;
};

// CG = channels per workgroup (the 1024 threads are CG channels x 1024 / CG row lanes).  Round 3: 32 instead of 64 wherever the
// 64 arrival counters of the workspace header allow it (C <= 2048): twice the workgroups and half the serial walk per thread --
// these 106 launches per ResNet-50 step are latency, not bandwidth (<= 3 MB of partial rows each) -- and two independent
// accumulator pairs per thread so that four loads are in flight instead of two.
template <int MODE, int CG>
__global__ __launch_bounds__(1024) void bn_reduce_finalize_kernel(const float* __restrict__ part, double* __restrict__ chunks,
                                                                  unsigned int* __restrict__ counters, int nrows, int C,
                                                                  int rows_per_chunk, BnFinalizeArgs a) {
  constexpr int RL = 1024 / CG;
  __shared__ double red[RL][2][CG];
  __shared__ int s_last;
  const int cc = threadIdx.x & (CG - 1);
  const int c = blockIdx.x * CG + cc;
  const int rl = threadIdx.x / CG;
  const int r0 = blockIdx.y * rows_per_chunk;
  const int r1 = min(nrows, r0 + rows_per_chunk);
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    double u1 = 0.0, u2 = 0.0;
    int r = r0 + rl;
    // Round 4: 16 rows (32 loads) requested at once, then added in the SAME order as the loop below adds them (the two
    // accumulator pairs take alternate rows): bit-identical sums, one L2 round trip instead of eight for a 512-row table --
    // these 106 launches per ResNet-50 step are a dependent latency chain, 7.3 / 6.1 us each for <= 3 MB of partial rows.
    constexpr int UN = 8;
    for (; r + (2 * UN - 1) * RL < r1; r += 2 * UN * RL) {
      float a0[UN], a1[UN], b0[UN], b1[UN];
#pragma unroll
      for (int k = 0; k < UN; ++k) {
        a0[k] = part[((long long)(r + 2 * k * RL) * 2 + 0) * C + c];
        a1[k] = part[((long long)(r + 2 * k * RL) * 2 + 1) * C + c];
        b0[k] = part[((long long)(r + (2 * k + 1) * RL) * 2 + 0) * C + c];
        b1[k] = part[((long long)(r + (2 * k + 1) * RL) * 2 + 1) * C + c];
      }
#pragma unroll
      for (int k = 0; k < UN; ++k) { s1 += (double)a0[k]; s2 += (double)a1[k]; u1 += (double)b0[k]; u2 += (double)b1[k]; }
    }
    for (; r + RL < r1; r += 2 * RL) {
      const float a0 = part[((long long)r * 2 + 0) * C + c], a1 = part[((long long)r * 2 + 1) * C + c];
      const float b0 = part[((long long)(r + RL) * 2 + 0) * C + c], b1 = part[((long long)(r + RL) * 2 + 1) * C + c];
      s1 += (double)a0; s2 += (double)a1; u1 += (double)b0; u2 += (double)b1;
    }
    if (r < r1) { s1 += (double)part[((long long)r * 2 + 0) * C + c]; s2 += (double)part[((long long)r * 2 + 1) * C + c]; }
    s1 += u1; s2 += u2;
  }
  red[rl][0][cc] = s1;
  red[rl][1][cc] = s2;
  __syncthreads();
  double t1 = 0.0, t2 = 0.0;
  if (gridDim.y == 1) {
    // a single chunk (<= 512 partial rows): no chunk row, no ticket -- the sums go straight to the finishing code
    if (threadIdx.x >= CG || c >= C) return;
#pragma unroll
    for (int j = 0; j < RL; ++j) { t1 += red[j][0][cc]; t2 += red[j][1][cc]; }
  } else {
  if (threadIdx.x < 2 * CG) {
    const int which = threadIdx.x / CG;
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < RL; ++j) s += red[j][which][cc];
    if (c < C) chunks[((long long)blockIdx.y * 2 + which) * C + c] = s;
  }
  // publish this block's chunk row, then take a ticket (cdna_hip_programming.md Guideline 16, counter form)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int prev = __hip_atomic_fetch_add(&counters[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (prev == gridDim.y - 1) ? 1 : 0;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&counters[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  // the last arriver folds the chunk rows with all row lanes (chunk k on lane k % RL, lanes then summed in lane order: a fixed
  // order)
  const int nchunks = gridDim.y;
  double p1 = 0.0, p2 = 0.0;
  if (c < C) {
    for (int k = rl; k < nchunks; k += RL) {
      p1 += chunks[((long long)k * 2 + 0) * C + c];
      p2 += chunks[((long long)k * 2 + 1) * C + c];
    }
  }
  red[rl][0][cc] = p1;
  red[rl][1][cc] = p2;
  __syncthreads();
  if (threadIdx.x >= CG || c >= C) return;
#pragma unroll
  for (int j = 0; j < RL; ++j) { t1 += red[j][0][cc]; t2 += red[j][1][cc]; }
  }
  if constexpr (MODE == 0) {
    const double mean = t1 / a.count;
    double var = t2 / a.count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
    const float meanf = (float)mean;
    a.mean_out[c] = meanf;
    a.invstd_out[c] = invstd;
    const float sc = a.gamma[c] * invstd;
    a.scale_out[c] = sc;
    a.shift_out[c] = a.beta[c] - meanf * sc;
    if (a.running_mean != nullptr) {
      const double unbiased = a.count > 1.0 ? var * (a.count / (a.count - 1.0)) : var;
      a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * meanf;
      a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unbiased;
    }
  } else {
    if (a.gy_mean != nullptr) t2 = (double)a.gy_invstd[c] * (t2 - (double)a.gy_mean[c] * t1);
    a.c1_out[c] = (float)(t1 / a.count);
    a.c2_out[c] = (float)(t2 / a.count);
    if (a.accumulate) { a.dgamma[c] += (float)t2; a.dbeta[c] += (float)t1; }
    else { a.dgamma[c] = (float)t2; a.dbeta[c] = (float)t1; }
  }
}

// out = act(y*scale[c] + shift[c] (+ residual)); 8 channels per thread
__global__ __launch_bounds__(256) void bn_apply_kernel(const bf16_t* __restrict__ y, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const bf16_t* __restrict__ residual,
                                                       bf16_t* __restrict__ out, unsigned char* __restrict__ maskbits,
                                                       long long nvec, int cpr, int relu,
                                                       const float* __restrict__ res_scale,
                                                       const float* __restrict__ res_shift) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  // the launcher makes stride a multiple of cpr, so this thread's channel group never changes
  const int cg = (int)(i % cpr) * 8;
  float sc[8], sh[8], rsc[8], rsh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = scale[cg + e]; sh[e] = shift[cg + e]; }
  // res_scale != nullptr: `residual` is the RAW conv output of the shortcut branch and its BatchNorm (no ReLU) is applied
  // here, rounded to bf16 exactly as the stored shortcut activation would have been
  const bool res_bn = res_scale != nullptr;
#pragma unroll
  for (int e = 0; e < 8; ++e) { rsc[e] = res_bn ? res_scale[cg + e] : 1.f; rsh[e] = res_bn ? res_shift[cg + e] : 0.f; }
  for (; i < nvec; i += stride) {
    const u32x4 v = bn_ld(y, i);
    float f[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f[2 * e] = fmaf(bf16_lo(v[e]), sc[2 * e], sh[2 * e]);
      f[2 * e + 1] = fmaf(bf16_hi(v[e]), sc[2 * e + 1], sh[2 * e + 1]);
    }
    if (residual != nullptr) {
      const u32x4 r = bn_ld(residual, i);
      if (res_bn) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f[2 * e] += bf16_to_f32(f32_to_bf16(fmaf(bf16_lo(r[e]), rsc[2 * e], rsh[2 * e])));
          f[2 * e + 1] += bf16_to_f32(f32_to_bf16(fmaf(bf16_hi(r[e]), rsc[2 * e + 1], rsh[2 * e + 1])));
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { f[2 * e] += bf16_lo(r[e]); f[2 * e + 1] += bf16_hi(r[e]); }
      }
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = (f[e] < 0.f) ? 0.f : f[e];   // NaN stays NaN, as torch.relu
    }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(f[2 * e], f[2 * e + 1]);
    bn_st(out, i, o);
    if (maskbits != nullptr) {   // bit e = [output element e > 0]: the ReLU mask the backward pass needs, 1 bit/element
      unsigned int bits = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) bits |= (f[e] > 0.f ? 1u : 0u) << e;
      maskbits[i] = (unsigned char)bits;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Max-pool backward folded into the BatchNorm backward that follows it (the ResNet stem: conv -> BN -> ReLU -> max-pool):
// with pg.idx set, `dout` is the gradient of the POOLED map [N][OH][OW][C] and the gradient of full-resolution pixel `row`
// is gathered on the fly -- the sum over the (<= 4) 3x3/s2/p1 windows that cover it and whose recorded argmax it is, rounded
// to bf16 exactly as icamd_maxpool3x3s2_bwd would have stored it -- so the 4x larger full-resolution gradient is never
// written or read.
struct PoolGather {
  const unsigned char* idx;   // nullptr: no pooling in front, dout is read directly
  int IH, IW, OH, OW;
  FastDiv dIW, dIH;
};

__device__ __forceinline__ u32x4 pool_gather8(const bf16_t* __restrict__ dout, const PoolGather& pg, unsigned int row,
                                              unsigned int cpr, unsigned int cg) {
  const unsigned int t1 = fdiv(row, pg.dIW);
  const int w = (int)(row - t1 * pg.IW);
  const unsigned int n = fdiv(t1, pg.dIH);
  const int h = (int)(t1 - n * pg.IH);
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  const int oh_lo = h >> 1, oh_hi = (h + 1) >> 1, ow_lo = w >> 1, ow_hi = (w + 1) >> 1;
  for (int oh = oh_lo; oh <= oh_hi; ++oh) {
    if (oh >= pg.OH) continue;
    const int r = h - (oh * 2 - 1);
    for (int ow = ow_lo; ow <= ow_hi; ++ow) {
      if (ow >= pg.OW) continue;
      const unsigned int code4 = (unsigned)(r * 3 + (w - (ow * 2 - 1))) * 0x01010101u;
      const unsigned int o = ((n * pg.OH + oh) * pg.OW + ow) * cpr + cg;
      const u32x2 iv = ((const u32x2*)pg.idx)[o];
      const u32x4 d = ((const u32x4*)dout)[o];
      const unsigned int m0 = iv[0] ^ code4, m1 = iv[1] ^ code4;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned int mb = ((e < 4 ? m0 : m1) >> ((e & 3) * 8)) & 0xffu;
        const float dv = (e & 1) ? bf16_hi(d[e >> 1]) : bf16_lo(d[e >> 1]);
        acc[e] += mb == 0u ? dv : 0.f;
      }
    }
  }
  u32x4 out;
#pragma unroll
  for (int e = 0; e < 4; ++e) out[e] = pack_bf16x2(acc[2 * e], acc[2 * e + 1]);
  return out;
}

// ------------------------------------------------------------------------------------------------
// BN backward, pass 1: per-channel partial sums of g and g*xhat, g = dout * [act > 0]
//   act == nullptr && relu : mask recomputed from y*scale+shift > 0 (no residual in front of the ReLU)
// rows are pixels; block handles `rows_per_block` rows; part[blk][2][C]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ act,
                                                            const bf16_t* __restrict__ y, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float* __restrict__ part,
                                                            const unsigned char* __restrict__ maskbits, long long rows, int C,
                                                            int rows_per_block, int relu, const PoolGather pg) {
  __shared__ float red[256 * 16];
  const int cpr = C >> 3;                 // 8-channel groups per row
  const int tid = threadIdx.x;
  // thread -> (channel group, row lane); when cpr > 256 the block loops over channel-group tiles
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = (rows < r0 + rows_per_block) ? rows : r0 + rows_per_block;
  for (int cg0 = 0; cg0 < cpr; cg0 += 256) {
    const int tcols = (cpr - cg0 < 256) ? (cpr - cg0) : 256;   // channel groups in this tile
    const int rlanes = 256 / tcols;                              // row lanes
    const int cgi = tid % tcols, rl = tid / tcols;
    float sg[8], sgx[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sg[e] = 0.f; sgx[e] = 0.f; }
    if (rl < rlanes) {
      const int c = (cg0 + cgi) * 8;
      float mu[8], is[8], sc[8], sh[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { mu[e] = mean[c + e]; is[e] = invstd[c + e]; sc[e] = scale[c + e]; sh[e] = shift[c + e]; }
      for (long long r = r0 + rl; r < r1; r += rlanes) {
        const long long off = r * cpr + cg0 + cgi;
        const u32x4 d = pg.idx != nullptr ? pool_gather8(dout, pg, (unsigned)r, (unsigned)cpr, (unsigned)(cg0 + cgi))
                                          : bn_ld(dout, off);
        const u32x4 yv = bn_ld(y, off);
        float g[8], yy[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          g[2 * e] = bf16_lo(d[e]); g[2 * e + 1] = bf16_hi(d[e]);
          yy[2 * e] = bf16_lo(yv[e]); yy[2 * e + 1] = bf16_hi(yv[e]);
        }
        if (relu) {
          if (maskbits != nullptr) {
            const unsigned int bits = maskbits[off];
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (!((bits >> e) & 1u)) g[e] = 0.f;
          } else if (act != nullptr) {
            const u32x4 a = bn_ld(act, off);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (!(bf16_lo(a[e]) > 0.f)) g[2 * e] = 0.f;
              if (!(bf16_hi(a[e]) > 0.f)) g[2 * e + 1] = 0.f;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (!(fmaf(yy[e], sc[e], sh[e]) > 0.f)) g[e] = 0.f;
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          sg[e] += g[e];
          sgx[e] += g[e] * ((yy[e] - mu[e]) * is[e]);
        }
      }
    }
    // cross-row-lane reduction through LDS: red[tid][16]
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = sg[e]; red[tid * 16 + 8 + e] = sgx[e]; }
    __syncthreads();
    // tcols*16 outputs; thread t sums over row lanes
    for (int o = tid; o < tcols * 16; o += 256) {
      const int cgo = o >> 4, e = o & 15;
      float s = 0.f;
      for (int l = 0; l < rlanes; ++l) s += red[(l * tcols + cgo) * 16 + e];
      const int which = e >> 3;
      part[((long long)blockIdx.x * 2 + which) * C + (cg0 + cgo) * 8 + (e & 7)] = s;
    }
    __syncthreads();
  }
}

// BN backward, pass 2: dy = scale * (g - c1 - xhat*c2); optionally also stores g (masked dout) in gout
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ act,
                                                           const bf16_t* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ c1,
                                                           const float* __restrict__ c2, bf16_t* __restrict__ dy,
                                                           bf16_t* __restrict__ gout, const unsigned char* __restrict__ maskbits,
                                                           long long nvec, int cpr, int relu, int reverse,
                                                           const PoolGather pg, const FastDiv dcpr) {
  // `reverse`: walk the tensors from the END.  The reduce pass that ran just before streamed dout and y front to back, so
  // their tails are what the 256 MB Infinity Cache (and L2) still hold: reading back to front meets those lines first.
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long k0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  // the launcher makes stride and nvec multiples of cpr, so this thread's channel group never changes (either direction)
  const int cg = (int)((reverse ? nvec - 1 - k0 : k0) % cpr) * 8;
  float mu[8], is[8], sc[8], sh[8], k1[8], k2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mu[e] = mean[cg + e]; is[e] = invstd[cg + e]; sc[e] = scale[cg + e]; sh[e] = shift[cg + e];
    k1[e] = c1[cg + e]; k2[e] = c2[cg + e];
  }
  for (; k0 < nvec; k0 += stride) {
    const long long i = reverse ? nvec - 1 - k0 : k0;
    u32x4 d;
    if (pg.idx != nullptr) {
      const unsigned int row = fdiv((unsigned)i, dcpr);
      d = pool_gather8(dout, pg, row, (unsigned)cpr, (unsigned)i - row * (unsigned)cpr);
    } else {
      d = bn_ld(dout, i);
    }
    const u32x4 yv = bn_ld(y, i);
    float g[8], yy[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      g[2 * e] = bf16_lo(d[e]); g[2 * e + 1] = bf16_hi(d[e]);
      yy[2 * e] = bf16_lo(yv[e]); yy[2 * e + 1] = bf16_hi(yv[e]);
    }
    if (relu) {
      if (maskbits != nullptr) {
        const unsigned int bits = maskbits[i];
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (!((bits >> e) & 1u)) g[e] = 0.f;
      } else if (act != nullptr) {
        const u32x4 a = bn_ld(act, i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (!(bf16_lo(a[e]) > 0.f)) g[2 * e] = 0.f;
          if (!(bf16_hi(a[e]) > 0.f)) g[2 * e + 1] = 0.f;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (!(fmaf(yy[e], sc[e], sh[e]) > 0.f)) g[e] = 0.f;
      }
    }
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = sc[e] * (g[e] - k1[e] - ((yy[e] - mu[e]) * is[e]) * k2[e]);
    u32x4 ov;
#pragma unroll
    for (int e = 0; e < 4; ++e) ov[e] = pack_bf16x2(o[2 * e], o[2 * e + 1]);
    bn_st(dy, i, ov);
    if (gout != nullptr) {
      u32x4 gv;
#pragma unroll
      for (int e = 0; e < 4; ++e) gv[e] = pack_bf16x2(g[2 * e], g[2 * e + 1]);
      bn_st(gout, i, gv);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Two BatchNorm backward passes that share one masked output gradient: a ResNet block with a projection shortcut feeds
// g = dout * [block output > 0] into BOTH the block's last BatchNorm (conv output yA) and the shortcut's BatchNorm (yB).
// Done as two separate icamd_bn_bwd calls, dout and its mask bits are read four times; here twice.
//   reduce: partA[blk] = (sum g, sum g*xhatA), partB[blk] = (sum g, sum g*xhatB)   (same layout as bn_bwd_reduce_kernel)
//   apply : dyA = scaleA*(g - c1A - xhatA*c2A), dyB likewise
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_bwd_dual_reduce_kernel(const bf16_t* __restrict__ dout,
                                                                 const unsigned char* __restrict__ maskbits,
                                                                 const bf16_t* __restrict__ yA, const float* __restrict__ meanA,
                                                                 const float* __restrict__ invstdA,
                                                                 const bf16_t* __restrict__ yB, const float* __restrict__ meanB,
                                                                 const float* __restrict__ invstdB, float* __restrict__ partA,
                                                                 float* __restrict__ partB, long long rows, int C,
                                                                 int rows_per_block) {
  __shared__ float red[256 * 24];
  const int cpr = C >> 3;
  const int tid = threadIdx.x;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = (rows < r0 + rows_per_block) ? rows : r0 + rows_per_block;
  for (int cg0 = 0; cg0 < cpr; cg0 += 256) {
    const int tcols = (cpr - cg0 < 256) ? (cpr - cg0) : 256;
    const int rlanes = 256 / tcols;
    const int cgi = tid % tcols, rl = tid / tcols;
    float sg[8], sa[8], sb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sg[e] = 0.f; sa[e] = 0.f; sb[e] = 0.f; }
    if (rl < rlanes) {
      const int c = (cg0 + cgi) * 8;
      float muA[8], isA[8], muB[8], isB[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { muA[e] = meanA[c + e]; isA[e] = invstdA[c + e]; muB[e] = meanB[c + e]; isB[e] = invstdB[c + e]; }
      for (long long r = r0 + rl; r < r1; r += rlanes) {
        const long long off = r * cpr + cg0 + cgi;
        const u32x4 d = bn_ld(dout, off);
        const u32x4 va = bn_ld(yA, off);
        const u32x4 vb = bn_ld(yB, off);
        const unsigned int bits = maskbits[off];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned int wd = d[e >> 1], wa = va[e >> 1], wb = vb[e >> 1];
          float g = (e & 1) ? bf16_hi(wd) : bf16_lo(wd);
          if (!((bits >> e) & 1u)) g = 0.f;
          const float ya = (e & 1) ? bf16_hi(wa) : bf16_lo(wa), yb = (e & 1) ? bf16_hi(wb) : bf16_lo(wb);
          sg[e] += g;
          sa[e] += g * ((ya - muA[e]) * isA[e]);
          sb[e] += g * ((yb - muB[e]) * isB[e]);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[tid * 24 + e] = sg[e]; red[tid * 24 + 8 + e] = sa[e]; red[tid * 24 + 16 + e] = sb[e]; }
    __syncthreads();
    for (int o = tid; o < tcols * 24; o += 256) {
      const int cgo = o / 24, e = o - cgo * 24;
      float s = 0.f;
      for (int l = 0; l < rlanes; ++l) s += red[(l * tcols + cgo) * 24 + e];
      const int ch = (cg0 + cgo) * 8 + (e & 7);
      if (e < 8) {
        partA[((long long)blockIdx.x * 2 + 0) * C + ch] = s;
        partB[((long long)blockIdx.x * 2 + 0) * C + ch] = s;
      } else if (e < 16) {
        partA[((long long)blockIdx.x * 2 + 1) * C + ch] = s;
      } else {
        partB[((long long)blockIdx.x * 2 + 1) * C + ch] = s;
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void bn_bwd_dual_apply_kernel(const bf16_t* __restrict__ dout,
                                                                const unsigned char* __restrict__ maskbits,
                                                                const bf16_t* __restrict__ yA, const float* __restrict__ meanA,
                                                                const float* __restrict__ invstdA, const float* __restrict__ scaleA,
                                                                const float* __restrict__ cA, bf16_t* __restrict__ dyA,
                                                                const bf16_t* __restrict__ yB, const float* __restrict__ meanB,
                                                                const float* __restrict__ invstdB, const float* __restrict__ scaleB,
                                                                const float* __restrict__ cB, bf16_t* __restrict__ dyB,
                                                                long long nvec, int cpr, int C, int reverse) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long k0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int cg = (int)((reverse ? nvec - 1 - k0 : k0) % cpr) * 8;   // see bn_bwd_apply_kernel
  float muA[8], isA[8], scA[8], k1A[8], k2A[8], muB[8], isB[8], scB[8], k1B[8], k2B[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    muA[e] = meanA[cg + e]; isA[e] = invstdA[cg + e]; scA[e] = scaleA[cg + e]; k1A[e] = cA[cg + e]; k2A[e] = cA[C + cg + e];
    muB[e] = meanB[cg + e]; isB[e] = invstdB[cg + e]; scB[e] = scaleB[cg + e]; k1B[e] = cB[cg + e]; k2B[e] = cB[C + cg + e];
  }
  for (; k0 < nvec; k0 += stride) {
    const long long i = reverse ? nvec - 1 - k0 : k0;
    const u32x4 d = bn_ld(dout, i);
    const u32x4 va = bn_ld(yA, i);
    const u32x4 vb = bn_ld(yB, i);
    const unsigned int bits = maskbits[i];
    float oa[8], ob[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned int wd = d[e >> 1], wa = va[e >> 1], wb = vb[e >> 1];
      float g = (e & 1) ? bf16_hi(wd) : bf16_lo(wd);
      if (!((bits >> e) & 1u)) g = 0.f;
      const float ya = (e & 1) ? bf16_hi(wa) : bf16_lo(wa), yb = (e & 1) ? bf16_hi(wb) : bf16_lo(wb);
      oa[e] = scA[e] * (g - k1A[e] - ((ya - muA[e]) * isA[e]) * k2A[e]);
      ob[e] = scB[e] * (g - k1B[e] - ((yb - muB[e]) * isB[e]) * k2B[e]);
    }
    u32x4 pa, pb;
#pragma unroll
    for (int e = 0; e < 4; ++e) { pa[e] = pack_bf16x2(oa[2 * e], oa[2 * e + 1]); pb[e] = pack_bf16x2(ob[2 * e], ob[2 * e + 1]); }
    bn_st(dyA, i, pa);
    bn_st(dyB, i, pb);
  }
}

// ------------------------------------------------------------------------------------------------
// max-pool 3x3 stride 2 pad 1 (torch scan order: rows then columns, first maximum wins, NaN propagates)
// ------------------------------------------------------------------------------------------------
// BNRELU: x is a raw conv output; every window element is first mapped to bf16(relu(x*scale[c] + shift[c])) -- exactly the
// tensor icamd_bn_apply would have stored -- so BatchNorm-apply + ReLU + max-pool of the ResNet stem is one pass that
// reads the conv output once and never writes the full-resolution activation.
// Index arithmetic is 32-bit with precomputed reciprocals (the launcher refuses >= 2^31 vectors): 64-bit `%` and `/` cost
// more VALU work per output than the nine window loads.  The grid's thread count is a multiple of C/8, so a thread's
// 8-channel group -- and its BatchNorm scale / shift -- never change.
struct PoolDivs { FastDiv cpr, ow, oh; };

template <bool BNRELU>
__global__ __launch_bounds__(256) void maxpool3x3s2_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out,
                                                               unsigned char* __restrict__ idx, int N, int IH, int IW,
                                                               int C, int OH, int OW, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, const PoolDivs dv) {
  const unsigned int cpr = (unsigned)C >> 3;
  const unsigned int total = (unsigned)N * OH * OW * cpr;
  const unsigned int stride = gridDim.x * blockDim.x;
  unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned int cg = i - fdiv(i, dv.cpr) * cpr;
  float sc[8], sh[8];
  if constexpr (BNRELU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = scale[cg * 8 + e]; sh[e] = shift[cg * 8 + e]; }
  }
  for (; i < total; i += stride) {
    unsigned int t = fdiv(i, dv.cpr);
    const unsigned int t1 = fdiv(t, dv.ow);
    const int ow = (int)(t - t1 * OW);
    const unsigned int n = fdiv(t1, dv.oh);
    const int oh = (int)(t1 - n * OH);
    if constexpr (BNRELU) {
      // The window elements are bf16(relu(.)) >= 0 (or NaN): their bit patterns order like the values, NaN above all, so
      // key = bits << 8 | (15 - position) makes ONE v_max_u32 pick the largest value, the FIRST position among equals,
      // and a NaN if there is one (torch.max_pool2d propagates NaN; with several NaNs in a window it records the last,
      // this the first -- a NaN step is skipped by the engine either way).
      unsigned int key[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) key[e] = 0u;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int ih = oh * 2 - 1 + r;
        if ((unsigned)ih >= (unsigned)IH) continue;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int iw = ow * 2 - 1 + s;
          if ((unsigned)iw >= (unsigned)IW) continue;
          const u32x4 v = ((const u32x4*)x)[((n * IH + ih) * IW + iw) * cpr + cg];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float a0 = fmaf(bf16_lo(v[e]), sc[2 * e], sh[2 * e]);
            float a1 = fmaf(bf16_hi(v[e]), sc[2 * e + 1], sh[2 * e + 1]);
            a0 = a0 < 0.f ? 0.f : a0;
            a1 = a1 < 0.f ? 0.f : a1;
            const unsigned int pk = pack_bf16x2(a0, a1) & 0x7fff7fffu;   // -0.0 (a < 0 is false for it) orders as +0.0
            const unsigned int k0 = ((pk & 0xffffu) << 8) | (unsigned)(15 - (r * 3 + s));
            const unsigned int k1 = ((pk >> 16) << 8) | (unsigned)(15 - (r * 3 + s));
            key[2 * e] = key[2 * e] > k0 ? key[2 * e] : k0;
            key[2 * e + 1] = key[2 * e + 1] > k1 ? key[2 * e + 1] : k1;
          }
        }
      }
      // (key 0 can only remain if the window is empty, which a 3x3/s2/p1 window over IH, IW >= 1 never is)
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = ((key[2 * e] >> 8) & 0xffffu) | ((key[2 * e + 1] >> 8) << 16);
      ((u32x4*)out)[i] = o;
      if (idx != nullptr) {
        u32x2 iv;
        iv[0] = (15u - (key[0] & 15u)) | ((15u - (key[1] & 15u)) << 8) | ((15u - (key[2] & 15u)) << 16) | ((15u - (key[3] & 15u)) << 24);
        iv[1] = (15u - (key[4] & 15u)) | ((15u - (key[5] & 15u)) << 8) | ((15u - (key[6] & 15u)) << 16) | ((15u - (key[7] & 15u)) << 24);
        ((u32x2*)idx)[i] = iv;
      }
    } else {
      float best[8];
      int bi[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0; }
      bool first = true;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int ih = oh * 2 - 1 + r;
        if ((unsigned)ih >= (unsigned)IH) continue;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int iw = ow * 2 - 1 + s;
          if ((unsigned)iw >= (unsigned)IW) continue;
          const u32x4 v = ((const u32x4*)x)[((n * IH + ih) * IW + iw) * cpr + cg];
          float f[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { f[2 * e] = bf16_lo(v[e]); f[2 * e + 1] = bf16_hi(v[e]); }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if (first || f[e] > best[e] || f[e] != f[e]) { best[e] = f[e]; bi[e] = r * 3 + s; }
          }
          first = false;
        }
      }
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(best[2 * e], best[2 * e + 1]);
      ((u32x4*)out)[i] = o;
      if (idx != nullptr) {
        u32x2 iv;
        iv[0] = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
        iv[1] = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
        ((u32x2*)idx)[i] = iv;
      }
    }
  }
}

// dx[n,h,w,c] = sum over the (<=4) windows that cover (h,w) and whose recorded argmax is (h,w).  dv: reciprocals of
// C/8, IW, IH.
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_kernel(const bf16_t* __restrict__ dout,
                                                               const unsigned char* __restrict__ idx,
                                                               bf16_t* __restrict__ dx, int N, int IH, int IW, int C,
                                                               int OH, int OW, const PoolDivs dv) {
  const unsigned int cpr = (unsigned)C >> 3;
  const unsigned int total = (unsigned)N * IH * IW * cpr;
  const unsigned int stride = gridDim.x * blockDim.x;
  unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned int cg = i - fdiv(i, dv.cpr) * cpr;
  for (; i < total; i += stride) {
    const unsigned int t = fdiv(i, dv.cpr);
    const unsigned int t1 = fdiv(t, dv.ow);
    const int w = (int)(t - t1 * IW);
    const unsigned int n = fdiv(t1, dv.oh);
    const int h = (int)(t1 - n * IH);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    const int oh_lo = h >> 1, oh_hi = (h + 1) >> 1;   // windows rows covering h: ceil((h-1)/2) .. floor((h+1)/2)
    const int ow_lo = w >> 1, ow_hi = (w + 1) >> 1;
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
      if (oh >= OH) continue;
      const int r = h - (oh * 2 - 1);
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        if (ow >= OW) continue;
        const int s = w - (ow * 2 - 1);
        const unsigned int code4 = (unsigned)(r * 3 + s) * 0x01010101u;
        const unsigned int o = ((n * OH + oh) * OW + ow) * cpr + cg;
        const u32x2 iv = ((const u32x2*)idx)[o];
        const u32x4 d = ((const u32x4*)dout)[o];
        // a byte of iv ^ code4 is zero where that channel's argmax is this pixel
        const unsigned int m0 = iv[0] ^ code4, m1 = iv[1] ^ code4;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned int mb = ((e < 4 ? m0 : m1) >> ((e & 3) * 8)) & 0xffu;
          const float dvv = (e & 1) ? bf16_hi(d[e >> 1]) : bf16_lo(d[e >> 1]);
          acc[e] += mb == 0u ? dvv : 0.f;
        }
      }
    }
    u32x4 ov;
#pragma unroll
    for (int e = 0; e < 4; ++e) ov[e] = pack_bf16x2(acc[2 * e], acc[2 * e + 1]);
    ((u32x4*)dx)[i] = ov;
  }
}

// global average pool: x[N][HW][C] -> out[N][C] (fp32 mean rounded once to bf16)
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out,
                                                          int N, int HW, int C) {
  const int cpr = C >> 3;
  const long long total = (long long)N * cpr;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int cg = (int)(i % cpr);
  const int n = (int)(i / cpr);
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int p = 0; p < HW; ++p) {
    const u32x4 v = ((const u32x4*)x)[((long long)n * HW + p) * cpr + cg];
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc[2 * e] += bf16_lo(v[e]); acc[2 * e + 1] += bf16_hi(v[e]); }
  }
  const float inv = 1.0f / (float)HW;
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(acc[2 * e] * inv, acc[2 * e + 1] * inv);
  ((u32x4*)out)[i] = o;
}

__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const bf16_t* __restrict__ dout, bf16_t* __restrict__ dx,
                                                          int N, int HW, int C) {
  const int cpr = C >> 3;
  const long long total = (long long)N * HW * cpr;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const float inv = 1.0f / (float)HW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int cg = (int)(i % cpr);
    const int n = (int)(i / ((long long)HW * cpr));
    const u32x4 d = ((const u32x4*)dout)[(long long)n * cpr + cg];
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(bf16_lo(d[e]) * inv, bf16_hi(d[e]) * inv);
    ((u32x4*)dx)[i] = o;
  }
}

// ------------------------------------------------------------------------------------------------
// Input packing: NCHW fp32 [B,3,H,W] -> NHWC bf16 [B,H,W,8] (channels 3..7 zero), with the batch-mode
// mixup / cutmix of timm.data.Mixup fused in (x <- lam*x + (1-lam)*x.flip(0), or a pasted box):
// the mix is done in fp32 on the fp32 pixels, then rounded once. mode 0 none, 1 mixup, 2 cutmix.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_input_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int B,
                                                         int Cin, int H, int W, int mode, float lam, int yl, int yh,
                                                         int xl, int xh) {
  const long long total = (long long)B * H * W;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long hw = (long long)H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int b = (int)(i / hw);
    const long long pix = i - (long long)b * hw;
    const int h = (int)(pix / W), w = (int)(pix - (long long)h * W);
    float f[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) f[c] = 0.f;
    const int fb = B - 1 - b;
    for (int c = 0; c < Cin; ++c) {
      float v = x[((long long)b * Cin + c) * hw + pix];
      if (mode == 1) {
        const float o = x[((long long)fb * Cin + c) * hw + pix];
        v = v * lam + o * (1.f - lam);
      } else if (mode == 2) {
        if (h >= yl && h < yh && w >= xl && w < xh) v = x[((long long)fb * Cin + c) * hw + pix];
      }
      f[c] = v;
    }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(f[2 * e], f[2 * e + 1]);
    ((u32x4*)out)[i] = o;
  }
}

// ResNet stem layout: fp32 NCHW -> bf16 [B][H][Wp][4] (RGB + one zero channel), Wp = W + 8 with 3 zero columns on the left and
// 5 on the right, so that the 7x7/2 window row of output column q is the 64 contiguous, 16 B-aligned bytes starting at padded
// column 2q (8 pixels x 4 channels; the 8th pixel meets a zero filter tap) and no load ever crosses the image border.
// One thread writes two adjacent padded pixels (16 B).  Same fused mixup / cutmix as pack_input_kernel.
__global__ __launch_bounds__(256) void pack_input_rgb4_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int B,
                                                              int Cin, int H, int W, int mode, float lam, int yl, int yh,
                                                              int xl, int xh) {
  const int Wp2 = (W + (W & 1) + 8) / 2;        // an odd width gets one more zero column: the row pitch stays even
  const long long total = (long long)B * H * Wp2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long hw = (long long)H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long row = i / Wp2;
    const int pp = (int)(i - row * Wp2);
    const int b = (int)(row / H), h = (int)(row - (long long)b * H);
    const int fb = B - 1 - b;
    float f[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) f[c] = 0.f;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int w = 2 * pp + e - 3;
      if (w >= 0 && w < W) {
        const long long pix = (long long)h * W + w;
        for (int c = 0; c < Cin; ++c) {
          float v = x[((long long)b * Cin + c) * hw + pix];
          if (mode == 1) {
            const float o = x[((long long)fb * Cin + c) * hw + pix];
            v = v * lam + o * (1.f - lam);
          } else if (mode == 2) {
            if (h >= yl && h < yh && w >= xl && w < xh) v = x[((long long)fb * Cin + c) * hw + pix];
          }
          f[4 * e + c] = v;
        }
      }
    }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(f[2 * e], f[2 * e + 1]);
    ((u32x4*)out)[i] = o;
  }
}

inline unsigned int grid_for(long long work_items, int threads, int multiple_of) {
  static const long long cap = []() { const char* e = getenv("ICAMD_EW_BLOCKS"); return e ? atoll(e) : 1024ll; }();
  long long blocks = (work_items + threads - 1) / threads;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  if (multiple_of > 1) blocks = (blocks + multiple_of - 1) / multiple_of * multiple_of;
  return (unsigned int)blocks;
}
// ICAMD_BN_REVERSE=1: back-to-front traversal of the second BatchNorm-backward pass.  Measured on MI355X (ResNet-50 bs 256,
// round 2): 5.50-5.53 ms/step of BatchNorm backward either way -- the pass is not helped by Infinity-Cache hits -- so it
// stays off.
inline int bn_reverse() {
  static const int v = []() { const char* e = getenv("ICAMD_BN_REVERSE"); return e ? atoi(e) : 0; }();
  return v;
}
inline int gcd_i(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

}  // namespace

// ---------------- host launchers (called from capi.hip) ----------------

static void chunking(int nrows, int* rows_per_chunk, int* nchunks) {
  int rpc = 128;
  int nc = (nrows + rpc - 1) / rpc;
  if (nrows <= 512) { rpc = nrows > 0 ? nrows : 1; nc = 1; }   // one workgroup per 64 channels, no ticket round
  if (nc > 64) { rpc = (nrows + 63) / 64; nc = (nrows + rpc - 1) / rpc; }
  *rows_per_chunk = rpc; *nchunks = nc;
}

// workspace head: 256 B of zero-initialised uint32 arrival counters (one per 64-channel group, C <= 4096) at a FIXED
// offset (so no other layer's data can ever land on them), then the [64][2][C] fp64 chunk rows
static unsigned int* counters_of(double* chunks, int C) { (void)C; return (unsigned int*)((char*)chunks - 256); }

// one launch of the reduce + finalize kernel: 32 channels per workgroup where the 64 counters of the header suffice (16 per
// workgroup measured slower: 0.58 vs 0.52 ms per step over the 53 forward launches -- 64 B row segments)
template <int MODE>
static void launch_reduce_finalize(const float* part, double* chunks, int nrows, int C, int rpc, int nc, const BnFinalizeArgs& a,
                                   hipStream_t s) {
  if (C <= 2048)
    hipLaunchKernelGGL((bn_reduce_finalize_kernel<MODE, 32>), dim3((unsigned)((C + 31) / 32), (unsigned)nc), dim3(1024), 0, s, part,
                       chunks, counters_of(chunks, C), nrows, C, rpc, a);
  else
    hipLaunchKernelGGL((bn_reduce_finalize_kernel<MODE, 64>), dim3((unsigned)((C + 63) / 64), (unsigned)nc), dim3(1024), 0, s, part,
                       chunks, counters_of(chunks, C), nrows, C, rpc, a);
}

int icamd_bn_finalize_launch(const float* part, int nrows, int C, double count, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, float momentum, float eps, float* mean,
                             float* invstd, float* scale, float* shift, double* chunks, hipStream_t s) {
  int rpc, nc;
  chunking(nrows, &rpc, &nc);
  BnFinalizeArgs a = {};
  a.gamma = gamma; a.beta = beta; a.running_mean = running_mean; a.running_var = running_var;
  a.momentum = momentum; a.eps = eps; a.mean_out = mean; a.invstd_out = invstd; a.scale_out = scale; a.shift_out = shift;
  a.count = count;
  launch_reduce_finalize<0>(part, chunks, nrows, C, rpc, nc, a, s);
  return icamd_launch_status();
}

// Generic fold of per-workgroup partial rows part[nrows][2][C]: sum1 -> out1, sum2 -> out2 (fixed order, fp64 across rows)
int icamd_sum_partials_launch(const float* part, int nrows, int C, float* out1, float* out2, int accumulate, double* chunks,
                              float* c1c2, hipStream_t s) {
  int rpc, nc;
  chunking(nrows, &rpc, &nc);
  BnFinalizeArgs a = {};
  a.dgamma = out2; a.dbeta = out1; a.c1_out = c1c2; a.c2_out = c1c2 + C; a.accumulate = accumulate; a.count = 1.0;
  launch_reduce_finalize<1>(part, chunks, nrows, C, rpc, nc, a, s);
  return icamd_launch_status();
}

int icamd_bn_eval_coeffs_launch(int C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                float* scale, float* shift, hipStream_t s) {
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, s, C, gamma, beta, rm, rv,
                     eps, scale, shift);
  return icamd_launch_status();
}

static unsigned int elementwise_grid(long long nvec, int cpr) {
  // total threads must be a multiple of cpr so a thread's channel group is loop-invariant
  const int mult = cpr / gcd_i(cpr, 256);
  return grid_for(nvec, 256, mult);
}

int icamd_bn_apply_launch(const bf16_t* y, const float* scale, const float* shift, const bf16_t* residual, bf16_t* out,
                          unsigned char* maskbits, long long numel, int C, int relu, hipStream_t s,
                          const float* res_scale, const float* res_shift) {
  if (C % 8 != 0 || numel % C != 0) return ICAMD_ERR_BAD_ARG;
  const long long nvec = numel / 8;
  const int cpr = C / 8;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(elementwise_grid(nvec, cpr)), dim3(256), 0, s, y, scale, shift, residual, out,
                     maskbits, nvec, cpr, relu, res_scale, res_shift);
  return icamd_launch_status();
}

int icamd_bn_bwd_rows_per_block(long long rows, int C) {
  // aim for ~N blocks (default 2048), at least 32 rows each
  static const long long nb = []() { const char* e = getenv("ICAMD_BNBWD_BLOCKS"); return e ? atoll(e) : 1024ll; }();
  long long rpb = (rows + nb - 1) / nb;
  if (rpb < 32) rpb = 32;
  (void)C;
  return (int)rpb;
}

int icamd_bn_bwd_launch(const bf16_t* dout, const bf16_t* act, const bf16_t* y, const float* mean, const float* invstd,
                        const float* scale, const float* shift, float* dgamma, float* dbeta, bf16_t* dy, bf16_t* gout,
                        const unsigned char* maskbits, long long rows, int C, int relu, int accumulate, float* part,
                        double* chunks, float* c1c2, hipStream_t s, const unsigned char* pool_idx, int pool_ih, int pool_iw) {
  if (C % 8 != 0) return ICAMD_ERR_BAD_ARG;
  PoolGather pg;
  memset(&pg, 0, sizeof(pg));
  if (pool_idx != nullptr) {   // dout is the pooled map's gradient: rows = N * pool_ih * pool_iw full-resolution pixels
    if (rows * (C / 8) >= (1ll << 31) || pool_ih < 1 || pool_iw < 1 || rows % ((long long)pool_ih * pool_iw) != 0)
      return ICAMD_ERR_BAD_ARG;
    pg.idx = pool_idx; pg.IH = pool_ih; pg.IW = pool_iw; pg.OH = (pool_ih - 1) / 2 + 1; pg.OW = (pool_iw - 1) / 2 + 1;
    pg.dIW = make_fastdiv((unsigned)pool_iw); pg.dIH = make_fastdiv((unsigned)pool_ih);
  }
  const FastDiv dcpr = make_fastdiv((unsigned)(C / 8));
  const int rpb = icamd_bn_bwd_rows_per_block(rows, C);
  const int nblk = (int)((rows + rpb - 1) / rpb);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)nblk), dim3(256), 0, s, dout, act, y, mean, invstd, scale, shift,
                     part, maskbits, rows, C, rpb, relu, pg);
  int rc = icamd_launch_status();
  if (rc) return rc;
  float* c1 = c1c2;
  float* c2 = c1c2 + C;
  {
    int rpc, nc;
    chunking(nblk, &rpc, &nc);
    BnFinalizeArgs a = {};
    a.dgamma = dgamma; a.dbeta = dbeta; a.c1_out = c1; a.c2_out = c2; a.accumulate = accumulate; a.count = (double)rows;
    launch_reduce_finalize<1>(part, chunks, nblk, C, rpc, nc, a, s);
    rc = icamd_launch_status();
    if (rc) return rc;
  }
  const long long nvec = rows * (C / 8);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(elementwise_grid(nvec, C / 8)), dim3(256), 0, s, dout, act, y, mean, invstd,
                     scale, shift, c1, c2, dy, gout, maskbits, nvec, C / 8, relu, bn_reverse(), pg, dcpr);
  return icamd_launch_status();
}

int icamd_bn_bwd_dual_launch(const bf16_t* dout, const unsigned char* maskbits, const bf16_t* yA, const float* meanA,
                             const float* invstdA, const float* scaleA, float* dgammaA, float* dbetaA, bf16_t* dyA,
                             const bf16_t* yB, const float* meanB, const float* invstdB, const float* scaleB, float* dgammaB,
                             float* dbetaB, bf16_t* dyB, long long rows, int C, int accumulate, float* partA, double* chunksA,
                             float* cA, float* partB, double* chunksB, float* cB, hipStream_t s) {
  if (C % 8 != 0) return ICAMD_ERR_BAD_ARG;
  const int rpb = icamd_bn_bwd_rows_per_block(rows, C);
  const int nblk = (int)((rows + rpb - 1) / rpb);
  hipLaunchKernelGGL(bn_bwd_dual_reduce_kernel, dim3((unsigned)nblk), dim3(256), 0, s, dout, maskbits, yA, meanA, invstdA, yB,
                     meanB, invstdB, partA, partB, rows, C, rpb);
  int rc = icamd_launch_status();
  if (rc) return rc;
  int rpc, nc;
  chunking(nblk, &rpc, &nc);
  for (int which = 0; which < 2; ++which) {
    BnFinalizeArgs a = {};
    a.dgamma = which ? dgammaB : dgammaA; a.dbeta = which ? dbetaB : dbetaA;
    a.c1_out = which ? cB : cA; a.c2_out = (which ? cB : cA) + C; a.accumulate = accumulate; a.count = (double)rows;
    double* chunks = which ? chunksB : chunksA;
    launch_reduce_finalize<1>(which ? partB : partA, chunks, nblk, C, rpc, nc, a, s);
    rc = icamd_launch_status();
    if (rc) return rc;
  }
  const long long nvec = rows * (C / 8);
  hipLaunchKernelGGL(bn_bwd_dual_apply_kernel, dim3(elementwise_grid(nvec, C / 8)), dim3(256), 0, s, dout, maskbits, yA, meanA,
                     invstdA, scaleA, cA, dyA, yB, meanB, invstdB, scaleB, cB, dyB, nvec, C / 8, C, bn_reverse());
  return icamd_launch_status();
}

// Reduce half of the BatchNorm backward alone, for an output gradient that is ALREADY masked (no ReLU handling): partial rows
// part[nblk][2][C] = (sum g, sum g * xhat) per block of rows; returns the number of rows written through *nblk_out.
int icamd_bn_bwd_reduce_launch(const bf16_t* g, const bf16_t* y, const float* mean, const float* invstd, float* part, long long rows,
                               int C, int* nblk_out, hipStream_t s) {
  if (C % 8 != 0) return ICAMD_ERR_BAD_ARG;
  PoolGather pg;
  memset(&pg, 0, sizeof(pg));
  const int rpb = icamd_bn_bwd_rows_per_block(rows, C);
  const int nblk = (int)((rows + rpb - 1) / rpb);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)nblk), dim3(256), 0, s, g, (const bf16_t*)nullptr, y, mean, invstd, mean, mean,
                     part, (const unsigned char*)nullptr, rows, C, rpb, 0, pg);
  *nblk_out = nblk;
  return icamd_launch_status();
}

// Finalize half of the BatchNorm backward alone: partial rows -> c1 = mean g, c2 = mean g * xhat (c1c2[0..C), [C..2C)) and the
// gamma / beta gradients.  The apply half then runs wherever the caller wants it (conv_fused_bwd.hip: inside the convolution's
// backward kernel).
int icamd_bn_bwd_finalize_launch(const float* part, int nrows, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                 long long rows, int C, int accumulate, double* chunks, float* c1c2, hipStream_t s, int sums_are_gy) {
  int rpc, nc;
  chunking(nrows, &rpc, &nc);
  BnFinalizeArgs a = {};
  a.dgamma = dgamma; a.dbeta = dbeta; a.c1_out = c1c2; a.c2_out = c1c2 + C; a.accumulate = accumulate; a.count = (double)rows;
  if (sums_are_gy) { a.gy_mean = mean; a.gy_invstd = invstd; }
  launch_reduce_finalize<1>(part, chunks, nrows, C, rpc, nc, a, s);
  return icamd_launch_status();
}

// BN backward from pass-1 partials produced elsewhere (the fused data-gradient epilogue): finalize + apply pass.
// g is already masked, so the apply kernel runs without a ReLU mask; shift is unused in that mode.
int icamd_bn_bwd_apply_launch(const float* part, int nrows, const bf16_t* g, const bf16_t* y, const float* mean,
                              const float* invstd, const float* scale, float* dgamma, float* dbeta, bf16_t* dy,
                              long long rows, int C, int accumulate, double* chunks, float* c1c2, hipStream_t s,
                              int sums_are_gy) {
  float* c1 = c1c2;
  float* c2 = c1c2 + C;
  int rc;
  {
    int rpc, nc;
    chunking(nrows, &rpc, &nc);
    BnFinalizeArgs a = {};
    a.dgamma = dgamma; a.dbeta = dbeta; a.c1_out = c1; a.c2_out = c2; a.accumulate = accumulate; a.count = (double)rows;
    if (sums_are_gy) { a.gy_mean = mean; a.gy_invstd = invstd; }
    launch_reduce_finalize<1>(part, chunks, nrows, C, rpc, nc, a, s);
    rc = icamd_launch_status();
    if (rc) return rc;
  }
  const long long nvec = rows * (C / 8);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(elementwise_grid(nvec, C / 8)), dim3(256), 0, s, g, (const bf16_t*)nullptr, y,
                     mean, invstd, scale, scale, c1, c2, dy, (bf16_t*)nullptr, (const unsigned char*)nullptr, nvec, C / 8, 0, 0, PoolGather{}, make_fastdiv((unsigned)(C / 8)));
  return icamd_launch_status();
}

static PoolDivs pool_divs(int C, int W, int H) {
  PoolDivs d;
  d.cpr = make_fastdiv((unsigned)(C / 8)); d.ow = make_fastdiv((unsigned)W); d.oh = make_fastdiv((unsigned)H);
  return d;
}

int icamd_maxpool_fwd_launch(const bf16_t* x, bf16_t* out, unsigned char* idx, int N, int IH, int IW, int C, int OH, int OW,
                             hipStream_t s) {
  if (C % 8 != 0 || (long long)N * IH * IW * (C / 8) >= (1ll << 31)) return ICAMD_ERR_BAD_ARG;
  const long long total = (long long)N * OH * OW * (C / 8);
  hipLaunchKernelGGL(maxpool3x3s2_fwd_kernel<false>, dim3(elementwise_grid(total, C / 8) * 4), dim3(256), 0, s, x, out, idx, N,
                     IH, IW, C, OH, OW, nullptr, nullptr, pool_divs(C, OW, OH));
  return icamd_launch_status();
}

int icamd_bn_relu_maxpool_fwd_launch(const bf16_t* y, const float* scale, const float* shift, bf16_t* out, unsigned char* idx,
                                     int N, int IH, int IW, int C, int OH, int OW, hipStream_t s) {
  if (C % 8 != 0 || (long long)N * IH * IW * (C / 8) >= (1ll << 31)) return ICAMD_ERR_BAD_ARG;
  const long long total = (long long)N * OH * OW * (C / 8);
  hipLaunchKernelGGL(maxpool3x3s2_fwd_kernel<true>, dim3(elementwise_grid(total, C / 8) * 4), dim3(256), 0, s, y, out, idx, N,
                     IH, IW, C, OH, OW, scale, shift, pool_divs(C, OW, OH));
  return icamd_launch_status();
}

int icamd_maxpool_bwd_launch(const bf16_t* dout, const unsigned char* idx, bf16_t* dx, int N, int IH, int IW, int C, int OH,
                             int OW, hipStream_t s) {
  if (C % 8 != 0 || (long long)N * IH * IW * (C / 8) >= (1ll << 31)) return ICAMD_ERR_BAD_ARG;
  const long long total = (long long)N * IH * IW * (C / 8);
  hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3(elementwise_grid(total, C / 8) * 4), dim3(256), 0, s, dout, idx, dx, N, IH,
                     IW, C, OH, OW, pool_divs(C, IW, IH));
  return icamd_launch_status();
}

int icamd_avgpool_fwd_launch(const bf16_t* x, bf16_t* out, int N, int HW, int C, hipStream_t s) {
  if (C % 8 != 0) return ICAMD_ERR_BAD_ARG;
  const long long total = (long long)N * (C / 8);
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, out, N, HW, C);
  return icamd_launch_status();
}

int icamd_avgpool_bwd_launch(const bf16_t* dout, bf16_t* dx, int N, int HW, int C, hipStream_t s) {
  if (C % 8 != 0) return ICAMD_ERR_BAD_ARG;
  const long long total = (long long)N * HW * (C / 8);
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for(total, 256, 1)), dim3(256), 0, s, dout, dx, N, HW, C);
  return icamd_launch_status();
}

int icamd_pack_input_launch(const float* x, bf16_t* out, int B, int Cin, int H, int W, int mode, float lam, int yl, int yh,
                            int xl, int xh, hipStream_t s) {
  if (Cin < 1 || Cin > 8) return ICAMD_ERR_BAD_ARG;
  const long long total = (long long)B * H * W;
  hipLaunchKernelGGL(pack_input_kernel, dim3(grid_for(total, 256, 1) * 2), dim3(256), 0, s, x, out, B, Cin, H, W, mode, lam,
                     yl, yh, xl, xh);
  return icamd_launch_status();
}


int icamd_pack_input_rgb4_launch(const float* x, bf16_t* out, int B, int Cin, int H, int W, int mode, float lam, int yl,
                                 int yh, int xl, int xh, hipStream_t s) {
  if (Cin < 1 || Cin > 3) return ICAMD_ERR_BAD_ARG;
  const long long total = (long long)B * H * ((W + (W & 1) + 8) / 2);
  hipLaunchKernelGGL(pack_input_rgb4_kernel, dim3(grid_for(total, 256, 1) * 2), dim3(256), 0, s, x, out, B, Cin, H, W, mode,
                     lam, yl, yh, xl, xh);
  return icamd_launch_status();
}


// ---- inference: fold a BatchNorm into the filters of the convolution in front of it ------------------------------
// w_folded[co][k] = bf16(w[co][k] * gamma[co] / sqrt(running_var[co] + eps)); shift[co] = beta - running_mean * scale.
// One workgroup per output channel.
namespace {
__global__ __launch_bounds__(256) void bn_fold_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ rm,
                                                      const float* __restrict__ rv, float eps, int K,
                                                      bf16_t* __restrict__ out, float* __restrict__ shift) {
  const int co = blockIdx.x;
  const float scale = gamma[co] / sqrtf(rv[co] + eps);
  if (threadIdx.x == 0) shift[co] = beta[co] - rm[co] * scale;
  const float* src = w + (long long)co * K;
  bf16_t* dst = out + (long long)co * K;
  for (int k = threadIdx.x; k < K; k += 256) dst[k] = f32_to_bf16(src[k] * scale);
}
}  // namespace

int icamd_bn_fold_launch(const float* w, const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                         int Cout, int K, bf16_t* w_folded, float* shift, hipStream_t s) {
  hipLaunchKernelGGL(bn_fold_kernel, dim3((unsigned)Cout), dim3(256), 0, s, w, gamma, beta, rm, rv, eps, K, w_folded, shift);
  return icamd_launch_status();
}
