// Minimal reproducer (round 4): does a packed-fp32 VALU instruction with an op_sel swizzle on a VGPR pair lose part of its
// result while another wave of the SIMD runs MFMA / LDS work?  Found through layernorm_bwd_kernel<32> (see make_variants.py /
// probe.cpp): hipcc emitted `v_pk_add_f32 vD, vD, vS op_sel:[0,1] op_sel_hi:[1,0]` for the column sums and, next to the ring
// weight gradient on a second stream, lanes 48-55 occasionally lost the low-half add.  Test infrastructure, not product.
// Every lane accumulates exact small integers through several instruction forms; the host checks the closed forms.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "icamd.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;

constexpr int NFORM = 8;

// forms (acc = accumulator pair, s = (1, 2), t = (3, 5), all exact integers):
// 0 control      v_pk_add_f32 acc, acc, s                                  -> (+1, +2)
// 1 LN's form    v_pk_add_f32 acc, acc, s op_sel:[0,1] op_sel_hi:[1,0]     -> (+2, +1)
// 2 hi,hi        v_pk_add_f32 acc, acc, s op_sel:[0,1] op_sel_hi:[1,1]     -> (+2, +2)
// 3 lo,lo        v_pk_add_f32 acc, acc, s op_sel:[0,0] op_sel_hi:[1,0]     -> (+1, +1)  (the broadcast form the tree uses a lot)
// 4 fma swz      v_pk_fma_f32 acc, s, t, acc op_sel:[0,1,0] op_sel_hi:[1,0,1] -> (+1*5, +2*3) = (+5, +6)
// 5 fma plain    v_pk_fma_f32 acc, s, t, acc                               -> (+3, +10)
// 6 mul swz      v_pk_mul_f32 tmp, s, t op_sel:[0,1] op_sel_hi:[1,0]; acc += tmp (two scalar adds) -> (+5, +6)
// 7 pk_mov swz   v_pk_mov_b32 tmp, s, s op_sel:[1,0]  (tmp = (s.hi, s.lo)); acc += tmp (scalar adds) -> (+2, +1)
__global__ __launch_bounds__(256) void pk_probe(float* __restrict__ out, int iters, int zero_high_lanes) {
  const int lane = threadIdx.x & 63;
  f32x2 s = {1.f, 2.f}, t = {3.f, 5.f};
  if (zero_high_lanes && (lane & 31) >= 24) { s = f32x2{0.f, 0.f}; t = s; }   // as in LN: the last 8 lanes of each half carry zeros
  asm volatile("" : "+v"(s), "+v"(t));
  f32x2 a[NFORM];
#pragma unroll
  for (int f = 0; f < NFORM; ++f) a[f] = f32x2{0.f, 0.f};
  for (int i = 0; i < iters; ++i) {
    f32x2 tmp6, tmp7;
    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(s));
    asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(a[1]) : "v"(s));
    asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,1]" : "+v"(a[2]) : "v"(s));
    asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,0] op_sel_hi:[1,0]" : "+v"(a[3]) : "v"(s));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(a[4]) : "v"(s), "v"(t));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[5]) : "v"(s), "v"(t));
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(tmp6) : "v"(s), "v"(t));
    asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[1,0]" : "=v"(tmp7) : "v"(s));
    a[6].x += tmp6.x; a[6].y += tmp6.y;
    a[7].x += tmp7.x; a[7].y += tmp7.y;
  }
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int f = 0; f < NFORM; ++f) { out[(tid * NFORM + f) * 2] = a[f].x; out[(tid * NFORM + f) * 2 + 1] = a[f].y; }
}

#define GPTR(p) ((const void __attribute__((address_space(1)))*)(p))
#define LPTR(p) ((void __attribute__((address_space(3)))*)(p))
typedef __attribute__((ext_vector_type(4))) short bf16x4;
// neighbours: mode 0 = MFMA only, 1 = LDS reads only, 2 = plain fp32 VALU only, 3 = MFMA + LDS,
// 5 = LDS-DMA 16 B per lane (global_load_lds_dwordx4), 6 = LDS-DMA 4 B per lane, 7 = ds_read_b64_tr_b16, 8 = plain global loads
__global__ __launch_bounds__(256) void neighbour(float* __restrict__ out, int iters, int mode, const float* __restrict__ src) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)i;
  __syncthreads();
  bf16x8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = (short)(0x3f80 + threadIdx.x); b[e] = (short)(0x3f80 + e); }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  float v = (float)threadIdx.x, acc = 0.f;
  for (int i = 0; i < iters; ++i) {
    if (mode == 0 || mode == 3) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
    }
    if (mode == 1 || mode == 3) {
      const f32x4 q = *(const f32x4*)&lds[((threadIdx.x * 4 + i * 64) & 8188)];
      acc += q[0] + q[1] + q[2] + q[3];
    }
    if (mode == 2) { v = v * 1.0001f + 0.5f; acc += v; }
    if (mode == 5) {
      const int wave = threadIdx.x >> 6;
      __builtin_amdgcn_global_load_lds(GPTR(src + ((long long)(blockIdx.x * 64 + (i & 63)) * 1024 + threadIdx.x * 4)), LPTR(lds + wave * 256 + (i & 3) * 1024), 16, 0, 0);
      if ((i & 7) == 7) { __builtin_amdgcn_s_waitcnt(0x0070); acc += lds[threadIdx.x]; }
    }
    if (mode == 6) {
      const int wave = threadIdx.x >> 6;
      __builtin_amdgcn_global_load_lds(GPTR(src + ((long long)(blockIdx.x * 64 + (i & 63)) * 1024 + threadIdx.x)), LPTR(lds + wave * 64 + (i & 3) * 1024), 4, 0, 0);
      if ((i & 7) == 7) { __builtin_amdgcn_s_waitcnt(0x0070); acc += lds[threadIdx.x]; }
    }
    if (mode == 7) {
      const bf16x4 q = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3)))*)&lds[((threadIdx.x * 2 + i * 64) & 8190)]);
      acc += (float)q[0] + (float)q[3];
    }
    if (mode == 9 || mode == 10) {   // MFMA with VGPR (not AGPR) accumulators, as every MFMA kernel of the library has them
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b));
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b));
      if (mode == 10) {
        const f32x4 q = *(const f32x4*)&lds[((threadIdx.x * 4 + i * 64) & 8188)];
        a[0] = (short)(a[0] + (short)q[0]);
      }
    }
    if (mode == 8) {
      const f32x4 q = *(const f32x4*)(src + ((long long)(blockIdx.x * 64 + (i & 63)) * 1024 + threadIdx.x * 4));
      acc += q[0] + q[3];
    }
  }
  out[(long long)blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + acc;
}

int main(int argc, char** argv) {
  const int nblk = 1024, iters = 8192;
  const long long nthreads = (long long)nblk * 256;
  float *out, *nout;
  CK(hipMalloc(&out, nthreads * NFORM * 2 * 4)); CK(hipMalloc(&nout, 4096ll * 256 * 4));
  hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
  std::vector<float> h(nthreads * NFORM * 2);
  const float inc[NFORM][2] = {{1, 2}, {2, 1}, {2, 2}, {1, 1}, {5, 6}, {3, 10}, {5, 6}, {2, 1}};
  const char* names[] = {"alone", "MFMA", "LDS reads", "fp32 VALU", "MFMA+LDS", "ring wgrad", "LDS-DMA x4", "LDS-DMA x1", "ds_read_tr", "global loads", "MFMA vgpr-acc", "MFMA vgpr-acc + LDS"};
  float* nsrc; CK(hipMalloc(&nsrc, 1024ll * 64 * 1024 * 4)); CK(hipMemset(nsrc, 0, 1024ll * 64 * 1024 * 4));
  icamd_conv_desc d = {256, 28, 28, 192, 28, 28, 768, 1, 1, 1, 0};
  const long long rows = 200704;
  const size_t wgb = icamd_conv2d_wgrad_workspace_bytes(&d);
  void *xa, *dya, *wgw; float *dw, *db;
  CK(hipMalloc(&xa, rows * 192 * 2)); CK(hipMalloc(&dya, rows * 768 * 2)); CK(hipMalloc(&wgw, wgb)); CK(hipMalloc(&dw, 768 * 192 * 4)); CK(hipMalloc(&db, 768 * 4));
  CK(hipMemset(xa, 0x3c, rows * 192 * 2)); CK(hipMemset(dya, 0x3c, rows * 768 * 2)); CK(hipMemset(wgw, 0, wgb));
  for (int zero_high = 0; zero_high < 2; ++zero_high)
    for (int nb = -1; nb < 11; ++nb) {
      if (nb >= 0 && nb <= 3) continue;
      if (nb >= 5 && nb <= 8) continue;
      long long bad[NFORM][2] = {{0}}, lane_hist[NFORM][4] = {{0}};
      const int reps = 6;
      for (int r = 0; r < reps; ++r) {
        if (nb == 4) {
          for (int k = 0; k < 3; ++k)
            if (icamd_conv2d_wgrad_bias(&d, xa, dya, dw, db, 0, wgw, wgb, s1) != 0) { printf("wgrad failed\n"); return 3; }
        } else if (nb >= 0) hipLaunchKernelGGL(neighbour, dim3(1024), dim3(256), 0, s1, nout, (nb >= 5 && nb <= 8) ? 6000 : 40000, nb, nsrc);
        hipLaunchKernelGGL(pk_probe, dim3(nblk), dim3(256), 0, s0, out, iters, zero_high);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
        for (long long t = 0; t < nthreads; ++t) {
          const int lane = (int)(t & 63);
          const bool zeroed = zero_high && (lane & 31) >= 24;
          for (int f = 0; f < NFORM; ++f)
            for (int k = 0; k < 2; ++k) {
              const float want = zeroed ? 0.f : inc[f][k] * iters;
              if (h[(t * NFORM + f) * 2 + k] != want) { bad[f][k]++; lane_hist[f][lane >> 4]++; }
            }
        }
      }
      printf("zero_high=%d neighbour=%-10s:", zero_high, names[nb + 1]);
      for (int f = 0; f < NFORM; ++f) printf(" f%d lo %lld hi %lld [q %lld %lld %lld %lld]", f, bad[f][0], bad[f][1], lane_hist[f][0], lane_hist[f][1], lane_hist[f][2], lane_hist[f][3]);
      printf("\n"); fflush(stdout);
    }
  return 0;
}
