// Hazard probe for layernorm_bwd_kernel<32> (round 4).  Test infrastructure, not product.
// Loads variants of the kernel as code objects (built by make_variants.py from the compiler's own assembly), runs each
// alone and next to the library's ring weight gradient on a second stream, and compares the per-workgroup partial rows with
// exact integer sums: dy[row][c] = (row % 49) + 1, so a missing / doubled row is named by the size of the error.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>
#include <cmath>
#include "icamd.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
// synthetic neighbours: mode 0 = MFMA only, 1 = LDS reads only, 2 = plain fp32 VALU only, 3 = MFMA + LDS
__global__ __launch_bounds__(256) void neighbour(float* __restrict__ out, int iters, int mode) {
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
  }
  out[(long long)blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + acc;
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }   // exact for small ints

int main(int argc, char** argv) {
  const long long rows = 200704; const int C = 192, nblk = 1024, rpw = 49;
  const char* kname = "_ZN12_GLOBAL__N_120layernorm_bwd_kernelILi32EEEvPKtS2_PKfS4_S4_S2_PtPfxii";
  std::vector<uint16_t> hdy(rows * C), hx(rows * C);
  for (long long r = 0; r < rows; ++r)
    for (int c = 0; c < C; ++c) {
      hdy[r * C + c] = f2bf((float)(r % 49 + 1));
      hx[r * C + c] = f2bf((float)((int)((r * 7 + c * 3) % 5) - 2));
    }
  std::vector<double> expect((size_t)nblk * 2 * C, 0.0);
  for (long long r = 0; r < rows; ++r) {
    const long long blk = r / (4 * rpw);
    for (int c = 0; c < C; ++c) {
      const double d = (double)(r % 49 + 1), xv = (double)((int)((r * 7 + c * 3) % 5) - 2);
      expect[(blk * 2 + 0) * C + c] += d;
      expect[(blk * 2 + 1) * C + c] += d * xv;
    }
  }
  uint16_t *dy, *x, *dx, *dx_ref; float *mean, *rstd, *gamma, *part;
  CK(hipMalloc(&dy, rows * C * 2)); CK(hipMalloc(&x, rows * C * 2)); CK(hipMalloc(&dx, rows * C * 2)); CK(hipMalloc(&dx_ref, rows * C * 2));
  CK(hipMalloc(&mean, rows * 4)); CK(hipMalloc(&rstd, rows * 4)); CK(hipMalloc(&gamma, C * 4)); CK(hipMalloc(&part, (size_t)nblk * 2 * C * 4));
  CK(hipMemcpy(dy, hdy.data(), rows * C * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(x, hx.data(), rows * C * 2, hipMemcpyHostToDevice));
  { std::vector<float> z(rows, 0.f), o(rows, 1.f), g(C, 1.f);
    CK(hipMemcpy(mean, z.data(), rows * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(rstd, o.data(), rows * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(gamma, g.data(), C * 4, hipMemcpyHostToDevice)); }
  // side-stream work: the 192 -> 768 pointwise weight gradient of ConvNeXt-T stage 1 at batch 256 (ring kernel)
  icamd_conv_desc d = {256, 28, 28, 192, 28, 28, 768, 1, 1, 1, 0};
  const size_t wgb = icamd_conv2d_wgrad_workspace_bytes(&d);
  void *xa, *dya, *wgw; float *dw, *db;
  CK(hipMalloc(&xa, rows * 192 * 2)); CK(hipMalloc(&dya, rows * 768 * 2)); CK(hipMalloc(&wgw, wgb)); CK(hipMalloc(&dw, 768 * 192 * 4)); CK(hipMalloc(&db, 768 * 4));
  CK(hipMemset(xa, 0x3c, rows * 192 * 2)); CK(hipMemset(dya, 0x3c, rows * 768 * 2)); CK(hipMemset(wgw, 0, wgb));
  hipStream_t main_s, side_s; CK(hipStreamCreate(&main_s)); CK(hipStreamCreate(&side_s));
  std::vector<float> got((size_t)nblk * 2 * C);
  std::vector<uint16_t> hdx(rows * C), hdx_ref(rows * C);
  const char* side_mode = getenv("PROBE_SIDE") ? getenv("PROBE_SIDE") : "wgrad";   // wgrad | copy | ln
  uint16_t* dx2 = nullptr; float* part2 = nullptr; void *cpa = nullptr, *cpb = nullptr;
  CK(hipMalloc(&dx2, rows * C * 2)); CK(hipMalloc(&part2, (size_t)nblk * 2 * C * 4)); CK(hipMalloc(&cpa, 1ll << 30)); CK(hipMalloc(&cpb, 1ll << 30));
  hipModule_t pmod; hipFunction_t pfn;   // the pinned kernel as a side-stream neighbour for PROBE_SIDE=ln
  CK(hipModuleLoad(&pmod, "variants/p0.co")); CK(hipModuleGetFunction(&pfn, pmod, kname));
  for (int v = 1; v < argc; ++v) {
    hipModule_t mod; hipFunction_t fn;
    CK(hipModuleLoad(&mod, argv[v])); CK(hipModuleGetFunction(&fn, mod, kname));
    const void* addend = nullptr; long long rows_ = rows; int C_ = C, rpw_ = rpw;
    void* args[] = {&dy, &x, &mean, &rstd, &gamma, &addend, &dx, &part, &rows_, &C_, &rpw_};
    {   // timing, alone
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int k = 0; k < 3; ++k) CK(hipModuleLaunchKernel(fn, nblk, 1, 1, 256, 1, 1, 0, main_s, args, nullptr));
      CK(hipEventRecord(e0, main_s));
      for (int k = 0; k < 20; ++k) CK(hipModuleLaunchKernel(fn, nblk, 1, 1, 256, 1, 1, 0, main_s, args, nullptr));
      CK(hipEventRecord(e1, main_s)); CK(hipEventSynchronize(e1));
      float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("TIMING %s: %.1f us per launch (rows %lld, C %d: %.0f MB of dy + x + dx)\n", argv[v], ms * 50.f, rows, C, rows * C * 6 / 1e6);
    }
    for (int with_side = 0; with_side < 2; ++with_side) {
      long long bad_entries = 0, bad_runs = 0, dx_bad_runs = 0; const int runs = with_side ? 8 : 3;
      int shown = 0; long long hist[2][8] = {{0}}, odd_diff = 0, pos_diff = 0;
      for (int it = 0; it < runs; ++it) {
        CK(hipMemsetAsync(part, 0xff, (size_t)nblk * 2 * C * 4, main_s)); CK(hipMemsetAsync(dx, 0xff, rows * C * 2, main_s));
        CK(hipStreamSynchronize(main_s));
        if (with_side) {
          if (!strcmp(side_mode, "wgrad")) {
            for (int k = 0; k < 3; ++k)
              if (icamd_conv2d_wgrad_bias(&d, xa, dya, dw, db, 0, wgw, wgb, side_s) != 0) { printf("wgrad failed\n"); return 3; }
          } else if (!strcmp(side_mode, "fwd")) {   // 192 -> 768 pointwise forward (y into the dya buffer)
            for (int k = 0; k < 3; ++k)
              if (icamd_conv2d_fwd(&d, xa, wgw, dya, nullptr, nullptr, nullptr, side_s) != 0) { printf("fwd failed\n"); return 3; }
          } else if (!strncmp(side_mode, "syn", 3)) {
            hipLaunchKernelGGL(neighbour, dim3(1024), dim3(256), 0, side_s, (float*)cpa, 40000, side_mode[3] - '0');
          } else if (!strcmp(side_mode, "copy")) {
            for (int k = 0; k < 3; ++k) CK(hipMemcpyAsync(cpb, cpa, 1ll << 30, hipMemcpyDeviceToDevice, side_s));
          } else {
            void* args2[] = {&dy, &x, &mean, &rstd, &gamma, &addend, &dx2, &part2, &rows_, &C_, &rpw_};
            for (int k = 0; k < 3; ++k) CK(hipModuleLaunchKernel(pfn, nblk, 1, 1, 256, 1, 1, 0, side_s, args2, nullptr));
          }
        }
        CK(hipModuleLaunchKernel(fn, nblk, 1, 1, 256, 1, 1, 0, main_s, args, nullptr));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), part, got.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hdx.data(), dx, rows * C * 2, hipMemcpyDeviceToHost));
        if (with_side == 0 && it == 0) hdx_ref = hdx;
        long long nb = 0;
        for (size_t i = 0; i < got.size(); ++i)
          if ((double)got[i] != expect[i]) {
            ++nb;
            { const double df = (double)got[i] - expect[i]; hist[(i / C) % 2][i % 8]++; odd_diff += (std::fmod(std::fabs(df), 2.0) != 0.0); pos_diff += df > 0; }
            if (shown < 3) {
              const int c = (int)(i % C), which = (int)((i / C) % 2), blk = (int)(i / (2 * C));
              printf("    %s side=%d run=%d blk=%d which=%d c=%d (lane %d elem %d) got %.1f want %.1f diff %.1f\n", argv[v], with_side, it, blk,
                     which, c, c / 8, c % 8, got[i], expect[i], (double)got[i] - expect[i]);
              ++shown;
            }
          }
        long long ndx = 0;
        for (long long i = 0; i < rows * C; ++i) ndx += hdx[i] != hdx_ref[i];
        bad_entries += nb; bad_runs += nb != 0; dx_bad_runs += ndx != 0;
      }
      if (bad_entries) {
        printf("    histogram by element: dbeta"); for (int e = 0; e < 8; ++e) printf(" %lld", hist[0][e]);
        printf(" | dgamma"); for (int e = 0; e < 8; ++e) printf(" %lld", hist[1][e]);
        printf(" | odd-valued diffs %lld, positive diffs %lld\n", odd_diff, pos_diff);
      }
      printf("VARIANT %s side=%s/%d: runs %d, runs with wrong partial rows %lld (entries %lld), runs with dx != first run %lld\n", argv[v], side_mode, with_side,
             runs, bad_runs, bad_entries, dx_bad_runs);
      fflush(stdout);
    }
    CK(hipModuleUnload(mod));
  }
  return 0;
}
