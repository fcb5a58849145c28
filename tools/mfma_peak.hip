// Sustained dense bf16 MFMA rate of the GPU this runs on: register-only v_mfma_f32_16x16x32_bf16 chains, no memory
// traffic, W waves per SIMD.  Gives the ceiling the GEMM-shaped kernels can be priced against on THIS board (clock under
// MFMA load), next to the data-sheet 2.5 PFLOP/s.   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, long long* clk) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x); b[i] = (short)(0x3f00 + i); }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

int main() {
  float* out; long long* clk;
  hipMalloc(&out, sizeof(float) * 256 * 2048);
  hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
    const int blocks = 256 * blocks_per_cu, iters = 20000;
    for (int rep = 0; rep < (blocks_per_cu == 2 ? 300 : 3); ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
      const double flop = (double)blocks * 4 * iters * 32 * 16 * 16 * 32 * 2;
      if (rep < 3 || rep % 50 == 49) printf("waves/SIMD %d: %.2f ms  %.1f TFLOP/s   shader clocks %lld / wall ticks %lld (100 MHz) -> %.0f MHz\n",
             blocks_per_cu, ms, flop / ms * 1e-9, h[0], h[1], (double)h[0] / ((double)h[1] / 100.0));
    }
  }
  return 0;
}
