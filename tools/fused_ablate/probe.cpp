// Ablation probe of conv_fused_bwd.hip (measurement only; the kernel source is compiled in with -DICAMD_FUSED_ABLATE=<bits>):
// times the fused conv3 + bn3 backward kernel alone at ResNet-50's two shapes, batch 256.  Usage: probe [reps]
#include "../../imageclassification_amd/csrc/conv_fused_bwd.hip"
#include <cstdio>
#include <vector>

static void fill(void* p, size_t bytes, unsigned seed) {   // bf16 patterns in [-2, 2): finite, varied
  std::vector<unsigned short> h(bytes / 2);
  unsigned s = seed;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (unsigned short)(((s >> 16) & 0x80ff) | 0x3f00); }
  (void)hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  const int shapes[2][3] = {{802816, 64, 256}, {200704, 128, 512}};
  for (auto& sh : shapes) {
    const int M = sh[0], CI = sh[1], CO = sh[2];
    void *g, *y, *x, *wt, *dx, *slab, *cf;
    (void)hipMalloc(&g, (size_t)M * CO * 2); (void)hipMalloc(&y, (size_t)M * CO * 2); (void)hipMalloc(&x, (size_t)M * CI * 2);
    (void)hipMalloc(&wt, (size_t)CI * CO * 2); (void)hipMalloc(&dx, (size_t)M * CI * 2);
    (void)hipMalloc(&slab, (size_t)256 * CO * CI * 4); (void)hipMalloc(&cf, 5 * CO * 4);
    fill(g, (size_t)M * CO * 2, 1); fill(y, (size_t)M * CO * 2, 2); fill(x, (size_t)M * CI * 2, 3); fill(wt, (size_t)CI * CO * 2, 4);
    std::vector<float> c(5 * CO, 0.5f);
    (void)hipMemcpy(cf, c.data(), c.size() * 4, hipMemcpyHostToDevice);
    FusedBwdParams p = {};
    p.g = (const bf16_t*)g; p.y = (const bf16_t*)y; p.x = (const bf16_t*)x; p.wt = (const bf16_t*)wt; p.dx = (bf16_t*)dx;
    p.slab = (float*)slab; p.mean = (float*)cf; p.invstd = (float*)cf + CO; p.scale = (float*)cf + 2 * CO; p.c1 = (float*)cf + 3 * CO;
    p.c2 = (float*)cf + 4 * CO; p.M = M; p.CI = CI; p.CO = CO;
    for (int i = 0; i < 3; ++i) if (icamd_conv1x1_bn_bwd_fused_launch(p, 0) != 0) { printf("launch failed\n"); return 1; }
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipDeviceSynchronize(); (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) (void)icamd_conv1x1_bn_bwd_fused_launch(p, 0);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / reps, gb = (4.0 * M * CO + 4.0 * M * CI) / 1e9;
    printf("ablate %d: %d->%d M %d: %.1f us  (%.2f GB -> %.0f GB/s)\n", ICAMD_FUSED_ABLATE, CI, CO, M, us, gb, gb / us * 1e6);
    (void)hipFree(g); (void)hipFree(y); (void)hipFree(x); (void)hipFree(wt); (void)hipFree(dx); (void)hipFree(slab); (void)hipFree(cf);
  }
  return 0;
}
