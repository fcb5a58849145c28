// How fast does a CU-persistent kernel read [T = 197][128 B] row sets when the rows are `stride` bytes apart (the attention
// kernels' view of q / k / v inside the token-major qkv tensor: stride 4608 B; of dO / O: 1536 B) against contiguous rows?
// One workgroup per CU walks (image, head) pairs like attention.hip; every lane loads 16 B, sums, and one value per workgroup is
// stored so that nothing is optimised away.  Usage: probe   (prints GB/s per stride and matrix count)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(1024) void walk(const uint4* __restrict__ base, long long head_stride16, long long row_stride16,
                                             long long img_stride16, int H, int T, int nheads, int nmat, long long mat_stride16,
                                             unsigned* out) {
  unsigned acc = 0;
  for (int hd = blockIdx.x; hd < nheads; hd += gridDim.x) {
    const int b = hd / H, h = hd - b * H;
    const uint4* p = base + b * img_stride16 + h * head_stride16;
    for (int m = 0; m < nmat; ++m)
      for (int i = threadIdx.x; i < T * 8; i += blockDim.x) {
        const uint4 v = p[m * mat_stride16 + (long long)(i >> 3) * row_stride16 + (i & 7)];
        acc += v.x ^ v.y ^ v.z ^ v.w;
      }
  }
  if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
int main() {
  const int B = 256, T = 197, H = 12;
  const size_t bytes = (size_t)B * T * 3 * H * 128;
  uint4* d; unsigned* o;
  hipMalloc(&d, bytes); hipMalloc(&o, 4096);
  hipMemset(d, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct Case { const char* name; long long head, row, img, mat; int nmat; } cases[] = {
    {"token-major qkv (row stride 4608 B), q|k|v", 8, 288, 288ll * T, 96, 3},
    {"token-major qkv, k|v only", 8, 288, 288ll * T, 96, 2},
    {"head-major [which][B][H][T][64], q|k|v", 8ll * T, 8, 8ll * T * H, 8ll * T * H * B, 3},
    {"head-major, k|v only", 8ll * T, 8, 8ll * T * H, 8ll * T * H * B, 2},
  };
  for (auto& c : cases) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int it = 0; it < 10; ++it)
        hipLaunchKernelGGL(walk, dim3(256), dim3(1024), 0, 0, d + (c.nmat == 2 ? c.mat : 0), c.head, c.row, c.img, H, T, B * H, c.nmat, c.mat, o);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("%-50s %7.1f us per pass, %6.0f GB/s\n", c.name, ms * 100, (double)B * H * c.nmat * T * 128 / (ms / 10 * 1e-3) / 1e9);
    }
  }
  return 0;
}
