// Input pipeline on the GPU: everything the reference's train / eval transforms do AFTER JPEG decoding
// (/root/reference/datasets.py:121-144: timm.create_transform(scale=(1,1), ratio=(1,1), vflip=0.5, color_jitter=0.3,
// interpolation='bicubic', re_prob=0.25, re_mode='pixel') for training; Resize -> ToTensor -> Normalize for eval), on uint8
// HWC images of arbitrary sizes, one launch sequence per batch:
//   1. resample_coeffs_kernel  per image and axis: Pillow's filter windows and 22-bit integer weights (Resample.c), computed
//                              in double with FMA contraction OFF so that they equal the CPU's bit for bit;
//   2. resize_h_kernel         crop window + horizontal pass -> uint8 (rounded and clamped, as Pillow stores the pass);
//   3. resize_v_kernel         vertical pass (+ horizontal / vertical flip folded into the store index) -> uint8 [B][H][W][3];
//   4. jitter_kernel           torchvision ColorJitter on PIL semantics: brightness / contrast / saturation in a per-image
//                              order, PIL.ImageEnhance arithmetic (Image.blend truncation / clamping, integer luma, contrast
//                              pivot = rounded mean luma of the image AT THAT POINT of the chain: one workgroup per image);
//   5. finalize_kernel         ToTensor + Normalize -> fp32 NCHW, timm RandomErasing(mode='pixel') box filled with N(0,1).
// Random decisions (flips, jitter order and factors, erase box, noise seed) are made on the host and arrive in the
// descriptors, so parity tests inject them.  Bit-exact against Pillow for steps 1-4 (tests/test_image_gpu.py).
#include "../../include/icamd.h"
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

#pragma clang fp contract(off)
__device__ double filter_weight(int filt, double x) {
#pragma clang fp contract(off)
  if (x < 0.0) x = -x;
  if (filt == 1) {         // bicubic, a = -0.5
    const double a = -0.5;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
  }
  return x < 1.0 ? 1.0 - x : 0.0;   // bilinear
}

// grid: (ceil(out/64), 2 axes, B); one thread per output sample of one axis
__global__ __launch_bounds__(64) void resample_coeffs_kernel(const icamd_image_desc* __restrict__ descs, int out_h, int out_w,
                                                             int filt, int kmax, int* __restrict__ tabs, long long per_image) {
#pragma clang fp contract(off)
  const int b = blockIdx.z, axis = blockIdx.y;
  const int out_size = axis == 0 ? out_w : out_h;
  const int xx = blockIdx.x * 64 + threadIdx.x;
  if (xx >= out_size) return;
  const icamd_image_desc d = descs[b];
  const int in_size = axis == 0 ? d.crop_w : d.crop_h;
  int* bounds = tabs + (long long)b * per_image + (axis == 0 ? 0 : (long long)out_w * (2 + kmax));
  int* kk = bounds + (long long)out_size * 2;
  const double scale = (double)in_size / (double)out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = (filt == 1 ? 2.0 : 1.0) * filterscale;
  const double ss = 1.0 / filterscale;
  const double center = (xx + 0.5) * scale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  if (xmax > kmax) xmax = kmax;   // cannot happen when the host sized kmax from the same formula; keeps stores in bounds
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) ww += filter_weight(filt, (x + xmin - center + 0.5) * ss);
  for (int x = 0; x < kmax; ++x) {
    int q = 0;
    if (x < xmax) {
      double w = filter_weight(filt, (x + xmin - center + 0.5) * ss);
      if (ww != 0.0) w = w / ww;
      q = w < 0 ? (int)(-0.5 + w * (double)(1 << PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << PRECISION_BITS));
    }
    kk[(long long)xx * kmax + x] = q;
  }
  bounds[xx * 2] = xmin;
  bounds[xx * 2 + 1] = xmax;
}

__device__ __forceinline__ unsigned char clip8(int v) {
  v >>= PRECISION_BITS;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// grid: (ceil(out_w*rows/256), 1, B): thread -> (row of the crop window, output column); tmp[b][row][ox][3]
__global__ __launch_bounds__(256) void resize_h_kernel(const unsigned char* __restrict__ src,
                                                       const icamd_image_desc* __restrict__ descs, int out_w, int kmax,
                                                       const int* __restrict__ tabs, long long per_image,
                                                       unsigned char* __restrict__ tmp, long long tmp_per_image) {
  const int b = blockIdx.z;
  const icamd_image_desc d = descs[b];
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)d.crop_h * out_w) return;
  const int row = (int)(idx / out_w), ox = (int)(idx - (long long)row * out_w);
  const int* bounds = tabs + (long long)b * per_image;
  const int* kk = bounds + (long long)out_w * 2 + (long long)ox * kmax;
  const int xmin = bounds[ox * 2], cnt = bounds[ox * 2 + 1];
  const unsigned char* line = src + d.src_offset + ((long long)(d.crop_top + row) * d.src_w + d.crop_left) * 3;
  unsigned char* o = tmp + (long long)b * tmp_per_image + ((long long)row * out_w + ox) * 3;
  if (d.crop_w == out_w) {   // Pillow skips a pass that does not change the size
    o[0] = line[ox * 3]; o[1] = line[ox * 3 + 1]; o[2] = line[ox * 3 + 2];
    return;
  }
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int t = 0; t < cnt; ++t) {
    const int w = kk[t];
    const unsigned char* px = line + (xmin + t) * 3;
    s0 += px[0] * w; s1 += px[1] * w; s2 += px[2] * w;
  }
  o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// grid: (ceil(out_h*out_w/256), 1, B); img[b][oy'][ox'][3] with the flips applied to the destination index
__global__ __launch_bounds__(256) void resize_v_kernel(const icamd_image_desc* __restrict__ descs, int out_h, int out_w, int kmax,
                                                       const int* __restrict__ tabs, long long per_image,
                                                       const unsigned char* __restrict__ tmp, long long tmp_per_image,
                                                       unsigned char* __restrict__ img) {
  const int b = blockIdx.z;
  const icamd_image_desc d = descs[b];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= out_h * out_w) return;
  const int oy = idx / out_w, ox = idx - oy * out_w;
  const int* bounds = tabs + (long long)b * per_image + (long long)out_w * (2 + kmax);
  const int* kk = bounds + (long long)out_h * 2 + (long long)oy * kmax;
  const int ymin = bounds[oy * 2], cnt = bounds[oy * 2 + 1];
  const unsigned char* col = tmp + (long long)b * tmp_per_image + (long long)ox * 3;
  unsigned char r0, r1, r2;
  if (d.crop_h == out_h) {
    const unsigned char* px = col + (long long)oy * out_w * 3;
    r0 = px[0]; r1 = px[1]; r2 = px[2];
  } else {
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int t = 0; t < cnt; ++t) {
      const int w = kk[t];
      const unsigned char* px = col + (long long)(ymin + t) * out_w * 3;
      s0 += px[0] * w; s1 += px[1] * w; s2 += px[2] * w;
    }
    r0 = clip8(s0); r1 = clip8(s1); r2 = clip8(s2);
  }
  const int dy = d.vflip ? out_h - 1 - oy : oy, dx = d.hflip ? out_w - 1 - ox : ox;
  unsigned char* o = img + ((long long)b * out_h * out_w + (long long)dy * out_w + dx) * 3;
  o[0] = r0; o[1] = r1; o[2] = r2;
}

__device__ __forceinline__ int luma(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }
// PIL Image.blend(degenerate, image, factor) on one uint8 sample (libImaging/Blend.c)
__device__ __forceinline__ int blend(int deg, int v, float f, bool inside) {
#pragma clang fp contract(off)
  const float t = (float)deg + f * ((float)v - (float)deg);   // separate multiply and add, as the C library's x86 build
  if (inside) return (int)t & 255;                     // (UINT8)(float): truncation toward zero, value already in [0, 255]
  return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}

// one workgroup per image, in place on img[b]
__global__ __launch_bounds__(256) void jitter_kernel(const icamd_image_desc* __restrict__ descs, int npix,
                                                     unsigned char* __restrict__ img) {
  __shared__ unsigned long long red[256];
  const icamd_image_desc d = descs[blockIdx.x];
  unsigned char* p = img + (long long)blockIdx.x * npix * 3;
  for (int k = 0; k < 3; ++k) {
    const int op = d.jitter_order[k];
    if (op < 0 || op > 2) continue;
    const float f = d.jitter_factor[op];
    const bool inside = f >= 0.f && f <= 1.f;
    int pivot = 0;
    if (op == 1) {                                     // contrast: rounded mean luma of the image as it is NOW
      unsigned long long s = 0;
      for (int i = threadIdx.x; i < npix; i += 256) s += (unsigned)luma(p[i * 3], p[i * 3 + 1], p[i * 3 + 2]);
      red[threadIdx.x] = s;
      __syncthreads();
      for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
      }
      pivot = (int)((double)red[0] / (double)npix + 0.5);   // int(ImageStat.Stat(L).mean[0] + 0.5)
      __syncthreads();
    }
    for (int i = threadIdx.x; i < npix; i += 256) {
      const int r = p[i * 3], g = p[i * 3 + 1], b = p[i * 3 + 2];
      const int dg = op == 0 ? 0 : (op == 1 ? pivot : luma(r, g, b));
      p[i * 3] = (unsigned char)blend(dg, r, f, inside);
      p[i * 3 + 1] = (unsigned char)blend(dg, g, f, inside);
      p[i * 3 + 2] = (unsigned char)blend(dg, b, f, inside);
    }
    __syncthreads();   // the next operation reads what every thread wrote
  }
}

// counter-based noise for the erased box: two rounds of a 32-bit mixer per draw, Box-Muller
__device__ __forceinline__ unsigned int mix32(unsigned int x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float normal01(unsigned int seed, unsigned int idx) {
  const unsigned int a = mix32(seed ^ (idx * 2u + 1u)), b = mix32((seed + 0x9e3779b9u) ^ (idx * 2u + 2u));
  const float u1 = ((float)(a >> 8) + 0.5f) * (1.f / 16777216.f), u2 = ((float)(b >> 8) + 0.5f) * (1.f / 16777216.f);
  return sqrtf(-2.f * __logf(u1)) * __cosf(6.283185307179586f * u2);
}

// grid: (ceil(H*W/256), 3 channels, B) -> out[b][c][y][x] fp32
__global__ __launch_bounds__(256) void finalize_kernel(const icamd_image_desc* __restrict__ descs, int out_h, int out_w,
                                                       const unsigned char* __restrict__ img, float m0, float m1, float m2,
                                                       float s0, float s1, float s2, float* __restrict__ out) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= out_h * out_w) return;
  const icamd_image_desc d = descs[b];
  const int y = idx / out_w, x = idx - y * out_w;
  const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
  float v = ((float)img[((long long)b * out_h * out_w + idx) * 3 + c] / 255.f - mean) / sd;
  if (d.erase_h > 0 && d.erase_w > 0 && y >= d.erase_top && y < d.erase_top + d.erase_h && x >= d.erase_left &&
      x < d.erase_left + d.erase_w)
    v = normal01(d.erase_seed, (unsigned)((c * out_h + y) * out_w + x));
  out[(((long long)b * 3 + c) * out_h + y) * out_w + x] = v;
}

}  // namespace

extern "C" {

// int32 table words per image: both axes' bounds + weights
static long long tab_words(int out_h, int out_w, int kmax) { return (long long)(out_h + out_w) * (2 + kmax); }

size_t icamd_image_pipeline_workspace_bytes(int B, int max_crop_h, int out_h, int out_w, int kmax) {
  if (B <= 0 || max_crop_h <= 0 || out_h <= 0 || out_w <= 0 || kmax <= 0) return 0;
  const size_t tabs = (size_t)B * tab_words(out_h, out_w, kmax) * 4;
  const size_t tmp = (size_t)B * max_crop_h * out_w * 3;
  const size_t img = (size_t)B * out_h * out_w * 3;
  return ((tabs + 255) / 256 + (tmp + 255) / 256 + (img + 255) / 256) * 256;
}

int icamd_image_pipeline(const uint8_t* src, const icamd_image_desc* descs, int B, int max_crop_h, int out_h, int out_w,
                         int filter, int kmax, const float* mean3, const float* std3, float* out_nchw, void* workspace,
                         size_t workspace_bytes, void* stream) {
  if (src == nullptr || descs == nullptr || out_nchw == nullptr || workspace == nullptr || mean3 == nullptr || std3 == nullptr)
    return ICAMD_ERR_BAD_ARG;
  if (B <= 0 || out_h <= 0 || out_w <= 0 || kmax <= 0 || (filter != 0 && filter != 1)) return ICAMD_ERR_BAD_ARG;
  const size_t need = icamd_image_pipeline_workspace_bytes(B, max_crop_h, out_h, out_w, kmax);
  if (need == 0 || workspace_bytes < need) return ICAMD_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const long long per_image = tab_words(out_h, out_w, kmax);
  int* tabs = (int*)workspace;
  const size_t tabs_b = ((size_t)B * per_image * 4 + 255) / 256 * 256;
  unsigned char* tmp = (unsigned char*)workspace + tabs_b;
  const long long tmp_per_image = (long long)max_crop_h * out_w * 3;
  const size_t tmp_b = ((size_t)B * tmp_per_image + 255) / 256 * 256;
  unsigned char* img = tmp + tmp_b;
  const int omax = out_h > out_w ? out_h : out_w;
  hipLaunchKernelGGL(resample_coeffs_kernel, dim3((unsigned)((omax + 63) / 64), 2, (unsigned)B), dim3(64), 0, s, descs, out_h,
                     out_w, filter, kmax, tabs, per_image);
  hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)(((long long)max_crop_h * out_w + 255) / 256), 1, (unsigned)B), dim3(256), 0,
                     s, src, descs, out_w, kmax, tabs, per_image, tmp, tmp_per_image);
  hipLaunchKernelGGL(resize_v_kernel, dim3((unsigned)((out_h * out_w + 255) / 256), 1, (unsigned)B), dim3(256), 0, s, descs, out_h,
                     out_w, kmax, tabs, per_image, tmp, tmp_per_image, img);
  hipLaunchKernelGGL(jitter_kernel, dim3((unsigned)B), dim3(256), 0, s, descs, out_h * out_w, img);
  hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((out_h * out_w + 255) / 256), 3, (unsigned)B), dim3(256), 0, s, descs, out_h,
                     out_w, img, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], out_nchw);
  return icamd_launch_status();
}

// the uint8 [B][H][W][3] image after resize / flips / jitter (what Pillow would hold before ToTensor): for parity tests
int icamd_image_pipeline_u8(const void* workspace, int B, int max_crop_h, int out_h, int out_w, int kmax, const uint8_t** img) {
  if (workspace == nullptr || img == nullptr) return ICAMD_ERR_BAD_ARG;
  const size_t tabs_b = ((size_t)B * tab_words(out_h, out_w, kmax) * 4 + 255) / 256 * 256;
  const size_t tmp_b = ((size_t)B * max_crop_h * out_w * 3 + 255) / 256 * 256;
  *img = (const uint8_t*)workspace + tabs_b + tmp_b;
  return ICAMD_OK;
}

}  // extern "C"
