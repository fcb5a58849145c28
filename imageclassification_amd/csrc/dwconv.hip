// ConvNeXt-specific kernels for gfx950: depthwise 7x7 convolution (forward, data gradient, weight gradient) and the
// layer-scale + stochastic-depth + residual tail of the block.  Block spec: /root/reference/semantic_segmentation/
// backbone/convnext.py:21-56 (dwconv 7x7 pad 3 groups=dim -> LayerNorm -> Linear 4x -> GELU -> Linear -> gamma ->
// drop_path -> residual); in the classification path the same block comes from timm's convnext_tiny (train.py:194).
//
// A depthwise conv is a 49-tap stencil per channel (no channel mixing): no MFMA, LDS-tiled, VALU-bound.  NHWC bf16
// in/out, fp32 accumulation, filters stored tap-major [7][7][C] (bf16 shadow of the fp32 master).
#include "common.h"
#include "icamd_internal.h"

namespace {

constexpr int TS = 8;              // output tile edge
constexpr int HS = TS + 6;         // halo edge
constexpr int CG = 32;             // channels per workgroup (4 vectors of 8)

// y[n,h,w,c] = bias[c] + sum_{r,s} x[n,h+r-3,w+s-3,c] * w[r][s][c]      (flip: taps mirrored = data gradient)
//            (+ addend[n,h,w,c])
//
// Register-window kernel.  A thread owns a strip of 7 horizontally adjacent outputs of one image row for one 8-channel
// vector and keeps their 56 fp32 accumulators in registers; for each kernel row it walks the 13 input vectors under the
// strip once (one 16 B LDS read and one bf16->fp32 unpack each) and feeds every vector to the up-to-7 outputs that see it
// through a different tap, with the row's seven tap vectors (fp32 in LDS) held in registers.  Per output vector that is
// 13 LDS reads and 13 unpacks per kernel row instead of 49 + 49 of each for the whole stencil done output-by-output, and
// the multiply-adds are written on float2 so they issue as v_pk_fma_f32: the stencil is VALU-bound (98 flop per output
// element, no MFMA shape), so instruction count is the roofline.
// A workgroup covers TR rows x NS strips (TR * NS = 64) of the [N*H] x W plane for 32 channels; rows are GLOBAL rows
// g = n*H + h, contiguous in NHWC memory across images, so small feature maps (14x14, 7x7) still fill the workgroup:
// the halo is staged from contiguous global rows and each thread skips the kernel rows that would leave its own image.
template <int NS>
__global__ __launch_bounds__(256, 2) void dwconv7_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                         const float* __restrict__ bias, const bf16_t* __restrict__ addend,
                                                         bf16_t* __restrict__ y, int N, int H, int W, int C, int flip) {
  constexpr int TR = 64 / NS;            // global rows per workgroup
  constexpr int HR = TR + 6;             // halo rows
  constexpr int HC = 7 * NS + 6;         // halo columns
  __shared__ __attribute__((aligned(16))) bf16_t sx[HR * HC][CG];
  __shared__ __attribute__((aligned(16))) float sw[49][CG];
  const int tiles_w = (W + 7 * NS - 1) / (7 * NS);
  const long long rows_total = (long long)N * H;
  int b = blockIdx.x;
  const int cgi = b % (C / CG); b /= (C / CG);
  const int tw = b % tiles_w;
  const long long g0 = (long long)(b / tiles_w) * TR;
  const int c0 = cgi * CG, w0 = tw * 7 * NS;
  const int tid = threadIdx.x;
  for (int i = tid; i < HR * HC * 4; i += 256) {
    const int v = i & 3, pix = i >> 2;
    const int hr = pix / HC, hc = pix - hr * HC;
    const long long g = g0 - 3 + hr;
    const int ww = w0 - 3 + hc;
    u32x4 val = {0u, 0u, 0u, 0u};
    if (g >= 0 && g < rows_total && (unsigned)ww < (unsigned)W) val = *(const u32x4*)(x + (g * W + ww) * C + c0 + v * 8);
    *(u32x4*)&sx[pix][v * 8] = val;
  }
  for (int i = tid; i < 49 * CG; i += 256) {
    const int t = i / CG, c = i - t * CG;
    sw[t][c] = bf16_to_f32(w[(long long)(flip ? 48 - t : t) * C + c0 + c]);
  }
  __syncthreads();
  const int v = tid & 3, strip = (tid >> 2) % NS, lr = (tid >> 2) / NS;
  const long long g = g0 + lr;
  const int h = (int)(g % H);
  f32x2 acc[7][4];
  {
    f32x2 b2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      b2[e] = (bias != nullptr) ? f32x2{bias[c0 + v * 8 + 2 * e], bias[c0 + v * 8 + 2 * e + 1]} : f32x2{0.f, 0.f};
#pragma unroll
    for (int o = 0; o < 7; ++o)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[o][e] = b2[e];
  }
#pragma unroll 1
  for (int r = 0; r < 7; ++r) {
    if ((unsigned)(h + r - 3) >= (unsigned)H) continue;   // that input row belongs to another image (or to none)
    f32x2 wv[7][4];
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      const f32x4 lo = *(const f32x4*)&sw[r * 7 + t][v * 8], hi = *(const f32x4*)&sw[r * 7 + t][v * 8 + 4];
      wv[t][0] = f32x2{lo[0], lo[1]}; wv[t][1] = f32x2{lo[2], lo[3]};
      wv[t][2] = f32x2{hi[0], hi[1]}; wv[t][3] = f32x2{hi[2], hi[3]};
    }
    const bf16_t* row = &sx[(lr + r) * HC + strip * 7][v * 8];
#pragma unroll
    for (int j = 0; j < 13; ++j) {
      const u32x4 xv = *(const u32x4*)(row + j * CG);
      f32x2 xu[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) xu[e] = f32x2{bf16_lo(xv[e]), bf16_hi(xv[e])};
#pragma unroll
      for (int o = 0; o < 7; ++o) {
        if (j - o >= 0 && j - o < 7) {   // compile-time: output o sees input j through tap s = j - o
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[o][e] = __builtin_elementwise_fma(xu[e], wv[j - o][e], acc[o][e]);
        }
      }
    }
  }
  if (g < rows_total) {
#pragma unroll
    for (int o = 0; o < 7; ++o) {
      const int ow = w0 + strip * 7 + o;
      if (ow < W) {
        const long long off = (g * W + ow) * C + c0 + v * 8;
        f32x2 r2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) r2[e] = acc[o][e];
        if (addend != nullptr) {
          const u32x4 a = *(const u32x4*)(addend + off);
#pragma unroll
          for (int e = 0; e < 4; ++e) { r2[e][0] += bf16_lo(a[e]); r2[e][1] += bf16_hi(a[e]); }
        }
        u32x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = pack_bf16x2(r2[e][0], r2[e][1]);
        *(u32x4*)(y + off) = ov;
      }
    }
  }
}

// dw[r][s][c] partial sums: grid = (C/32) * nb workgroups; workgroup (cg, j) walks the spatial tiles j, j+nb, ... of
// every image; thread (v, t) (t < 49) accumulates tap t for channel vector v over the tile's 64 pixels.
// part[j][49][C] fp32
__global__ __launch_bounds__(256) void dwconv7_wgrad_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                            float* __restrict__ part, int N, int H, int W, int C, int nb) {
  __shared__ __attribute__((aligned(16))) bf16_t sx[HS * HS][CG];
  __shared__ __attribute__((aligned(16))) bf16_t sd[TS * TS][CG];
  const int cgi = blockIdx.x % (C / CG), j = blockIdx.x / (C / CG);
  const int c0 = cgi * CG;
  const int tiles_w = (W + TS - 1) / TS, tiles_h = (H + TS - 1) / TS;
  const int ntiles = N * tiles_h * tiles_w;
  const int tid = threadIdx.x;
  const int v = tid & 3, t = tid >> 2;       // t in [0, 64): taps 0..48 active
  const int r = t / 7, s = t - r * 7;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int tile = j; tile < ntiles; tile += nb) {
    int q = tile;
    const int tw = q % tiles_w; q /= tiles_w;
    const int th = q % tiles_h;
    const int n = q / tiles_h;
    const int h0 = th * TS, w0 = tw * TS;
    __syncthreads();
    for (int i = tid; i < HS * HS * 4; i += 256) {
      const int vv = i & 3, pix = i >> 2;
      const int hh = h0 - 3 + pix / HS, ww = w0 - 3 + pix % HS;
      u32x4 val = {0u, 0u, 0u, 0u};
      if ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W)
        val = *(const u32x4*)(x + (((long long)n * H + hh) * W + ww) * C + c0 + vv * 8);
      *(u32x4*)&sx[pix][vv * 8] = val;
    }
    {
      const int vv = tid & 3, pix = tid >> 2;
      const int hh = h0 + pix / TS, ww = w0 + pix % TS;
      u32x4 val = {0u, 0u, 0u, 0u};
      if (hh < H && ww < W) val = *(const u32x4*)(dy + (((long long)n * H + hh) * W + ww) * C + c0 + vv * 8);
      *(u32x4*)&sd[pix][vv * 8] = val;
    }
    __syncthreads();
    if (t < 49) {
#pragma unroll 8
      for (int pix = 0; pix < TS * TS; ++pix) {
        const int ph = pix / TS, pw = pix % TS;
        const u32x4 dv = *(const u32x4*)&sd[pix][v * 8];
        const u32x4 xv = *(const u32x4*)&sx[(ph + r) * HS + pw + s][v * 8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[2 * e] = fmaf(bf16_lo(dv[e]), bf16_lo(xv[e]), acc[2 * e]);
          acc[2 * e + 1] = fmaf(bf16_hi(dv[e]), bf16_hi(xv[e]), acc[2 * e + 1]);
        }
      }
    }
  }
  if (t < 49) {
    float* dst = part + ((long long)j * 49 + t) * C + c0 + v * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[e] = acc[e];
  }
}

// out = inp + keep[b] * gamma[c] * z      (rows_per_image rows of C channels per sample; keep may be NULL)
__global__ __launch_bounds__(256) void layerscale_fwd_kernel(const bf16_t* __restrict__ z, const bf16_t* __restrict__ inp,
                                                             const float* __restrict__ gamma, const float* __restrict__ keep,
                                                             bf16_t* __restrict__ out, long long nvec, int cpr,
                                                             long long vec_per_image) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    const int cg = (int)(i % cpr) * 8;
    const float k = keep != nullptr ? keep[i / vec_per_image] : 1.f;
    const u32x4 zv = ((const u32x4*)z)[i];
    const u32x4 iv = ((const u32x4*)inp)[i];
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      o[e] = pack_bf16x2(fmaf(k * gamma[cg + 2 * e], bf16_lo(zv[e]), bf16_lo(iv[e])),
                         fmaf(k * gamma[cg + 2 * e + 1], bf16_hi(zv[e]), bf16_hi(iv[e])));
    ((u32x4*)out)[i] = o;
  }
}

// dz = dout * keep[b] * gamma[c]; partial column sums of dout * z * keep[b] (-> dgamma) in part[blk][2][C] (row 1 zero)
__global__ __launch_bounds__(256) void layerscale_bwd_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ z,
                                                             const float* __restrict__ gamma, const float* __restrict__ keep,
                                                             bf16_t* __restrict__ dz, float* __restrict__ part, long long rows,
                                                             int C, int rows_per_block, long long rows_per_image) {
  __shared__ float red[256 * 8];
  const int cpr = C >> 3;
  const int tid = threadIdx.x;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = (rows < r0 + rows_per_block) ? rows : r0 + rows_per_block;
  for (int cg0 = 0; cg0 < cpr; cg0 += 256) {
    const int tcols = (cpr - cg0 < 256) ? (cpr - cg0) : 256;
    const int rlanes = 256 / tcols;
    const int cgi = tid % tcols, rl = tid / tcols;
    float s[8], g8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; g8[e] = (rl < rlanes) ? gamma[(cg0 + cgi) * 8 + e] : 0.f; }
    if (rl < rlanes) {
      for (long long r = r0 + rl; r < r1; r += rlanes) {
        const long long off = r * cpr + cg0 + cgi;
        const float k = keep != nullptr ? keep[r / rows_per_image] : 1.f;
        const u32x4 dv = ((const u32x4*)dout)[off];
        const u32x4 zv = ((const u32x4*)z)[off];
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d0 = bf16_lo(dv[e]) * k, d1 = bf16_hi(dv[e]) * k;
          s[2 * e] += d0 * bf16_lo(zv[e]);
          s[2 * e + 1] += d1 * bf16_hi(zv[e]);
          o[e] = pack_bf16x2(d0 * g8[2 * e], d1 * g8[2 * e + 1]);
        }
        ((u32x4*)dz)[off] = o;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[tid * 8 + e] = s[e];
    __syncthreads();
    for (int o = tid; o < tcols * 8; o += 256) {
      const int cgo = o >> 3, e = o & 7;
      float t = 0.f;
      for (int l = 0; l < rlanes; ++l) t += red[(l * tcols + cgo) * 8 + e];
      part[((long long)blockIdx.x * 2 + 0) * C + (cg0 + cgo) * 8 + e] = t;
      part[((long long)blockIdx.x * 2 + 1) * C + (cg0 + cgo) * 8 + e] = 0.f;
    }
    __syncthreads();
  }
}

}  // namespace

int icamd_dwconv7_launch(const bf16_t* x, const bf16_t* w, const float* bias, const bf16_t* addend, bf16_t* y, int N, int H,
                         int W, int C, int flip, hipStream_t s) {
  if (C % CG != 0) return ICAMD_ERR_UNSUPPORTED;
  // strips of 7 outputs per row and workgroup: as many as the row needs, up to 8 (TR = 64 / NS global rows each)
  const int ns = W <= 7 ? 1 : (W <= 14 ? 2 : (W <= 28 ? 4 : 8));
  const long long rows = (long long)N * H;
  const long long blocks = ((rows + 64 / ns - 1) / (64 / ns)) * ((W + 7 * ns - 1) / (7 * ns)) * (C / CG);
  if (blocks <= 0 || blocks >= (1ll << 31)) return ICAMD_ERR_BAD_ARG;
  const dim3 grid((unsigned)blocks), block(256);
  if (ns == 1) hipLaunchKernelGGL(dwconv7_kernel<1>, grid, block, 0, s, x, w, bias, addend, y, N, H, W, C, flip);
  else if (ns == 2) hipLaunchKernelGGL(dwconv7_kernel<2>, grid, block, 0, s, x, w, bias, addend, y, N, H, W, C, flip);
  else if (ns == 4) hipLaunchKernelGGL(dwconv7_kernel<4>, grid, block, 0, s, x, w, bias, addend, y, N, H, W, C, flip);
  else hipLaunchKernelGGL(dwconv7_kernel<8>, grid, block, 0, s, x, w, bias, addend, y, N, H, W, C, flip);
  return icamd_launch_status();
}

int icamd_dwconv7_wgrad_blocks(int N, int H, int W, int C) {
  const long long tiles = (long long)N * ((H + TS - 1) / TS) * ((W + TS - 1) / TS);
  long long nb = 1024 / (C / CG);
  if (nb < 1) nb = 1;
  if (nb > tiles) nb = tiles;
  return (int)nb;
}

int icamd_dwconv7_wgrad_launch(const bf16_t* x, const bf16_t* dy, float* part, float* dw, int N, int H, int W, int C,
                               int accumulate, hipStream_t s) {
  if (C % CG != 0) return ICAMD_ERR_UNSUPPORTED;
  const int nb = icamd_dwconv7_wgrad_blocks(N, H, W, C);
  hipLaunchKernelGGL(dwconv7_wgrad_kernel, dim3((unsigned)(nb * (C / CG))), dim3(256), 0, s, x, dy, part, N, H, W, C, nb);
  // fold the nb partial rows with the slab reducer (many workgroups, fixed order) -- 49*C is a multiple of 4
  return icamd_slab_reduce_launch(part, dw, 49ll * C, nb, accumulate, s);
}

static unsigned int ls_grid(long long nvec) {
  long long blocks = (nvec + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  return (unsigned int)(blocks < 1 ? 1 : blocks);
}

int icamd_layerscale_fwd_launch(const bf16_t* z, const bf16_t* inp, const float* gamma, const float* keep, bf16_t* out,
                                long long rows, int C, long long rows_per_image, hipStream_t s) {
  if (C % 8 != 0) return ICAMD_ERR_BAD_ARG;
  const long long nvec = rows * (C / 8);
  hipLaunchKernelGGL(layerscale_fwd_kernel, dim3(ls_grid(nvec)), dim3(256), 0, s, z, inp, gamma, keep, out, nvec, C / 8,
                     rows_per_image * (C / 8));
  return icamd_launch_status();
}

int icamd_layerscale_bwd_blocks(long long rows) {
  long long rpb = (rows + 1023) / 1024;
  if (rpb < 32) rpb = 32;
  return (int)((rows + rpb - 1) / rpb);
}

int icamd_layerscale_bwd_launch(const bf16_t* dout, const bf16_t* z, const float* gamma, const float* keep, bf16_t* dz,
                                float* part, long long rows, int C, long long rows_per_image, hipStream_t s) {
  if (C % 8 != 0) return ICAMD_ERR_BAD_ARG;
  const int nblk = icamd_layerscale_bwd_blocks(rows);
  const int rpb = (int)((rows + nblk - 1) / nblk);
  hipLaunchKernelGGL(layerscale_bwd_kernel, dim3((unsigned)nblk), dim3(256), 0, s, dout, z, gamma, keep, dz, part, rows, C, rpb,
                     rows_per_image);
  return icamd_launch_status();
}
