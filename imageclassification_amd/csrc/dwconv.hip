// ConvNeXt-specific kernels for gfx950: depthwise 7x7 convolution (forward, data gradient, weight gradient) and the
// layer-scale + stochastic-depth + residual tail of the block.  Block spec: /root/reference/semantic_segmentation/
// backbone/convnext.py:21-56 (dwconv 7x7 pad 3 groups=dim -> LayerNorm -> Linear 4x -> GELU -> Linear -> gamma ->
// drop_path -> residual); in the classification path the same block comes from timm's convnext_tiny (train.py:194).
//
// A depthwise conv is a 49-tap stencil per channel (no channel mixing): no MFMA, LDS-tiled, VALU-bound.  NHWC bf16
// in/out, fp32 accumulation, filters stored tap-major [7][7][C] (bf16 shadow of the fp32 master).
#include "common.h"
#include "icamd_internal.h"
#include <cstdlib>

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

// ------------------------------------------------------------------------------------------------------------------------
// Register sliding-window form (round 3).  The tile kernel above staged a halo tile in LDS and re-read every input vector
// once per kernel row: 13 LDS reads + 13 unpacks per 49 multiply-adds, 296 us for stage 0 of ConvNeXt-T at batch 256 against
// 38 us of HBM time and ~45 us of VALU time.  Here a thread owns TWO channels (one packed fp32 pair) of a strip of 7 output
// columns and walks DOWN the image: the 49 taps (98 VGPRs) and a ring of 7 x 7 output accumulators (98 VGPRs) stay in
// registers, every input row is loaded ONCE (13 dwords straight from global memory -- the lanes of a wave are consecutive
// channel pairs, 256 B per pixel, so no LDS is needed for coalescing) and feeds all seven kernel rows: 343 v_pk_fma_f32 per
// 13 loads + 26 unpacks.  Output row h - 3 is complete when input row h has been consumed; the next row's loads are in
// flight under the current row's arithmetic.  flip = the data gradient (taps mirrored).
// ------------------------------------------------------------------------------------------------------------------------
template <bool ADD>
__global__ __launch_bounds__(256, 2) void dwconv7_rows_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                              const float* __restrict__ bias, const bf16_t* __restrict__ addend,
                                                              bf16_t* __restrict__ y, int N, int H, int W, int C, int flip,
                                                              int nstrips, int hsplit, unsigned int total, unsigned int last_off) {
  // blockIdx.y = which part of the image rows (uniform: every row test below is a scalar branch); blockIdx.x * 256 + tid walks
  // (channel pair, strip, image) with the channel pairs fastest: the lanes of a wave read consecutive dwords of a pixel
  const unsigned int t = blockIdx.x * 256u + threadIdx.x;
  if (t >= total) return;
  const int cpr = C >> 1;
  const int cp = (int)(t % (unsigned)cpr);
  unsigned int q = t / (unsigned)cpr;
  const int strip = (int)(q % (unsigned)nstrips);
  const int n = (int)(q / (unsigned)nstrips);
  const int rows_per = (H + hsplit - 1) / hsplit;
  const int h0 = blockIdx.y * rows_per, h1 = (H < h0 + rows_per) ? H : h0 + rows_per;
  if (h0 >= h1) return;
  const int c = cp * 2, w0 = strip * 7;
  // the taps: kernel rows 0..5 in registers (84 VGPRs), row 6 in LDS (this thread's own 7 pairs, read back once per input
  // row: all 49 in registers leave hipcc four VGPRs short and it spills two taps INTO the row loop)
  __shared__ f32x2 w6[7][256];
  f32x2 wv[42];
#pragma unroll
  for (int k = 0; k < 49; ++k) {
    const unsigned int raw = *(const unsigned int*)(w + (long long)(flip ? 48 - k : k) * C + c);
    if (k < 42) wv[k] = f32x2{bf16_lo(raw), bf16_hi(raw)};
    else w6[k - 42][threadIdx.x] = f32x2{bf16_lo(raw), bf16_hi(raw)};
  }
  const f32x2 b2 = (bias != nullptr) ? f32x2{bias[c], bias[c + 1]} : f32x2{0.f, 0.f};
  unsigned int colmask = 0;                      // bit j: input column w0 - 3 + j is inside the image
#pragma unroll
  for (int j = 0; j < 13; ++j) colmask |= ((unsigned)(w0 - 3 + j) < (unsigned)W ? 1u : 0u) << j;
  // 32-bit byte offsets from the (uniform) tensor bases (the launcher checks the tensors are < 4 GB).  Columns outside the
  // image are read from a CLAMPED offset (any valid dword) and replaced by zero with a select: no exec-mask branch per load.
  const unsigned int cb = (unsigned)C * 2u, rowb = (unsigned)W * cb;
  const unsigned int img = (unsigned)n * (unsigned)H * rowb;
  const unsigned int xo = img + (unsigned)(w0 - 3) * cb + (unsigned)c * 2u;   // + h * rowb + j * cb (wraps for w0 < 3: clamped)
  const unsigned int yo = img + (unsigned)w0 * cb + (unsigned)c * 2u;
  const unsigned char* const xB = (const unsigned char*)x;
  const unsigned char* const aB = (const unsigned char*)addend;
  unsigned char* const yB = (unsigned char*)y;
  const int lo = h0 - 3 > 0 ? h0 - 3 : 0, hi_end = h1 + 3 < H ? h1 + 3 : H;   // input rows that exist and matter
  f32x2 acc[7][7];
  // ADD (the data gradient's residual addend): an output row's accumulators START from its addend row, loaded when the ring
  // slot is recycled -- a whole input row of arithmetic before the slot's first multiply-add -- instead of 7 exposed loads in
  // front of every row's stores
  auto init_slot = [&](auto slotc, int ho) {
    constexpr int S = decltype(slotc)::value;
    if (ADD && ho < h1) {
      const unsigned int orow = yo + (unsigned)ho * rowb;
#pragma unroll
      for (int o = 0; o < 7; ++o) {
        const unsigned int off = orow + (unsigned)o * cb;
        const unsigned int a = ld_stream((const unsigned int*)(aB + (off < last_off ? off : last_off)));
        acc[S][o] = f32x2{bf16_lo(a), bf16_hi(a)};            // (columns past W: never stored)
      }
    } else {
#pragma unroll
      for (int o = 0; o < 7; ++o) acc[S][o] = b2;
    }
  };
  static_for<0, 7>([&](auto sc) {                             // slot s first serves the output row ho >= h0 with ho % 7 == s
    constexpr int S = decltype(sc)::value;
    init_slot(sc, h0 + ((S - h0 % 7) + 7) % 7);
  });
  unsigned int nxt[13];
  auto load_row = [&](int h) {
    const unsigned int ro = xo + (unsigned)h * rowb;
#pragma unroll
    for (int j = 0; j < 13; ++j) {
      const unsigned int off = ro + (unsigned)j * cb;
      // columns outside the image read the zero page: a select AFTER the load (rounds 3-4) made hipcc wait for every load of the
      // next row in front of this row's 49 multiply-adds -- the row was never "in flight under the arithmetic" (round 5, ISA)
      const unsigned char* src = ((colmask >> j) & 1u) ? xB + (off < last_off ? off : last_off) : (const unsigned char*)icamd_zero_page;
      nxt[j] = ld_stream((const unsigned int*)src);
    }
  };
  load_row(lo);
  for (int base = lo / 7 * 7; base < h1 + 3; base += 7) {
    static_for<0, 7>([&](auto kc) {
      constexpr int K = decltype(kc)::value;     // = hi % 7
      const int hi = base + K;
      if (hi >= lo && hi < h1 + 3) {
        if (hi < hi_end) {
          f32x2 xu[13];
#pragma unroll
          for (int j = 0; j < 13; ++j) xu[j] = f32x2{bf16_lo(nxt[j]), bf16_hi(nxt[j])};
          if (hi + 1 < hi_end) load_row(hi + 1);             // in flight under this row's arithmetic
          static_for<0, 7>([&](auto rc) {
            constexpr int R = decltype(rc)::value;           // kernel row: output row ho = hi - R + 3, ring slot ho % 7
            constexpr int SLOT = (K - R + 3 + 7) % 7;
            const int ho = hi - R + 3;
            if (ho >= h0 && ho < h1) {
#pragma unroll
              for (int sx = 0; sx < 7; ++sx) {                 // tap outer, output inner: seven independent accumulators in a row
                f32x2 wt;
                if constexpr (R < 6) wt = wv[R * 7 + sx];
                else wt = w6[sx][threadIdx.x];
#pragma unroll
                for (int o = 0; o < 7; ++o) acc[SLOT][o] = __builtin_elementwise_fma(xu[o + sx], wt, acc[SLOT][o]);
              }
            }
          });
        }
        const int ho = hi - 3;                               // complete: its last input row (kernel row 6) was hi
        if (ho >= h0) {
          constexpr int SLOT = (K + 4) % 7;
          const unsigned int orow = yo + (unsigned)ho * rowb;
#pragma unroll
          for (int o = 0; o < 7; ++o)
            if (w0 + o < W) *(unsigned int*)(yB + (orow + (unsigned)o * cb)) = pack_bf16x2(acc[SLOT][o][0], acc[SLOT][o][1]);
          init_slot(std::integral_constant<int, SLOT>{}, ho + 7);
        }
      }
    });
  }
}

// Weight gradient in the same form: dw[r][s][c] = sum dy[n,h,w,c] * x[n,h+r-3,w+s-3,c].  A thread owns two channels of a
// 7-column strip of one image and walks down it: input row h of x meets dy row h - r + 3 for every kernel row r.  The tap
// accumulators (7 packed pairs per kernel row) and a ring of the dy rows in reach stay in registers; with all seven kernel
// rows at once that is 98 + 98 VGPRs + the unpacked x row, which hipcc does not fit into 256 (hundreds of spills), so the walk
// is done TWICE -- kernel rows 0..3, then 4..6 -- each pass with its own ring (4 / 3 dy rows): 196 / 147 v_pk_fma_f32 per 13
// + 7 loads; the second read of x comes from L2.  The threads of a workgroup that hold the same channel pair are summed
// through LDS in a fixed order; one partial [49][C] row per workgroup, folded by the slab reducer.
template <int R0, int NR>
__device__ __forceinline__ void dw_wgrad_pass(const unsigned char* __restrict__ xB, const unsigned char* __restrict__ dB,
                                              float* __restrict__ part, float* red, bool active, unsigned int xo, unsigned int dyo,
                                              unsigned int cb, unsigned int rowb, unsigned int colmask, int w0, int H, int W, int C,
                                              int cpr_wg, int groups, int cslice, long long ib, float* __restrict__ part_b) {
  const int tid = threadIdx.x;
  f32x2 acc[NR * 7];
#pragma unroll
  for (int k = 0; k < NR * 7; ++k) acc[k] = f32x2{0.f, 0.f};
  // round 5: the first pass loads every dy row of the strip exactly once (rows 0..2 ahead of the walk, row h + 3 at x row h), so
  // the bias gradient sum(dy) falls out of it for seven packed adds per row (it was a separate pass over dy: icamd_colsum_rows)
  f32x2 bacc = f32x2{0.f, 0.f};
  if (active) {
    f32x2 dr[NR][7];                                         // ring: dy row hd lives in slot hd mod NR
    auto load_dy = [&](auto slotc, int hd) {
      constexpr int S = decltype(slotc)::value;
      const unsigned int ro = dyo + (unsigned)hd * rowb;
#pragma unroll
      for (int o = 0; o < 7; ++o) {
        const unsigned int v = (hd >= 0 && hd < H && w0 + o < W) ? ld_stream((const unsigned int*)(dB + (ro + (unsigned)o * cb))) : 0u;
        dr[S][o] = f32x2{bf16_lo(v), bf16_hi(v)};
        if constexpr (R0 == 0) bacc = bacc + dr[S][o];
      }
    };
    // what x row 0 needs besides the row it loads itself: dy rows 4 - R0 - NR .. 2 - R0 (zeros where outside the image)
    static_for<0, NR - 1>([&](auto ic) {
      constexpr int HD = 4 - R0 - NR + decltype(ic)::value;
      load_dy(std::integral_constant<int, ((HD % NR) + NR) % NR>{}, HD);
    });
    for (int base = 0; base < H; base += NR) {
      static_for<0, NR>([&](auto kc) {
        constexpr int K = decltype(kc)::value;               // = h mod NR
        const int h = base + K;
        if (h < H) {
          load_dy(std::integral_constant<int, (((K - R0 + 3) % NR) + NR) % NR>{}, h - R0 + 3);   // the newest row in reach
          const unsigned int ro = xo + (unsigned)h * rowb;
          f32x2 xu[13];
#pragma unroll
          for (int j = 0; j < 13; ++j) {
            // (the address-select form of the forward kernel's loads measured 2 % slower here at 56 x 56: 227.7 -> 232.5 us)
            const unsigned int raw = ((colmask >> j) & 1u) ? ld_stream((const unsigned int*)(xB + (ro + (unsigned)j * cb))) : 0u;
            xu[j] = f32x2{bf16_lo(raw), bf16_hi(raw)};
          }
          static_for<0, NR>([&](auto rc) {
            constexpr int RR = decltype(rc)::value;          // kernel row R0 + RR: dy row h - (R0 + RR) + 3 (zeros if outside)
            constexpr int SLOT = (((K - (R0 + RR) + 3) % NR) + NR) % NR;
#pragma unroll
            for (int o = 0; o < 7; ++o)                        // seven independent accumulators (taps) per output column
#pragma unroll
              for (int sx = 0; sx < 7; ++sx)
                acc[RR * 7 + sx] = __builtin_elementwise_fma(dr[SLOT][o], xu[o + sx], acc[RR * 7 + sx]);
          });
        }
      });
    }
  }
  // sum the `groups` threads that share a channel pair (fixed order): one kernel row of taps per pass through LDS
  static_for<0, NR>([&](auto rc) {
    constexpr int RR = decltype(rc)::value;
    __syncthreads();
#pragma unroll
    for (int sx = 0; sx < 7; ++sx) {
      red[(sx * 2 + 0) * 256 + tid] = acc[RR * 7 + sx][0];
      red[(sx * 2 + 1) * 256 + tid] = acc[RR * 7 + sx][1];
    }
    __syncthreads();
    for (int idx = tid; idx < 7 * 2 * cpr_wg; idx += 256) {
      const int k2 = idx / cpr_wg, l = idx - k2 * cpr_wg;   // k2 = tap-in-row * 2 + (channel of the pair)
      float sum = 0.f;
      for (int g = 0; g < groups; ++g) sum += red[k2 * 256 + g * cpr_wg + l];
      part[((long long)ib * 49 + (R0 + RR) * 7 + (k2 >> 1)) * C + (cslice * cpr_wg + l) * 2 + (k2 & 1)] = sum;
    }
  });
  if constexpr (R0 == 0) {
    if (part_b != nullptr) {      // the bias partial row of this workgroup, same fixed-order sum over the groups
      __syncthreads();
      red[tid] = bacc[0];
      red[256 + tid] = bacc[1];
      __syncthreads();
      for (int idx = tid; idx < 2 * cpr_wg; idx += 256) {
        const int k2 = idx / cpr_wg, l = idx - k2 * cpr_wg;
        float sum = 0.f;
        for (int g = 0; g < groups; ++g) sum += red[k2 * 256 + g * cpr_wg + l];
        part_b[(long long)ib * C + (cslice * cpr_wg + l) * 2 + k2] = sum;
      }
    }
  }
}

__global__ __launch_bounds__(256, 2) void dwconv7_wgrad_rows_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                                    float* __restrict__ part, float* __restrict__ part_b, int N,
                                                                    int H, int W, int C, int nstrips, int groups) {
  // workgroup = `groups` (n, strip) items x cpr_wg channel pairs, cpr_wg * groups == 256; blockIdx.x = cslice + nslices * item_block
  __shared__ float red[7 * 2 * 256];   // one kernel row of taps at a time (14 KB)
  const int cpr = C >> 1;
  const int cpr_wg = 256 / groups;
  const int nslices = cpr / cpr_wg;
  const int cslice = blockIdx.x % nslices;
  const long long ib = blockIdx.x / nslices;
  const int tid = threadIdx.x;
  const int cpl = tid % cpr_wg, grp = tid / cpr_wg;
  const int c = (cslice * cpr_wg + cpl) * 2;
  const long long item = ib * groups + grp;                 // (n, strip)
  const bool active = item < (long long)N * nstrips;
  const int strip = active ? (int)(item % nstrips) : 0, n = active ? (int)(item / nstrips) : 0;
  const int w0 = strip * 7;
  unsigned int colmask = 0;
#pragma unroll
  for (int j = 0; j < 13; ++j) colmask |= ((unsigned)(w0 - 3 + j) < (unsigned)W ? 1u : 0u) << j;
  const unsigned int cb = (unsigned)C * 2u, rowb = (unsigned)W * cb;
  const unsigned int img = (unsigned)n * (unsigned)H * rowb;
  const unsigned int xo = img + (unsigned)(w0 - 3) * cb + (unsigned)c * 2u;
  const unsigned int dyo = img + (unsigned)w0 * cb + (unsigned)c * 2u;
  dw_wgrad_pass<0, 4>((const unsigned char*)x, (const unsigned char*)dy, part, red, active, xo, dyo, cb, rowb, colmask, w0, H, W, C,
                      cpr_wg, groups, cslice, ib, part_b);
  dw_wgrad_pass<4, 3>((const unsigned char*)x, (const unsigned char*)dy, part, red, active, xo, dyo, cb, rowb, colmask, w0, H, W, C,
                      cpr_wg, groups, cslice, ib, nullptr);
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
    const u32x4 zv = ld_stream((const u32x4*)z + i);
    const u32x4 iv = ld_stream((const u32x4*)inp + i);
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
        const u32x4 dv = ld_stream((const u32x4*)dout + off);
        const u32x4 zv = ld_stream((const u32x4*)z + off);
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

static int dw_rows_mode() {   // ICAMD_DWCONV_ROWS=0: the LDS tile kernels of round 1/2 (tests compare both)
  static const int m = [] { const char* e = getenv("ICAMD_DWCONV_ROWS"); return e ? atoi(e) : 1; }();
  return m;
}
// The register sliding-window kernels address x / dy with 32-bit byte offsets: tensors of 4 GB and more take the LDS tile
// kernels (64-bit addressing).  ONE predicate for the forward, the weight gradient, its block count and its workspace size.
static bool dw_rows_usable(int N, int H, int W, int C) {
  return dw_rows_mode() && (long long)N * H * W * C * 2 < (1ll << 32);
}

int icamd_dwconv7_launch(const bf16_t* x, const bf16_t* w, const float* bias, const bf16_t* addend, bf16_t* y, int N, int H,
                         int W, int C, int flip, hipStream_t s) {
  if (C % CG != 0) return ICAMD_ERR_UNSUPPORTED;
  if (dw_rows_usable(N, H, W, C)) {
    const int nstrips = (W + 6) / 7;
    // threads = channel pairs x strips x images: 98 304 = 1 536 waves on every ConvNeXt-T stage at batch 256; 246 VGPRs
    // allow two waves per SIMD.  The image rows are cut in two from 28 rows up (6 halo rows are re-read per part; measured
    // at batch 256: 56 x 56 x 96 forward 151 -> 137 us, 28 x 28 x 192 68 -> 61; 14 x 14 and 7 x 7 lose), and further while
    // the grid would leave most of the chip idle.  The kernel is VALU-bound: 343 v_pk_fma_f32 (4 cycles each on a wave64) +
    // ~100 other instructions per input row and thread, two waves per SIMD, 1.5 rounds of the grid at stage 0 -- 55 TFLOP/s of
    // the 157 TFLOP/s fp32 vector peak, 2.3-2.5 TB/s.  (A second row of loads in flight -- 21 taps in LDS to pay for the
    // registers -- changed nothing: 137 -> 142 us.)
    const long long per_part = (long long)(C / 2) * nstrips * N;
    int hsplit = H >= 28 ? 2 : 1;
    while (per_part * hsplit < 1536ll * 64 / 2 && H / (hsplit * 2) >= 7) hsplit *= 2;
    static const int force = [] { const char* e = getenv("ICAMD_DW_HSPLIT"); return e ? atoi(e) : 0; }();
    if (force > 0) hsplit = force;
    const long long blocks = (per_part + 255) / 256;
    if (blocks <= 0 || blocks >= (1ll << 31)) return ICAMD_ERR_BAD_ARG;
    const unsigned int last_off = (unsigned int)((long long)N * H * W * C * 2 - 4);
    const dim3 grid((unsigned)blocks, (unsigned)hsplit);
    if (addend != nullptr)
      hipLaunchKernelGGL(dwconv7_rows_kernel<true>, grid, dim3(256), 0, s, x, w, bias, addend, y, N, H, W, C, flip, nstrips,
                         hsplit, (unsigned)per_part, last_off);
    else
      hipLaunchKernelGGL(dwconv7_rows_kernel<false>, grid, dim3(256), 0, s, x, w, bias, addend, y, N, H, W, C, flip, nstrips,
                         hsplit, (unsigned)per_part, last_off);
    return icamd_launch_status();
  }
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

// rows form: workgroup = 256 threads = groups x (256 / groups) channel pairs; groups = 256 / min(C/2, 256) so that a
// workgroup's lanes cover whole pixels' channel runs where C/2 < 256 (C/2 must then divide 256: 16, 32, 64, 128, 256) --
// else (C/2 = 48, 96, 192, 384: ConvNeXt's dims) the largest power-of-two slice that divides C/2
static void dw_wgrad_rows_geometry(int N, int W, int C, int* nstrips, int* groups, int* nslices, long long* iblocks) {
  const int cpr = C / 2;
  int slice = 256;
  while (cpr % slice != 0) slice >>= 1;            // C % 32 == 0  =>  slice >= 16
  *groups = 256 / slice;
  *nslices = cpr / slice;
  *nstrips = (W + 6) / 7;
  const long long items = (long long)N * *nstrips;
  *iblocks = (items + *groups - 1) / *groups;
}

int icamd_dwconv7_wgrad_blocks(int N, int H, int W, int C) {
  if (dw_rows_usable(N, H, W, C)) {
    int nstrips, groups, nslices; long long ib;
    dw_wgrad_rows_geometry(N, W, C, &nstrips, &groups, &nslices, &ib);
    return (int)ib;
  }
  const long long tiles = (long long)N * ((H + TS - 1) / TS) * ((W + TS - 1) / TS);
  long long nb = 1024 / (C / CG);
  if (nb < 1) nb = 1;
  if (nb > tiles) nb = tiles;
  return (int)nb;
}

// dbias != NULL (round 5): sum(dy) per channel out of the same pass (rows kernel only: icamd_dwconv7_wgrad_bias_supported_cxx);
// its partial rows [ib][C] follow the [ib][49][C] filter partials in the workspace
bool icamd_dwconv7_wgrad_bias_supported_cxx(int N, int H, int W, int C) { return C % CG == 0 && dw_rows_usable(N, H, W, C); }

int icamd_dwconv7_wgrad_launch(const bf16_t* x, const bf16_t* dy, float* part, float* dw, float* dbias, int N, int H, int W, int C,
                               int accumulate, hipStream_t s) {
  if (C % CG != 0) return ICAMD_ERR_UNSUPPORTED;
  if (dw_rows_usable(N, H, W, C)) {
    int nstrips, groups, nslices; long long ib;
    dw_wgrad_rows_geometry(N, W, C, &nstrips, &groups, &nslices, &ib);
    if (ib * nslices >= (1ll << 31)) return ICAMD_ERR_BAD_ARG;
    float* part_b = dbias != nullptr ? part + ib * 49 * C : nullptr;
    hipLaunchKernelGGL(dwconv7_wgrad_rows_kernel, dim3((unsigned)(ib * nslices)), dim3(256), 0, s, x, dy, part, part_b, N, H, W, C,
                       nstrips, groups);
    int rc = icamd_launch_status();
    if (rc) return rc;
    rc = icamd_slab_reduce_launch(part, dw, 49ll * C, (int)ib, accumulate, s);
    if (rc || dbias == nullptr) return rc;
    return icamd_slab_reduce_launch(part_b, dbias, (long long)C, (int)ib, accumulate, s);
  }
  if (dbias != nullptr) return ICAMD_ERR_UNSUPPORTED;
  const int nb = icamd_dwconv7_wgrad_blocks(N, H, W, C);
  hipLaunchKernelGGL(dwconv7_wgrad_kernel, dim3((unsigned)(nb * (C / CG))), dim3(256), 0, s, x, dy, part, N, H, W, C, nb);
  // fold the nb partial rows with the slab reducer (many workgroups, fixed order) -- 49*C is a multiple of 4
  return icamd_slab_reduce_launch(part, dw, 49ll * C, nb, accumulate, s);
}

// ---- round 5: layer scale folded into the Mlp's second Linear layer (DESIGN.md section 5, round-5 finding 10) -----------------
namespace {
// out = x + keep[n] * gamma[c] * (a W2^T + b2)[m, c] with keep in {0, cb}: the GEMM runs on W2' = cb * gamma[c] * W2[c, :] and
// b2' = cb * gamma[c] * b2[c] with x as its residual addend, and the (few) dropped samples are put right afterwards.

// job row (8 x int64): filter offset (elements, the same in the fp32 parameter arena and in its bf16 shadow), gamma offset, bias
// offset, folded-bias offset, C (filter rows), K (row length, % 4 == 0), first row of the job in the grid, float bits of cb
__global__ __launch_bounds__(256) void layerscale_fold_kernel(const float* __restrict__ params, bf16_t* __restrict__ shadow,
                                                              float* __restrict__ fold_bias, const long long* __restrict__ jobs,
                                                              int njobs) {
  const int row = (int)blockIdx.x;
  int j = 0;
  while (j + 1 < njobs && (int)jobs[(j + 1) * 8 + 6] <= row) ++j;
  const long long* jb = jobs + (long long)j * 8;
  const int c = row - (int)jb[6], K = (int)jb[5];
  if (c >= (int)jb[4]) return;
  const float g = params[jb[1] + c] * __int_as_float((int)jb[7]);
  const float* w = params + jb[0] + (long long)c * K;
  bf16_t* o = shadow + jb[0] + (long long)c * K;
  for (int k = (int)threadIdx.x * 4; k < K; k += 1024) {
    const f32x4 v = *(const f32x4*)(w + k);
    u32x2 r;
    r[0] = pack_bf16x2(g * v[0], g * v[1]);
    r[1] = pack_bf16x2(g * v[2], g * v[3]);
    *(u32x2*)(o + k) = r;
  }
  if (threadIdx.x == 0) fold_bias[jb[3] + c] = g * params[jb[2] + c];
}

// For every sample n with keep[n] == 0: dst1[n] = src1 ? src1[n] : 0 (bytes1 per sample) and dst2[n] = 0 (bytes2 per sample).
// grid (chunks, samples): the workgroups of kept samples leave at once.
__global__ __launch_bounds__(256) void rows_fix_kernel(const float* __restrict__ keep, unsigned char* __restrict__ dst1,
                                                       const unsigned char* __restrict__ src1, long long bytes1,
                                                       unsigned char* __restrict__ dst2, long long bytes2) {
  const int n = (int)blockIdx.y;
  if (keep[n] != 0.f) return;
  const long long stride = (long long)gridDim.x * 256 * 16;
  const u32x4 zero = {0u, 0u, 0u, 0u};
  if (dst1 != nullptr)
    for (long long o = ((long long)blockIdx.x * 256 + threadIdx.x) * 16; o < bytes1; o += stride)
      *(u32x4*)(dst1 + n * bytes1 + o) = src1 != nullptr ? *(const u32x4*)(src1 + n * bytes1 + o) : zero;
  if (dst2 != nullptr)
    for (long long o = ((long long)blockIdx.x * 256 + threadIdx.x) * 16; o < bytes2; o += stride)
      *(u32x4*)(dst2 + n * bytes2 + o) = zero;
}

// partial[n][c] = sum over the rows of sample n of dy[m][c] where keep[n] == 0, else 0 (one workgroup per sample, fixed order)
__global__ __launch_bounds__(256) void dropped_colsum_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ keep,
                                                             long long rows_per_image, int C, float* __restrict__ partial) {
  __shared__ float red[256 * 8];
  const int n = (int)blockIdx.x, tid = (int)threadIdx.x;
  const int cpr = C >> 3;
  const bool dropped = keep[n] == 0.f;   // workgroup-uniform
  for (int cg0 = 0; cg0 < cpr; cg0 += 256) {
    const int tcols = (cpr - cg0 < 256) ? (cpr - cg0) : 256;
    const int rlanes = 256 / tcols;
    const int cgi = tid % tcols, rl = tid / tcols;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    if (dropped && rl < rlanes) {
      const u32x4* base = (const u32x4*)dy + (long long)n * rows_per_image * cpr + cg0 + cgi;
      for (long long r = rl; r < rows_per_image; r += rlanes) {
        const u32x4 v = base[r * cpr];
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[2 * e] += bf16_lo(v[e]); s[2 * e + 1] += bf16_hi(v[e]); }
      }
    }
    if (dropped) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red[tid * 8 + e] = s[e];
      __syncthreads();
    }
    for (int o = tid; o < tcols * 8; o += 256) {
      const int cgo = o >> 3, e = o & 7;
      float t = 0.f;
      if (dropped)
        for (int l = 0; l < rlanes; ++l) t += red[(l * tcols + cgo) * 8 + e];
      partial[(long long)n * C + (cg0 + cgo) * 8 + e] = t;
    }
    if (dropped) __syncthreads();
  }
}

// One workgroup per output channel c of the folded layer:  S = colsum_all[c] - sum_n dropped[n][c];
//   dw[c][:] (+)= cb * gamma[c] * G[c][:],  dbias[c] (+)= cb * gamma[c] * S,  dgamma[c] (+)= cb * (<G[c][:], w[c][:]> + bias[c] * S)
__global__ __launch_bounds__(256) void layerscale_param_grads_kernel(const float* __restrict__ G, const float* __restrict__ w,
                                                                     const float* __restrict__ bias, const float* __restrict__ gamma,
                                                                     const float* __restrict__ colsum_all,
                                                                     const float* __restrict__ dropped, int n_images, float cb, int C,
                                                                     int K, float* __restrict__ dw, float* __restrict__ dbias,
                                                                     float* __restrict__ dgamma, int accumulate) {
  __shared__ float red[256], red2[256];
  const int c = (int)blockIdx.x, tid = (int)threadIdx.x;
  const float gc = gamma[c] * cb;
  float dot = 0.f, sd = 0.f;
  const float* g = G + (long long)c * K;
  const float* wr = w + (long long)c * K;
  float* o = dw + (long long)c * K;
  for (int k = tid * 4; k < K; k += 1024) {
    const f32x4 gv = *(const f32x4*)(g + k), wv = *(const f32x4*)(wr + k);
    dot += gv[0] * wv[0] + gv[1] * wv[1] + gv[2] * wv[2] + gv[3] * wv[3];
    f32x4 r = {gc * gv[0], gc * gv[1], gc * gv[2], gc * gv[3]};
    if (accumulate) { const f32x4 old = *(const f32x4*)(o + k); r[0] += old[0]; r[1] += old[1]; r[2] += old[2]; r[3] += old[3]; }
    *(f32x4*)(o + k) = r;
  }
  if (dropped != nullptr)
    for (int n = tid; n < n_images; n += 256) sd += dropped[(long long)n * C + c];
  red[tid] = dot; red2[tid] = sd;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if (tid < h) { red[tid] += red[tid + h]; red2[tid] += red2[tid + h]; }
    __syncthreads();
  }
  if (tid == 0) {
    const float S = colsum_all[c] - red2[0];
    const float db = gc * S, dg = cb * (red[0] + bias[c] * S);
    dbias[c] = accumulate ? dbias[c] + db : db;
    dgamma[c] = accumulate ? dgamma[c] + dg : dg;
  }
}
}  // namespace

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

int icamd_layerscale_fold_launch(const float* params, bf16_t* shadow, float* fold_bias, const long long* jobs, int njobs,
                                 int total_rows, hipStream_t s) {
  hipLaunchKernelGGL(layerscale_fold_kernel, dim3((unsigned)total_rows), dim3(256), 0, s, params, shadow, fold_bias, jobs, njobs);
  return icamd_launch_status();
}

int icamd_rows_fix_launch(const float* keep, int n_images, void* dst1, const void* src1, long long bytes1, void* dst2,
                          long long bytes2, hipStream_t s) {
  const long long big = bytes1 > bytes2 ? bytes1 : bytes2;
  long long chunks = (big + 65535) / 65536;       // 16 vectors per thread
  if (chunks < 1) chunks = 1;
  if (chunks > 64) chunks = 64;
  hipLaunchKernelGGL(rows_fix_kernel, dim3((unsigned)chunks, (unsigned)n_images), dim3(256), 0, s, keep, (unsigned char*)dst1,
                     (const unsigned char*)src1, bytes1, (unsigned char*)dst2, bytes2);
  return icamd_launch_status();
}

int icamd_dropped_colsum_launch(const bf16_t* dy, const float* keep, int n_images, long long rows_per_image, int C, float* partial,
                                hipStream_t s) {
  hipLaunchKernelGGL(dropped_colsum_kernel, dim3((unsigned)n_images), dim3(256), 0, s, dy, keep, rows_per_image, C, partial);
  return icamd_launch_status();
}

int icamd_layerscale_param_grads_launch(const float* G, const float* w, const float* bias, const float* gamma,
                                        const float* colsum_all, const float* dropped, int n_images, float cb, int C, int K,
                                        float* dw, float* dbias, float* dgamma, int accumulate, hipStream_t s) {
  hipLaunchKernelGGL(layerscale_param_grads_kernel, dim3((unsigned)C), dim3(256), 0, s, G, w, bias, gamma, colsum_all, dropped,
                     n_images, cb, C, K, dw, dbias, dgamma, accumulate);
  return icamd_launch_status();
}
