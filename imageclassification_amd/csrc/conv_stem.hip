// ResNet stem (7x7 / stride 2 / pad 3, 3 -> 64 channels) forward on the rgb4 layout, filter resident in registers
// (reference: timm ResNet.conv1 under model(samples), /root/reference/engine.py:48,51).
//
// Layout (icamd_pack_input_rgb4): image [N][H][W+8][4] bf16 (3 zero columns left, 5 right, RGB + one zero channel); filter
// [Cout][8 rows][8 pixels][4] (row 7 and pixel 7 zero).  The window row r of output (oh, ow) is the 64 contiguous bytes of
// padded pixels 2*ow .. 2*ow+7 of image row 2*oh - 3 + r: the k index of the implicit GEMM is r*32 + pixel*4 + channel.
//
// conv_igemm's KMODE 3 stages, per 128-pixel tile, 4 k-steps of [128 pixels][64 B] gathered rows plus the filter: every image
// row is fetched ~3.5 times and the filter once per tile -- 14.2 M L2 requests per launch, TCC 90 % busy, 243 us against a
// 112 us roof.  Here (the scheme of conv3x3_c64_resident_kernel / conv1x1_resident.hip):
//   * a wave keeps its 32 output channels' filter rows 0..6 in 56 VGPRs (row 7 is zero and is skipped: 7 k-steps, not 8);
//   * persistent workgroups; a tile is TWO output rows of one image; its 9 image rows are one contiguous 9 * (W+8) * 8 B
//     range, copied to LDS as it lies (LDS-DMA, double-buffered, rows outside the image from the zero page);
//   * the A fragment of (output row, 16 pixels, kernel row r) is 16 B per lane at stride 16 B -- conflict-free without any
//     swizzle -- at ONE per-lane base address plus an immediate offset (buffer, r, fragment);
//   * four waves = 2 output rows x 2 channel halves; wave-private epilogue through two 16-row LDS patches; BatchNorm
//     partial sums in registers across tiles, one partial row per workgroup.
#ifndef ICAMD_STEM_NT
#define ICAMD_STEM_NT 0   // cache policy of the once-read LDS-DMA streams of this unit: 0 default, 2 non-temporal (round 5 A/B)
#endif
#include "common.h"
#include "icamd_internal.h"
#include <cstdlib>

namespace {

// MF = OW / 16 fragments per output row; PITCH = (W + 8) * 8 bytes per image row.
template <int MF>
__global__ __launch_bounds__(256, 2) void stem7x7s2_resident_kernel(const StemParams p, const int ntiles) {
  constexpr int OW = MF * 16, W = 2 * OW, PITCH = (W + 8) * 8;
  constexpr int A_ROWS = 9;
  constexpr int A_BYTES = (A_ROWS * PITCH + 1023) / 1024 * 1024;
  constexpr int NINST = A_BYTES / 1024;
  constexpr int IPW = (NINST + 3) / 4;
  constexpr int EROW = 64;                       // 32 channels per wave
  constexpr int E_WAVE = 2 * 16 * EROW;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * A_BYTES + 4 * E_WAVE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wm = wave >> 1;      // channel half, output row of the tile
  const int fr = lane & 15, fq = lane >> 4;
  const bf16_t* zero = (const bf16_t*)icamd_zero_page;
  const unsigned lds_base = (unsigned)(uintptr_t)LPTR(smem);
  unsigned char* const sE = smem + 2 * A_BYTES + wave * E_WAVE;
  const int OH = p.OH, H = p.H;
  const int tiles_per_image = OH >> 1;

  // ---- staging: the tile's 9 image rows 2*oh0 - 3 .. 2*oh0 + 5 are contiguous in memory; instruction q = j*4 + wave
  auto stage = [&](int tile, int buf) {
    const int n = tile / tiles_per_image, oh0 = (tile - n * tiles_per_image) * 2;
    const int ih0 = 2 * oh0 - 3;
    const bf16_t* base = p.x + ((long long)n * H + ih0) * (PITCH / 2);
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
      const int q = j * 4 + wave;
      if (q < NINST) {
        const int byte = q * 1024 + lane * 16;
        const int row = byte / PITCH;             // 0..8 (9 and up: the rounding tail, never read)
        const int ih = ih0 + row;
        const bf16_t* src = (row < A_ROWS && (unsigned)ih < (unsigned)H) ? base + (byte >> 1) : zero;
        __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(smem + buf * A_BYTES + q * 1024), 16, 0, ICAMD_STEM_NT);
      }
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) stage(tile, 0);             // in flight under the filter loads

  // ---- the filter: fragment (r, j) = channel wn*32 + j*16 + fr, k = r*32 + fq*8 .. +7
  bf16x8 wf[7][2];
#pragma unroll
  for (int r = 0; r < 7; ++r)
#pragma unroll
    for (int j = 0; j < 2; ++j) wf[r][j] = *(const bf16x8*)(p.w + (wn * 32 + j * 16 + fr) * 256 + r * 32 + fq * 8);
  __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): the filter and the first tile have arrived

  // one read address per lane: image row 2*wm (+ r), padded pixel 2*fr (+ 32*i), 16 B chunk fq
  const unsigned ra = lds_base + (unsigned)(2 * wm * PITCH + fr * 16 + fq * 16);

  f32x2 s1[4], s2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { s1[e] = f32x2{0.f, 0.f}; s2[e] = f32x2{0.f, 0.f}; }
  const bool want_stats = p.stats != nullptr;
  const bool epi = p.bias != nullptr || p.relu;          // uniform
  f32x4 bias4[2];                                         // MFMA layout: a lane owns channels wn*32 + j*16 + 4*fq .. +3
#pragma unroll
  for (int j = 0; j < 2; ++j)
    bias4[j] = p.bias != nullptr ? *(const f32x4*)(p.bias + wn * 32 + j * 16 + 4 * fq) : f32x4{0.f, 0.f, 0.f, 0.f};

  auto do_tile = [&](auto bufc, int t) {
    constexpr int BUF = decltype(bufc)::value;
    f32x4 acc[2][MF];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < MF; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    static_for<0, 7>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      bf16x8 xf[MF];
      static_for<0, MF>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        xf[i] = lds_read128_off<BUF * A_BYTES + r * PITCH + i * 256>(ra);
      });
      static_for<0, MF>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(xf[i]) : "n"(MF - 1 - i) : "memory");
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[r][j], xf[i], acc[j][i], 0, 0, 0);
      });
    });

    // ---- epilogue, wave-private: this wave's output row (n, oh0 + wm), 32 channels at wn*32
    const int n = t / tiles_per_image, oh = (t - n * tiles_per_image) * 2 + wm;
    bf16_t* orow = p.y + (((long long)n * OH + oh) * OW) * 64 + wn * 32;
    static_for<0, (MF + 1) / 2>([&](auto hc) {
      constexpr int h = decltype(hc)::value;
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = h * 2 + ii;
        if (i < MF) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            f32x4 v = acc[j][i];
            if (epi) {   // inference (round 4): + folded BatchNorm shift, ReLU, one rounding -- evaluate()'s stem ran on conv_igemm
              v += bias4[j];
              if (p.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];
              }
            }
            u32x2 pk;
            pk[0] = pack_bf16x2(v[0], v[1]);
            pk[1] = pack_bf16x2(v[2], v[3]);
            const int slot = j * 4 + fq;          // 8 B slot of the 64 B row; 16 B chunk = slot >> 1
            const int ch = ((slot >> 1) ^ (fr >> 2)) & 3;
            *(u32x2*)(sE + ii * 16 * EROW + fr * EROW + (ch << 4) + ((slot & 1) << 3)) = pk;
          }
        }
      }
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {            // 16 rows per read instruction (4 lanes per 64 B row)
        const int i = h * 2 + rr;
        if (i < MF) {
          const int prow = lane >> 2, c = lane & 3;
          const int ch = (c ^ (prow >> 2)) & 3;
          const u32x4 o = *(const u32x4*)(sE + (rr * 16 + prow) * EROW + (ch << 4));
          *(u32x4*)(orow + (i * 16 + prow) * 64 + c * 8) = o;
          if (want_stats) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const f32x2 v = {bf16_lo(o[e]), bf16_hi(o[e])};
              s1[e] += v;
              s2[e] = __builtin_elementwise_fma(v, v, s2[e]);
            }
          }
        }
      }
    });
  };

  const int step = gridDim.x;
  for (; tile < ntiles; tile += 2 * step) {
    // this tile has landed for every wave, every wave is done with the other buffer; behind the loads in the queue are
    // the previous tile's MF row stores
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MF) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tile + step < ntiles) stage(tile + step, 1);
    do_tile(std::integral_constant<int, 0>{}, tile);
    if (tile + step >= ntiles) break;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MF) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tile + 2 * step < ntiles) stage(tile + 2 * step, 0);
    do_tile(std::integral_constant<int, 1>{}, tile + step);
  }

  if (want_stats) {
    // one partial row per workgroup (row blockIdx.x of the [ceil(M/128)] table); rows no workgroup owns are zero
    __syncthreads();
    float* red = (float*)smem;                    // [2 rows x 16 lane groups][2][64]
    const int c = lane & 3, g = wm * 16 + (lane >> 2);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[(g * 2 + 0) * 64 + wn * 32 + c * 8 + 2 * e] = s1[e][0];
      red[(g * 2 + 0) * 64 + wn * 32 + c * 8 + 2 * e + 1] = s1[e][1];
      red[(g * 2 + 1) * 64 + wn * 32 + c * 8 + 2 * e] = s2[e][0];
      red[(g * 2 + 1) * 64 + wn * 32 + c * 8 + 2 * e + 1] = s2[e][1];
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, cc = tid & 63;
      float s = 0.f;
#pragma unroll 8
      for (int k = 0; k < 32; ++k) s += red[(k * 2 + which) * 64 + cc];
      p.stats[((long long)blockIdx.x * 2 + which) * 64 + cc] = s;
      const int nrows = (p.N * OH * OW + 127) / 128;
      for (int r = blockIdx.x + gridDim.x; r < nrows; r += gridDim.x) p.stats[((long long)r * 2 + which) * 64 + cc] = 0.f;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Stem weight gradient, same staging: dw[co][kr*32 + px*4 + ch] = sum over output pixels of dy[.][co] * window[.][k].
// A workgroup (8 waves) owns the WHOLE 64 x 224 filter gradient (kernel rows 0..6; row 7 of the arena layout is written as
// zeros) for a range of tiles: wave = (16-channel fragment of co) x (16-element half of a kernel row's 32 k), 7 accumulator
// tiles = 28 VGPRs.  A tile is two output rows = 32*MF pixels = MF reduction steps; both operands are read transposed
// (ds_read_b64_tr_b16, the reduction index is the pixel): the window operand straight from the linear copy of the 9 image
// rows (a pixel's window is 16 B further than its neighbour's: rows of the transposed read overlap, no bank conflicts), dy
// from [pixel][64] rows with the 32 B-block swizzle of conv_wgrad.hip.  Per lane 2*MF + 2 read addresses, constants of the
// whole kernel; kernel row, half, buffer and step are immediate offsets.  Slabs [workgroup][64][256] + slab_reduce_kernel
// (which clears the entries of window pixel 7, as for the implicit-GEMM form).
template <int MF>
__global__ __launch_bounds__(512, 2) void stem7x7s2_wgrad_resident_kernel(const StemWgradParams p, const int ntiles) {
  constexpr int OW = MF * 16, W = 2 * OW, PITCH = (W + 8) * 8;
  constexpr int A_ROWS = 9;
  constexpr int X_BYTES = (A_ROWS * PITCH + 1023) / 1024 * 1024;
  constexpr int XINST = X_BYTES / 1024;
  constexpr int TPX = 2 * OW;                    // pixels per tile
  constexpr int Y_BYTES = TPX * 128;
  constexpr int YINST = Y_BYTES / 1024;
  constexpr int YBASE = 2 * X_BYTES;
  static_assert(YBASE + 2 * Y_BYTES <= 160 * 1024 && (MF - 1) * 4096 < 65536 && X_BYTES + 6 * PITCH < 65536, "LDS map / offsets");
  __shared__ __attribute__((aligned(16))) unsigned char smem[YBASE + 2 * Y_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cf = wave & 3, kh = wave >> 2;       // channel fragment of co, 16-element half of a kernel row
  const bf16_t* zero = (const bf16_t*)icamd_zero_page;
  const unsigned lds_base = (unsigned)(uintptr_t)LPTR(smem);
  const int OH = p.OH, H = p.H;
  const int tiles_per_image = OH >> 1;

  // ---- staging (8 waves): x as in the forward kernel; dy rows keep their 32 B blocks at block ^ key(row mod 16)
  auto stage = [&](int tile, int buf) {
    const int n = tile / tiles_per_image, oh0 = (tile - n * tiles_per_image) * 2;
    const int ih0 = 2 * oh0 - 3;
    const bf16_t* xb = p.x + ((long long)n * H + ih0) * (PITCH / 2);
#pragma unroll
    for (int j = 0; j < (XINST + 7) / 8; ++j) {
      const int q = j * 8 + wave;
      if (q < XINST) {
        const int byte = q * 1024 + lane * 16;
        const int row = byte / PITCH;
        const int ih = ih0 + row;
        const bf16_t* src = (row < A_ROWS && (unsigned)ih < (unsigned)H) ? xb + (byte >> 1) : zero;
        __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(smem + buf * X_BYTES + q * 1024), 16, 0, ICAMD_STEM_NT);
      }
    }
    const bf16_t* yb = p.dy + (((long long)n * OH + oh0) * OW) * 64;     // two output rows: TPX contiguous pixels
#pragma unroll
    for (int j = 0; j < (YINST + 7) / 8; ++j) {
      const int q = j * 8 + wave;
      if (q < YINST) {
        const int row = q * 8 + (lane >> 3);      // pixel of the tile
        const int key = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
        const int chunk = ((((lane & 7) >> 1) ^ key) << 1) | (lane & 1);
        __builtin_amdgcn_global_load_lds(GPTR(yb + row * 64 + chunk * 8), LPTR(smem + YBASE + buf * Y_BYTES + q * 1024), 16, 0, ICAMD_STEM_NT);
      }
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) stage(tile, 0);

  // ---- read roles (ds_read_b64_tr_b16: lane -> reduction row 8g + q (+4), 8 B at 8*pq of a 32 B block)
  const int g = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
  unsigned xa[MF][2], ya[2][2];   // dy: [buffer][row select] (its buffers lie beyond the 16-bit immediate)
#pragma unroll
  for (int st = 0; st < MF; ++st)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int pxl = st * 32 + 8 * g + q + 4 * s2;        // pixel of the tile
      const int orow = pxl >= OW ? 1 : 0, ow = pxl - orow * OW;
      xa[st][s2] = lds_base + (unsigned)(2 * orow * PITCH + ow * 16 + kh * 32 + 8 * pq);
    }
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const int row = 8 * g + q + 4 * s2;
    const int key = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
    ya[0][s2] = lds_base + (unsigned)(YBASE + row * 128 + ((cf ^ key) << 5) + 8 * pq);
    ya[1][s2] = ya[0][s2] + (unsigned)Y_BYTES;
  }

  f32x4 acc[7];
#pragma unroll
  for (int r = 0; r < 7; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto do_tile = [&](auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
    static_for<0, MF>([&](auto stc) {
      constexpr int st = decltype(stc)::value;
      const bf16x8 yf = tr_read_pair_off<st * 4096>(ya[BUF][0], ya[BUF][1]);
      bf16x8 xf[7];
      static_for<0, 7>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        xf[r] = tr_read_pair_off<BUF * X_BYTES + r * PITCH>(xa[st][0], xa[st][1]);
      });
      static_for<0, 7>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        // reads return in issue order: kernel row r may start once at most 2 * (6 - r) reads are outstanding
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(xf[r]) : "n"(2 * (6 - r)) : "memory");
        acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[r], yf, acc[r], 0, 0, 0);
      });
    });
  };

  const int step = gridDim.x;
  for (; tile < ntiles; tile += 2 * step) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // this tile landed for every wave; the other buffers are free again
    if (tile + step < ntiles) stage(tile + step, 1);
    do_tile(std::integral_constant<int, 0>{});
    if (tile + step >= ntiles) break;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tile + 2 * step < ntiles) stage(tile + 2 * step, 0);
    do_tile(std::integral_constant<int, 1>{});
  }

  // acc[r]: rows = k elements kh*16 + 4*(lane>>4) .. +3 of kernel row r, column = output channel cf*16 + (lane & 15)
  float* slab = p.slab + (long long)blockIdx.x * 64 * 256;
  const int co = cf * 16 + (lane & 15);
#pragma unroll
  for (int r = 0; r < 7; ++r) *(f32x4*)(slab + co * 256 + r * 32 + kh * 16 + 4 * (lane >> 4)) = acc[r];
  *(f32x4*)(slab + co * 256 + 7 * 32 + kh * 16 + 4 * (lane >> 4)) = f32x4{0.f, 0.f, 0.f, 0.f};   // kernel row 7 of the layout
}

int mode() {
  static const int m = [] { const char* e = getenv("ICAMD_STEM_RESIDENT"); return e ? atoi(e) : 1; }();
  return m;
}

template <int MF>
int launch(const StemParams& p, hipStream_t stream) {
  const int ntiles = p.N * (p.OH / 2);
  int grid = 2 * icamd_num_cus();
  const int nrows = (p.N * p.OH * p.OW + 127) / 128;
  if (grid > ntiles) grid = ntiles;
  if (grid > nrows) grid = nrows;
  hipLaunchKernelGGL((stem7x7s2_resident_kernel<MF>), dim3((unsigned)grid), dim3(256), 0, stream, p, ntiles);
  return icamd_launch_status();
}

template <int MF>
int launch_wgrad(const StemWgradParams& p, int grid, hipStream_t stream) {
  hipLaunchKernelGGL((stem7x7s2_wgrad_resident_kernel<MF>), dim3((unsigned)grid), dim3(512), 0, stream, p,
                     p.N * (p.OH / 2));
  return icamd_launch_status();
}

}  // namespace

// one workgroup (and one slab) per CU, at most one per tile
int icamd_stem_wgrad_resident_splits(int N, int H) {
  const int ntiles = N * (H / 4), cus = icamd_num_cus();
  return ntiles < cus ? ntiles : cus;
}

int icamd_stem_wgrad_resident_launch(StemWgradParams& p, int S, hipStream_t stream) {
  if (!icamd_stem_resident_wanted(p.N, p.H, p.W, 64)) return ICAMD_ERR_UNSUPPORTED;
  p.OH = p.H / 2; p.OW = p.W / 2;
  switch (p.OW / 16) {
    case 4: return launch_wgrad<4>(p, S, stream);
    case 5: return launch_wgrad<5>(p, S, stream);
    case 6: return launch_wgrad<6>(p, S, stream);
    case 7: return launch_wgrad<7>(p, S, stream);
    default: return launch_wgrad<8>(p, S, stream);
  }
}

// 64 output channels, square-ish maps whose output width is 64..128 in steps of 16 and whose output height is even
bool icamd_stem_resident_wanted(int N, int H, int W, int Cout) {
  if (mode() == 0 || Cout != 64 || W % 2 != 0 || H % 2 != 0) return false;
  const int OW = W / 2, OH = H / 2;
  return OW % 16 == 0 && OW >= 64 && OW <= 128 && OH % 2 == 0 && (long long)N * OH * OW < (1ll << 30);
}

int icamd_stem_resident_launch(StemParams& p, hipStream_t stream) {
  if (!icamd_stem_resident_wanted(p.N, p.H, p.W, 64)) return ICAMD_ERR_UNSUPPORTED;
  p.OH = p.H / 2; p.OW = p.W / 2;
  switch (p.OW / 16) {
    case 4: return launch<4>(p, stream);
    case 5: return launch<5>(p, stream);
    case 6: return launch<6>(p, stream);
    case 7: return launch<7>(p, stream);
    default: return launch<8>(p, stream);
  }
}
