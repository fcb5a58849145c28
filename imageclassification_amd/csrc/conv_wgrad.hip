// Convolution weight gradient for gfx950: dW[co][tap][ci] = sum_m dY[m][co] * X[pixel(m)+tap][ci]
// (m = (n,oh,ow) runs over N*OH*OW output pixels). bf16 operands, fp32 MFMA accumulation, fp32 result.
//
// Replaces the wgrad half of `loss.backward()` in the reference's step (/root/reference/engine.py:64,72).
//
// GEMM view: D[kk][co] = sum_m Xcol[m][kk] * dY[m][co]: both operands are stored reduction-major in memory
// (a pixel row is contiguous in channels), so they are staged as [m][kk] / [m][co] LDS tiles by LDS-DMA
// (global_load_lds 16 B per lane; the Xcol rows are gathered per lane, out-of-image taps read zeros) and
// fed to v_mfma_f32_16x16x32_bf16 through ds_read_b64_tr_b16 (hardware transposed read): no transposed
// copy of any activation is ever made. 32 B column blocks of every tile row are XOR-swizzled (on the DMA
// source address and on the read) so the transposed reads of a half-wave hit 8 distinct bank groups.
// The m reduction is split over S workgroups per output tile; partial tiles go to an fp32 slab and a
// second kernel sums the S slabs in a fixed order (bitwise reproducible, no float atomics).
#include "common.h"
#include "icamd_internal.h"
#include <stdlib.h>

namespace {

constexpr int BKR = 64;  // reduction rows (pixels) per stage

template <int W>  // W = tile row width in elements (64 or 128)
__device__ __forceinline__ int tr_swz(int row) {
  if constexpr (W == 128) return (row & 3) | (((row >> 3) & 1) << 2);
  else return ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
}

__device__ __forceinline__ bf16x8 tr_read_pair(const unsigned char* p0, const unsigned char* p1) {
  bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3)))*)p0);
  bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3)))*)p1);
  return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

#ifndef ICAMD_WGRAD_STAGES
#define ICAMD_WGRAD_STAGES 1   // 1: single LDS stage + 3-4 workgroups per CU; 2: double buffer
#endif
#ifndef ICAMD_WGRAD_WAVES_PER_SIMD
#define ICAMD_WGRAD_WAVES_PER_SIMD 3
#endif

template <int BMK, int BNC>
__global__ __launch_bounds__(256, (ICAMD_WGRAD_STAGES == 1 ? ICAMD_WGRAD_WAVES_PER_SIMD : 2)) void conv_wgrad_kernel(const WgradParams p) {
  constexpr int NSTAGE = ICAMD_WGRAD_STAGES;
  constexpr int X_BYTES = BKR * BMK * 2;
  constexpr int Y_BYTES = BKR * BNC * 2;
  constexpr int STAGE_BYTES = X_BYTES + Y_BYTES;
  constexpr int XROWB = BMK * 2, YROWB = BNC * 2;
  constexpr int XCPR = BMK / 8, YCPR = BNC / 8;        // 16 B chunks per row
  constexpr int XRPI = 64 / XCPR, YRPI = 64 / YCPR;    // rows per wave-instruction
  constexpr int XJ = BMK / 32, YJ = BNC / 32;          // staging instructions per wave
  constexpr int KR = BMK / 32, CR = BNC / 32;          // 16-wide fragments per wave (kk, co)
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * STAGE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave & 1, wc = wave >> 1;

  // Block order: 8 consecutive pixel splits form a group; inside a group block ids go tile-major with the split as the
  // low 3 bits, so (blocks b and b+8 share an XCD) all output tiles of one split run on ONE XCD at about the same time and
  // the split's dY / X pixels are fetched from HBM once and re-read from that XCD's L2 by the other tiles.
  const unsigned int nt = (unsigned)(p.ntiles_k * p.ntiles_c);
  const unsigned int grp = blockIdx.x / (8u * nt);
  const unsigned int rem = blockIdx.x - grp * 8u * nt;
  const unsigned int ns = min(8u, (unsigned)p.S - grp * 8u);
  const unsigned int tile = rem / ns;
  const int split = (int)(grp * 8u + (rem - tile * ns));
  const int tile_c = (int)(tile % (unsigned)p.ntiles_c);
  const int tile_k = (int)(tile / (unsigned)p.ntiles_c);
  const int k0 = tile_k * BMK, c0 = tile_c * BNC;
  const int m_begin = split * p.rows_per_split;
  const int m_end = min(p.M, m_begin + p.rows_per_split);

  const bf16_t* __restrict__ x = p.x;
  const bf16_t* __restrict__ dy = p.dy;
  const bf16_t* zero = (const bf16_t*)icamd_zero_page;

  // ---- per-lane constant part of the Xcol gather: which (tap, ci) this lane's chunk is, per instruction j ----
  int x_row[XJ], x_dh[XJ], x_dw[XJ], x_ci[XJ];
#pragma unroll
  for (int j = 0; j < XJ; ++j) {
    const int row = (wave * XJ + j) * XRPI + lane / XCPR;
    const int pc = lane % XCPR;
    const int lc = pc ^ (tr_swz<BMK>(row) << 1);
    const int kk = k0 + lc * 8;
    x_row[j] = row;
    if (kk < p.Ktot) {
      const unsigned int t = fdiv((unsigned)kk, p.divCin);
      const unsigned int r = fdiv(t, p.divKW);
      x_ci[j] = kk - t * p.Cin;
      x_dh[j] = (int)r - p.pad;
      x_dw[j] = (int)(t - r * p.KW) - p.pad;
    } else {
      x_ci[j] = -1; x_dh[j] = 0; x_dw[j] = 0;
    }
  }
  int y_row[YJ], y_co[YJ];
#pragma unroll
  for (int j = 0; j < YJ; ++j) {
    const int row = (wave * YJ + j) * YRPI + lane / YCPR;
    const int pc = lane % YCPR;
    const int lc = pc ^ (tr_swz<BNC>(row) << 1);
    const int co = c0 + lc * 8;
    y_row[j] = row;
    y_co[j] = (co < p.Cout) ? co : -1;
  }

  auto stage = [&](int mbase, int buf) {
    unsigned char* sX = smem + buf * STAGE_BYTES;
    unsigned char* sY = sX + X_BYTES;
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      const int m = mbase + x_row[j];
      const bf16_t* src = zero;
      if (p.pointwise) {
        // 1x1 / stride 1 / no padding: output pixel m IS input pixel m -- no index arithmetic at all
        if (m < m_end && x_ci[j] >= 0) src = x + ((long long)m * p.Cin + x_ci[j]);
      } else if (m < m_end && x_ci[j] >= 0) {
        const unsigned int n = fdiv((unsigned)m, p.divHW);
        const unsigned int rem = m - n * (p.OH * p.OW);
        const unsigned int oh = fdiv(rem, p.divW);
        const unsigned int ow = rem - oh * p.OW;
        const int ih = (int)oh * p.stride + x_dh[j], iw = (int)ow * p.stride + x_dw[j];
        if ((unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW)
          src = x + (((n * p.IH + ih) * p.IW + iw) * p.Cin + x_ci[j]);
      }
      __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(sX + ((wave * XJ + j) * XRPI) * XROWB), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < YJ; ++j) {
      const int m = mbase + y_row[j];
      const bf16_t* src = (m < m_end && y_co[j] >= 0) ? dy + ((long long)m * p.Cout + y_co[j]) : zero;
      __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(sY + ((wave * YJ + j) * YRPI) * YROWB), 16, 0, 0);
    }
  };

  f32x4 acc[KR][CR];
#pragma unroll
  for (int i = 0; i < KR; ++i)
#pragma unroll
    for (int j = 0; j < CR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // Bias gradient (column sums of dY) for free: the workgroups of the first filter-column tile multiply their dY
  // fragments once more by an all-ones operand (every accumulator row then holds sum_m dY[m][co]); one wave per dY
  // fragment column does it.  +CR MFMAs per 16*CR on 1/ntiles_k of the workgroups instead of a separate pass over dY.
  const bool do_bias = p.bias_slab != nullptr && tile_k == 0 && wk == 0;   // wave-uniform
  f32x4 bacc[CR];
#pragma unroll
  for (int j = 0; j < CR; ++j) bacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16x8 ones = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};

  // transposed-read lane roles: 16-lane group g reads reduction rows 8g..8g+7 (two 4-row blocks),
  // lane 4q+pq of the group addresses row q, columns 4pq..4pq+3 of the 16-column block
  const int g = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;

  const int nsteps = (m_end - m_begin + BKR - 1) / BKR;
  auto compute = [&](int buf) {
    const unsigned char* sX = smem + buf * STAGE_BYTES;
    const unsigned char* sY = sX + X_BYTES;
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const int r0 = k2 * 32 + 8 * g + q, r1 = r0 + 4;
      bf16x8 xf[KR], yf[CR];
#pragma unroll
      for (int i = 0; i < KR; ++i) {
        const int lb = wk * KR + i;
        xf[i] = tr_read_pair(sX + r0 * XROWB + ((lb ^ tr_swz<BMK>(r0)) << 5) + 8 * pq,
                             sX + r1 * XROWB + ((lb ^ tr_swz<BMK>(r1)) << 5) + 8 * pq);
      }
#pragma unroll
      for (int j = 0; j < CR; ++j) {
        const int lb = wc * CR + j;
        yf[j] = tr_read_pair(sY + r0 * YROWB + ((lb ^ tr_swz<BNC>(r0)) << 5) + 8 * pq,
                             sY + r1 * YROWB + ((lb ^ tr_swz<BNC>(r1)) << 5) + 8 * pq);
      }
#pragma unroll
      for (int i = 0; i < KR; ++i)
#pragma unroll
        for (int j = 0; j < CR; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], yf[j], acc[i][j], 0, 0, 0);
      if (do_bias) {
#pragma unroll
        for (int j = 0; j < CR; ++j) bacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, yf[j], bacc[j], 0, 0, 0);
      }
    }
  };
  if constexpr (NSTAGE == 2) {
    if (nsteps > 0) stage(m_begin, 0);
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
      const int buf = st & 1;
      if (st + 1 < nsteps) stage(m_begin + (st + 1) * BKR, buf ^ 1);
      compute(buf);
      __syncthreads();
    }
  } else {
    for (int st = 0; st < nsteps; ++st) {
      stage(m_begin + st * BKR, 0);
      __syncthreads();   // vmcnt(0) + barrier: stage landed
      compute(0);
      __syncthreads();   // reads done before the refill
    }
  }

  // D[kk][co]: lane holds co = lane&15, kk = 4*(lane>>4) + reg  -> one 16 B fp32 store per fragment
  float* slab = p.slab + (long long)split * p.Cout * p.Ktot;
#pragma unroll
  for (int i = 0; i < KR; ++i)
#pragma unroll
    for (int j = 0; j < CR; ++j) {
      const int kk = k0 + (wk * KR + i) * 16 + 4 * (lane >> 4);
      const int co = c0 + (wc * CR + j) * 16 + (lane & 15);
      if (kk < p.Ktot && co < p.Cout) *(f32x4*)(slab + (long long)co * p.Ktot + kk) = acc[i][j];
    }
  if (do_bias && (lane >> 4) == 0) {
#pragma unroll
    for (int j = 0; j < CR; ++j) {
      const int co = c0 + (wc * CR + j) * 16 + (lane & 15);
      if (co < p.Cout) p.bias_slab[(long long)split * p.Cout + co] = bacc[j][0];
    }
  }
}

// out[i] = (accumulate ? out[i] : 0) + sum_s slab[s][i]; 16 B per lane; fixed summation order.
// A block owns OUTS consecutive float4 outputs and splits the S slabs over 256/OUTS slab lanes (4 loads in
// flight per thread), then folds the lanes through LDS in lane order.
template <int OUTS>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                                          long long n4, int S, int accumulate) {
  constexpr int LANES = 256 / OUTS;
  __shared__ f32x4 red[256];
  const int o = threadIdx.x % OUTS, l = threadIdx.x / OUTS;
  const long long i = (long long)blockIdx.x * OUTS + o;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  if (i < n4) {
    const f32x4* p = (const f32x4*)slab + i;
    int k = l;
    for (; k + 3 * LANES < S; k += 4 * LANES) {
      a0 += p[(long long)k * n4];
      a1 += p[(long long)(k + LANES) * n4];
      a2 += p[(long long)(k + 2 * LANES) * n4];
      a3 += p[(long long)(k + 3 * LANES) * n4];
    }
    for (; k < S; k += LANES) a0 += p[(long long)k * n4];
  }
  red[threadIdx.x] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (l == 0 && i < n4) {
    f32x4 s = accumulate ? ((const f32x4*)out)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < LANES; ++j) s += red[j * OUTS + o];
    ((f32x4*)out)[i] = s;
  }
}

template <int BMK, int BNC>
int launch(const WgradParams& p, hipStream_t stream) {
  dim3 grid((unsigned)(p.ntiles_k * p.ntiles_c * p.S));
  hipLaunchKernelGGL((conv_wgrad_kernel<BMK, BNC>), grid, dim3(256), 0, stream, p);
  return icamd_launch_status();
}

}  // namespace

static inline int wgrad_tile(int n) { return n <= 64 ? 64 : 128; }

void icamd_wgrad_plan(int M, int Cout, int Ktot, int* S, int* rows_per_split) {
  // Split of the pixel reduction: each workgroup should run ~64 stages (4096 pixels) -- long enough to amortise its
  // prologue and its fp32 slab tile, short enough to balance -- while the grid stays within [512, 2048] workgroups
  // (2-8 per CU).  Measured on MI355X over the ResNet-50 shapes (profiles/r01 notes).
  static const int rows_target = []() { const char* e = getenv("ICAMD_WGRAD_ROWS"); return e ? atoi(e) : 4096; }();
  const int bmk = wgrad_tile(Ktot), bnc = wgrad_tile(Cout);
  const int tiles = ((Ktot + bmk - 1) / bmk) * ((Cout + bnc - 1) / bnc);
  int s = (M + rows_target - 1) / rows_target;
  static const int blocks_min = []() { const char* e = getenv("ICAMD_WGRAD_BLOCKS_MIN"); return e ? atoi(e) : 512; }();
  const int smin = (blocks_min + tiles - 1) / tiles, smax = (2048 + tiles - 1) / tiles;
  if (s < smin) s = smin;
  if (s > smax) s = smax;
  const int scap = (M + BKR - 1) / BKR;
  if (s > scap) s = scap;
  if (s < 1) s = 1;
  int rows = (M + s - 1) / s;
  rows = (rows + BKR - 1) / BKR * BKR;
  *rows_per_split = rows;
  *S = (M + rows - 1) / rows;
}

int icamd_wgrad_launch(WgradParams& p, hipStream_t stream) {
  if (p.Cin % 8 != 0 || p.Cout % 8 != 0) return ICAMD_ERR_UNSUPPORTED;
  if ((long long)p.N * p.IH * p.IW * p.Cin >= (1ll << 31)) return ICAMD_ERR_UNSUPPORTED;
  if (p.M <= 0 || p.M >= (1 << 30)) return ICAMD_ERR_BAD_ARG;
  const int bmk = wgrad_tile(p.Ktot), bnc = wgrad_tile(p.Cout);
  p.ntiles_k = (p.Ktot + bmk - 1) / bmk;
  p.ntiles_c = (p.Cout + bnc - 1) / bnc;
  p.divHW = make_fastdiv((unsigned)(p.OH * p.OW));
  p.divW = make_fastdiv((unsigned)p.OW);
  p.divCin = make_fastdiv((unsigned)p.Cin);
  p.divKW = make_fastdiv((unsigned)p.KW);
  p.pointwise = (p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0) ? 1 : 0;
  if (bmk == 64) return bnc == 64 ? launch<64, 64>(p, stream) : launch<64, 128>(p, stream);
  return bnc == 64 ? launch<128, 64>(p, stream) : launch<128, 128>(p, stream);
}

int icamd_slab_reduce_launch(const float* slab, float* out, long long n, int S, int accumulate, hipStream_t stream) {
  if (n % 4 != 0) return ICAMD_ERR_BAD_ARG;
  const long long n4 = n / 4;
  if (n4 >= 64 * 1024) {   // large filters: 64 outputs x 4 slab lanes per block
    hipLaunchKernelGGL(slab_reduce_kernel<64>, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, stream, slab, out, n4, S,
                       accumulate);
  } else {                 // small filters, many slabs: 16 outputs x 16 slab lanes per block
    hipLaunchKernelGGL(slab_reduce_kernel<16>, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, stream, slab, out, n4, S,
                       accumulate);
  }
  return icamd_launch_status();
}
