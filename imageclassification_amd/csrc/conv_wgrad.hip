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
  if constexpr (W >= 128) return (row & 3) | (((row >> 3) & 1) << 2);   // 256-wide rows: same key on the low 3 block bits
  else return ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
}

__device__ __forceinline__ bf16x8 tr_read_pair(const unsigned char* p0, const unsigned char* p1) {
  bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3)))*)p0);
  bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3)))*)p1);
  return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// Workgroup -> (output tile, pixel split).  Dispatch is round-robin over the 8 XCDs (blocks b and b + 8 share one; speed only,
// never correctness): XCD x runs blocks x, x + 8, ... in that order.
//   xcd_chunk = 1 (round 5): the split-major list L = split * tiles + tile is cut into 8 CONTIGUOUS chunks, one per XCD, walked in
//     dispatch order -- an XCD works on at most two neighbouring splits at a time and every tile of a split that it owns reads the
//     split's pixel rows from ITS L2.  This is a bijection for any grid size.
//   xcd_chunk = 0 (rounds 1-4): groups of 8 consecutive splits, the split as the low 3 bits of the block id inside a group.  Right
//     only while S is a multiple of 8: with S = 7 (ViT-B/16's 768 x 3072 weight gradients: 36 tiles of 256 x 256, one workgroup per
//     CU) block b = tile * 7 + split lands on XCD (tile * 7 + split) % 8, every split's tiles are spread over all eight L2s and each
//     of them fetches the split's rows again: PMC 891 MB fetched per launch for 310 MB of operands (profiles/r04_pmc_traffic_vit.json).
__device__ __forceinline__ void wgrad_block_order(const WgradParams& p, unsigned int& tile, int& split) {
  const unsigned int nt = (unsigned)(p.ntiles_k * p.ntiles_c);
  if (p.xcd_chunk) {
    const unsigned int nblk = gridDim.x, xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
    const unsigned int q = nblk >> 3, r = nblk & 7u;
    const unsigned int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    split = (int)(L / nt);
    tile = L - (unsigned)split * nt;
    return;
  }
  const unsigned int grp = blockIdx.x / (8u * nt);
  const unsigned int rem = blockIdx.x - grp * 8u * nt;
  const unsigned int ns = min(8u, (unsigned)p.S - grp * 8u);
  tile = rem / ns;
  split = (int)(grp * 8u + (rem - tile * ns));
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

  // Block order (wgrad_block_order): all output tiles of one pixel split run on ONE XCD at about the same time, so the split's
  // dY / X pixels are fetched from HBM once and re-read from that XCD's L2 by the other tiles.
  unsigned int tile;
  int split;
  wgrad_block_order(p, tile, split);
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
    if (p.stem7) {
      // k = row*32 + pixel*4 + channel over the [N][H][W+8][4] image: a 16 B chunk is two pixels of one kernel row; row 7
      // does not exist (zero page), pixel 7's gradient is cleared by the slab reducer
      const int r = kk >> 5;
      x_ci[j] = (kk < p.Ktot && r < 7) ? 0 : -1;
      x_dh[j] = r - 3;
      x_dw[j] = (kk & 31) >> 2;
    } else if (kk < p.Ktot) {
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
        // stem7: IW is the padded pitch and iw a padded column (always inside the row)
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

// ------------------------------------------------------------------------------------------------------------------
// Ring-pipelined form of the same kernel (the default): the pixel reduction is walked in 32-row stages through a ring of
// NSLOT LDS slots that LDS-DMA fills NSLOT-1 stages ahead.  A wave waits only for ITS OWN loads of the stage it is about
// to read (counted s_waitcnt vmcnt: the younger stages stay in flight across the barrier), one raw s_barrier per stage
// both publishes that stage and frees the slot read one stage earlier, which is refilled at once.  All LDS lives in one
// array and the loop contains no register-destination global load, so the compiler never drains the DMA queue
// (cdna_hip_programming.md, "Pipelining across barriers").  The single-stage kernel above exposed the full L2/HBM latency
// of every 64-row step behind a block-wide barrier (19.8 % MFMA-busy on MI355X, profiles/r01_pmc_mfma_busy.json).
// ------------------------------------------------------------------------------------------------------------------
constexpr int RKR = 32;   // reduction rows (pixels) per ring stage

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most `younger` whole stages (LPW LDS-DMA instructions each) of this wave are still in flight
template <int LPW> __device__ __forceinline__ void ring_wait(int younger) {
  if (younger >= 3) wait_vmcnt<3 * LPW>();
  else if (younger == 2) wait_vmcnt<2 * LPW>();
  else if (younger == 1) wait_vmcnt<LPW>();
  else wait_vmcnt<0>();
}

// ds_read_b64_tr_b16 through inline asm: hipcc puts `s_waitcnt vmcnt(0)` in front of the BUILTIN form of this read
// whenever an LDS-DMA is in flight (it cannot tell the slots apart), which would drain the ring every stage.  The asm
// form is invisible to that pass; its completion is waited for by hand (lgkmcnt(0) + sched_barrier, cdna_hip_programming.md
// 5.7 / rule 18).  EXEC is all ones everywhere these are issued (no divergent control flow in the main loop).
__device__ __forceinline__ bf16x8 tr_read_pair_asm(unsigned a0, unsigned a1) {
  bf16x4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// Tile shapes.  Measured on MI355X (profiles/README.md, round 2): every deep layer of ResNet-50 ran the 128x128-tile
// kernel at the SAME rate of LDS-DMA traffic, 8.8-8.9 TB/s (L2 -> LDS; each operand row is re-read once per tile of the
// other operand), whatever the pipelining -- i.e. the kernel's roof is that path, and its arithmetic intensity against it
// is 2*BMK*BNC*32 flop per (BMK+BNC)*64 staged bytes = 64 flop/B for 128x128 (-> ~570 TFLOP/s), 85 for 256x128, 128 for
// 256x256.  The ring kernel is therefore built for larger output tiles: WK x WC waves, each owning a
// (BMK/WK) x (BNC/WC) sub-tile of 128x64 or 64x64 accumulators.
template <int BMK, int BNC, int WK, int WC, int NSLOT, int WPS>
__global__ __launch_bounds__(64 * WK * WC, WPS) void conv_wgrad_ring_kernel(const WgradParams p) {
  constexpr int NW = WK * WC;                           // waves per workgroup
  constexpr int X_BYTES = RKR * BMK * 2;
  constexpr int Y_BYTES = RKR * BNC * 2;
  constexpr int STAGE_BYTES = X_BYTES + Y_BYTES;
  constexpr int XROWB = BMK * 2, YROWB = BNC * 2;
  constexpr int XCPR = BMK / 8, YCPR = BNC / 8;        // 16 B chunks per row
  constexpr int XRPI = (64 / XCPR) > 0 ? (64 / XCPR) : 1, YRPI = (64 / YCPR) > 0 ? (64 / YCPR) : 1;   // rows per wave-instruction
  static_assert(XCPR <= 64 && YCPR <= 64, "a row must not exceed one wave-instruction (1 KiB)");
  constexpr int XJ = BMK / 16 / NW, YJ = BNC / 16 / NW;   // LDS-DMA instructions per wave and stage
  static_assert(XJ >= 1 && YJ >= 1 && XJ * NW * 16 == BMK && YJ * NW * 16 == BNC, "staging split");
  constexpr int LPW = XJ + YJ;
  constexpr int KR = BMK / WK / 16, CR = BNC / WC / 16;   // 16-wide fragments per wave (kk, co)
  constexpr int KH2 = KR / 2;                              // the MFMA block is issued in two halves (see below)
  static_assert(NSLOT >= 2 && NSLOT <= 5 && KR % 2 == 0, "ring depth / fragment split");
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSLOT * STAGE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: LDS-DMA destinations stay scalar
  const int wk = wave % WK, wc = wave / WK;

  // block order: see conv_wgrad_kernel (all tiles of one pixel split on one XCD)
  unsigned int tile;
  int split;
  wgrad_block_order(p, tile, split);
  const int tile_c = (int)(tile % (unsigned)p.ntiles_c);
  const int tile_k = (int)(tile / (unsigned)p.ntiles_c);
  const int k0 = tile_k * BMK, c0 = tile_c * BNC;
  const int m_begin = split * p.rows_per_split;
  const int m_end = min(p.M, m_begin + p.rows_per_split);

  const bf16_t* __restrict__ x = p.x;
  const bf16_t* __restrict__ dy = p.dy;
  const bf16_t* zero = (const bf16_t*)icamd_zero_page;

  int x_row[XJ], x_dh[XJ], x_dw[XJ], x_ci[XJ];
#pragma unroll
  for (int j = 0; j < XJ; ++j) {
    const int row = (wave * XJ + j) * XRPI + lane / XCPR;
    const int pc = lane % XCPR;
    const int lc = pc ^ (tr_swz<BMK>(row) << 1);
    const int kk = k0 + lc * 8;
    x_row[j] = row;
    if (p.stem7) {
      // k = row*32 + pixel*4 + channel over the [N][H][W+8][4] image: a 16 B chunk is two pixels of one kernel row; row 7
      // does not exist (zero page), pixel 7's gradient is cleared by the slab reducer
      const int r = kk >> 5;
      x_ci[j] = (kk < p.Ktot && r < 7) ? 0 : -1;
      x_dh[j] = r - 3;
      x_dw[j] = (kk & 31) >> 2;
    } else if (kk < p.Ktot) {
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

  auto stage = [&](int mbase, int slot) {
    unsigned char* sX = smem + slot * STAGE_BYTES;
    unsigned char* sY = sX + X_BYTES;
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      const int m = mbase + x_row[j];
      const bf16_t* src = zero;
      if (p.pointwise) {
        if (m < m_end && x_ci[j] >= 0) src = x + ((long long)m * p.Cin + x_ci[j]);
      } else if (m < m_end && x_ci[j] >= 0) {
        const unsigned int n = fdiv((unsigned)m, p.divHW);
        const unsigned int rm = m - n * (p.OH * p.OW);
        const unsigned int oh = fdiv(rm, p.divW);
        const unsigned int ow = rm - oh * p.OW;
        const int ih = (int)oh * p.stride + x_dh[j], iw = (int)ow * p.stride + x_dw[j];
        // stem7: IW is the padded pitch and iw a padded column (always inside the row)
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
  const bool do_bias = p.bias_slab != nullptr && tile_k == 0 && wk == 0;   // wave-uniform
  f32x4 bacc[CR];
#pragma unroll
  for (int j = 0; j < CR; ++j) bacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16x8 ones = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};

  const int g = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
  const int r0 = 8 * g + q, r1 = r0 + 4;
  // per-lane read offsets inside a slot (the stage is exactly one 32-deep MFMA step)
  unsigned xoff0[KR], xoff1[KR], yoff0[CR], yoff1[CR];
#pragma unroll
  for (int i = 0; i < KR; ++i) {
    const int lb = wk * KR + i;
    xoff0[i] = r0 * XROWB + ((lb ^ tr_swz<BMK>(r0)) << 5) + 8 * pq;
    xoff1[i] = r1 * XROWB + ((lb ^ tr_swz<BMK>(r1)) << 5) + 8 * pq;
  }
#pragma unroll
  for (int j = 0; j < CR; ++j) {
    const int lb = wc * CR + j;
    yoff0[j] = X_BYTES + r0 * YROWB + ((lb ^ tr_swz<BNC>(r0)) << 5) + 8 * pq;
    yoff1[j] = X_BYTES + r1 * YROWB + ((lb ^ tr_swz<BNC>(r1)) << 5) + 8 * pq;
  }

  const unsigned lds_base = (unsigned)(uintptr_t)LPTR(smem);
  const int nsteps = (m_end - m_begin + RKR - 1) / RKR;
  int fill_slot = 0, fill_m = m_begin, issued = 0;
  auto issue = [&]() {
    stage(fill_m, fill_slot);
    fill_m += RKR;
    fill_slot = fill_slot == NSLOT - 1 ? 0 : fill_slot + 1;
    ++issued;
  };
#pragma unroll
  for (int k = 0; k < NSLOT - 1; ++k)
    if (k < nsteps) issue();
  int read_slot = 0;
  for (int t = 0; t < nsteps; ++t) {
    ring_wait<LPW>(issued - 1 - t);              // this wave's part of stage t has landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                // stage t complete for every wave; the slot of stage t-1 is free
    if (issued < nsteps) issue();                // refill it with stage t + NSLOT - 1
    const unsigned sl = lds_base + (unsigned)(read_slot * STAGE_BYTES);
    read_slot = read_slot == NSLOT - 1 ? 0 : read_slot + 1;
    // fragments of the first half, MFMAs of the first half while the second half's fragments arrive
    bf16x8 xf[KR], yf[CR];
#pragma unroll
    for (int j = 0; j < CR; ++j) yf[j] = tr_read_pair_asm(sl + yoff0[j], sl + yoff1[j]);
#pragma unroll
    for (int i = 0; i < KH2; ++i) xf[i] = tr_read_pair_asm(sl + xoff0[i], sl + xoff1[i]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = KH2; i < KR; ++i) xf[i] = tr_read_pair_asm(sl + xoff0[i], sl + xoff1[i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < KH2; ++i)
#pragma unroll
      for (int j = 0; j < CR; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], yf[j], acc[i][j], 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = KH2; i < KR; ++i)
#pragma unroll
      for (int j = 0; j < CR; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], yf[j], acc[i][j], 0, 0, 0);
    if (do_bias) {
#pragma unroll
      for (int j = 0; j < CR; ++j) bacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, yf[j], bacc[j], 0, 0, 0);
    }
  }

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

// ------------------------------------------------------------------------------------------------------------------------
// 256 x 256 tile, eight phases per pair of 64-row reduction tiles: gemm_nt.hip's 8-phase schedule (cdna_hip_programming.md "The
// 256^2 8-phase template") carried to the weight gradient of pointwise layers (round 5; ViT-B/16's Linear layers: M = 50 432 rows,
// 768 x {768, 2304, 3072} filters -- the ring kernel above ran them at 700-765 TFLOP/s where the 8-phase GEMM does 850-920 on the
// same shapes).  D[kk][co] = sum_m X[m][kk] dY[m][co]:
//   * 8 waves, wr = wave >> 2 owns filter columns kk wr*128 .. +128, wc = wave & 3 output channels co wc*64 .. +64: 128 x 64 fp32
//     accumulators per wave = four QUADRANTS of 64 kk x 32 co, one per phase: 16 MFMAs (4 kk fragments x 2 co fragments x 2 steps of
//     32 rows) between two raw s_barriers;
//   * a reduction tile = 64 rows of both operands, staged as four PIECES of 16 KB cut along what a phase starts to need:
//       X_h0 = columns {wr*128 + [0, 64)}   (phase 1)      dY_n0 = channels {wc*64 + [0, 32)}  (phase 1, kept for phase 4)
//       dY_n1 = channels {wc*64 + [32, 64)} (phase 2)      X_h1 = columns {wr*128 + [64, 128)} (phase 3)
//     a piece = [64 rows][256 B] -- whole 128 B (X) / 64 B (dY) segments of an operand row per 8 / 4 lanes, 32 B blocks XOR-swizzled
//     by key(row) = row[1:0] | row[3] << 2 on the source side -- read with ds_read_b64_tr_b16 (the reduction index is the slow
//     memory dimension of both operands: no transposed copy is ever made).  A lane's two reads of a fragment are 4 rows = 1 KiB apart
//     with the SAME key, fragments differ in the block bits only: one address register per fragment and buffer, everything else is
//     the instruction's immediate offset;
//   * LDS holds two reduction tiles (128 KB); every phase issues ONE piece, six pieces ahead, with a scalar base (the tile's first
//     row) and a 32-bit per-lane offset; counted s_waitcnt vmcnt(6); the two wave groups run one barrier apart (gemm_nt.hip).
// Requires whole tiles: Ktot, Cout multiples of 256, rows per split multiples of 64 (the launcher falls back to the ring kernel).
// ------------------------------------------------------------------------------------------------------------------------
constexpr int W8_PIECE = 16384, W8_KTILE = 65536;

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void w8_glds16_sbase(const void* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(voff), "s"(sbase) : "memory", "m0");
}
#pragma clang diagnostic pop

__device__ __forceinline__ void w8_wait_pieces_younger(int n) {   // leave the n youngest pieces (2 LDS-DMA each) in flight
  if (n >= 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (n == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (n == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// both transposed reads of one fragment: rows r and r + 4 (1 KiB further) of a 256 B-row piece
template <int OFF>
__device__ __forceinline__ bf16x8 w8_tr_frag(unsigned a) {
  static_assert(OFF >= 0 && OFF + 1024 < 65536, "ds_read offset is a 16-bit field");
  bf16x4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "n"(OFF + 1024));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__global__ __launch_bounds__(512, 2) void conv_wgrad_8phase_kernel(const WgradParams p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * W8_KTILE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  unsigned int tile;
  int split;
  wgrad_block_order(p, tile, split);
  const int tile_c = (int)(tile % (unsigned)p.ntiles_c), tile_k = (int)(tile / (unsigned)p.ntiles_c);
  const int k0 = tile_k * 256, c0 = tile_c * 256;
  const int m_begin = split * p.rows_per_split;
  const int m_end = min(p.M, m_begin + p.rows_per_split);
  const int nk = (m_end - m_begin) >> 6;            // whole 64-row reduction tiles (launcher)
  const int last_piece = 4 * nk - 1;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LPTR(smem));
  const unsigned xpitch = (unsigned)p.Cin * 2u, ypitch = (unsigned)p.Cout * 2u;
  const unsigned char* const Xb = (const unsigned char*)p.x + (size_t)m_begin * xpitch;
  const unsigned char* const Yb = (const unsigned char*)p.dy + (size_t)m_begin * ypitch;

  // ---- transposed-read roles: 16-lane group g4 reads rows 8 g4 + q (and + 4), 8 B at 8 pq inside a 32 B block; key(row) is a lane
  // constant (rows advance by 32 per step and by 4 between the two reads: bits 0, 1 and 3 never change)
  const int g4 = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
  const int r0 = 8 * g4 + q;
  const int key = (r0 & 3) | (((r0 >> 3) & 1) << 2);
  unsigned aX[2][4], aY[2][2];     // [reduction-tile buffer][fragment]
#pragma unroll
  for (int b = 0; b < 2; ++b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) aX[b][i] = lds_base + (unsigned)(b * W8_KTILE + r0 * 256 + (((wr * 4 + i) ^ key) << 5) + 8 * pq);
#pragma unroll
    for (int j = 0; j < 2; ++j) aY[b][j] = lds_base + (unsigned)(b * W8_KTILE + r0 * 256 + (((wc * 2 + j) ^ key) << 5) + 8 * pq);
  }

  // ---- staging roles: instruction j of this wave is instruction qi = wave * 2 + j of a piece: piece rows qi * 4 .. + 4, sixteen
  // lanes per row; the lane's 16 B chunk pc holds the operand's chunk pc ^ (key(row) << 1) of the piece row
  unsigned off[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = (wave * 2 + j) * 4 + (lane >> 4);
      const int lc = (lane & 15) ^ ((((r & 3) | (((r >> 3) & 1) << 2))) << 1);
      const bool isX = (t == 0 || t == 3);
      if (isX) off[t][j] = (unsigned)r * xpitch + (unsigned)(k0 + (lc >> 3) * 128 + (t == 3 ? 64 : 0) + (lc & 7) * 8) * 2u;
      else off[t][j] = (unsigned)r * ypitch + (unsigned)(c0 + (lc >> 2) * 64 + (t == 2 ? 32 : 0) + (lc & 3) * 8) * 2u;
    }
  auto stage1 = [&](auto tc, auto bc, auto jc, int ktile) {   // instruction J of piece type T of reduction tile `ktile` into buffer B
    constexpr int T = decltype(tc)::value, B = decltype(bc)::value, J = decltype(jc)::value;
    const unsigned char* kb = (T == 0 || T == 3) ? Xb + (size_t)ktile * 64 * xpitch : Yb + (size_t)ktile * 64 * ypitch;
    w8_glds16_sbase(kb, off[T][J], lds_base + (unsigned)(B * W8_KTILE + T * W8_PIECE + (wave * 2 + J) * 1024));
  };
  // prologue: pieces 0..5 (the whole first reduction tile and the first half of the second)
  static_for<0, 6>([&](auto sc) {
    constexpr int S = decltype(sc)::value;
    if (S < 4 || nk > 1) {
      stage1(std::integral_constant<int, (S & 3)>{}, std::integral_constant<int, (S >> 2)>{}, std::integral_constant<int, 0>{}, S >> 2);
      stage1(std::integral_constant<int, (S & 3)>{}, std::integral_constant<int, (S >> 2)>{}, std::integral_constant<int, 1>{}, S >> 2);
    }
  });
  w8_wait_pieces_younger((nk > 1 ? 5 : 3) - 1);      // pieces 0 and 1 have landed (this wave's part)

  f32x4 acc[4][8];     // [co fragment][kk fragment]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // bias gradient (column sums of dY): the workgroups of the first filter-column tile multiply the dY fragments of ONE channel half
  // per wave (wr = 0: half 0, wr = 1: half 1 -- 4 more MFMAs per reduction tile in either wave group) by an all-ones operand
  const bool do_bias = p.bias_slab != nullptr && tile_k == 0;
  f32x4 bacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const bf16x8 ones = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();      // the stagger: group 1 runs one barrier behind group 0
  bf16x8 af[8], b0[4], b1[4];   // X fragments [step*4 + i] of the current column half; dY fragments [step*2 + j] of both channel halves

  auto phase = [&](auto phc, auto steadyc, int g) {
    constexpr int PH = decltype(phc)::value;       // 0..7: reduction-tile parity PH >> 2, phase PH & 3
    constexpr bool STEADY = decltype(steadyc)::value;
    constexpr int BUF = PH >> 2, P = PH & 3;
    // ---- load section: the fragments this phase starts to need, then the wait that retires what the NEXT phase reads
    if constexpr (P == 0) {
      static_for<0, 8>([&](auto c) { constexpr int x = decltype(c)::value; af[x] = w8_tr_frag<0 * W8_PIECE + (x >> 2) * 8192>(aX[BUF][x & 3]); });
      static_for<0, 4>([&](auto c) { constexpr int x = decltype(c)::value; b0[x] = w8_tr_frag<1 * W8_PIECE + (x >> 1) * 8192>(aY[BUF][x & 1]); });
    } else if constexpr (P == 1) {
      static_for<0, 4>([&](auto c) { constexpr int x = decltype(c)::value; b1[x] = w8_tr_frag<2 * W8_PIECE + (x >> 1) * 8192>(aY[BUF][x & 1]); });
    } else if constexpr (P == 2) {
      static_for<0, 8>([&](auto c) { constexpr int x = decltype(c)::value; af[x] = w8_tr_frag<3 * W8_PIECE + (x >> 2) * 8192>(aX[BUF][x & 3]); });
    }
    // pieces <= g + 2 (what phase g + 1 starts to read) have landed; issued so far: pieces <= g + 5
    if constexpr (STEADY) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else w8_wait_pieces_younger((g + 5 < last_piece ? g + 5 : last_piece) - (g + 2));
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4]), "+v"(af[5]), "+v"(af[6]), "+v"(af[7]),
                   "+v"(b0[0]), "+v"(b0[1]), "+v"(b0[2]), "+v"(b0[3]), "+v"(b1[0]), "+v"(b1[1]), "+v"(b1[2]), "+v"(b1[3])
                 :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // ---- compute section: one quadrant, 16 MFMAs; among them the phase's piece (g + 6: type (PH + 2) & 3, reduction tile (g + 6) >> 2)
    constexpr int MH = (P >= 2) ? 1 : 0, NH = (P == 1 || P == 2) ? 1 : 0;
    const bool do_stage = STEADY || g + 6 <= last_piece;
    const int ktile = (g + 6) >> 2;
    __builtin_amdgcn_s_setprio(1);
    static_for<0, 16>([&](auto mc) {
      constexpr int x = decltype(mc)::value;
      constexpr int ks = x >> 3, i = (x >> 1) & 3, j = x & 1;
      if constexpr (x == 3 || x == 10) {
        __builtin_amdgcn_sched_barrier(0);
        if (do_stage) stage1(std::integral_constant<int, ((PH + 2) & 3)>{}, std::integral_constant<int, (((PH + 6) >> 2) & 1)>{},
                             std::integral_constant<int, (x == 3 ? 0 : 1)>{}, ktile);
        __builtin_amdgcn_sched_barrier(0);
      }
      acc[NH * 2 + j][MH * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks * 4 + i], NH ? b1[ks * 2 + j] : b0[ks * 2 + j],
                                                                           acc[NH * 2 + j][MH * 4 + i], 0, 0, 0);
    });
    if constexpr (P == 0 || P == 1) {
      if (do_bias && wr == P) {
#pragma unroll
        for (int x = 0; x < 4; ++x)
          bacc[x & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, P ? b1[x] : b0[x], bacc[x & 1], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };

  int g = 0;
  int t = 0;
  for (; t + 2 <= nk && g + 13 <= last_piece; t += 2) {   // both reduction tiles' phases stage pieces that exist
    static_for<0, 8>([&](auto phc) { phase(phc, std::true_type{}, g + decltype(phc)::value); });
    g += 8;
  }
  for (; t + 2 <= nk; t += 2) {
    static_for<0, 8>([&](auto phc) { phase(phc, std::false_type{}, g + decltype(phc)::value); });
    g += 8;
  }
  if (nk & 1) static_for<0, 4>([&](auto phc) { phase(phc, std::false_type{}, g + decltype(phc)::value); });
  if (wr == 0) __builtin_amdgcn_s_barrier();      // re-align the two groups

  // D[kk][co]: lane holds co = lane & 15, kk = 4 * (lane >> 4) + reg of a 16 x 16 tile -> one 16 B fp32 store per fragment
  float* slab = p.slab + (long long)split * p.Cout * p.Ktot;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj)
#pragma unroll
    for (int ii = 0; ii < 8; ++ii) {
      const int co = c0 + wc * 64 + jj * 16 + (lane & 15);
      const int kk = k0 + wr * 128 + ii * 16 + 4 * (lane >> 4);
      *(f32x4*)(slab + (long long)co * p.Ktot + kk) = acc[jj][ii];
    }
  if (do_bias && (lane >> 4) == 0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) p.bias_slab[(long long)split * p.Cout + c0 + wc * 64 + wr * 32 + j * 16 + (lane & 15)] = bacc[j][0];
  }
}

// out[i] = (accumulate ? out[i] : 0) + sum_s slab[s][i]; 16 B per lane; fixed summation order.
// A block owns OUTS consecutive float4 outputs and splits the S slabs over 256/OUTS slab lanes (4 loads in
// flight per thread), then folds the lanes through LDS in lane order.
#ifndef ICAMD_SLAB_NT
#define ICAMD_SLAB_NT 1   // round 5: the slab fold reads every slab once (ResNet-50 18.02-18.09 -> 17.93-17.94 ms, two A/B pairs)
#endif
__device__ __forceinline__ f32x4 slab_ld(const f32x4* p) {
  if constexpr (ICAMD_SLAB_NT != 0) return __builtin_nontemporal_load(p);
  else return *p;
}
template <int OUTS>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                                          long long n4, int S, int accumulate, int stem7_mask) {
  constexpr int LANES = 256 / OUTS;
  __shared__ f32x4 red[256];
  const int o = threadIdx.x % OUTS, l = threadIdx.x / OUTS;
  const long long i = (long long)blockIdx.x * OUTS + o;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  if (i < n4) {
    const f32x4* p = (const f32x4*)slab + i;
    int k = l;
    for (; k + 3 * LANES < S; k += 4 * LANES) {
      a0 += slab_ld(p + (long long)k * n4);
      a1 += slab_ld(p + (long long)(k + LANES) * n4);
      a2 += slab_ld(p + (long long)(k + 2 * LANES) * n4);
      a3 += slab_ld(p + (long long)(k + 3 * LANES) * n4);
    }
    for (; k < S; k += LANES) a0 += slab_ld(p + (long long)k * n4);
  }
  red[threadIdx.x] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (l == 0 && i < n4) {
    f32x4 s = accumulate ? ((const f32x4*)out)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < LANES; ++j) s += red[j * OUTS + o];
    // stem filters [Cout][8 rows][8 pixels][4]: pixel 7 is padding of the layout (its "gradient" is the product with the
    // next window's first pixel): the filter entry must stay zero, so its gradient is zero
    if (stem7_mask && (i & 7) == 7) s = f32x4{0.f, 0.f, 0.f, 0.f};
    ((f32x4*)out)[i] = s;
  }
}

// Tile choice.  ICAMD_WGRAD_RING=0 selects the single-stage 128x128 kernel of round 1 (kept for A/B runs).
int ring_mode() {
  static const int d = [] { const char* e = getenv("ICAMD_WGRAD_RING"); return e ? atoi(e) : 1; }();
  return d;
}
// widest tile side whose last tile is not mostly padding: 256 when the extent is a multiple of 256 or large, else 128 / 64
int pick_side(int n) {
  if (n <= 64) return 64;
  if (n <= 128) return 128;
  if (n % 256 == 0 || n >= 1024) return 256;
  return ((n + 255) / 256 * 256 - n) * 4 <= n ? 256 : 128;     // at most 25 % padding for the 256 side
}

// ------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 weight gradient with the input staged ONCE per pixel chunk ("halo" form of conv3x3_halo.hip):
//   dw[co][tap][ci] = sum_px dy[px][co] * x[px + (dh-1)*W + (dw-1)][ci] * [tap inside the image]
// The implicit-GEMM kernels above re-gather x once per tap (9x through L2; PMC: TCC busy 74-91 %).  Here a workgroup owns
// one 64 (co) x 64 (ci) block of the filter for ALL NINE taps -- 9 x 64 x 64 fp32 accumulators = 72 VGPRs in each of its
// eight waves (wave = 32-channel half of co x 16-channel quarter of ci) -- and a range of pixels, walked in chunks of 128:
//   * per chunk, LDS-DMA stages the raster range [chunk - W - 1, chunk + 128 + W + 1) of x (64 channels, 128 B rows) and
//     the chunk's 128 rows of dy, double-buffered; both operands are read with ds_read_b64_tr_b16 (the reduction index,
//     the pixel, is the slow memory dimension of both), 32 B blocks XOR-swizzled by a key of (row mod 16) so the reads
//     are conflict-free at any tap offset -- and constant over all steps, since a step advances 32 rows;
//   * a tap is an offset dh*W + dw on the row of the x read: 18 per-lane byte offsets computed once per KERNEL; pixels
//     whose tap leaves the image read a row of zeros instead (one v_cndmask per read);
//   * per 32-pixel step and wave: 22 transposed reads, 18 MFMAs (v_mfma_f32_16x16x32_bf16).
// The ResNet-50 3x3 layers at batch 256 all become 256 workgroups x 3136 pixels (1 / 4 / 16 / 64 filter blocks x
// 256 / 64 / 16 / 4 pixel splits), 37.7 MB of fp32 slabs, reduced in fixed order by slab_reduce_kernel.
constexpr int HW_CH = 128;                 // pixels per chunk
constexpr int HW_XS = 256;                 // staged x rows per buffer: HW_CH + 2W + 2 <= 256
constexpr int HW_XB = HW_XS * 128, HW_YB = HW_CH * 128;

// LDS map: [dy buffer 0 | dy buffer 1 | x buffer 0 | x buffer 1 | 128 B of zeros].  A read's buffer and step are an IMMEDIATE
// offset (buffer * size + step * 32 rows; x: <= 45056, dy: <= 28672, inside the 16-bit field), so the 18 + 4 per-lane read
// addresses are constants of the whole kernel and cost no VALU work inside the loop.
constexpr int HW_XBASE = 2 * HW_YB;
constexpr int HW_ZERO = 2 * HW_YB + 2 * HW_XB;

struct HaloLane {
  unsigned xa[9][2];   // x read addresses (tap, row select), buffer 0, step 0
  unsigned ya[2][2];   // dy read addresses (channel fragment, row select), buffer 0, step 0
  unsigned zaddr;      // the zero row
};

template <int N>
__device__ __forceinline__ void halo_wait_tap(bf16x8& xf, bf16x8& y0, bf16x8& y1) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(xf), "+v"(y0), "+v"(y1) : "n"(N) : "memory");
}

template <int BUF, int ST>
__device__ __forceinline__ void halo_wgrad_step(const WgradParams& p, const HaloLane& h, int px0, f32x4 (&acc)[9][2]) {
  // tap validity of this lane's two pixel rows (px0 + 32*ST and + 4)
  bool hv0[2], hv2[2], wv0[2], wv2[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    unsigned int m = (unsigned)(px0 + ST * 32 + s * 4);
    if (m >= (unsigned)p.M) m = 0;           // rows past M carry dy = 0; any in-range pixel will do
    const unsigned int n = fdiv(m, p.divHW);
    const unsigned int rem = m - n * (p.OH * p.OW);
    const int hh = (int)fdiv(rem, p.divW);
    const int ww = (int)(rem - (unsigned)hh * p.OW);
    hv0[s] = hh > 0; hv2[s] = hh + 1 < p.OH; wv0[s] = ww > 0; wv2[s] = ww + 1 < p.OW;
  }
  constexpr int XOFF = BUF * HW_XB + ST * 32 * 128;
  constexpr int YOFF = BUF * HW_YB + ST * 32 * 128;
  const unsigned zsel = h.zaddr - (unsigned)XOFF;
  bf16x8 yf[2], xf[9];
#pragma unroll
  for (int j = 0; j < 2; ++j) yf[j] = tr_read_pair_off<YOFF>(h.ya[j][0], h.ya[j][1]);
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int dh = t / 3, dw = t % 3;
    unsigned a[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bool okh = dh == 0 ? hv0[s] : (dh == 2 ? hv2[s] : true);
      const bool okw = dw == 0 ? wv0[s] : (dw == 2 ? wv2[s] : true);
      a[s] = (okh && okw) ? h.xa[t][s] : zsel;
    }
    xf[t] = tr_read_pair_off<XOFF>(a[0], a[1]);
  }
  // Progressive waits: the reads return in issue order (4 for dy, then 2 per tap), so tap t may start once at most
  // 2 * (8 - t) reads are outstanding -- the MFMAs of the first taps run under the LDS latency of the last ones.  Each wait
  // names the fragment it releases as an in/out operand, which pins that tap's MFMAs below it.
  __builtin_amdgcn_sched_barrier(0);
  halo_wait_tap<15>(xf[0], yf[0], yf[1]);   // lgkmcnt is a 4-bit field: 15 (not 16) outstanding is the weakest first wait
#pragma unroll
  for (int j = 0; j < 2; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[0], yf[j], acc[0][j], 0, 0, 0);
#define ICAMD_HALO_TAP(T)                                                                                        \
  halo_wait_tap<2 * (8 - T)>(xf[T], yf[0], yf[1]);                                                               \
  acc[T][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[T], yf[0], acc[T][0], 0, 0, 0);                         \
  acc[T][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[T], yf[1], acc[T][1], 0, 0, 0);
  ICAMD_HALO_TAP(1) ICAMD_HALO_TAP(2) ICAMD_HALO_TAP(3) ICAMD_HALO_TAP(4)
  ICAMD_HALO_TAP(5) ICAMD_HALO_TAP(6) ICAMD_HALO_TAP(7) ICAMD_HALO_TAP(8)
#undef ICAMD_HALO_TAP
}

__global__ __launch_bounds__(512, 2) void conv3x3_wgrad_halo_kernel(const WgradParams p, const int nblk_ci, const int nblk) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[HW_ZERO + 128];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int coh = wave & 1, cif = wave >> 1;
  const int split = blockIdx.x / nblk, blk = blockIdx.x - split * nblk;
  const int cob = blk / nblk_ci, cib = blk - cob * nblk_ci;
  const int W = p.OW;
  const int p_begin = split * p.rows_per_split;
  const int p_end = (p.M < p_begin + p.rows_per_split) ? p.M : p_begin + p.rows_per_split;
  const bf16_t* zero = (const bf16_t*)icamd_zero_page;
  const unsigned lds_base = (unsigned)(uintptr_t)LPTR(smem);
  if (tid < 8) *(u32x4*)(smem + HW_ZERO + tid * 16) = u32x4{0u, 0u, 0u, 0u};

  // ---- staging roles.  An LDS-DMA instruction writes 8 rows of 128 B; instruction j of wave w covers rows (j*8 + w)*8..+7.
  // Row r keeps its 32 B blocks at block ^ key(r), key(r) = ((r>>1)&1) | (((r>>3)&1)<<1): r mod 16 = (w&1)*8 + lane>>3.
  const int st_row = wave * 8 + (lane >> 3);                    // + j*64
  const int st_key = (((lane >> 3) >> 1) & 1) | ((wave & 1) << 1);
  const int st_chunk = ((((lane & 7) >> 1) ^ st_key) << 1) | (lane & 1);   // 16 B source chunk of this lane's LDS position
  auto stage = [&](int cpx, int buf) {
    unsigned char* xb = smem + HW_XBASE + buf * HW_XB;
    unsigned char* yb = smem + buf * HW_YB;
#pragma unroll
    for (int j = 0; j < 4; ++j) {        // x: 256 rows = pixels cpx - W - 1 ...
      const int pix = cpx - (W + 1) + j * 64 + st_row;
      const bf16_t* src = (pix >= 0 && pix < p.M) ? p.x + ((long long)pix * p.Cin + cib * 64 + st_chunk * 8) : zero;
      __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(xb + (j * 8 + wave) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {        // dy: the chunk's 128 rows; rows past the split's end are zeros
      const int pix = cpx + j * 64 + st_row;
      const bf16_t* src = pix < p_end ? p.dy + ((long long)pix * p.Cout + cob * 64 + st_chunk * 8) : zero;
      __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(yb + (j * 8 + wave) * 1024), 16, 0, 0);
    }
  };

  // ---- read roles (ds_read_b64_tr_b16: lane -> row 8g + q (+4), 8 B at 8*pq inside a 32 B channel block)
  const int g = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
  const int r0 = 8 * g + q;
  HaloLane h;
  h.zaddr = lds_base + (unsigned)HW_ZERO;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int row = r0 + 4 * s + (t / 3) * W + (t % 3);       // staged row of pixel (chunk + r) + (dh-1)*W + (dw-1)
      const int key = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
      h.xa[t][s] = lds_base + (unsigned)(HW_XBASE + row * 128 + ((cif ^ key) << 5) + 8 * pq);
    }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int row = r0 + 4 * s;
      const int key = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
      h.ya[j][s] = lds_base + (unsigned)(row * 128 + (((coh * 2 + j) ^ key) << 5) + 8 * pq);
    }

  f32x4 acc[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (p_begin < p_end) stage(p_begin, 0);
  for (int cpx = p_begin; cpx < p_end; cpx += 2 * HW_CH) {
    // ---- even chunk: buffers 0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // this chunk landed for every wave; the other buffers are free again
    if (cpx + HW_CH < p_end) stage(cpx + HW_CH, 1);
    halo_wgrad_step<0, 0>(p, h, cpx + r0, acc);
    halo_wgrad_step<0, 1>(p, h, cpx + r0, acc);
    halo_wgrad_step<0, 2>(p, h, cpx + r0, acc);
    halo_wgrad_step<0, 3>(p, h, cpx + r0, acc);
    if (cpx + HW_CH >= p_end) break;
    // ---- odd chunk: buffers 1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (cpx + 2 * HW_CH < p_end) stage(cpx + 2 * HW_CH, 0);
    halo_wgrad_step<1, 0>(p, h, cpx + HW_CH + r0, acc);
    halo_wgrad_step<1, 1>(p, h, cpx + HW_CH + r0, acc);
    halo_wgrad_step<1, 2>(p, h, cpx + HW_CH + r0, acc);
    halo_wgrad_step<1, 3>(p, h, cpx + HW_CH + r0, acc);
  }

  // acc[t][j]: rows = input channels 4*(lane>>4) .. +3 of this wave's 16, column = output channel lane & 15
  float* slab = p.slab + (long long)split * p.Cout * p.Ktot;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int co = cob * 64 + coh * 32 + j * 16 + (lane & 15);
      const int kk = t * p.Cin + cib * 64 + cif * 16 + 4 * (lane >> 4);
      *(f32x4*)(slab + (long long)co * p.Ktot + kk) = acc[t][j];
    }
}

template <int BMK, int BNC, int WK, int WC, int NSLOT, int WPS>
int launch_ring(const WgradParams& p, hipStream_t stream) {
  dim3 grid((unsigned)(p.ntiles_k * p.ntiles_c * p.S));
  hipLaunchKernelGGL((conv_wgrad_ring_kernel<BMK, BNC, WK, WC, NSLOT, WPS>), grid, dim3(64 * WK * WC), 0, stream, p);
  return icamd_launch_status();
}

template <int BMK, int BNC>
int launch(const WgradParams& p, hipStream_t stream) {
  dim3 grid((unsigned)(p.ntiles_k * p.ntiles_c * p.S));
  hipLaunchKernelGGL((conv_wgrad_kernel<BMK, BNC>), grid, dim3(256), 0, stream, p);
  return icamd_launch_status();
}

}  // namespace

void icamd_wgrad_tile(long long M, int Ktot, int Cout, int* bmk, int* bnc) {
  // 256x256 ring kernel where it wins (MI355X, round-2 per-layer table in profiles/README.md): both sides multiples of 256
  // and at least ~50 GFLOP of work (ResNet's deep 3x3 and strided 1x1 layers: 100-106 us -> 78-84 us).  Smaller problems
  // are dominated by the fp32 slab traffic of the pixel split, which grows with the tile area, and by the 32-row stage's
  // barrier rate: they stay on the single-stage 128 / 64 tiles.
  // (round 3) also the >= 50 GFLOP pointwise layers whose sides are multiples of 64 but not of 256 -- ConvNeXt-T's 192 <-> 768
  // and 384 <-> 1536 Linear layers -- on the 128 x 256 / 256 x 128 ring shapes: 130-140 -> 112-124 us and 100-104 -> 87-91 us.
  static const double work_floor = [] { const char* e = getenv("ICAMD_WGRAD_WORK"); return e ? atof(e) : 2.5e10; }();   // (A/B runs)
  const bool work = (double)M * Ktot * Cout >= work_floor;
  const bool both256 = Ktot % 256 == 0 && Cout % 256 == 0;
  const bool wide = Ktot % 64 == 0 && Cout % 64 == 0 && (Ktot < Cout ? Ktot : Cout) >= 192 &&
                    (pick_side(Ktot) == 256 || pick_side(Cout) == 256);
  const bool big = ring_mode() != 0 && work && (both256 || wide);
  if (big || ring_mode() == 2) { *bmk = pick_side(Ktot); *bnc = pick_side(Cout); return; }   // mode 2: every ring shape (tests)
  *bmk = Ktot <= 64 ? 64 : 128;
  *bnc = Cout <= 64 ? 64 : 128;
}

// mode 1: the ring kernel serves exactly the tiles with a 256 side (icamd_wgrad_tile hands those out for big problems only)
static bool use_ring(int bmk, int bnc) { return ring_mode() == 2 || (ring_mode() == 1 && (bmk == 256 || bnc == 256)); }

// workgroups of this tile shape that fit one CU (LDS- or register-limited; must match the launch table below)
static int wgs_per_cu(int bmk, int bnc) {
  if (!use_ring(bmk, bnc)) return 3;
  if (bmk == 256 && bnc == 256) return 1;
  if (bmk + bnc >= 320) return 2;
  return 3;
}

void icamd_wgrad_plan(int M, int Cout, int Ktot, int* S, int* rows_per_split) {
  // Split of the pixel reduction over S workgroups per output tile.  Cost model (MI355X, round-2 measurements): the grid
  // runs in ceil(tiles*S / resident workgroups) rounds, a round lasts as long as one workgroup: its pixel rows plus a
  // fixed prologue + slab-store overhead worth ~256 rows; S is chosen to minimise rounds x that, within [1, 8 rounds] and
  // at least 16 stages per workgroup where M allows.
  int bmk, bnc;
  icamd_wgrad_tile(M, Ktot, Cout, &bmk, &bnc);
  const int tiles = ((Ktot + bmk - 1) / bmk) * ((Cout + bnc - 1) / bnc);
  if (!use_ring(bmk, bnc)) {
    // single-stage kernel (round 1): ~64 stages (4096 pixels) per workgroup, grid within [512, 2048] workgroups
    int s = (M + 4095) / 4096;
    static const int minwg = [] { const char* e = getenv("ICAMD_WGRAD_MINWG"); return e ? atoi(e) : 512; }();
    const int smin = (minwg + tiles - 1) / tiles, smax = (2048 + tiles - 1) / tiles;
    if (s < smin) s = smin;
    if (s > smax) s = smax;
    const int cap = (M + BKR - 1) / BKR;
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    int r = (M + s - 1) / s;
    r = (r + BKR - 1) / BKR * BKR;
    *rows_per_split = r;
    *S = (M + r - 1) / r;
    return;
  }
  const int resident = 256 * wgs_per_cu(bmk, bnc);
  static const int overhead_rows = []() { const char* e = getenv("ICAMD_WGRAD_OVERHEAD_ROWS"); return e ? atoi(e) : 256; }();
  // 256 x 256 tiles: whole 64-row reduction tiles per split, what the 8-phase kernel walks (the ring kernel takes any multiple of 32)
  const int gran = (bmk == 256 && bnc == 256) ? 64 : RKR;
  int scap = (M + gran - 1) / gran;
  const int smax = (8 * resident + tiles - 1) / tiles;
  if (scap > smax) scap = smax;
  if (scap < 1) scap = 1;
  long long best_cost = -1;
  int best_s = 1;
  for (int s = 1; s <= scap; ++s) {
    int rows = (M + s - 1) / s;
    rows = (rows + gran - 1) / gran * gran;
    const int s_eff = (M + rows - 1) / rows;
    const long long rounds = ((long long)tiles * s_eff + resident - 1) / resident;
    const long long cost = rounds * (rows + overhead_rows);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_s = s; }
  }
  int rows = (M + best_s - 1) / best_s;
  rows = (rows + gran - 1) / gran * gran;
  *rows_per_split = rows;
  *S = (M + rows - 1) / rows;
}

static int halo_wgrad_mode() {
  static const int m = [] { const char* e = getenv("ICAMD_WGRAD_HALO"); return e ? atoi(e) : 1; }();
  return m;
}

// 3x3 / stride 1 / pad 1 with both channel counts multiples of 64 and a staged row range within 256 rows
bool icamd_wgrad_halo_wanted(int KH, int KW, int stride, int pad, int H, int W, int Cin, int Cout, long long M) {
  if (halo_wgrad_mode() == 0) return false;
  return KH == 3 && KW == 3 && stride == 1 && pad == 1 && Cin % 64 == 0 && Cout % 64 == 0 && H >= 2 && W >= 2 &&
         HW_CH + 2 * W + 2 <= HW_XS && M < (1ll << 30);
}

// pixel split of the halo kernel: (filter blocks) x S workgroups ~ one per CU, whole chunks per split
void icamd_wgrad_halo_plan(int M, int Cin, int Cout, int* S, int* rows_per_split) {
  const int nblk = (Cin / 64) * (Cout / 64);
  int s = (256 + nblk - 1) / nblk;
  const int cap = (M + HW_CH - 1) / HW_CH;
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  int rows = (M + s - 1) / s;
  rows = (rows + HW_CH - 1) / HW_CH * HW_CH;
  *rows_per_split = rows;
  *S = (M + rows - 1) / rows;
}

int icamd_wgrad_halo_launch(WgradParams& p, hipStream_t stream) {
  if (!icamd_wgrad_halo_wanted(p.KH, p.KW, p.stride, p.pad, p.OH, p.OW, p.Cin, p.Cout, p.M) || p.bias_slab != nullptr)
    return ICAMD_ERR_UNSUPPORTED;
  p.divHW = make_fastdiv((unsigned)(p.OH * p.OW));
  p.divW = make_fastdiv((unsigned)p.OW);
  const int nci = p.Cin / 64, nblk = nci * (p.Cout / 64);
  hipLaunchKernelGGL(conv3x3_wgrad_halo_kernel, dim3((unsigned)(nblk * p.S)), dim3(512), 0, stream, p, nci, nblk);
  return icamd_launch_status();
}

int icamd_wgrad_launch(WgradParams& p, hipStream_t stream) {
  if ((!p.stem7 && p.Cin % 8 != 0) || p.Cout % 8 != 0) return ICAMD_ERR_UNSUPPORTED;
  if ((long long)p.N * p.IH * p.IW * p.Cin >= (1ll << 31)) return ICAMD_ERR_UNSUPPORTED;
  if (p.M <= 0 || p.M >= (1 << 30)) return ICAMD_ERR_BAD_ARG;
  int bmk, bnc;
  icamd_wgrad_tile(p.M, p.Ktot, p.Cout, &bmk, &bnc);
  p.ntiles_k = (p.Ktot + bmk - 1) / bmk;
  p.ntiles_c = (p.Cout + bnc - 1) / bnc;
  p.divHW = make_fastdiv((unsigned)(p.OH * p.OW));
  p.divW = make_fastdiv((unsigned)p.OW);
  p.divCin = make_fastdiv((unsigned)p.Cin);
  p.divKW = make_fastdiv((unsigned)p.KW);
  p.pointwise = (!p.stem7 && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0) ? 1 : 0;
  // ICAMD_WGRAD_XCD=0: the order of rounds 1-4 (A/B); both orders are bijections on any device
  static const int xcd_chunk = [] { const char* e = getenv("ICAMD_WGRAD_XCD"); return (e ? atoi(e) : 1) && icamd_num_xccs() == 8; }();
  p.xcd_chunk = xcd_chunk;
  if (!use_ring(bmk, bnc)) {
    if (bmk == 64) return bnc == 64 ? launch<64, 64>(p, stream) : launch<64, 128>(p, stream);
    return bnc == 64 ? launch<128, 64>(p, stream) : launch<128, 128>(p, stream);
  }
  // 256 x 256 tiles of pointwise layers with whole tiles everywhere: the 8-phase kernel (ICAMD_WGRAD_8PHASE=0: the ring kernel, A/B)
  static const int eight = [] { const char* e = getenv("ICAMD_WGRAD_8PHASE"); return e ? atoi(e) : 1; }();
  if (eight && bmk == 256 && bnc == 256 && p.pointwise && p.Ktot % 256 == 0 && p.Cout % 256 == 0 && p.M % 64 == 0 &&
      p.rows_per_split % 64 == 0 && (long long)p.M * p.Cin < (1ll << 30) && (long long)p.M * p.Cout < (1ll << 30)) {
    hipLaunchKernelGGL(conv_wgrad_8phase_kernel, dim3((unsigned)(p.ntiles_k * p.ntiles_c * p.S)), dim3(512), 0, stream, p);
    return icamd_launch_status();
  }
  //                                         BMK  BNC  WK WC NSLOT WPS   per wave   LDS/workgroup  workgroups/CU
  if (bmk == 256 && bnc == 256) return launch_ring<256, 256, 2, 4, 4, 2>(p, stream);   // 128x64     128 KB         1
  if (bmk == 256 && bnc == 128) return launch_ring<256, 128, 2, 2, 3, 2>(p, stream);   // 128x64      72 KB         2
  if (bmk == 128 && bnc == 256) return launch_ring<128, 256, 1, 4, 3, 2>(p, stream);   // 128x64      72 KB         2
  if (bmk == 256 && bnc == 64) return launch_ring<256, 64, 4, 1, 3, 2>(p, stream);     //  64x64      60 KB         2
  if (bmk == 64 && bnc == 256) return launch_ring<64, 256, 1, 4, 3, 2>(p, stream);     //  64x64      60 KB         2
  if (bmk == 128 && bnc == 128) return launch_ring<128, 128, 2, 2, 3, 3>(p, stream);   //  64x64      48 KB         3
  if (bmk == 128 && bnc == 64) return launch_ring<128, 64, 2, 2, 3, 3>(p, stream);     //  64x32      36 KB         3
  if (bmk == 64 && bnc == 128) return launch_ring<64, 128, 2, 2, 3, 3>(p, stream);     //  32x64      36 KB         3
  return launch_ring<64, 64, 2, 2, 3, 3>(p, stream);                                    //  32x32      24 KB         3
}

int icamd_slab_reduce_launch(const float* slab, float* out, long long n, int S, int accumulate, hipStream_t stream,
                             int stem7_mask) {
  if (n % 4 != 0) return ICAMD_ERR_BAD_ARG;
  const long long n4 = n / 4;
  if (n4 >= 64 * 1024) {   // large filters: 64 outputs x 4 slab lanes per block
    hipLaunchKernelGGL(slab_reduce_kernel<64>, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, stream, slab, out, n4, S,
                       accumulate, stem7_mask);
  } else {                 // small filters, many slabs: 16 outputs x 16 slab lanes per block
    hipLaunchKernelGGL(slab_reduce_kernel<16>, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, stream, slab, out, n4, S,
                       accumulate, stem7_mask);
  }
  return icamd_launch_status();
}
