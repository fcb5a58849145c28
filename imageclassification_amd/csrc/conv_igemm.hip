// Implicit-GEMM convolution for gfx950: NHWC bf16 activations, [Cout][taps][Cin] bf16 filters,
// fp32 MFMA accumulation (v_mfma_f32_16x16x32_bf16), one kernel for forward and data-gradient.
//
// Replaces what ATen dispatches for `model(samples)` / `loss.backward()` in the reference's step
// (/root/reference/engine.py:48,51,64,72 through timm's conv layers, train.py:194).
//
// GEMM view: out[m][co] = sum_k A[m][k] * W[co][k], m = (n,p,q) output pixel, k = (tap, ci).
//   * A rows are gathered straight from the NHWC tensor into LDS by LDS-DMA (global_load_lds, 16 B per
//     lane, per-lane source address = the gather); out-of-image taps read a 256 B zero page.
//   * LDS tiles are [row][64 k] bf16 (128 B rows); the 16 B chunk index is XOR-swizzled with (row>>1)&7 on
//     the SOURCE address (LDS destination stays lane-linear) and on the ds_read_b128 fragment reads.
//   * MFMA operands are swapped (A-operand = filter rows, B-operand = pixel rows) so each lane ends with 4
//     consecutive output channels of one pixel; the tile is transposed through LDS in fp32 and leaves as
//     whole 16 B / 256 B-coalesced bf16 rows, rounded exactly once after bias / addend are added in fp32.
//   * Optional epilogue: per-channel sum and sum-of-squares of the ROUNDED outputs (BatchNorm batch
//     statistics), written as one deterministic partial row per m-tile (no float atomics).
// The same kernel runs the data gradient: "input" = dY, filters = the [Cin][taps][Cout] transposed copy,
// taps negated; stride-2 gradients run as 4 parity classes (ostr=2, only the taps that hit each class).
#include "common.h"
#include "icamd_internal.h"
#include <cstdlib>

namespace {

constexpr int BM = 128;
constexpr int BK = 64;

#ifndef ICAMD_IGEMM_STAGES
#define ICAMD_IGEMM_STAGES 1   // 1: single LDS stage, overlap comes from 4 workgroups per CU; 2: double buffer, 2 per CU
#endif

// EPI 0: out = [relu](acc (+bias)(+addend)); optional statistics of the rounded outputs (BatchNorm forward).
// EPI 2: EPI 0 with every optional operand compiled OUT (no bias, addend, ReLU, GELU; statistics still optional): the plain
//        training forward and the plain data gradient, i.e. most launches of a ResNet step, run ~1/10 of the epilogue code.
// EPI 3: EPI 2 plus a full-size addend (optionally gated by 1-bit ReLU masks) of a stride-1 problem, read COALESCED through
//        wave-private LDS patches: the residual data gradients of the bottleneck blocks (16 launches of a ResNet-50 step).
// EPI 1: data-gradient with the next BatchNorm-backward fused in: g = (acc + addend) * [ReLU mask], out = g, and the
//        partial rows hold sum(g) and sum(g * xhat), xhat from the BN input `bnb_y` (BatchNorm backward, pass 1).
// KMODE 3: the ResNet stem (7x7 stride 2 pad 3 on the [N][H][W+8][4] layout of icamd_pack_input_rgb4): the reduction is
//          8 kernel rows x (8 pixels x 4 channels) = 256 with zero filter entries for row 7, pixel 7 and channel 3; a 64-wide
//          k-step is two kernel rows, each one contiguous 64 B piece of the padded image -- no per-chunk tap arithmetic, no
//          border case along W (1.74x the algorithmic MFMA work instead of 2.67x for the 8-channel general path, half the
//          input bytes, and the gather costs two adds per load).
template <int BN, int KMODE, int EPI>   // KMODE 0: uniform taps; 1: per-chunk taps; 2: single tap with a partial last k-step
__global__ __launch_bounds__(256, (ICAMD_IGEMM_STAGES == 1 ? (EPI == 1 ? 3 : 4) : 2)) void conv_igemm_kernel(const IgemmParams p) {
  constexpr int NSTAGE = ICAMD_IGEMM_STAGES;
  constexpr int A_BYTES = BM * BK * 2;
  constexpr int B_BYTES = BN * BK * 2;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int EPI_BYTES = BM * BN * 2;   // the rounded bf16 output tile goes through LDS once
  constexpr int SMEM_BYTES = (NSTAGE * STAGE_BYTES > EPI_BYTES) ? NSTAGE * STAGE_BYTES : EPI_BYTES;
  constexpr int NJ = BN / 32;   // filter-row fragments per wave (wave covers BN/2 channels)
  constexpr int BROWS = BN / 32;  // B staging instructions per wave
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;

  // XCD-aware tile order: blocks b and b+8 share an XCD (L2); give each XCD a contiguous run of tiles,
  // n-tiles of one m-tile adjacent, so the gathered A rows are fetched from HBM once per XCD.
  const unsigned int nblk = gridDim.x;
  unsigned int L;
  {
    const unsigned int xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
    const unsigned int q = nblk >> 3, r = nblk & 7u;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int ntn = p.ntiles_n;
  const int tile_m = L / ntn, tile_n = L - tile_m * ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const bf16_t* __restrict__ in = p.in;
  const bf16_t* __restrict__ wt = p.wt;
  const bf16_t* zero = (const bf16_t*)icamd_zero_page;

  // ---- per-lane gather state for the 4 A rows this lane stages ----
  int a_base[4], a_ih0[4], a_iw0[4], a_lc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = wave * 32 + j * 8 + (lane >> 3);
    const int m = m0 + row;
    a_lc[j] = ((lane & 7) ^ ((row >> 1) & 7)) * 8;  // logical k offset (elements) of this lane's chunk
    if (m < p.M) {
      const unsigned int n = fdiv((unsigned)m, p.divPQ);
      const unsigned int rem = m - n * (p.P * p.Q);
      const unsigned int pp = fdiv(rem, p.divQ);
      const unsigned int qq = rem - pp * p.Q;
      a_ih0[j] = pp * p.istr;
      a_iw0[j] = qq * p.istr;
      a_base[j] = ((n * p.IH + a_ih0[j]) * p.IW + a_iw0[j]) * p.Cin;
    } else {
      a_ih0[j] = -(1 << 20);
      a_iw0[j] = -(1 << 20);
      a_base[j] = 0;
    }
  }
  int b_off[BROWS];  // element offset of this lane's filter row (+ its swizzled chunk), or -1
#pragma unroll
  for (int j = 0; j < BROWS; ++j) {
    const int row = wave * (BN / 4) + j * 8 + (lane >> 3);
    const int co = n0 + row;
    const int lc = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
    b_off[j] = (co < p.Cout) ? co * p.Ktot + lc : -1;
  }

  auto stage = [&](int ks, int buf) {
    unsigned char* sA = smem + buf * STAGE_BYTES;
    unsigned char* sB = sA + A_BYTES;
    if constexpr (KMODE == 3) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 2 * ks + (a_lc[j] >> 5);                    // kernel row of this lane's chunk
        const int ih = a_ih0[j] - 3 + r;
        const bool ok = r < 7 && (unsigned)ih < (unsigned)p.IH;   // a_ih0 is hugely negative for rows past M
        const bf16_t* src = ok ? in + (a_base[j] + (r - 3) * p.IW * 4 + (a_lc[j] & 31)) : zero;
        __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(sA + (wave * 32 + j * 8) * 128), 16, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < BROWS; ++j) {
        const bf16_t* src = b_off[j] >= 0 ? wt + (b_off[j] + ks * BK) : zero;
        __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(sB + (wave * (BN / 4) + j * 8) * 128), 16, 0, 0);
      }
    } else if constexpr (KMODE != 1) {
      // all 64 k of this step share one tap (Cin % 64 == 0, or a single tap): tap index and channel offset are wave-uniform
      const int kk0 = ks * BK;
      const int t = kk0 / p.Cin;          // uniform
      const int ci0 = kk0 - t * p.Cin;
      const int dh = p.dh[t], dw = p.dw[t];
      const int tapoff = (dh * p.IW + dw) * p.Cin + ci0;
      const int woff = p.wtap[t] * p.Cin + ci0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ih = a_ih0[j] + dh, iw = a_iw0[j] + dw;
        // kk0 + chunk < Cin only matters for single-tap problems whose channel count is not a multiple of 64 (ConvNeXt's
        // 96-channel pointwise layers, one parity class of a 2x2/2 data gradient): the last k-step is then partly past the
        // end of the tap's channels and reads zeros.  (NOT Ktot: a strided data gradient's filter row holds KH*KW taps of
        // which this launch uses one.)
        const bool ok = ((unsigned)ih < (unsigned)p.IH) && ((unsigned)iw < (unsigned)p.IW) &&
                        (KMODE != 2 || kk0 + a_lc[j] < p.Cin);
        const bf16_t* src = ok ? in + (a_base[j] + tapoff + a_lc[j]) : zero;
        __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(sA + (wave * 32 + j * 8) * 128), 16, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < BROWS; ++j) {
        bool bok = b_off[j] >= 0;
        if constexpr (KMODE == 2) {
          const int row = wave * (BN / 4) + j * 8 + (lane >> 3);
          bok = bok && (kk0 + ((lane & 7) ^ ((row >> 1) & 7)) * 8 < p.Cin);
        }
        const bf16_t* src = bok ? wt + (b_off[j] + woff) : zero;
        __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(sB + (wave * (BN / 4) + j * 8) * 128), 16, 0, 0);
      }
    } else {
      // General channel counts (Cin % 8 == 0, e.g. the 8-channel padded RGB stem or ConvNeXt's 96): a 64-wide k-step
      // may straddle taps, so every 16 B chunk derives its own (tap, channel) from k = tap*Cin + ci; taps follow the
      // regular rule dh = sign*(r - pad) (sign = -1 for a stride-1 data gradient)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned int kk = (unsigned)(ks * BK + a_lc[j]);
        const unsigned int t = fdiv(kk, p.divCin);
        const int ci = (int)(kk - t * p.Cin);
        const unsigned int r = fdiv(t, p.divKW);
        const int dh = p.tap_sign * ((int)r - p.pad), dw = p.tap_sign * ((int)(t - r * p.KW) - p.pad);
        const int ih = a_ih0[j] + dh, iw = a_iw0[j] + dw;
        const bool ok = ((int)t < p.ntaps) && ((unsigned)ih < (unsigned)p.IH) && ((unsigned)iw < (unsigned)p.IW);
        const bf16_t* src = ok ? in + (a_base[j] + (dh * p.IW + dw) * p.Cin + ci) : zero;
        __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(sA + (wave * 32 + j * 8) * 128), 16, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < BROWS; ++j) {
        const int row = wave * (BN / 4) + j * 8 + (lane >> 3);
        const int lc = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
        // b_off already holds co*Ktot + lc; the filter row is k-contiguous in the same (tap, ci) order
        const bf16_t* src = (b_off[j] >= 0 && ks * BK + lc < p.Ktot) ? wt + (b_off[j] + ks * BK) : zero;
        __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(sB + (wave * (BN / 4) + j * 8) * 128), 16, 0, 0);
      }
    }
  };

  f32x4 acc[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  const int sw = fr >> 1;  // (row>>1)&7 for every fragment row this lane reads (rows are 16-aligned + fr)

  const int nks = p.ksteps;
  auto compute = [&](int buf) {
    const unsigned char* sA = smem + buf * STAGE_BYTES;
    const unsigned char* sB = sA + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int chunk = ((kk * 4 + fq) ^ sw) * 16;
      bf16x8 xf[4], wf[NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        xf[i] = *(const bf16x8*)(sA + (wm * 64 + i * 16 + fr) * 128 + chunk);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        wf[j] = *(const bf16x8*)(sB + (wn * (BN / 2) + j * 16 + fr) * 128 + chunk);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[j][i], 0, 0, 0);
    }
  };
  if constexpr (NSTAGE == 2) {
    if (nks > 0) stage(0, 0);
    __syncthreads();  // emits vmcnt(0): stage 0 has landed
    for (int ks = 0; ks < nks; ++ks) {
      const int buf = ks & 1;
      if (ks + 1 < nks) stage(ks + 1, buf ^ 1);
      compute(buf);
      __syncthreads();  // next stage landed (vmcnt(0)) and everyone is done reading this buffer
    }
  } else {
    for (int ks = 0; ks < nks; ++ks) {
      stage(ks, 0);
      __syncthreads();  // vmcnt(0) + barrier: the stage has landed for every wave
      compute(0);
      __syncthreads();  // all fragment reads done before the buffer is refilled / reused by the epilogue
    }
  }

  // ---- epilogue ----------------------------------------------------------------------------------------
  // (1) in the MFMA layout (a lane owns 4 consecutive channels of one pixel): add bias / addend in fp32, round
  //     ONCE to bf16, and drop the tile into LDS as [128 m][BN] bf16 (16 B chunks XOR-swizzled by m&15);
  // (2) coalesced pass: each thread takes 8 channels (16 B) of a row from LDS, applies the fused BatchNorm-
  //     backward mask / accumulates the per-channel statistics, and stores 16 B rows (256 B per 16 lanes).
  //     Everything pass (2) reads from global memory (BN input, mask) is prefetched before the barrier.
  // Loads are never placed under a per-lane condition (hipcc would branch around each and wait vmcnt(0) per
  // load): invalid lanes read a clamped, in-bounds address and discard the value.
  constexpr int ROWB = BN * 2;         // bytes per bf16 tile row
  constexpr int CPR = BN / 8;          // 8-channel groups per row
  constexpr int RPP = 256 / CPR;       // rows per pass
  constexpr int NPASS = BM / RPP;
  const int cp = tid % CPR, rg = tid / CPR;
  const int co = n0 + cp * 8;
  const bool co_ok = co < p.Cout;  // Cout % 8 == 0 (host-checked)

  // pass-(2) row offsets (elements; the host checks the output tensor has < 2^31 elements) and prefetches
  int roff[NPASS];
  u32x4 yv[EPI == 1 ? NPASS : 1], mv[EPI == 1 ? NPASS : 1];
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const int m = m0 + ps * RPP + rg;
    roff[ps] = -1;
    if (m < p.M && co_ok) {
      const unsigned int n = fdiv((unsigned)m, p.divPQ);
      const unsigned int rem = m - n * (p.P * p.Q);
      const unsigned int pp = fdiv(rem, p.divQ);
      const unsigned int qq = rem - pp * p.Q;
      const int pix = (n * p.OH + pp * p.ostr + p.ooff_h) * p.OW + qq * p.ostr + p.ooff_w;
      roff[ps] = pix * p.Cout + co;
    }
    if constexpr (EPI == 1) {
      const int ro = roff[ps] >= 0 ? roff[ps] : 0;
      yv[ps] = *(const u32x4*)(p.bnb_y + ro);
      mv[ps] = *(const u32x4*)((p.bnb_mask != nullptr ? p.bnb_mask : p.bnb_y) + ro);
    }
  }

  // (1) MFMA-layout adds + rounding + LDS write
  {
    const bool has_addend = EPI != 2 && p.addend != nullptr;   // wave-uniform
    int moff[4];
    unsigned long long abw[4] = {~0ull, ~0ull, ~0ull, ~0ull};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      moff[i] = 0;   // rows past M read pixel 0 (valid memory); their results are never stored
      const int m = m0 + wm * 64 + i * 16 + fr;
      if constexpr (EPI == 3) {   // stride-1, full-size addend: the output pixel index IS m
        if (m < p.M) moff[i] = m * p.Cout;
      } else if (has_addend && m < p.M) {
        const unsigned int n = fdiv((unsigned)m, p.divPQ);
        const unsigned int rem = m - n * (p.P * p.Q);
        const unsigned int pp = fdiv(rem, p.divQ);
        const unsigned int qq = rem - pp * p.Q;
        const unsigned int oh = pp * p.ostr + p.ooff_h, ow = qq * p.ostr + p.ooff_w;
        if (p.addend_sub2) {   // addend lives on the even pixel grid only (gradient sent back by a 1x1 stride-2 shortcut)
          if ((oh | ow) & 1u) abw[i] = 0ull;
          else moff[i] = ((n * ((p.OH + 1) >> 1) + (oh >> 1)) * ((p.OW + 1) >> 1) + (ow >> 1)) * p.Cout;
        } else {
          moff[i] = ((n * p.OH + oh) * p.OW + ow) * p.Cout;
        }
      }
    }
    // ReLU-mask bits of the addend: the wave's BN/2 channels of one pixel are BN/16 consecutive bytes -> one load per row
    if (has_addend && p.addend_bits != nullptr) {
      const int cw0 = n0 + wn * (BN / 2);
      const int cwc = cw0 < p.Cout ? cw0 : 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned char* bp = p.addend_bits + ((moff[i] + cwc) >> 3);
        if constexpr (BN == 128) abw[i] = *(const unsigned long long*)bp;
        else abw[i] = *(const unsigned int*)bp;
      }
    }
    if constexpr (EPI == 3) {
      // Coalesced addend: in the MFMA layout a lane's 4 channels are 8 B and a wave-load touches 16 rows x 32 B; here the
      // wave reads its 64 rows x 128 B as 16 B per lane (8 rows x 128 B per instruction), parks them in a wave-private
      // 8 KB patch of the (now idle) stage buffer and picks its fragments back in the MFMA layout.
      static_assert(EPI != 3 || BN == 128, "EPI 3 is instantiated for 128-channel tiles");
      unsigned char* patch = smem + wave * 8192;
      const int cw0 = n0 + wn * 64;
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        const int row = h * 8 + (lane >> 3), chunk = lane & 7;
        const int m = m0 + wm * 64 + row;
        const int mc = m < p.M ? m : 0;
        const u32x4 v = *(const u32x4*)(p.addend + (long long)mc * p.Cout + cw0 + chunk * 8);
        *(u32x4*)(patch + row * 128 + ((chunk ^ (row & 7)) << 4)) = v;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = i * 16 + fr;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const u32x2 a = *(const u32x2*)(patch + row * 128 + ((((2 * j + (fq >> 1)) ^ (row & 7)) & 7) << 4) + ((fq & 1) << 3));
          const unsigned int ab = (unsigned int)(abw[i] >> (8 * (j * 2 + (fq >> 1)) + 4 * (fq & 1))) & 0xfu;
          acc[j][i][0] += (ab & 1u) ? bf16_lo(a[0]) : 0.f;
          acc[j][i][1] += (ab & 2u) ? bf16_hi(a[0]) : 0.f;
          acc[j][i][2] += (ab & 4u) ? bf16_lo(a[1]) : 0.f;
          acc[j][i][3] += (ab & 8u) ? bf16_hi(a[1]) : 0.f;
        }
      }
      __syncthreads();   // every wave is done with its patch before the output tile overwrites the buffer
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int cl = wn * (BN / 2) + j * 16 + 4 * fq;      // tile-local channel of this lane's 4 values
      const int cg = n0 + cl;
      const int cgc = cg < p.Cout ? cg : 0;                 // clamped: loads stay in bounds, values unused
      f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
      if (EPI < 2 && p.bias != nullptr) b4 = *(const f32x4*)(p.bias + cgc);
      u32x2 av[4];
      unsigned int ab[4] = {0xffu, 0xffu, 0xffu, 0xffu};
      if (EPI != 3 && has_addend) {
#pragma unroll
        for (int i = 0; i < 4; ++i) av[i] = *(const u32x2*)(p.addend + moff[i] + cgc);
        // addend is a ReLU layer's output gradient: bit = [that output > 0]; this lane's nibble within the row word
#pragma unroll
        for (int i = 0; i < 4; ++i) ab[i] = (unsigned int)(abw[i] >> (8 * (j * 2 + (fq >> 1)) + 4 * (fq & 1))) & 0xfu;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 v = acc[j][i] + b4;
        if (EPI != 3 && has_addend) {
          const u32x2 a = av[i];
          v[0] += (ab[i] & 1u) ? bf16_lo(a[0]) : 0.f;
          v[1] += (ab[i] & 2u) ? bf16_hi(a[0]) : 0.f;
          v[2] += (ab[i] & 4u) ? bf16_lo(a[1]) : 0.f;
          v[3] += (ab[i] & 8u) ? bf16_hi(a[1]) : 0.f;
        }
        if (EPI < 2 && p.relu) {   // inference epilogue (BatchNorm folded into filters + bias): NaN passes through like torch.relu
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];
        }
        const int ml = wm * 64 + i * 16 + fr;
        u32x2 pk;
        pk[0] = pack_bf16x2(v[0], v[1]);
        pk[1] = pack_bf16x2(v[2], v[3]);
        const int slot = cl >> 2;                           // 8 B slot in the row; 16 B chunk = slot >> 1
        *(u32x2*)(smem + ml * ROWB + ((((slot >> 1) ^ (ml & 15)) & (CPR - 1)) << 4) + ((slot & 1) << 3)) = pk;
      }
    }
  }
  __syncthreads();

  // Round 4: the pre-GELU activations z of the "x gelu'(z)" form for ALL passes up front.  Read inside the pass loop (rounds 2-3)
  // every load sat behind the previous pass's store -- the compiler may not hoist a load above a store to memory that could
  // alias -- and each of the 8 passes of a tile paid a full load latency (ConvNeXt-T's fc2 data gradient: 1.55-2x its twin
  // without GELU at every stage).  The accumulators are dead here (the tile is in LDS), so the registers are free.
  u32x4 zv[EPI < 2 ? NPASS : 1];
  if constexpr (EPI < 2) {
    if (p.gelu_z != nullptr) {
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) zv[ps] = *(const u32x4*)(p.gelu_z + (roff[ps] >= 0 ? roff[ps] : 0));
    }
  }
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  float bmu[8], bis[8], bsc[8], bsh[8];
  if constexpr (EPI == 1) {
    const int cc = co_ok ? co : 0;
    const bool from_y = p.bnb_mask == nullptr && p.bnb_relu;   // uniform
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bmu[e] = p.bnb_mean[cc + e];
      bis[e] = p.bnb_invstd[cc + e];
      bsc[e] = from_y ? p.bnb_scale[cc + e] : 0.f;
      bsh[e] = from_y ? p.bnb_shift[cc + e] : 0.f;
    }
  }

  // (2) coalesced pass
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const int ml = ps * RPP + rg;
    u32x4 o = *(const u32x4*)(smem + ml * ROWB + (((cp ^ (ml & 15)) & (CPR - 1)) << 4));
    if (roff[ps] >= 0) {
      if constexpr (EPI == 1) {
        float yy[8], gg[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          yy[2 * e] = bf16_lo(yv[ps][e]); yy[2 * e + 1] = bf16_hi(yv[ps][e]);
          gg[2 * e] = bf16_lo(o[e]); gg[2 * e + 1] = bf16_hi(o[e]);
        }
        if (p.bnb_relu) {
          if (p.bnb_mask != nullptr) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (!(bf16_lo(mv[ps][e]) > 0.f)) { gg[2 * e] = 0.f; o[e] &= 0xffff0000u; }
              if (!(bf16_hi(mv[ps][e]) > 0.f)) { gg[2 * e + 1] = 0.f; o[e] &= 0x0000ffffu; }
            }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              if (!(fmaf(yy[e], bsc[e], bsh[e]) > 0.f)) {
                gg[e] = 0.f;
                o[e >> 1] &= (e & 1) ? 0x0000ffffu : 0xffff0000u;
              }
            }
          }
        }
        *(u32x4*)(p.out + roff[ps]) = o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s1[e] += gg[e];
          s2[e] += gg[e] * ((yy[e] - bmu[e]) * bis[e]);
        }
      } else {
        if constexpr (EPI < 2) {
          if (p.gelu_z != nullptr) o = gelu_bwd8(o, zv[ps]);
          if (p.gelu_inplace) o = gelu8(o);
        }
        *(u32x4*)(p.out + roff[ps]) = o;
        if constexpr (EPI < 2) {
          if (p.gelu_out != nullptr) *(u32x4*)(p.gelu_out + roff[ps]) = gelu8(o);
        }
        if (p.stats != nullptr) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = bf16_lo(o[e]), hi = bf16_hi(o[e]);
            s1[2 * e] += lo; s2[2 * e] += lo * lo;
            s1[2 * e + 1] += hi; s2[2 * e + 1] += hi * hi;
          }
        }
      }
    }
  }

  if (p.stats != nullptr) {
    __syncthreads();  // all tile reads done; reuse LDS for the cross-row-group reduction
    float* red = (float*)smem;  // [RPP][2][BN]
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(rg * 2 + 0) * BN + cp * 8 + e] = s1[e];
      red[(rg * 2 + 1) * BN + cp * 8 + e] = s2[e];
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, c = tid - which * BN;
      float s = 0.f;
#pragma unroll 4
      for (int g = 0; g < RPP; ++g) s += red[(g * 2 + which) * BN + c];
      if (n0 + c < p.Cout) p.stats[((long long)tile_m * 2 + which) * p.Cout + n0 + c] = s;
    }
  }
}

template <int BN, int KMODE, int EPI>   // KMODE 0: uniform taps; 1: per-chunk taps; 2: single tap with a partial last k-step
int launch(const IgemmParams& p, hipStream_t stream) {
  const int ntm = (p.M + BM - 1) / BM;
  dim3 grid((unsigned)(ntm * p.ntiles_n));
  hipLaunchKernelGGL((conv_igemm_kernel<BN, KMODE, EPI>), grid, dim3(256), 0, stream, p);
  return icamd_launch_status();
}

}  // namespace

int icamd_igemm_pick_bn(int Cout) { return Cout <= 64 ? 64 : 128; }

int icamd_igemm_launch(IgemmParams& p, hipStream_t stream) {
  if (p.Cout % 8 != 0) return ICAMD_ERR_UNSUPPORTED;
  // "general" path: per-chunk taps (the 8-channel stem, ConvNeXt's 2x2 downsample at 96 channels, ...).  Single-tap
  // problems stay on the uniform-tap path whatever their channel count: a k-step cannot straddle taps there.
  const bool cin8 = !p.stem7 && (p.Cin % 64 != 0) && p.ntaps != 1;
  if (!p.stem7 && p.Cin % 8 != 0) return ICAMD_ERR_UNSUPPORTED;
  if (cin8 && !p.regular_taps) return ICAMD_ERR_UNSUPPORTED;   // strided data gradients need Cout % 64 == 0
  if (cin8 && p.Ktot != p.ntaps * p.Cin) return ICAMD_ERR_BAD_ARG;
  if ((long long)p.N * p.IH * p.IW * p.Cin >= (1ll << 31)) return ICAMD_ERR_UNSUPPORTED;
  if ((long long)p.Cout * p.Ktot >= (1ll << 31)) return ICAMD_ERR_UNSUPPORTED;
  if ((long long)p.N * p.OH * p.OW * p.Cout >= (1ll << 31)) return ICAMD_ERR_UNSUPPORTED;
  if (p.M <= 0 || p.ntaps < 0 || (!cin8 && p.ntaps > ICAMD_MAX_TAPS)) return ICAMD_ERR_BAD_ARG;
  const int bn = icamd_igemm_pick_bn(p.Cout);
  p.ntiles_n = (p.Cout + bn - 1) / bn;
  p.ksteps = (p.ntaps * p.Cin + BK - 1) / BK;
  p.divPQ = make_fastdiv((unsigned)(p.P * p.Q));
  p.divQ = make_fastdiv((unsigned)p.Q);
  p.divCin = make_fastdiv((unsigned)p.Cin);
  p.divKW = make_fastdiv((unsigned)(p.KW > 0 ? p.KW : 1));
  if (p.tap_sign == 0) p.tap_sign = 1;
  if (p.bnb_y != nullptr) {
    if (cin8 || p.stats == nullptr || p.bnb_mean == nullptr || p.bnb_invstd == nullptr) return ICAMD_ERR_BAD_ARG;
    if (p.bnb_relu && p.bnb_mask == nullptr && (p.bnb_scale == nullptr || p.bnb_shift == nullptr)) return ICAMD_ERR_BAD_ARG;
    if (p.Cin % 64 != 0) return ICAMD_ERR_UNSUPPORTED;
    return bn == 64 ? launch<64, 0, 1>(p, stream) : launch<128, 0, 1>(p, stream);
  }
  if (p.stem7) {   // set up by icamd_stem7x7s2_fwd: Cin = 4, IW = padded row pitch, istr = 2, Ktot = 256, 4 k-steps
    p.ksteps = 4;
    return bn == 64 ? launch<64, 3, 0>(p, stream) : launch<128, 3, 0>(p, stream);
  }
  const bool tail = !cin8 && (p.Cin % 64 != 0);   // single tap, channel count not a multiple of the k-step
  const bool plain = p.bias == nullptr && p.addend == nullptr && !p.relu && p.gelu_out == nullptr && !p.gelu_inplace &&
                     p.gelu_z == nullptr;
  static const bool lean_on = [] { const char* e = getenv("ICAMD_IGEMM_LEAN"); return !(e && atoi(e) == 0); }();
  const bool addend_only = p.bias == nullptr && p.addend != nullptr && !p.addend_sub2 && !p.relu && p.gelu_out == nullptr &&
                           !p.gelu_inplace && p.gelu_z == nullptr && p.ostr == 1 && p.ooff_h == 0 && p.ooff_w == 0 &&
                           p.P == p.OH && p.Q == p.OW && p.stats == nullptr;
  static const bool epi3_on = [] { const char* e = getenv("ICAMD_IGEMM_LEAN"); return !(e && atoi(e) == 2); }();
  if (lean_on && epi3_on && addend_only && bn == 128 && p.Cout % 128 == 0 && !cin8 && !tail) return launch<128, 0, 3>(p, stream);
  if (lean_on && plain && !cin8 && !tail) return bn == 64 ? launch<64, 0, 2>(p, stream) : launch<128, 0, 2>(p, stream);
  if (bn == 64) return cin8 ? launch<64, 1, 0>(p, stream) : (tail ? launch<64, 2, 0>(p, stream) : launch<64, 0, 0>(p, stream));
  return cin8 ? launch<128, 1, 0>(p, stream) : (tail ? launch<128, 2, 0>(p, stream) : launch<128, 0, 0>(p, stream));
}
