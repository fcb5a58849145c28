// 3x3 / stride 1 / padding 1 convolution for gfx950 with the input tile staged ONCE per 64-channel slice ("halo" form):
// forward and data gradient of the ResNet 3x3 layers (reference: timm Bottleneck.conv2 / BasicBlock.conv1,2 under
// model(samples) and loss.backward(), /root/reference/engine.py:48,51,64,72).
//
// Why: rocprofv3 counters on MI355X (profiles/README.md, round 2) show the implicit-GEMM kernel of conv_igemm.hip is bound
// by L2 request bandwidth on these layers (TCC busy 79-87 %, MFMA busy 25-32 %): with a 128-pixel tile it re-gathers every
// input pixel once per tap (9x) and re-reads the whole filter once per tile, ~0.9-1.4 GB of L2 -> LDS traffic for layers
// whose operands are 50-230 MB.  Here
//   * an output tile is BM (256 or 128) CONSECUTIVE pixels in (n, h, w) raster order; all nine taps of those pixels lie
//     in the raster range [m0 - W - 1, m0 + BM + W] of the input, which is staged once per 64-channel slice as
//     [slot][64 ch] rows of 128 B (LDS-DMA; the 16 B chunks of a row are ROTATED by slot & 7, which keeps ds_read_b128
//     conflict-free for a 16-slot fragment at ANY alignment and is additive in the tap offset): 1.06-1.45x instead of 9x;
//   * a tap is then only an offset dh*W + dw on the slot index of the A-fragment reads; taps that leave the image (the
//     raster neighbour is a pixel of another row / image) are zeroed per lane from a 9-bit validity mask;
//   * the filter streams through a ring of NB LDS slots, one (tap, 32-channel half slice) stage of [BN][32] per slot,
//     filled NB-1 stages ahead with counted s_waitcnt vmcnt and one raw s_barrier per stage (the gemm_nt.hip scheme);
//   * four waves as WM (pixels) x WN (channels); a wave owns up to 128 x 64 accumulators, so a stage is up to 32 MFMAs
//     (v_mfma_f32_16x16x32_bf16) per wave for 12 ds_read_b128.
// L2 -> LDS bytes per flop drop 3-4x and the layer becomes MFMA-bound.
// The data gradient of a stride-1 3x3 convolution is the same computation on dY with the transposed filter
// [Cin][3][3][Cout] and mirrored taps (flip = 1).
// Epilogue as conv_igemm.hip: lanes own 4 consecutive channels of a pixel, bias added in fp32, one rounding, bf16 tile
// through LDS, 16 B coalesced row stores, optional ReLU, optional per-channel sum / sum-of-squares of the rounded outputs
// as one partial row per tile (BatchNorm statistics, no float atomics).
#ifndef ICAMD_C64_NT
#define ICAMD_C64_NT 0   // cache policy of the once-read LDS-DMA streams of this unit: 0 default, 2 non-temporal (round 5 A/B)
#endif
#include "common.h"
#include "icamd_internal.h"
#include <cstdlib>

namespace {

template <int N> __device__ __forceinline__ void halo_wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// 16 B chunk swizzle of the filter stage's 64 B rows (gemm_nt.hip): chunk' = chunk ^ swzq((row >> 2) & 3)
__device__ __forceinline__ int swzq(int blk) { return (0x78 >> (2 * blk)) & 3; }

// WN: waves along the channel dimension (1 or 2; 4 / WN along pixels); MFR: 16-pixel fragments per wave along M;
// LA: A-staging LDS-DMA instructions per wave and slice (8 slots each): 4 * LA * 8 > BM + 2 * W + 2; NB: filter ring depth.
// KS: channels per filter stage (64: [BN][64] stages of 128 B rows; 32: [BN][32] stages of 64 B rows, twice as many barriers
// but a ring twice as deep in the same LDS).  Measured on MI355X: KS = 64 is faster for BN = 128, KS = 32 for BN = 64.
template <int BN, int WN, int MFR, int LA, int NB, int KS, int WPS>
__global__ __launch_bounds__(256, WPS) void conv3x3_halo_kernel(const Halo3x3Params p) {
  constexpr int WM = 4 / WN;
  constexpr int BM = WM * MFR * 16;
  constexpr int NJ = BN / WN / 16;             // 16-channel fragments per wave along N
  constexpr int A_SLOTS = 4 * LA * 8;
  constexpr int ZERO_SLOT = A_SLOTS - 1;       // the last slot is staged from the zero page: every tap that leaves the image
  constexpr int A_BYTES = A_SLOTS * 128;       // reads it (the launcher sizes LA so that the tile needs < A_SLOTS slots)
  constexpr int B_STAGE = BN * KS * 2;         // one tap of KS channels: [BN][KS] bf16
  constexpr int LB = BN * KS / 2048;           // filter LDS-DMA instructions per wave and stage (1 KiB each, 4 waves)
  static_assert(KS == 32 || KS == 64, "filter stage width");
  constexpr int RING = NB * B_STAGE;
  constexpr int EPI_BYTES = BM * BN * 2;
  constexpr int LDS_BYTES = (A_BYTES + RING > EPI_BYTES) ? A_BYTES + RING : EPI_BYTES;
  static_assert(NB >= 2 && NB <= 8 && (WN == 1 || WN == 2) && LB >= 1, "configuration");
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
  unsigned char* const sA = smem;
  unsigned char* const sB = smem + A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int fr = lane & 15, fq = lane >> 4;

  // XCD-aware tile order (blocks b and b+8 share an XCD): the channel tiles of one pixel tile stay on one XCD
  const unsigned int nblk = gridDim.x;
  unsigned int L;
  {
    const unsigned int xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
    const unsigned int q = nblk >> 3, r = nblk & 7u;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = L / p.ntiles_n, tile_n = L - tile_m * p.ntiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int W = p.W, C = p.C;
  const bf16_t* __restrict__ in = p.in;
  const bf16_t* __restrict__ wt = p.wt;
  const bf16_t* zero = (const bf16_t*)icamd_zero_page;

  // ---- A staging roles: instruction j of this wave covers slots (wave*LA + j)*8 .. +7; lane -> (slot, swizzled chunk).
  // The source offset is recomputed per slice (twelve multiply-adds once per nine stages) instead of living in registers.
  const int a_slot_lane = lane >> 3;
  // ---- B staging roles: an instruction covers 1 KiB of the stage = 8 rows of 128 B (KS 64) or 16 rows of 64 B (KS 32)
  int b_src[LB];
#pragma unroll
  for (int j = 0; j < LB; ++j) {
    int row, lc;
    if constexpr (KS == 64) {
      row = (wave * LB + j) * 8 + (lane >> 3);
      lc = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
    } else {
      row = (wave * LB + j) * 16 + (lane >> 2);
      lc = ((lane & 3) ^ swzq((row >> 2) & 3)) * 8;
    }
    const int co = n0 + row;
    b_src[j] = (co < p.Cout) ? co * 9 * C + lc : -1;
  }
  auto stage_a = [&](int slice) {
#pragma unroll
    for (int j = 0; j < LA; ++j) {
      const int slot = (wave * LA + j) * 8 + a_slot_lane;
      const int pix = m0 - (W + 1) + slot;
      const int lc = (((lane & 7) - slot) & 7) * 8;      // position pos holds logical chunk (pos - slot) & 7
      const bf16_t* src = (pix >= 0 && pix < p.M && slot != ZERO_SLOT) ? in + ((long long)pix * C + slice * 64 + lc) : zero;
      __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(sA + (wave * LA + j) * 1024), 16, 0, 0);
    }
  };
  auto stage_b = [&](int slice, int tap, int kk, int slot) {
    const int wtap = p.flip ? 8 - tap : tap;
    const int off = wtap * C + slice * 64 + kk * 32;   // kk = 0 for 64-channel stages
#pragma unroll
    for (int j = 0; j < LB; ++j) {
      const bf16_t* src = b_src[j] >= 0 ? wt + (b_src[j] + off) : zero;
      __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(sB + slot * B_STAGE + (wave * LB + j) * 1024), 16, 0, 0);
    }
  };

  // ---- per-lane pixel state of the MFR fragments this lane reads: slot of the CENTRE tap and the 9-bit tap validity
  int slot0[MFR];
  unsigned int vmask[MFR];
#pragma unroll
  for (int i = 0; i < MFR; ++i) {
    const int ml = wm * (BM / WM) + i * 16 + fr;
    const int m = m0 + ml;
    slot0[i] = ml + W + 1;
    unsigned int mk = 0;
    if (m < p.M) {
      const unsigned int n = fdiv((unsigned)m, p.divHW);
      const unsigned int rem = m - n * (p.H * W);
      const int h = (int)fdiv(rem, p.divW);
      const int w = (int)(rem - (unsigned)h * W);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int dh = t / 3 - 1, dw = t % 3 - 1;
        if ((unsigned)(h + dh) < (unsigned)p.H && (unsigned)(w + dw) < (unsigned)W) mk |= 1u << t;
      }
    }
    vmask[i] = mk;
  }
  f32x4 acc[NJ][MFR];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < MFR; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nslices = C >> 6;
  constexpr int SPT = 64 / KS;                 // filter stages per tap
  const int nstages = nslices * 9 * SPT;
  // prologue: slice 0 of the input, then the first NB-1 filter stages
  stage_a(0);
  int issue_q = 0, issue_slice = 0, issue_tap = 0, issue_kk = 0, fill = 0;
  auto issue_b = [&]() {
    stage_b(issue_slice, issue_tap, issue_kk, fill);
    fill = fill == NB - 1 ? 0 : fill + 1;
    ++issue_q;
    if (++issue_kk == SPT) {
      issue_kk = 0;
      if (++issue_tap == 9) { issue_tap = 0; ++issue_slice; }
    }
  };
#pragma unroll
  for (int k = 0; k < NB - 1; ++k)
    if (issue_q < nstages) issue_b();

  // filter fragment: row (channel) wn*(BN/WN) + j*16 + fr, 16 B chunk fq (+4 for the second half of a 64-wide stage)
  const int b_row = wn * (BN / WN) + fr;
  const int b_frag = KS == 64 ? b_row * 128 : b_row * 64 + ((fq ^ swzq((fr >> 2) & 3)) << 4);
  const int bsw = (fr >> 1) & 7;
  int read = 0, q = 0;
  for (int slice = 0; slice < nslices; ++slice) {
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      const int trow = tap >= 6 ? 2 : (tap >= 3 ? 1 : 0);
      const int doff = (trow - 1) * W + (tap - 3 * trow - 1);
      const unsigned int tbit = 1u << tap;
      // A-fragment byte addresses of this tap: invalid lanes read the zero row (one select on the slot instead of four on
      // the data); the second 32-channel half of the slice is the same address with chunk bit 2 flipped
      int aaddr[MFR];
#pragma unroll
      for (int i = 0; i < MFR; ++i) {
        const int sl = (vmask[i] & tbit) ? slot0[i] + doff : ZERO_SLOT;
        aaddr[i] = sl * 128 + (((fq + sl) & 7) << 4);
      }
#pragma unroll
      for (int st = 0; st < SPT; ++st, ++q) {
        // Wait for this wave's part of filter stage q (and, at the first stage of a slice, of the input tile, issued before
        // it).  In flight behind it: the younger filter stages -- the ring keeps NB-1 ahead, fewer at the very end.
        {
          const int younger = issue_q - 1 - q;
          if (younger >= 4) halo_wait_vmcnt<4 * LB>();
          else if (younger == 3) halo_wait_vmcnt<3 * LB>();
          else if (younger == 2) halo_wait_vmcnt<2 * LB>();
          else if (younger == 1) halo_wait_vmcnt<LB>();
          else halo_wait_vmcnt<0>();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // stage q landed for every wave; the ring slot of stage q-1 is free
        if (issue_q < nstages) issue_b();
        const unsigned char* sb = sB + read * B_STAGE + b_frag;
        read = read == NB - 1 ? 0 : read + 1;
#pragma unroll
        for (int kk = (KS == 64 ? 0 : st); kk < (KS == 64 ? 2 : st + 1); ++kk) {
          bf16x8 wf[NJ];
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            if constexpr (KS == 64) wf[j] = *(const bf16x8*)(sb + j * 2048 + (((kk * 4 + fq) ^ bsw) << 4));
            else wf[j] = *(const bf16x8*)(sb + j * 1024);
          }
#pragma unroll
          for (int i = 0; i < MFR; ++i) {
            const bf16x8 xf = *(const bf16x8*)(sA + (kk ? (aaddr[i] ^ 64) : aaddr[i]));
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf, acc[j][i], 0, 0, 0);
          }
        }
      }
    }
    if (slice + 1 < nslices) {
      // every wave is done with this slice's input tile once it has passed the next barrier; the tile is re-staged after a
      // dedicated one (single buffer: the other workgroup on the CU covers the reload latency)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      stage_a(slice + 1);
      // these LA instructions are YOUNGER than the filter stages already in flight, so the counted waits above (which only
      // allow for filter stages) would let the next stage start too early: drain here, the filter ring refills behind it
      halo_wait_vmcnt<0>();
    }
  }
  __syncthreads();   // all fragment reads done: LDS becomes the output tile

  // ---- epilogue: MFMA layout (lane: pixel = fr, 4 consecutive channels) -> bias -> [relu] -> bf16 -> LDS [BM][BN] ----
  constexpr int ROWB = BN * 2, CPR = BN / 8;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int cl = wn * (BN / WN) + j * 16 + 4 * fq;
    const int cg = n0 + cl;
    const int cgc = cg < p.Cout ? cg : 0;
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr) b4 = *(const f32x4*)(p.bias + cgc);
#pragma unroll
    for (int i = 0; i < MFR; ++i) {
      const int ml = wm * (BM / WM) + i * 16 + fr;
      f32x4 v = acc[j][i] + b4;
      if (p.relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];
      }
      u32x2 pk;
      pk[0] = pack_bf16x2(v[0], v[1]);
      pk[1] = pack_bf16x2(v[2], v[3]);
      const int slot = cl >> 2;
      *(u32x2*)(smem + ml * ROWB + ((((slot >> 1) ^ ml) & (CPR - 1)) << 4) + ((slot & 1) << 3)) = pk;
    }
  }
  __syncthreads();
  constexpr int RPP = 256 / CPR;           // rows per pass of the whole workgroup
  constexpr int NPASS = BM / RPP;
  const int cp = tid & (CPR - 1), rg = tid / CPR;
  const int co = n0 + cp * 8;
  const bool co_ok = co < p.Cout;
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const int ml = ps * RPP + rg;
    const int m = m0 + ml;
    const u32x4 o = *(const u32x4*)(smem + ml * ROWB + (((cp ^ ml) & (CPR - 1)) << 4));
    if (m < p.M && co_ok) {
      *(u32x4*)(p.out + (long long)m * p.Cout + co) = o;
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
  if (p.stats != nullptr) {
    // One partial row per TILE (row tile_m).  The consumer sums ceil(M / 128) rows (icamd_conv2d_stats_rows), which is at
    // most twice the tile count for BM >= 128: the rows this grid does not produce are zero-filled by the tile whose index
    // they exceed the tile count by.
    __syncthreads();
    float* red = (float*)smem;             // [RPP][2][BN]
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(rg * 2 + 0) * BN + cp * 8 + e] = s1[e];
      red[(rg * 2 + 1) * BN + cp * 8 + e] = s2[e];
    }
    __syncthreads();
    const int ntm = (p.M + BM - 1) / BM, nrows = (p.M + 127) / 128;
    for (int idx = tid; idx < 2 * BN; idx += 256) {
      const int which = idx / BN, c = idx - which * BN;
      float s = 0.f;
#pragma unroll 4
      for (int g = 0; g < RPP; ++g) s += red[(g * 2 + which) * BN + c];
      if (n0 + c < p.Cout) {
        p.stats[((long long)tile_m * 2 + which) * p.Cout + n0 + c] = s;
        if (ntm + tile_m < nrows) p.stats[((long long)(ntm + tile_m) * 2 + which) * p.Cout + n0 + c] = 0.f;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 64 -> 64 channels (ResNet layer1 conv2; every BasicBlock convolution of ResNet-18/34 at 56 x 56): the whole filter is
// 64 x 576 bf16 = 72 KB.  A wave that owns HALF of the output channels needs 18 k-steps x 2 channel fragments x 4 VGPRs =
// 144 registers for its half, so every wave keeps its filter half in REGISTERS for the life of the kernel and the
// workgroups are PERSISTENT (two per CU, tiles b, b + grid, ...):
//   * nothing but the input halo tile goes through LDS: 128 + 2W + 2 slots of 128 B per 128-pixel tile, double-buffered,
//     the next tile's LDS-DMA in flight under the current tile's MFMAs; the filter operand costs no LDS reads and no L2
//     traffic after the first microseconds (the implicit-GEMM kernel re-gathers the input per tap and re-reads the filter
//     per tile: 7x the L2 -> LDS bytes);
//   * four waves = 2 pixel halves x 2 channel halves, 64 pixels x 32 channels of accumulators per wave; one s_barrier per
//     tile; the epilogue (MFMA layout -> 64 B half rows through a wave-private LDS patch, row stores, BatchNorm partial
//     sums in registers) has no workgroup barrier, and the second workgroup of the CU runs its MFMAs meanwhile;
//   * BatchNorm statistics leave as ONE partial row per workgroup (the rows of the [ceil(M/128)] table that no workgroup
//     owns are zero-filled).
// ACT: bias / ReLU epilogue of the inference form (icamd_conv2d_fwd_act); the training form carries neither.
template <bool ACT>
__global__ __launch_bounds__(256, 2) void conv3x3_c64_resident_kernel(const Halo3x3Params p, const int ntiles) {
  constexpr int BM = 128, LA = 8;
  constexpr int A_BYTES = 32 * LA * 128;       // 256 slots; slot 255 is staged from the zero page
  constexpr int E_BYTES = 4 * 64 * 64;
  constexpr int ZERO_SLOT = 32 * LA - 1;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * A_BYTES + E_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int fr = lane & 15, fq = lane >> 4;
  const int W = p.W, H = p.H;
  const bf16_t* __restrict__ in = p.in;
  const bf16_t* zero = (const bf16_t*)icamd_zero_page;
  unsigned char* const sE = smem + 2 * A_BYTES + wave * 4096;
  const unsigned lds_base = (unsigned)(uintptr_t)LPTR(smem);

  // Staging: instruction j of wave w covers slots (j*4 + w)*8 .. +7 (1 KiB of LDS); slot s holds pixel m0 - (W+1) + s with
  // its 16 B chunks rotated by s & 7 -- and s & 7 is lane >> 3 for every instruction, so a lane's source is one pointer
  // advanced by 32 pixels per instruction.
  const int a_slot = wave * 8 + (lane >> 3);
  const int a_lc = (((lane & 7) - (lane >> 3)) & 7) * 8;
  const bool a_zero_lane = a_slot == 31;       // in the last instruction: slot 255, the zero row
  auto stage_a = [&](int m0, unsigned char* dst) {
    const int pix0 = m0 - (W + 1) + a_slot;
    const bf16_t* src0 = in + ((long long)pix0 * 64 + a_lc);
    if (m0 - (W + 1) >= 0 && m0 - (W + 1) + LA * 32 <= p.M) {   // uniform: every tile but the tensor's first and last ones
#pragma unroll
      for (int j = 0; j < LA; ++j) {
        const bf16_t* src = src0 + j * (32 * 64);
        if (j == LA - 1) src = a_zero_lane ? zero : src;
        __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(dst + (j * 4 + wave) * 1024), 16, 0, ICAMD_C64_NT);
      }
    } else {
#pragma unroll
      for (int j = 0; j < LA; ++j) {
        const int pix = pix0 + j * 32;
        const bf16_t* src = (pix >= 0 && pix < p.M && !(j == LA - 1 && a_zero_lane)) ? src0 + j * (32 * 64) : zero;
        __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(dst + (j * 4 + wave) * 1024), 16, 0, ICAMD_C64_NT);
      }
    }
  };

  int buf = 0;
  int tile = blockIdx.x;
  if (tile < ntiles) stage_a(tile * BM, smem);   // in flight under the filter loads below

  // ---- this wave's filter half, once: fragment (k-step ks = tap*2 + half, channel fragment j): row wn*32 + j*16 + fr
  bf16x8 wf[18][2];
#pragma unroll
  for (int ks = 0; ks < 18; ++ks) {
    const int tap = ks >> 1;
    const int wtap = p.flip ? 8 - tap : tap;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      wf[ks][j] = *(const bf16x8*)(p.wt + ((wn * 32 + j * 16 + fr) * 9 + wtap) * 64 + (ks & 1) * 32 + fq * 8);
  }
  // The filter must have ARRIVED before the tile loop: left to the compiler, its s_waitcnt vmcnt(0) would sit at the first
  // use inside the loop, behind the next tile's LDS-DMA, and drain that every tile.  (The builtin, not inline asm: the
  // compiler's wait-count bookkeeping sees it.)  simm16 = vmcnt 0, expcnt 7, lgkmcnt 15.
  __builtin_amdgcn_s_waitcnt(0x0F70);

  f32x2 s1[4], s2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { s1[e] = f32x2{0.f, 0.f}; s2[e] = f32x2{0.f, 0.f}; }

  // (the s_waitcnt above also covered the first tile's input, which has no stores behind its loads: the counted wait at the
  // top of the loop would let it through)
  for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
    const int m0 = tile * BM;
    // this tile's input has landed for every wave, and every wave is done reading the other buffer (previous tile).
    // (in flight behind the input loads: the previous tile's 4 row stores -- every tile but a workgroup's LAST has all
    // 128 rows, so exactly 4 store instructions per wave follow the loads this wait is for)
    halo_wait_vmcnt<4>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tile + (int)gridDim.x < ntiles) stage_a((tile + gridDim.x) * BM, smem + (buf ^ 1) * A_BYTES);

    // ---- per-lane pixel state of the four fragments: (h, w) of pixel m0 + wm*64 + i*16 + fr; tap validity as lane masks:
    // rows x columns.
    // W % 8 == 0 (launcher), so the chunk rotation (fq + slot) & 7 does not depend on the tap's ROW: three addresses per
    // fragment (columns -1, 0, +1) at tap row -1; the tap row adds dh * W * 128 bytes.
    const unsigned sa = lds_base + (unsigned)(buf * A_BYTES);
    const unsigned zaddr = sa + (unsigned)(ZERO_SLOT * 128);
    // Fragment i sits 16 slots = 2048 B behind fragment 0 with the SAME rotation (16 = 0 mod 8): its reads are fragment 0's
    // address with an immediate offset of i * 2048, so three addresses serve all four fragments; lanes whose tap leaves the
    // image take the zero row's address minus that offset instead.
    unsigned abase[3];
    bool hv0[4], hv2[4], wv0[4], wv2[4];
    {
      const int ml0 = wm * 64 + fr;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const int sl = ml0 + d;                  // slot of (row -1, column d-1): centre slot ml + W + 1, minus W, plus d - 1
        abase[d] = sa + (unsigned)((sl << 7) | (((fq + sl) & 7) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned int m = (unsigned)(m0 + ml0 + i * 16);
        if (m >= (unsigned)p.M) m = 0;           // rows past M are never stored; any in-range pixel will do
        const unsigned int n = fdiv(m, p.divHW);
        const unsigned int rem = m - n * (H * W);
        const int h = (int)fdiv(rem, p.divW);
        const int w = (int)(rem - (unsigned)h * W);
        hv0[i] = h > 0; hv2[i] = h + 1 < H; wv0[i] = w > 0; wv2[i] = w + 1 < W;
      }
    }
    f32x4 acc[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // A-fragment reads as inline asm: hipcc's waitcnt pass would otherwise put s_waitcnt vmcnt(0) in front of every C++ LDS
    // load while the next tile's LDS-DMA is in flight (it cannot tell the two buffers apart).  Completion is waited for by
    // hand; the fragments of k-step ks+1 are requested before the 8 MFMAs of k-step ks.
    bf16x8 xf[2][4];
    unsigned aaddr[4];
    auto tap_addresses = [&](int tap) {
      const int dh = tap / 3, dw = tap % 3;      // 0, 1, 2 = -1, 0, +1
      const unsigned t = abase[dw] + (unsigned)(dh * W * 128);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool okh = dh == 0 ? hv0[i] : (dh == 2 ? hv2[i] : true);
        const bool okw = dw == 0 ? wv0[i] : (dw == 2 ? wv2[i] : true);
        aaddr[i] = (okh && okw) ? t : zaddr - (unsigned)(i * 2048);
      }
    };
#define ICAMD_READ4(dst, X)                                                                      \
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst[0]) : "v"(aaddr[0] ^ (X)));                      \
  asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(dst[1]) : "v"(aaddr[1] ^ (X)));          \
  asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(dst[2]) : "v"(aaddr[2] ^ (X)));          \
  asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(dst[3]) : "v"(aaddr[3] ^ (X)));
    tap_addresses(0);
    ICAMD_READ4(xf[0], 0u)
#pragma unroll
    for (int ks = 0; ks < 18; ++ks) {
      // the wait names the four fragments as in/out operands: their consumers cannot be scheduled above it
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(xf[ks & 1][0]), "+v"(xf[ks & 1][1]), "+v"(xf[ks & 1][2]), "+v"(xf[ks & 1][3])::"memory");
      if (ks + 1 < 18) {
        if (((ks + 1) & 1) == 0) {
          tap_addresses((ks + 1) >> 1);
          ICAMD_READ4(xf[(ks + 1) & 1], 0u)
        } else {
          ICAMD_READ4(xf[(ks + 1) & 1], 64u)     // second 32-channel half: chunk position bit 2 flipped
        }
      }
      __builtin_amdgcn_sched_barrier(0);         // the next reads are in flight BEFORE this k-step's MFMAs
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][j], xf[ks & 1][i], acc[j][i], 0, 0, 0);
    }
#undef ICAMD_READ4

    // ---- epilogue, wave-private: MFMA layout -> [bias, relu] -> bf16 -> this wave's [64 px][32 ch] LDS patch (64 B rows,
    // 16 B chunk ^= (row >> 2) & 3) -> 64 B half-row stores
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
      if constexpr (ACT) {
        if (p.bias != nullptr) b4 = *(const f32x4*)(p.bias + wn * 32 + j * 16 + 4 * fq);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pl = i * 16 + fr;
        f32x4 v = acc[j][i];
        if constexpr (ACT) {
          v += b4;
          if (p.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];
          }
        }
        u32x2 pk;
        pk[0] = pack_bf16x2(v[0], v[1]);
        pk[1] = pack_bf16x2(v[2], v[3]);
        const int slot = j * 4 + fq;              // 8 B slot of the 64 B row; 16 B chunk = slot >> 1
        *(u32x2*)(sE + pl * 64 + ((((slot >> 1) ^ (pl >> 2)) & 3) << 4) + ((slot & 1) << 3)) = pk;
      }
    }
    {
      const int pl0 = lane >> 2;                  // 16 rows per instruction, 4 lanes (chunks) per row
      const int mrow = m0 + wm * 64 + pl0;
      bf16_t* orow = p.out + ((long long)mrow * 64 + wn * 32 + (lane & 3) * 8);
      const bool want_stats = p.stats != nullptr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int pl = r * 16 + pl0;
        const u32x4 o = *(const u32x4*)(sE + pl * 64 + ((((lane & 3) ^ (pl >> 2)) & 3) << 4));
        if (mrow + r * 16 < p.M) {
          *(u32x4*)(orow + r * (16 * 64)) = o;
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
    }
  }

  if (p.stats != nullptr) {
    // one partial row per workgroup; rows no workgroup owns are zero (the consumer sums ceil(M / 128) rows)
    __syncthreads();
    float* red = (float*)smem;               // [32 lane groups: wm*16 + lane>>2][2][64]
    const int g = wm * 16 + (lane >> 2);
    const int c0 = wn * 32 + (lane & 3) * 8;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[(g * 2 + 0) * 64 + c0 + 2 * e] = s1[e][0];
      red[(g * 2 + 0) * 64 + c0 + 2 * e + 1] = s1[e][1];
      red[(g * 2 + 1) * 64 + c0 + 2 * e] = s2[e][0];
      red[(g * 2 + 1) * 64 + c0 + 2 * e + 1] = s2[e][1];
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, c = tid & 63;
      float s = 0.f;
#pragma unroll 8
      for (int k = 0; k < 32; ++k) s += red[(k * 2 + which) * 64 + c];
      p.stats[((long long)blockIdx.x * 2 + which) * 64 + c] = s;
      const int nrows = (p.M + 127) / 128;
      for (int r = blockIdx.x + gridDim.x; r < nrows; r += gridDim.x) p.stats[((long long)r * 2 + which) * 64 + c] = 0.f;
    }
  }
}

int launch_resident_c64(const Halo3x3Params& p, hipStream_t stream) {
  const int ntiles = (p.M + 127) / 128;
  const int slots = 2 * icamd_num_cus();                // two workgroups per CU
  const int grid = ntiles < slots ? ntiles : slots;
  if (p.bias != nullptr || p.relu)
    hipLaunchKernelGGL((conv3x3_c64_resident_kernel<true>), dim3((unsigned)grid), dim3(256), 0, stream, p, ntiles);
  else
    hipLaunchKernelGGL((conv3x3_c64_resident_kernel<false>), dim3((unsigned)grid), dim3(256), 0, stream, p, ntiles);
  return icamd_launch_status();
}

template <int BN, int WN, int MFR, int LA, int NB, int KS, int WPS>
int launch_halo(const Halo3x3Params& p, hipStream_t stream) {
  constexpr int BM = (4 / WN) * MFR * 16;
  const int ntm = (p.M + BM - 1) / BM;
  hipLaunchKernelGGL((conv3x3_halo_kernel<BN, WN, MFR, LA, NB, KS, WPS>), dim3((unsigned)(ntm * p.ntiles_n)), dim3(256), 0,
                     stream, p);
  return icamd_launch_status();
}

int halo_mode() {
  static const int m = [] { const char* e = getenv("ICAMD_CONV3X3_HALO"); return e ? atoi(e) : 1; }();
  return m;
}

}  // namespace

// register-resident filter kernel: exactly 64 -> 64 channels, W a multiple of 8, tile + halo within 255 slots
static bool resident_c64_ok(int W, int C, int Cout) {
  static const int on = [] { const char* e = getenv("ICAMD_CONV3X3_RESIDENT"); return e ? atoi(e) : 1; }();
  return on != 0 && C == 64 && Cout == 64 && W % 8 == 0 && 128 + 2 * W + 2 <= 255;
}

bool icamd_halo3x3_wanted(int N, int H, int W, int C, int Cout) {
  if (halo_mode() == 0) return false;
  // Default: the layers where it is faster than the implicit-GEMM kernel on MI355X at batch 256 (profiles/README.md,
  // round 2): >= 256 channels (256->256 at 14x14: 72 -> 63 us; 512->512 at 7x7: 85 -> 70 us).  At 64 / 128 channels the two
  // tie (the input tile is re-staged per 64-channel slice with nothing to overlap), so those stay on conv_igemm.hip.
  // ICAMD_CONV3X3_HALO=2 / 3 force the 128- / 256-pixel tiles for every eligible layer (tests).
  if (C % 64 != 0 || Cout % 8 != 0 || W < 3 || H < 3) return false;
  if (resident_c64_ok(W, C, Cout)) return true;
  if (halo_mode() == 1 && C < 256) return false;
  if ((long long)N * H * W >= (1ll << 30)) return false;
  if (2 * W + 3 + 128 > 4 * 8 * 8) return false;        // the 128-pixel tile (always a candidate) stages <= 256 slots (LA <= 8)
  return true;
}

int icamd_halo3x3_launch(Halo3x3Params& p, hipStream_t stream) {
  if (!icamd_halo3x3_wanted(p.N, p.H, p.W, p.C, p.Cout)) return ICAMD_ERR_UNSUPPORTED;
  p.M = p.N * p.H * p.W;
  p.divHW = make_fastdiv((unsigned)(p.H * p.W));
  p.divW = make_fastdiv((unsigned)p.W);
  if (resident_c64_ok(p.W, p.C, p.Cout)) return launch_resident_c64(p, stream);
  const int bn = p.Cout <= 64 ? 64 : 128;
  p.ntiles_n = (p.Cout + bn - 1) / bn;
  // Pixel-tile height: 256, 224 or 128, whichever finishes the grid in the fewest (rounds x tile) at two workgroups per CU
  // (e.g. 256 -> 256 at 14 x 14, batch 256: 392 tiles of 256 fill 77 % of one round, 448 tiles of 224 fill 88 % of it).
  int bm = 128;
  if (halo_mode() == 3) bm = 256;
  else if (halo_mode() != 2) {
    long long best = -1;
    for (int cand : {256, 224, 128}) {
      if (cand == 224 && bn != 128) continue;
      if ((cand + 2 * p.W + 3 + 31) / 32 > (cand == 128 ? 8 : 12)) continue;   // tile + halo must fit the staged slots
      const long long tiles = (long long)((p.M + cand - 1) / cand) * p.ntiles_n;
      const long long cost = ((tiles + 511) / 512) * cand + (cand == 128 ? 16 : 0);   // small tiles re-read the filter more
      if (best < 0 || cost < best) { best = cost; bm = cand; }
    }
  }
  const bool big = bm != 128;
  const int slots = bm + 2 * p.W + 2 + 1;                 // + the zero row
  const int la = (slots + 31) / 32;                       // instructions per wave (8 slots each, 4 waves)
  if (la > 12) return ICAMD_ERR_UNSUPPORTED;
  // LDS per workgroup = 4 KB * LA (input tile) + filter ring <= 80 KB: two workgroups per CU
  if (bn == 128) {   // 2 x 2 waves, 128 x 64 (or 64 x 64) accumulators per wave; 16 KB filter stages of 64 channels
    if (bm == 224) {
      if (la <= 8) return launch_halo<128, 2, 7, 8, 3, 64, 2>(p, stream);      // W <= 14: 32 + 48 = 80 KB, ring of three
      if (la <= 10) return launch_halo<128, 2, 7, 10, 2, 64, 2>(p, stream);
      return launch_halo<128, 2, 7, 12, 2, 64, 2>(p, stream);
    }
    if (big) {
      if (la <= 9) return launch_halo<128, 2, 8, 9, 2, 64, 2>(p, stream);      // 36 + 32 = 68 KB
      if (la <= 10) return launch_halo<128, 2, 8, 10, 2, 64, 2>(p, stream);    // 40 + 32 = 72 KB
      return launch_halo<128, 2, 8, 12, 2, 64, 2>(p, stream);                  // 48 + 32 = 80 KB
    }
    if (la <= 5) return launch_halo<128, 2, 4, 5, 3, 64, 2>(p, stream);        // 20 + 48 = 68 KB
    if (la <= 6) return launch_halo<128, 2, 4, 6, 3, 64, 2>(p, stream);
    return launch_halo<128, 2, 4, 8, 3, 64, 2>(p, stream);                     // 32 + 48 = 80 KB
  }
  // Cout <= 64: 4 x 1 waves, every wave reads all four filter fragments (64 x 64 accumulators per wave); 4 KB stages of 32
  if (big) {
    if (la <= 9) return launch_halo<64, 1, 4, 9, 6, 32, 2>(p, stream);         // 36 + 24 = 60 KB
    if (la <= 10) return launch_halo<64, 1, 4, 10, 6, 32, 2>(p, stream);
    return launch_halo<64, 1, 4, 12, 6, 32, 2>(p, stream);                     // 48 + 24 = 72 KB
  }
  if (la <= 5) return launch_halo<64, 1, 2, 5, 6, 32, 3>(p, stream);
  if (la <= 6) return launch_halo<64, 1, 2, 6, 6, 32, 3>(p, stream);
  return launch_halo<64, 1, 2, 8, 6, 32, 2>(p, stream);   // 32 + 24 = 56 KB: two workgroups per CU
}
