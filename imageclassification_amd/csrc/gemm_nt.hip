// Large-tile dense GEMM for gfx950 used by the pointwise (1x1, stride 1) convolution forward / data-gradient when the
// problem is big enough to be MFMA-bound (ViT's Linear layers, ResNet's deep 1x1 layers):
//   out[m][n] = [relu](sum_k A[m][k] * B[n][k]  (+ bias[n]) (+ addend[m][n])),  A = activations [M][K], B = filters [N][K], bf16.
//
// 256 x TN output tile per workgroup (TN = 128: 4 wavefronts, two workgroups per CU; TN = 256: 8 wavefronts, one per
// CU); every wave owns 128 x 64 = 32 accumulator tiles of v_mfma_f32_16x16x32_bf16.  K is walked in 32-wide stages
// through a ring of three LDS slots ((256 + TN) rows x 64 B, 16 B chunks XOR-swizzled per 4-row block on the LDS-DMA
// source address and on the ds_read_b128 reads).  Iteration t multiplies the fragments of stage t, which are already in
// registers, while it reads the fragments of stage t+1 from LDS (each activation fragment back into its own registers
// right after its four MFMAs, the filter fragments into the other of two register sets) and while stages t+2 and t+3
// are in flight from global memory.  Each wave waits only for ITS loads of stage t+1 with a counted s_waitcnt vmcnt and
// one raw s_barrier per stage orders the ring; all LDS lives in one array (cdna_hip_programming.md "Pipelining across
// barriers").  With TN = 128 the two resident workgroups drift out of phase, so one fills the MFMA pipe and the HBM
// write queue while the other sits in its barrier, prologue or epilogue.
// Epilogue as in conv_igemm.hip: bias / addend added in fp32 in the MFMA layout (a lane owns 4 consecutive channels of
// one row), one rounding, bf16 tile through LDS, 16 B coalesced row stores.
#include "common.h"
#include "icamd_internal.h"
#include <cstdlib>

namespace {

constexpr int TM = 256, TK = 32;

// 16 B chunk swizzle of the 64 B stage rows: chunk' = chunk ^ swz((row >> 2) & 3).  ds_read_b128 is served in the lane
// groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS): with lane = 16 * chunk + row a group
// holds rows 0-3 and 12-15 of one chunk and rows 4-11 of the next, so the per-4-row keys must make {s0, s3, 1^s1, 1^s2}
// and {s1, s2, 1^s0, 1^s3} both permutations of 0..3: s = (0, 2, 3, 1).
__device__ __forceinline__ int swz(int blk) { return (0x78 >> (2 * blk)) & 3; }

// 16 B store that does not stay in the XCD's L2 (MI355X_MICROARCH.md: plain / nt stores keep the line, sc1 drops it): the
// output tile is never re-read by this kernel, and 32 resident workgroups x 128 KB of output would otherwise push the operand
// lines the other workgroups of the XCD are about to share out of its 4 MB.
__device__ __forceinline__ void store16_sc1(void* ptr, const u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(ptr), "v"(v) : "memory");
}

template <int LPW>
__device__ __forceinline__ void wait_loads(int stages_younger) {   // LPW LDS-DMA loads per wave and stage
  static_assert(LPW == 4 || LPW == 6, "vmcnt immediates below");
  if constexpr (LPW == 4) {
    if (stages_younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (stages_younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    if (stages_younger >= 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (stages_younger == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

// ---- epilogue shared by both kernels: MFMA layout -> bias/addend -> bf16 -> LDS [256][TN] (chunk ^= row & (CPR-1)) ->
// coalesced rows.  acc[j][i]: rows wm*128 + i*16 + fr, channels wn*64 + j*16 + 4*fq .. +3.  The caller has made sure every
// wave is done reading the operand tiles (the output tile reuses that LDS).
template <int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmNtParams& p, unsigned char* smem, f32x4 (&acc)[4][8], int wm, int wn,
                                              int m0, int n0, int tile_m) {
  constexpr int ROWB = TN * 2, CPR = TN / 8;  // epilogue tile: bytes and 16 B chunks per row
  const int tid = threadIdx.x, lane = tid & 63;
  const int fr = lane & 15, fq = lane >> 4;

  // ---- epilogue: MFMA layout -> bias/addend -> bf16 -> LDS [256][TN] (chunk ^= row & (CPR-1)) -> coalesced rows ----
  long long aoff[8];      // addend row offsets of this lane's 8 rows; < 0: the row has no addend
  if (p.addend != nullptr) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int ml = wm * 128 + i * 16 + fr;
      const unsigned int mr = (m0 + ml < p.M) ? m0 + ml : 0;
      if (p.sub2_h > 0) {   // addend on the even pixel grid only (see icamd_conv2d_dgrad_sub2)
        const unsigned int n = fdiv(mr, p.divHW);
        const unsigned int rem = mr - n * (p.sub2_h * p.sub2_w);
        const unsigned int hh = fdiv(rem, p.divW);
        const unsigned int ww = rem - hh * p.sub2_w;
        const long long r = ((long long)n * ((p.sub2_h + 1) >> 1) + (hh >> 1)) * ((p.sub2_w + 1) >> 1) + (ww >> 1);
        aoff[i] = ((hh | ww) & 1u) ? -1 : r * p.N;
      } else {
        aoff[i] = (long long)mr * p.N;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int cl = wn * 64 + j * 16 + 4 * fq;
    const int cg = n0 + cl;
    const int cgc = cg < p.N ? cg : 0;
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr) b4 = *(const f32x4*)(p.bias + cgc);
    // The eight addend vectors of this channel group are requested TOGETHER (round 5): as `if (addend && aoff >= 0) { load; use }`
    // per row hipcc emitted branch -> global_load -> s_waitcnt vmcnt(0) -> ds_write thirty-two times in a row (seen in the ISA) --
    // 32 dependent trips to memory per tile, ~7 us of a 26 us tile on the residual-add launches.  Rows without an addend (the
    // even-grid form) read the zero page; so do their mask bytes.
    u32x2 av[8];
    unsigned int ab8[8];
    if (p.addend != nullptr) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bf16_t* src = aoff[i] >= 0 ? p.addend + aoff[i] + cgc : (const bf16_t*)icamd_zero_page;
        av[i] = *(const u32x2*)src;
        ab8[i] = 0xfu;
      }
      if (p.addend_bits != nullptr) {   // this lane's 4 channels are one nibble of the element's mask byte
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const unsigned char* bp = aoff[i] >= 0 ? p.addend_bits + ((aoff[i] + cgc) >> 3) : (const unsigned char*)icamd_zero_page;
          ab8[i] = ((unsigned int)*bp >> (4 * (fq & 1))) & 0xfu;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int ml = wm * 128 + i * 16 + fr;
      f32x4 v = acc[j][i] + b4;
      if (p.addend != nullptr) {
        v[0] += (ab8[i] & 1u) ? bf16_lo(av[i][0]) : 0.f;
        v[1] += (ab8[i] & 2u) ? bf16_hi(av[i][0]) : 0.f;
        v[2] += (ab8[i] & 4u) ? bf16_lo(av[i][1]) : 0.f;
        v[3] += (ab8[i] & 8u) ? bf16_hi(av[i][1]) : 0.f;
      }
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
  constexpr int RPP = (TN * 2) / CPR;   // rows per pass of the whole workgroup
  const int cp = tid & (CPR - 1), rg = tid / CPR;
  const int co = n0 + cp * 8;
  const bool want_stats = p.stats != nullptr;   // uniform
  f32x2 s1[4], s2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { s1[e] = f32x2{0.f, 0.f}; s2[e] = f32x2{0.f, 0.f}; }
  // (round 4) the x gelu'(z) form reads z four passes ahead of the stores: inside the loop every load sat behind the previous
  // pass's store (possible alias) and each pass paid a full load latency
  constexpr int NPS = TM / RPP, ZG = 4;
  u32x4 zv[ZG];
  const bool has_z = p.gelu_z != nullptr;
#pragma unroll 1
  for (int pg = 0; pg < NPS; pg += ZG) {
    if (has_z) {
#pragma unroll
      for (int u = 0; u < ZG; ++u) {
        const int m = m0 + (pg + u) * RPP + rg;
        zv[u] = (m < p.M && co < p.N) ? *(const u32x4*)(p.gelu_z + (long long)m * p.N + co) : u32x4{0u, 0u, 0u, 0u};
      }
    }
#pragma unroll
  for (int u = 0; u < ZG; ++u) {
    const int ps = pg + u;
    const int ml = ps * RPP + rg;
    const int m = m0 + ml;
    u32x4 o = *(const u32x4*)(smem + ml * ROWB + (((cp ^ ml) & (CPR - 1)) << 4));
    if (m < p.M && co < p.N) {
      const long long off = (long long)m * p.N + co;
      if (has_z) o = gelu_bwd8(o, zv[u]);
      if (p.gelu_inplace) o = gelu8(o);
      if (p.out_policy == 1) store16_sc1(p.out + off, o);
      else if (p.out_policy == 2) __builtin_nontemporal_store(o, (u32x4*)(p.out + off));
      else *(u32x4*)(p.out + off) = o;
      if (p.gelu_out != nullptr) *(u32x4*)(p.gelu_out + off) = gelu8(o);
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
  if (want_stats) {
    // BatchNorm statistics of the rounded outputs: one partial row per 256-row tile (row tile_m of the [ceil(M/128)] table
    // the consumer sums); the rows no tile owns are zero-filled by the tile whose index they exceed the tile count by
    __syncthreads();
    float* red = (float*)smem;             // [RPP][2][TN]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[(rg * 2 + 0) * TN + cp * 8 + 2 * e] = s1[e][0];
      red[(rg * 2 + 0) * TN + cp * 8 + 2 * e + 1] = s1[e][1];
      red[(rg * 2 + 1) * TN + cp * 8 + 2 * e] = s2[e][0];
      red[(rg * 2 + 1) * TN + cp * 8 + 2 * e + 1] = s2[e][1];
    }
    __syncthreads();
    const int ntm = (p.M + TM - 1) / TM, nrows = (p.M + 127) / 128;
    for (int idx = tid; idx < 2 * TN; idx += TN * 2) {
      const int which = idx / TN, c = idx - which * TN;
      float s = 0.f;
#pragma unroll 4
      for (int g = 0; g < RPP; ++g) s += red[(g * 2 + which) * TN + c];
      if (n0 + c < p.N) {
        p.stats[((long long)tile_m * 2 + which) * p.N + n0 + c] = s;
        if (ntm + tile_m < nrows) p.stats[((long long)(ntm + tile_m) * 2 + which) * p.N + n0 + c] = 0.f;
      }
    }
  }
}

template <int TN>
__global__ __launch_bounds__(TN * 2, 2) void gemm_nt_kernel(const GemmNtParams p) {
  constexpr int NSLOT = TN == 256 ? 4 : 3;   // ring depth: one 8-wave workgroup per CU can afford a fourth slot
  constexpr int NW = TN / 32;                 // waves: 2 along M x TN/64 along N
  constexpr int STAGE = (TM + TN) * TK * 2;   // bytes per ring slot
  constexpr int AI = 16 / NW, BI = (TN / 16) / NW, LPW = AI + BI;   // LDS-DMA instructions per wave and stage
  constexpr int ROWB = TN * 2;                // epilogue tile: bytes per row
  constexpr int LDSB = NSLOT * STAGE > TM * ROWB ? NSLOT * STAGE : TM * ROWB;   // the output tile reuses the ring
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDSB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: LDS-DMA destinations stay in SGPRs
  const int wm = wave & 1, wn = wave >> 1;
  const int fr = lane & 15, fq = lane >> 4;

  // XCD-aware tile order (blocks b and b+8 share an XCD): consecutive n-tiles of one m-tile stay on one XCD
  const unsigned int nblk = gridDim.x;
  unsigned int L;
  {
    const unsigned int xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
    const unsigned int q = nblk >> 3, r = nblk & 7u;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = L / p.ntiles_n, tile_n = L - tile_m * p.ntiles_n;
  const int m0 = tile_m * TM, n0 = tile_n * TN;
  const unsigned char* zero = (const unsigned char*)icamd_zero_page;

  // staging roles: a wave-instruction moves 16 rows x 64 B.  Rows past M / N read the zero page with a zero k-advance,
  // so the loop carries only pointer += step.
  const unsigned char* src[LPW];
  int step[LPW];
#pragma unroll
  for (int j = 0; j < LPW; ++j) {
    const bool isA = j < AI;
    const int row = (isA ? (wave * AI + j) : (wave * BI + j - AI)) * 16 + (lane >> 2);
    const int lc = ((lane & 3) ^ swz((row >> 2) & 3)) * 8;
    const bool valid = isA ? (m0 + row < p.M) : (n0 + row < p.N);
    const bf16_t* base = isA ? p.A + (long long)(m0 + row) * p.K : p.B + (long long)(n0 + row) * p.K;
    src[j] = valid ? (const unsigned char*)(base + lc) : zero;
    step[j] = valid ? TK * 2 : 0;
  }
  int fill_slot = 0;
  auto stage = [&]() {   // issues the next stage in k order into the next ring slot
    unsigned char* s = smem + fill_slot * STAGE;
#pragma unroll
    for (int j = 0; j < AI; ++j) __builtin_amdgcn_global_load_lds(GPTR(src[j]), LPTR(s + (wave * AI + j) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < BI; ++j)
      __builtin_amdgcn_global_load_lds(GPTR(src[AI + j]), LPTR(s + TM * 64 + (wave * BI + j) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < LPW; ++j) src[j] += step[j];
    fill_slot = fill_slot == NSLOT - 1 ? 0 : fill_slot + 1;
  };

  f32x4 acc[4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / TK;
  stage();
  if (nk > 1) stage();
  if (nk > 2) stage();
  if (NSLOT > 3 && nk > 3) stage();
  const int chunk = (fq ^ swz(fr >> 2)) * 16;   // fragment rows are 16-aligned + fr
  const int a_off = (wm * 128 + fr) * 64 + chunk, b_off = TM * 64 + (wn * 64 + fr) * 64 + chunk;
  bf16x8 xf[8], wf[4], wnext[4];
  wait_loads<LPW>(nk - 1 < NSLOT - 1 ? nk - 1 : NSLOT - 1);
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 8; ++i) xf[i] = *(const bf16x8*)(smem + a_off + i * 1024);
#pragma unroll
  for (int j = 0; j < 4; ++j) wf[j] = *(const bf16x8*)(smem + b_off + j * 1024);
  int read_slot = 1;   // slot of stage t+1
  // Ring invariant at the top of iteration t: stages <= t+2 issued; the fragments of stage t were requested from slot
  // t % 3 during iteration t-1.  lgkmcnt(0) before the barrier makes every wave's reads of that slot complete, so after
  // the barrier stage t+3 may overwrite it.
  auto iter = [&](int t, bf16x8 (&wc)[4], bf16x8 (&wn)[4]) {
    wait_loads<LPW>(nk - 2 - t < NSLOT - 2 ? nk - 2 - t : NSLOT - 2);   // this wave's part of stage t+1 has landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + NSLOT < nk) stage();
    const unsigned char* sn = smem + read_slot * STAGE;
    read_slot = read_slot == NSLOT - 1 ? 0 : read_slot + 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[j], xf[i], acc[j][i], 0, 0, 0);
      xf[i] = *(const bf16x8*)(sn + a_off + i * 1024);
      if (i < 2) {
        wn[2 * i] = *(const bf16x8*)(sn + b_off + (2 * i) * 1024);
        wn[2 * i + 1] = *(const bf16x8*)(sn + b_off + (2 * i + 1) * 1024);
      }
    }
    // pin the LDS reads between the MFMA groups (the scheduler otherwise sinks all twelve below the last MFMA and the
    // next iteration opens with a long lgkmcnt stall)
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
    for (int i = 2; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
  };
  int t = 0;
  for (; t + 2 < nk; t += 2) {
    iter(t, wf, wnext);
    iter(t + 1, wnext, wf);
  }
  if (t + 1 < nk) {   // one pipelined iteration left: the last fragments end up in wnext
    iter(t, wf, wnext);
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[j] = wnext[j];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[j][i], 0, 0, 0);
  __syncthreads();   // all fragment reads done: the ring becomes the output tile
  gemm_epilogue<TN>(p, smem, acc, wm, wn, m0, n0, tile_m);
}

// ------------------------------------------------------------------------------------------------------------------------
// 256 x 256 tile, eight phases per pair of 64-deep K tiles (cdna_hip_programming.md "The 256^2 8-phase template"), for the
// MFMA-bound problems (ViT-B/16's Linear layers: M = 50 432, N, K in {768, 2304, 3072}).
//
//   * 8 waves, wr = wave >> 2 owns rows wr*128 .. +128, wc = wave & 3 owns channels wc*64 .. +64: 128 x 64 accumulators per
//     wave = four QUADRANTS of 64 rows x 32 channels, one per phase: 16 MFMAs (4 row fragments x 2 channel fragments x 2
//     k-steps) between two raw s_barriers.
//   * The K tile is staged as four PIECES of 16 KB, cut so that a piece is exactly what one phase starts to need:
//       A_mh0 = rows {wr*128 + [0, 64)}  (phase 1)    B_nh0 = channels {wc*64 + [0, 32)}   (phase 1, kept for phase 4)
//       B_nh1 = channels {wc*64 + [32, 64)} (phase 2)  A_mh1 = rows {wr*128 + [64, 128)}    (phase 3)
//     a piece = [128 rows][128 B = the K tile's 64 elements] (16 B chunks XOR-swizzled by (row >> 1) & 7), filled by 16
//     LDS-DMA wave-instructions (2 per wave) of 8 rows x one whole 128 B line each.  LDS holds two K tiles (128 KB).  Every
//     phase issues ONE piece, six pieces ahead of the phase that runs: a piece is issued 5 phases before its first read and
//     >= 2 phases after the last read of the piece it overwrites (two K tiles earlier).
//   * The two wave groups (wr = 0 / 1: the two waves of every SIMD) run ONE BARRIER apart: while one group is between its
//     barriers issuing 16 MFMAs (and, among them, the phase's two LDS-DMA instructions), the other is in its load section
//     (ds_read_b128 of the next fragments and a counted s_waitcnt vmcnt(6) that leaves three pieces in flight).  A piece is
//     read one phase after the wait that retires it (both groups have then passed a barrier behind their own waits).
//   * Fragment reads are inline-asm ds_read_b128 with immediate offsets off per-lane base addresses (hipcc would put
//     vmcnt(0) in front of C++ LDS loads while LDS-DMA is in flight); the LDS-DMA is inline asm too, in its scalar-base form
//     (the builtin always takes a 64-bit per-lane address: 16 more VGPRs); all LDS is one array.
//   * Measured (round 3, M = 50 432, N = 2304, K = 768, random operands): 834 TFLOP/s (the ring kernel: 659; the vendor GEMM
//     behind torch.matmul: 955).  Ablations of the same instruction stream: made to re-read ONE K tile, so that every LDS-DMA
//     hits L2, it runs at 1 270 TFLOP/s -- the rate of its MFMAs alone (no staging and no fragment reads: 1 263; without the
//     barriers too: 1 273); on the real operands 29 % of the L2 requests miss (the 32 workgroups of an XCD share each A line
//     9 ways and each B line 3-4 ways, so ~20 % of the staged bytes are first touches even in lockstep) and the waves sit in
//     s_waitcnt / s_barrier for 54 % of their cycles.
//     Tried on top and NOT kept: a persistent form (one workgroup per CU walking its tiles, the next tile's first six pieces
//     issued under a wave-private epilogue): 5 % slower than letting the dispatcher place one workgroup per tile; on that
//     form, L2 prefetch touches (one global_load_ubyte per wave and K tile over the 512 lines of the K tile four tiles ahead):
//     11 % slower again -- the touches sit in the same in-order vmcnt queue as the LDS-DMA the load sections wait for;
//     LDS-DMA issued in the load section instead of among the MFMAs: the same; a ring of ten piece slots (all 160 KB of LDS,
//     two more pieces of lead): 5-11 % slower; the K tiles walked from a per-tile rotated start (so that the workgroups that
//     share a line do not all miss on it together): 3-6 % slower -- the lockstep requests merge in L2; the tiles of the last,
//     partial round over the CUs (N = 768: 591 tiles = 2.31 rounds) cut 3 ways in K, the last part adding the others' fp32
//     partial tiles in part order (sc0 sc1 stores / loads + a relaxed counter; with a release / acquire fence pair instead
//     every part paid 40-60 us for writing back its XCD's L2): 252.6 -> 246.9 / 254.8 / 254.7 us at K = 3072, 204 -> 192-203
//     at K = 2304, 73 -> 87 at K = 768 -- inside the run-to-run spread, because the 79 tiles of a partial round run faster
//     than the tiles of a full one (the shared L2 -> LDS delivery is the bound, not the count of busy CUs).
//     Round 5, the same split once more with the parts of a tile placed on ONE XCD (workgroups 8 apart; HW_REG_XCC_ID confirmed
//     workgroup b on XCD b mod 8 in every launch) so that the partial tiles meet in that XCD's L2 -- plain stores, vmcnt(0), a
//     relaxed device-scope ticket, the last arriver sums the parts in part order: correct, bit-stable, and slower again: K = 3072
//     251-258 -> 260-266 us, K = 2304 174-189 -> 197-213, K = 768 75 -> 92-100.  The 237 parts write 60 MB of partial tiles at the
//     same moment, 7.7 MB per XCD into a 4 MB L2, and read them back: ~35 us at the very end of the kernel with nothing left to
//     overlap it, against the 25-30 us the shorter tail saves.  The vendor GEMM on these shapes (rocprofv3: Custom_Cijk_..._SK3_...
//     MT256x256x64, 922-1060 TFLOP/s) is a Stream-K kernel: every CU walks an equal, contiguous share of ALL K iterations, so the
//     same volume of partial tiles is exchanged all along the run, under other CUs' MFMAs -- a persistent form of this kernel,
//     which round 3 measured 5 % slower than one workgroup per tile, would have to come first.  (Round 5, same day: the plain persistent
//     loop again -- tile loop around the whole body, thread index laundered per tile so that no lane constant is carried across
//     the epilogue: 246 registers, no spill -- against one workgroup per tile in the same build on one box: K = 3072 259-268 vs
//     273-278 us, N = 2304 206-217 vs 213-222, 768 x 768 and N = 3072 equal; against THIS kernel (231 registers) in the ViT-B/16 step,
//     three rounds on one box: 35.59-35.69 ms both ways -- not slower any more, not a gain by itself; and a
//     Stream-K walk hands every CU two partial tiles, 134 MB of fp32 partials per launch against 80-160 MB of operands.)
// ------------------------------------------------------------------------------------------------------------------------
constexpr int PIECE = 16384, KTILE = 65536;   // bytes: one piece, one K tile (pieces in stream order A_mh0, B_nh0, B_nh1, A_mh1)

// LDS-DMA of 16 B per lane with a SCALAR 64-bit base and a 32-bit per-lane byte offset.  M0 = the wave-uniform LDS
// destination, as the builtin would set it; hipcc re-materialises M0 before every use of its own.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void glds16_sbase(const void* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(voff), "s"(sbase) : "memory", "m0");
}
#pragma clang diagnostic pop

__device__ __forceinline__ void wait_pieces_younger(int n) {   // leave the n youngest pieces (2 LDS-DMA each) in flight
  if (n >= 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (n == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (n == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(512, 2) void gemm_nt_8phase_kernel(const GemmNtParams p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * KTILE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int nk = p.K / 64;
  const int last_piece = 4 * nk - 1;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LPTR(smem));
  // fragment rows are 16-aligned + fr, so the swizzle key (row >> 1) & 7 is a lane constant; k-step 1 is chunk ^ 4: a second
  // base address (XOR 64) instead of an immediate.  With lane = 16*fq + fr every ds_read_b128 lane group covers the 16 slots
  // of the 256 B bank span exactly once.
  const unsigned sw = (unsigned)((fr >> 1) & 7);
  const unsigned c0 = (((unsigned)fq ^ sw) & 7u) << 4;          // chunk (0*4 + fq) ^ sw
  const unsigned aA0 = lds_base + (unsigned)((wr * 64 + fr) * 128) + c0, aA1 = aA0 + KTILE;
  const unsigned aB0 = lds_base + (unsigned)((wc * 32 + fr) * 128) + c0, aB1 = aB0 + KTILE;
  const unsigned char* const Ab = (const unsigned char*)p.A;
  const unsigned char* const Bb = (const unsigned char*)p.B;
  const unsigned rowb = (unsigned)p.K * 2u;                      // operand row pitch in bytes

  // XCD-aware tile order (blocks b and b + 8 share an XCD): consecutive n-tiles of one m-tile stay on one XCD
  unsigned int L;
  {
    const unsigned int nblk = gridDim.x;
    const unsigned int xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
    const unsigned int q = nblk >> 3, r = nblk & 7u;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // Tiles in groups of group_n n-tiles, inside a group the n-tile fastest, then the m-tile: the 32 workgroups an XCD runs at a
  // time cover (32 / group_n) m-tiles x group_n n-tiles, and the group's B rows (group_n x 256 x K: 1.5 MB at K = 768, group_n =
  // 4) stay in the XCD's 4 MB L2 while the A rows stream past; with all n-tiles in one group (N = 2304: 3.5 MB of B) every
  // pass over a few m-tiles re-fetched B from beyond L2.
  int tile_m, tile_n;
  {
    const int gn = p.group_n;
    const int ntm = (int)(gridDim.x / (unsigned)p.ntiles_n);
    const int per_group = ntm * gn;
    const int g = (int)L / per_group;
    const int rem = (int)L - g * per_group;
    const int width = (p.ntiles_n - g * gn < gn) ? p.ntiles_n - g * gn : gn;   // the last group may be narrower
    tile_m = rem / width;
    tile_n = g * gn + (rem - tile_m * width);
  }
  const int m0 = tile_m * 256, n0 = tile_n * 256;

  // staging roles: instruction j of this wave is instruction q = wave*2 + j of a piece: piece rows q*8 .. +8, EIGHT lanes per
  // row = one whole 128 B line of the operand per row (the 16 B chunks permuted on the SOURCE side).  Per lane only a 32-bit
  // byte offset from the operand's base (M*K*2 < 4 GB is checked by the launcher); the K tile enters through the scalar base.
  // Rows past M / channels past N read the LAST valid row instead: valid memory, and what they produce is never stored.
  unsigned off[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int q = wave * 2 + j;
      const int r = q * 8 + (lane >> 3);
      const int lc = ((lane & 7) ^ ((r >> 1) & 7)) * 16;
      const bool isA = (t == 0 || t == 3);
      const int grow = isA ? m0 + (r >> 6) * 128 + (t == 3 ? 64 : 0) + (r & 63) : n0 + (r >> 5) * 64 + (t == 2 ? 32 : 0) + (r & 31);
      const int lim = isA ? p.M - 1 : p.N - 1;
      off[t][j] = (unsigned)(grow < lim ? grow : lim) * rowb + (unsigned)lc;
    }
  auto stage1 = [&](auto tc, auto bc, auto jc, int ktile) {   // instruction J of piece type T of K tile `ktile` into buffer B
    constexpr int T = decltype(tc)::value, B = decltype(bc)::value, J = decltype(jc)::value;
    const unsigned char* kb = ((T == 0 || T == 3) ? Ab : Bb) + (size_t)ktile * 128;
    glds16_sbase(kb, off[T][J], lds_base + (unsigned)(B * KTILE + T * PIECE + (wave * 2 + J) * 1024));
  };
  // prologue: pieces 0..5 (the whole first K tile and the first half of the second)
  static_for<0, 6>([&](auto sc) {
    constexpr int S = decltype(sc)::value;
    if (S < 4 || nk > 1) {
      stage1(std::integral_constant<int, (S & 3)>{}, std::integral_constant<int, (S >> 2)>{}, std::integral_constant<int, 0>{}, S >> 2);
      stage1(std::integral_constant<int, (S & 3)>{}, std::integral_constant<int, (S >> 2)>{}, std::integral_constant<int, 1>{}, S >> 2);
    }
  });
  wait_pieces_younger((nk > 1 ? 5 : 3) - 1);      // pieces 0 and 1 have landed (this wave's part)

  f32x4 acc[4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();      // the stagger: group 1 runs one barrier behind group 0
  bf16x8 af[8], b0[4], b1[4];   // A fragments [ks*4 + i] of the current row half; B fragments [ks*2 + j] of both channel halves

  // STEADY: every piece this phase and the next ones touch exists (no end-of-K checks, constant wait counts)
  auto phase = [&](auto phc, auto steadyc, int g) {
    constexpr int PH = decltype(phc)::value;       // 0..7: K-tile parity PH >> 2, phase PH & 3
    constexpr bool STEADY = decltype(steadyc)::value;
    constexpr int BUF = PH >> 2, P = PH & 3;
    const unsigned aA = BUF ? aA1 : aA0, aB = BUF ? aB1 : aB0;
    // ---- load section: the fragments this phase starts to need, then the wait that retires what the NEXT phase reads
    if constexpr (P == 0) {
      static_for<0, 8>([&](auto c) { constexpr int x = decltype(c)::value; af[x] = lds_read128_off<0 * PIECE + (x & 3) * 2048>((x >> 2) ? aA ^ 64u : aA); });
      static_for<0, 4>([&](auto c) { constexpr int x = decltype(c)::value; b0[x] = lds_read128_off<1 * PIECE + (x & 1) * 2048>((x >> 1) ? aB ^ 64u : aB); });
    } else if constexpr (P == 1) {
      static_for<0, 4>([&](auto c) { constexpr int x = decltype(c)::value; b1[x] = lds_read128_off<2 * PIECE + (x & 1) * 2048>((x >> 1) ? aB ^ 64u : aB); });
    } else if constexpr (P == 2) {
      static_for<0, 8>([&](auto c) { constexpr int x = decltype(c)::value; af[x] = lds_read128_off<3 * PIECE + (x & 3) * 2048>((x >> 2) ? aA ^ 64u : aA); });
    }
    // pieces <= g + 2 (what phase g + 1 starts to read) have landed; issued so far: pieces <= g + 5
    if constexpr (STEADY) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else wait_pieces_younger((g + 5 < last_piece ? g + 5 : last_piece) - (g + 2));
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4]), "+v"(af[5]), "+v"(af[6]), "+v"(af[7]),
                   "+v"(b0[0]), "+v"(b0[1]), "+v"(b0[2]), "+v"(b0[3]), "+v"(b1[0]), "+v"(b1[1]), "+v"(b1[2]), "+v"(b1[3])
                 :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // ---- compute section: one quadrant, 16 MFMAs; among them the phase's piece (g + 6: type (PH + 2) & 3, K tile (g + 6) >> 2)
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
      acc[NH * 2 + j][MH * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(NH ? b1[ks * 2 + j] : b0[ks * 2 + j], af[ks * 4 + i],
                                                                           acc[NH * 2 + j][MH * 4 + i], 0, 0, 0);
    });
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };

  int g = 0;
  int t = 0;
  for (; t + 2 <= nk && g + 13 <= last_piece; t += 2) {   // both K tiles' phases stage pieces that exist
    static_for<0, 8>([&](auto phc) { phase(phc, std::true_type{}, g + decltype(phc)::value); });
    g += 8;
  }
  for (; t + 2 <= nk; t += 2) {
    static_for<0, 8>([&](auto phc) { phase(phc, std::false_type{}, g + decltype(phc)::value); });
    g += 8;
  }
  if (nk & 1) static_for<0, 4>([&](auto phc) { phase(phc, std::false_type{}, g + decltype(phc)::value); });
  if (wr == 0) __builtin_amdgcn_s_barrier();      // re-align the two groups
  __syncthreads();                                 // every fragment read is done: the operand tiles become the output tile
  gemm_epilogue<256>(p, smem, acc, wr, wc, m0, n0, tile_m);
}

// ICAMD_GEMM_TN: 128 / 256 force the ring kernel at that tile width; 8 (default) = the 8-phase 256 x 256 kernel wherever K is a
// multiple of 64 (else the 128-wide ring)
int tile_n_width() {
  static const int tn = [] { const char* e = getenv("ICAMD_GEMM_TN"); const int v = e ? atoi(e) : 8; return v == 256 || v == 128 ? v : 8; }();
  return tn;
}

}  // namespace

bool icamd_gemm_nt_wanted(long long M, int N, int K) {
  static const int mode = [] { const char* e = getenv("ICAMD_GEMM_NT"); return e ? atoi(e) : 1; }();
  if (mode == 0 || K % TK != 0 || N % 8 != 0 || M >= (1ll << 31)) return false;
  if (mode == 2) return true;   // forced (tests)
  // MFMA-bound problems only: enough K to amortise the big-tile prologue / epilogue and enough tiles to fill 256 CUs.
  // Measured on MI355X, round 2 (tools/one_layer.py under rocprofv3): at K = 256 / 512 (ResNet-50's deep 1x1 layers) the
  // 128x128 implicit-GEMM kernel is 5-15 % FASTER (256->1024 at 14x14: 52.6 vs 58.2 us; 2048->512 data gradient at 7x7:
  // 40.2 vs 44.3 us); this kernel wins from ViT's K = 768 up.
  // (K >= 2048: the 7x7 layers of ResNet-50 -- 2048 -> 512 forward 52.9 -> 40.0 us, 512 -> 2048 data gradient 52.0 -> 39.1 us --
  // have only 98 tile pairs but sixty-four ring stages each)
  const long long pairs = ((M + TM - 1) / TM) * ((N + 255) / 256);
  // (round 3, K = 1024 / N = 256 at 14x14 -- ResNet-50 layer3's 1024 -> 256 forward and 256 -> 1024 data gradient: 196 tiles,
  // one per CU, sixteen K tiles each: 38.7 -> 30.6 us and 37.5 -> 29.5 us against the resident-filter kernel)
  static const bool k1024 = [] { const char* e = getenv("ICAMD_GEMM_K1024"); return !(e && atoi(e) == 0); }();
  if (k1024 && K == 1024 && N == 256 && pairs >= 160 && pairs <= 256) return true;
  static const int mink = [] { const char* e = getenv("ICAMD_GEMM_MINK"); return e ? atoi(e) : 768; }();   // (A/B: 384 = ConvNeXt-T stage 2's fc1)
  return K >= mink && N >= 256 && (pairs >= 256 || (K >= 2048 && pairs >= 96));
}

int icamd_gemm_nt_launch(GemmNtParams& p, hipStream_t stream) {
  if (p.K % TK != 0 || p.N % 8 != 0 || p.M <= 0) return ICAMD_ERR_UNSUPPORTED;
  int tn = tile_n_width();
  // (the 8-phase kernel addresses its operands with 32-bit byte offsets)
  const bool eight = tn == 8 && p.K % 64 == 0 && (long long)p.M * p.K < (1ll << 31) && (long long)p.N * p.K < (1ll << 31);
  if (tn == 8) tn = eight ? 256 : 128;
  p.ntiles_n = (p.N + tn - 1) / tn;
  if (p.sub2_h > 0) {
    p.divHW = make_fastdiv((unsigned)(p.sub2_h * p.sub2_w));
    p.divW = make_fastdiv((unsigned)p.sub2_w);
  }
  const long long tiles = (long long)((p.M + TM - 1) / TM) * p.ntiles_n;
  // output stores non-temporal by default (round 5): the output tile is never re-read by this kernel, and as plain stores the 32
  // resident workgroups' 128 KB tiles competed with the operand lines their XCD shares (ViT-B/16 36.84 -> 36.30-36.33 ms in two A/B
  // pairs on one box, ConvNeXt-T neutral; isolated N = 2304 / K = 768: 850 -> 887 TFLOP/s).  ICAMD_GEMM_OUT_POLICY=0: plain, 1: sc1.
  static const int out_policy = [] { const char* e = getenv("ICAMD_GEMM_OUT_POLICY"); return e ? atoi(e) : 2; }();
  p.out_policy = out_policy;
  if (eight) {
    static const int gn = [] { const char* e = getenv("ICAMD_GEMM_GROUP_N"); return e ? atoi(e) : 4; }();
    p.group_n = gn > 0 && gn < p.ntiles_n ? gn : p.ntiles_n;
    hipLaunchKernelGGL(gemm_nt_8phase_kernel, dim3((unsigned)tiles), dim3(512), 0, stream, p);
  }
  else if (tn == 256) hipLaunchKernelGGL(gemm_nt_kernel<256>, dim3((unsigned)tiles), dim3(512), 0, stream, p);
  else hipLaunchKernelGGL(gemm_nt_kernel<128>, dim3((unsigned)tiles), dim3(256), 0, stream, p);
  return icamd_launch_status();
}
