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

template <int TN>
__global__ __launch_bounds__(TN * 2, 2) void gemm_nt_kernel(const GemmNtParams p) {
  constexpr int NSLOT = TN == 256 ? 4 : 3;   // ring depth: one 8-wave workgroup per CU can afford a fourth slot
  constexpr int NW = TN / 32;                 // waves: 2 along M x TN/64 along N
  constexpr int STAGE = (TM + TN) * TK * 2;   // bytes per ring slot
  constexpr int AI = 16 / NW, BI = (TN / 16) / NW, LPW = AI + BI;   // LDS-DMA instructions per wave and stage
  constexpr int ROWB = TN * 2, CPR = TN / 8;  // epilogue tile: bytes and 16 B chunks per row
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
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int ml = wm * 128 + i * 16 + fr;
      f32x4 v = acc[j][i] + b4;
      if (p.addend != nullptr && aoff[i] >= 0) {
        const u32x2 a = *(const u32x2*)(p.addend + aoff[i] + cgc);
        unsigned int ab = 0xfu;
        if (p.addend_bits != nullptr)   // this lane's 4 channels are one nibble of the element's mask byte
          ab = ((unsigned int)p.addend_bits[(aoff[i] + cgc) >> 3] >> (4 * (fq & 1))) & 0xfu;
        v[0] += (ab & 1u) ? bf16_lo(a[0]) : 0.f;
        v[1] += (ab & 2u) ? bf16_hi(a[0]) : 0.f;
        v[2] += (ab & 4u) ? bf16_lo(a[1]) : 0.f;
        v[3] += (ab & 8u) ? bf16_hi(a[1]) : 0.f;
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
#pragma unroll 4
  for (int ps = 0; ps < TM / RPP; ++ps) {
    const int ml = ps * RPP + rg;
    const int m = m0 + ml;
    u32x4 o = *(const u32x4*)(smem + ml * ROWB + (((cp ^ ml) & (CPR - 1)) << 4));
    if (m < p.M && co < p.N) {
      const long long off = (long long)m * p.N + co;
      if (p.gelu_z != nullptr) o = gelu_bwd8(o, *(const u32x4*)(p.gelu_z + off));
      if (p.gelu_inplace) o = gelu8(o);
      *(u32x4*)(p.out + off) = o;
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

int tile_n_width() {
  static const int tn = [] { const char* e = getenv("ICAMD_GEMM_TN"); return e && atoi(e) == 256 ? 256 : 128; }();
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
  return K >= 768 && N >= 256 && (pairs >= 256 || (K >= 2048 && pairs >= 96));
}

int icamd_gemm_nt_launch(GemmNtParams& p, hipStream_t stream) {
  if (p.K % TK != 0 || p.N % 8 != 0 || p.M <= 0) return ICAMD_ERR_UNSUPPORTED;
  const int tn = tile_n_width();
  p.ntiles_n = (p.N + tn - 1) / tn;
  if (p.sub2_h > 0) {
    p.divHW = make_fastdiv((unsigned)(p.sub2_h * p.sub2_w));
    p.divW = make_fastdiv((unsigned)p.sub2_w);
  }
  const long long tiles = (long long)((p.M + TM - 1) / TM) * p.ntiles_n;
  if (tn == 256) hipLaunchKernelGGL(gemm_nt_kernel<256>, dim3((unsigned)tiles), dim3(512), 0, stream, p);
  else hipLaunchKernelGGL(gemm_nt_kernel<128>, dim3((unsigned)tiles), dim3(256), 0, stream, p);
  return icamd_launch_status();
}
