// Pointwise (1x1, stride 1) convolution forward / data gradient for K <= 512 with the FILTER RESIDENT IN REGISTERS
// (reference: timm Bottleneck.conv1 / conv3 / downsample under model(samples) and loss.backward(),
// /root/reference/engine.py:48,51,64,72):
//   out[m][n] = sum_k A[m][k] * B[n][k]  (+ addend[m][n] [* mask bit]),  A = activations [M][K], B = filters [N][K], bf16.
//
// Why (profiles/README.md, round 2): on these layers conv_igemm.hip's 128x128 tiles are bound by L2 requests -- every tile
// re-reads its 128 filter rows and every activation row is re-read once per channel tile -- and by per-tile fixed cost
// (4 to 16 k-steps between a prologue and an epilogue).  Here, as in conv3x3_c64_resident_kernel:
//   * a wave keeps its 16*NF filter rows for ALL of K in registers (NF * K/32 fragments <= 128 VGPRs), loaded once;
//   * workgroups are persistent over a range of rows (two per CU, phases drift apart): only the activation tile goes
//     through LDS, double-buffered by LDS-DMA, one s_barrier per tile; 16 B chunks XOR-swizzled by the row so that the
//     per-lane read addresses are FOUR constants for the whole kernel -- buffer, fragment and k-step are the instruction's
//     immediate offset -- and the tile loop contains no address arithmetic;
//   * the four waves split the channels (WN = 4: the workgroup covers 64*NF channels, every wave reads the same activation
//     fragments) or, for 64-channel outputs, the rows (WN = 1);
//   * wave-private epilogue: MFMA layout -> [addend] -> bf16 -> two 16-row LDS patches -> 16 B row stores; BatchNorm partial
//     sums stay in registers across tiles and leave as one partial row per workgroup;
//   * the addend of the residual data gradients (full-size, optionally gated by 1-bit ReLU masks, or the even-grid form of
//     icamd_conv2d_dgrad_sub2) is brought by LDS-DMA into a wave-private patch at the start of the tile -- whole rows, 16 B
//     per lane, no registers, zero page for rows that have none -- and read back in the MFMA layout after the MFMAs.
//     (Loaded in the MFMA layout into registers instead, 8 B per lane, the same launches were SLOWER than conv_igemm:
//     data gradients 4.27 -> 4.59 ms per step; staged this way 4.22 -> 3.88 ms.)
#include "common.h"
#include "icamd_internal.h"
#include <cstdlib>

namespace {

// 16 B per lane LDS-DMA with the cache policy as a (wave-uniform) run-time choice: the builtin's policy operand is an immediate
__device__ __forceinline__ void pw_glds16(const void* src, void* dst, bool nt) {
  if (nt) __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(dst), 16, 0, 2);
  else __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(dst), 16, 0, 0);
}

// KS: k-steps of 32 (K = 32*KS); NF: 16-channel fragments per wave; MF: 16-row fragments per wave; WN: waves across the
// channels (4 / WN across the rows); ADD: addend epilogue compiled in.
// EXT: bias, GELU epilogues, a row pitch different from K and zero-extended filter columns (K = 96 run as KS = 4).
// BNR (with ADD): the output is g = the previous block's masked output gradient, and the pass-1 sums of that block's last
// BatchNorm backward (sum g, sum g * xhat) are accumulated on the way out: the wave's sub-tile of the BatchNorm's input y comes
// by LDS-DMA into a second wave-private patch at the START of the tile (exactly as the addend does: whole rows, no registers,
// landing under the MFMAs), the mask words are one 64-bit load per row, and the store pass -- which already holds 8 channels
// of one row per lane -- reads the matching 16 B of y back and keeps the sums in registers across tiles.  bn_bwd_reduce (a full
// read of dout and y) disappears for that BatchNorm; this kernel reads y once instead.
// EPI (round 4): the inference epilogue of evaluate()'s BatchNorm-folded forward (engine.py:145-225): + bias[n] in fp32, + addend
// (the ADD patch), ReLU, ONE rounding -- the arithmetic of conv_igemm's fused epilogue, which these launches used to run on.
// (round 4's ABL instantiation -- the cost model of a fused data-gradient + weight-gradient kernel, profiles/r04_ablate_fused_dgrad_wgrad.txt --
// was measurement-only code and left the product in round 5; `git show 08b0279:imageclassification_amd/csrc/conv1x1_resident.hip` has it.)
// EXT: 0 = off, 1 = with the z patch of the "x gelu'(z)" form, 2 = forward forms only (no z patch: the LDS it would take buys
// twice the rows per tile instead)
template <int KS, int NF, int MF, int WN, bool ADD, int EXT = 0, bool BNR = false, bool EPI = false>
__global__ __launch_bounds__(256, 2) void conv1x1_resident_kernel(const PwResidentParams p) {
  constexpr int WM = 4 / WN;
  constexpr int TM = WM * MF * 16;              // rows per tile
  constexpr int ROWB = KS * 64;                 // bytes per activation row
  constexpr int A_BYTES = TM * ROWB;
  constexpr int CW = NF * 16;                   // channels per wave
  constexpr int EROW = CW * 2;                  // bytes per row of the epilogue patch (128 or 64)
  constexpr int E_WAVE = 2 * 16 * EROW;         // two 16-row patches per wave
  constexpr int NV = KS < 4 ? KS : 4;           // address variants (k-step bits that the row swizzle touches)
  constexpr int NINST = A_BYTES / 1024;         // LDS-DMA instructions per tile
  constexpr int IPW = NINST / 4;                // per wave
  static_assert(NINST % 4 == 0 && (NF == 2 || NF == 4) && KS >= 2 && KS <= 16, "configuration");
  constexpr int P_WAVE = ADD ? MF * 16 * EROW : 0;   // ADD: this wave's [MF*16 rows][CW] addend patch (LDS-DMA target)
  // BNR: the same shape for the BatchNorm input y.  EXT (round 4): the same patch carries the pre-GELU activations z of the
  // "data gradient x gelu'(z)" form: read per row group from global memory inside the store pass they cost a full load latency
  // four times per tile (455 us for 1.39 GB at ConvNeXt-T's stage 0); as a patch they land under the MFMAs.
  constexpr int Y_WAVE = (BNR || EXT == 1) ? MF * 16 * EROW : 0;
  static_assert(!BNR || (ADD && NF == 4 && WN == 4 && !EXT), "bnred: 256-channel workgroups with an addend");
  constexpr int X_BYTES = 0;
  constexpr int YB = EXT == 1 ? 2 : 1;                    // EXT: the z patch of the NEXT tile travels with that tile's activations
  static_assert(2 * A_BYTES + 4 * E_WAVE + 4 * P_WAVE + YB * 4 * Y_WAVE + X_BYTES <= 80 * 1024, "two workgroups per CU");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * A_BYTES + 4 * E_WAVE + 4 * P_WAVE + YB * 4 * Y_WAVE + X_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave % WN, wm = wave / WN;
  // non-temporal LDS-DMA for the once-read streams: the activation tile when one channel tile covers a row range (bit 0), the
  // addend / BatchNorm-input / z patches always (bit 1: a workgroup's own channels, nobody else reads them)
  const bool nt_loads = (p.nt_loads & 1) != 0, nt_patch = (p.nt_loads & 2) != 0;
  const int fr = lane & 15, fq = lane >> 4;
  // Workgroup -> (row range `split`, channel tile `tile_n`).  Round 4: XCD-aware.  The ntiles_n channel tiles of one row range
  // stage the SAME activation tiles; numbered consecutively (rounds 2-3) they land on ntiles_n different XCDs (dispatch is
  // round-robin: workgroup w runs on XCD w % 8) and every one fetches the tile beyond its own L2 -- PMC on ConvNeXt-T's 96 -> 384
  // layers (3 tiles): 638 MB fetched per launch for 359 MB of operands, L2 hit rate 36 %.  Placed 8 apart they share an XCD, run
  // side by side (all workgroups of the grid are resident) and share the tile in its L2.
  int tile_n, split;
  {
    const int ntn = p.ntiles_n, S = (int)gridDim.x / ntn, G = p.xcd_groups ? (S / 8) * 8 : 0;
    const int w = (int)blockIdx.x;
    if (w < G * ntn) {
      const int r = w % (8 * ntn);
      tile_n = r / 8;
      split = (w / (8 * ntn)) * 8 + (r & 7);
    } else {
      const int r = w - G * ntn;
      tile_n = r % ntn;
      split = G + r / ntn;
    }
  }
  const int n0 = (tile_n * WN + wn) * CW;       // this wave's first channel
  const int m_begin = split * p.rows_per_split;
  const int m_end = (p.M < m_begin + p.rows_per_split) ? p.M : m_begin + p.rows_per_split;
  const bf16_t* zero = (const bf16_t*)icamd_zero_page;
  const unsigned lds_base = (unsigned)(uintptr_t)LPTR(smem);
  unsigned char* const sE = smem + 2 * A_BYTES + wave * E_WAVE;

  // Row swizzle of the 16 B chunks: position = chunk ^ sw(row); 128 B rows alternate between the two halves of the 256 B
  // bank span, so their key is (row >> 1) & 7; wider rows all start on bank 0 and use row & 15.
  auto sw = [](int row) { return ROWB == 128 ? (row >> 1) & 7 : row & 15; };

  // ---- staging: instruction j of wave w is instruction q = j*4 + w of the tile: 1 KiB = 1024 / ROWB rows (or part of one)
  auto stage = [&](int m0, int buf) {
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
      const int q = j * 4 + wave;
      const int byte = q * 1024 + lane * 16;
      const int row = byte / ROWB;
      const int pos = (byte % ROWB) >> 4;
      const int chunk = pos ^ sw(row);
      const int m = m0 + row;
      long long srow = m;
      if constexpr (!EXT && !ADD && !BNR) {
        if (p.gat_ow > 0) {     // stride-2 gather: the source row of output pixel (n, oh, ow) is input pixel (n, 2 oh, 2 ow)
          const unsigned int n = fdiv((unsigned)m, p.gdivHW);
          const unsigned int rem = (unsigned)m - n * (unsigned)(p.gat_oh * p.gat_ow);
          const unsigned int oh = fdiv(rem, p.gdivW);
          const unsigned int ow = rem - oh * (unsigned)p.gat_ow;
          srow = ((long long)n * p.gat_ih + 2 * oh) * p.gat_iw + 2 * ow;
        }
      }
      const bf16_t* src = m < m_end ? p.A + (srow * p.K + chunk * 8) : zero;
      if constexpr (EXT) {
        // columns >= Ktrue of a staged row come from the zero page (round 5, ADVICE r4: read as the first columns of the NEXT row
        // and multiplied by the zero filter columns, one Inf / NaN in row m + 1 -- possibly the next image -- poisoned every output
        // channel of row m; a Linear layer keeps a non-finite value in its own row).  Same number of LDS-DMA instructions.
        src = (m < m_end && chunk * 8 < p.Ktrue) ? p.A + ((long long)m * p.lda + chunk * 8) : zero;
      }
      pw_glds16(src, smem + buf * A_BYTES + q * 1024, nt_loads);
    }
  };
  // EXT, data gradient x gelu'(z): this wave's [MF*16 rows][CW] sub-tile of z, staged ONE TILE AHEAD with the activations (round 4,
  // second form: issued at the start of its own tile the patch had only the tile's 64 MFMAs to land under and the store pass waited
  // for it -- 188 us at K = 192 either way; the loop-top wait that retires the tile's activations retires the patch with them)
  auto stage_z = [&](int tm0, int buf) {
    if constexpr (EXT == 1) {
      if (p.gelu_z != nullptr) {
        constexpr int LPRA = EROW / 16, RPIA = 64 / LPRA;
        unsigned char* const dst = smem + 2 * A_BYTES + 4 * E_WAVE + 4 * P_WAVE + (buf * 4 + wave) * Y_WAVE;
        const int mwz = tm0 + wm * MF * 16;
#pragma unroll
        for (int q = 0; q < Y_WAVE / 1024; ++q) {
          const int row = q * RPIA + lane / LPRA, pos = lane % LPRA;
          const int chunk = pos ^ (row & (LPRA - 1));
          const int m = mwz + row;
          const bf16_t* src = m < p.M ? p.gelu_z + ((long long)m * p.N + n0 + chunk * 8) : zero;
          pw_glds16(src, dst + q * 1024, nt_patch);
        }
      }
    }
  };
  int m0 = m_begin;
  if (m0 < m_end) { stage(m0, 0); stage_z(m0, 0); }   // in flight under the filter loads

  // ---- the filter: fragment (ks, j) = rows n0 + j*16 + fr, 8 input channels at ks*32 + fq*8
  bf16x8 wf[KS][NF];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      if constexpr (EXT) {
        const int k = ks * 32 + fq * 8;
        wf[ks][j] = k < p.Ktrue ? *(const bf16x8*)(p.B + (long long)(n0 + j * 16 + fr) * p.Ktrue + k) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      } else {
        wf[ks][j] = *(const bf16x8*)(p.B + (long long)(n0 + j * 16 + fr) * p.K + ks * 32 + fq * 8);
      }
    }
  // arrived before the loop (the builtin: hipcc's wait-count bookkeeping sees it); also covers the first tile
  __builtin_amdgcn_s_waitcnt(0x0F70);

  // ---- read addresses: row wm*MF*16 + fr (+16*i by immediate), chunk ks*4 + fq at position chunk ^ SW.  SW is constant
  // per lane (fragments are 16 rows apart); its low two bits meet fq, its next two meet ks & 3: NV variants.
  unsigned ra[NV];
  {
    const int row = wm * MF * 16 + fr;
    const int SW = sw(row);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int pos = ((((v ^ (SW >> 2)) & (NV - 1)) << 2) | (fq ^ (SW & 3)));
      ra[v] = lds_base + (unsigned)(row * ROWB + (pos << 4));
    }
  }

  f32x2 s1[4], s2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { s1[e] = f32x2{0.f, 0.f}; s2[e] = f32x2{0.f, 0.f}; }
  f32x4 bias4[(EXT || EPI) ? NF : 1];      // MFMA layout: a lane owns channels n0 + j*16 + 4*fq .. +3 of its rows
  if constexpr (EXT || EPI) {
#pragma unroll
    for (int j = 0; j < NF; ++j)
      bias4[j] = p.bias != nullptr ? *(const f32x4*)(p.bias + n0 + j * 16 + 4 * fq) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool want_stats = BNR || (!ADD && p.stats != nullptr);   // the launcher never pairs statistics with an addend

  constexpr int STORES = MF * (EROW / 64);       // row-store instructions per wave and tile
  auto tile = [&](auto bufc, int tm0) {
    constexpr int BUF = decltype(bufc)::value;
    f32x4 acc[NF][MF];
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int i = 0; i < MF; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ADD: the tile's addend sub-tile of THIS wave goes to its LDS patch by LDS-DMA (16 B per lane, whole rows: coalesced;
    // no registers; lands under the MFMAs below; rows past M and, for the even-grid form, odd pixels come from the zero
    // page, so no mask is needed for them); the 1-bit masks of the wave's 16*NF channels are one word per row.
    const int mw = tm0 + wm * MF * 16;
    unsigned long long abw[ADD ? MF : 1];
    unsigned long long pbw[BNR ? MF : 1];          // BNR: the previous block's ReLU mask words of this lane's rows
    bool has_add = false;
    unsigned char* const sP = smem + 2 * A_BYTES + 4 * E_WAVE + wave * P_WAVE;
    unsigned char* const sY = smem + 2 * A_BYTES + 4 * E_WAVE + 4 * P_WAVE + ((EXT == 1 ? BUF * 4 : 0) + wave) * Y_WAVE;
    if constexpr (ADD) {
      has_add = true;
      constexpr int LPRA = EROW / 16, RPIA = 64 / LPRA;      // lanes per patch row, rows per instruction
#pragma unroll
      for (int q = 0; q < P_WAVE / 1024; ++q) {
        const int row = q * RPIA + lane / LPRA, pos = lane % LPRA;
        const int chunk = pos ^ (row & (LPRA - 1));
        const int m = mw + row;
        const bf16_t* src = zero;
        if (m < p.M) {
          if (p.sub2_h > 0) {     // addend on the even pixel grid only (icamd_conv2d_dgrad_sub2)
            const unsigned int n = fdiv((unsigned)m, p.divHW);
            const unsigned int rem = m - n * (p.sub2_h * p.sub2_w);
            const unsigned int hh = fdiv(rem, p.divW);
            const unsigned int ww = rem - hh * p.sub2_w;
            if (((hh | ww) & 1u) == 0u)
              src = p.addend + ((((long long)n * ((p.sub2_h + 1) >> 1) + (hh >> 1)) * ((p.sub2_w + 1) >> 1) + (ww >> 1)) * p.N +
                                n0 + chunk * 8);
          } else {
            src = p.addend + ((long long)m * p.N + n0 + chunk * 8);
          }
        }
        pw_glds16(src, sP + q * 1024, nt_patch);
      }
      if constexpr (BNR) {
#pragma unroll
        for (int q = 0; q < Y_WAVE / 1024; ++q) {
          const int row = q * RPIA + lane / LPRA, pos = lane % LPRA;
          const int chunk = pos ^ (row & (LPRA - 1));
          const int m = mw + row;
          const bf16_t* src = m < p.M ? p.bn_y + ((long long)m * p.N + n0 + chunk * 8) : zero;
          pw_glds16(src, sY + q * 1024, nt_patch);
        }
#pragma unroll
        for (int i = 0; i < MF; ++i) {
          const int m = mw + i * 16 + fr;
          pbw[i] = *(const unsigned long long*)(p.bn_bits + (((long long)(m < p.M ? m : 0) * p.N + n0) >> 3));
        }
      }
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        abw[i] = ~0ull;
        if (p.addend_bits != nullptr) {
          const int m = mw + i * 16 + fr;
          const unsigned char* bp = p.addend_bits + (((long long)(m < p.M ? m : 0) * p.N + n0) >> 3);
          if constexpr (NF == 4) abw[i] = *(const unsigned long long*)bp;
          else abw[i] = *(const unsigned int*)bp;
        }
      }
    }
    static_for<0, KS>([&](auto ksc) {
      constexpr int ks = decltype(ksc)::value;
      bf16x8 xf[MF];
      static_for<0, MF>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        xf[i] = lds_read128_off<BUF * A_BYTES + i * 16 * ROWB + (ks >> 2) * 256>(ra[ks & (NV - 1)]);
      });
      static_for<0, MF>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(xf[i]) : "n"(MF - 1 - i) : "memory");
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][j], xf[i], acc[j][i], 0, 0, 0);
      });
    });

    // ---- epilogue, wave-private, two row fragments at a time through the wave's LDS patches
    if constexpr (ADD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's addend patch (and bit words) landed
    static_for<0, (MF + 1) / 2>([&](auto hc) {
      constexpr int h = decltype(hc)::value;
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = h * 2 + ii;
        if (i < MF) {
#pragma unroll
          for (int j = 0; j < NF; ++j) {
            f32x4 v = acc[j][i];
            if constexpr (EXT || EPI) v += bias4[j];
            if constexpr (ADD) {
              if (has_add) {
                constexpr int LPRA = EROW / 16;
                const int prow = i * 16 + fr, aslot = j * 4 + fq;
                const u32x2 a = *(const u32x2*)(sP + prow * EROW + ((((aslot >> 1) ^ prow) & (LPRA - 1)) << 4) + ((aslot & 1) << 3));
                const unsigned int bb = (unsigned int)(abw[i] >> (8 * (j * 2 + (fq >> 1)) + 4 * (fq & 1))) & 0xfu;
                v[0] += (bb & 1u) ? bf16_lo(a[0]) : 0.f;
                v[1] += (bb & 2u) ? bf16_hi(a[0]) : 0.f;
                v[2] += (bb & 4u) ? bf16_lo(a[1]) : 0.f;
                v[3] += (bb & 8u) ? bf16_hi(a[1]) : 0.f;
              }
            }
            if constexpr (EPI) {
              if (p.relu) {   // as conv_igemm's epilogue and torch.relu: NaN stays NaN
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (v[e] < 0.f) ? 0.f : v[e];
              }
            }
            if constexpr (BNR) {   // g = d(block output) * [block output > 0]
              const unsigned int pb = (unsigned int)(pbw[i] >> (8 * (j * 2 + (fq >> 1)) + 4 * (fq & 1))) & 0xfu;
              v[0] = (pb & 1u) ? v[0] : 0.f;
              v[1] = (pb & 2u) ? v[1] : 0.f;
              v[2] = (pb & 4u) ? v[2] : 0.f;
              v[3] = (pb & 8u) ? v[3] : 0.f;
            }
            u32x2 pk;
            pk[0] = pack_bf16x2(v[0], v[1]);
            pk[1] = pack_bf16x2(v[2], v[3]);
            const int slot = j * 4 + fq;          // 8 B slot of the row; 16 B chunk = slot >> 1
            const int ch = NF == 4 ? (((slot >> 1) ^ fr) & 7) : (((slot >> 1) ^ (fr >> 2)) & 3);
            *(u32x2*)(sE + ii * 16 * EROW + fr * EROW + (ch << 4) + ((slot & 1) << 3)) = pk;
          }
        }
      }
      constexpr int LPR = EROW / 16;              // lanes per row: 8 or 4
      constexpr int RPR = 64 / LPR;               // rows per read instruction: 8 or 16
#pragma unroll
      for (int r = 0; r < 32 / RPR; ++r) {
        const int prow = r * RPR + lane / LPR;    // row of the two-patch pair, 0..31
        const int i = h * 2 + (prow >> 4);
        if (i < MF) {
          const int rr = prow & 15, c = lane % LPR;
          const int ch = NF == 4 ? ((c ^ rr) & 7) : ((c ^ (rr >> 2)) & 3);
          u32x4 o = *(const u32x4*)(sE + prow * EROW + (ch << 4));
          const int m = mw + i * 16 + rr;
          if (m < m_end) {
            if constexpr (EXT) {
              const long long off = (long long)m * p.N + n0 + c * 8;
              if constexpr (EXT == 1) {
                if (p.gelu_z != nullptr) {
                  const int pr = i * 16 + rr;                // row of the wave's z patch; chunk c sits at c ^ (row & (LPR - 1))
                  o = gelu_bwd8(o, *(const u32x4*)(sY + pr * EROW + (((c ^ pr) & (LPR - 1)) << 4)));
                }
              }
              if (p.gelu_inplace) o = gelu8(o);
              if (p.gelu_out != nullptr) *(u32x4*)(p.gelu_out + off) = gelu8(o);
            }
            *(u32x4*)(p.out + (long long)m * p.N + n0 + c * 8) = o;
            if constexpr (BNR) {
              const int pr = i * 16 + rr;                  // row of the wave's y patch; chunk c sits at c ^ (row & (LPR - 1))
              const u32x4 yv = *(const u32x4*)(sY + pr * EROW + (((c ^ pr) & (LPR - 1)) << 4));
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const f32x2 g = {bf16_lo(o[e]), bf16_hi(o[e])};
                const f32x2 yy = {bf16_lo(yv[e]), bf16_hi(yv[e])};
                s1[e] += g;                                  // sum g and sum g * y: the finalize turns them into sum g * xhat
                s2[e] = __builtin_elementwise_fma(g, yy, s2[e]);
              }
            }
            if constexpr (!ADD) {
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
    });
  };

  // The tile's staging loads are older than the previous tile's row stores in the in-order vmcnt queue: leave exactly those
  // stores in flight.  A forward that writes z AND gelu(z) (EXT, round 4) issues 2 * STORES of them -- with the single count it
  // waited for half of its own stores to reach memory before every tile.
  const bool two_outputs = EXT && p.gelu_out != nullptr;
  auto wait_tile_loads = [&]() {
    if (two_outputs) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * STORES) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STORES) : "memory");
  };
  for (; m0 < m_end; m0 += 2 * TM) {
    // this tile has landed for every wave and every wave is done with the other buffer; behind the tile's loads in the
    // queue: the previous tile's STORES row stores (full tiles always issue all of them; a ragged tile is a range's last)
    wait_tile_loads();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (m0 + TM < m_end) { stage(m0 + TM, 1); stage_z(m0 + TM, 1); }
    tile(std::integral_constant<int, 0>{}, m0);
    if (m0 + TM >= m_end) break;
    wait_tile_loads();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (m0 + 2 * TM < m_end) { stage(m0 + 2 * TM, 0); stage_z(m0 + 2 * TM, 0); }
    tile(std::integral_constant<int, 1>{}, m0 + TM);
  }

  if (want_stats) {
    // one partial row per workgroup (row `split` of the [ceil(M/128)] table); rows no workgroup owns are zero
    __syncthreads();
    constexpr int LPR = EROW / 16, G = 64 / LPR;   // lane groups per wave that share a channel group
    float* red = (float*)smem;                      // [WM * G][2][WN * CW]
    const int c = lane % LPR, g = wm * G + lane / LPR;
    constexpr int COLS = WN * CW;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[(g * 2 + 0) * COLS + wn * CW + c * 8 + 2 * e] = s1[e][0];
      red[(g * 2 + 0) * COLS + wn * CW + c * 8 + 2 * e + 1] = s1[e][1];
      red[(g * 2 + 1) * COLS + wn * CW + c * 8 + 2 * e] = s2[e][0];
      red[(g * 2 + 1) * COLS + wn * CW + c * 8 + 2 * e + 1] = s2[e][1];
    }
    __syncthreads();
    const int nrows = (p.M + 127) / 128, S = gridDim.x / p.ntiles_n;
    for (int idx = tid; idx < 2 * COLS; idx += 256) {
      const int which = idx / COLS, cc = idx - which * COLS;
      float s = 0.f;
#pragma unroll 4
      for (int k = 0; k < WM * G; ++k) s += red[(k * 2 + which) * COLS + cc];
      const int col = tile_n * COLS + cc;
      float* const table = BNR ? p.bn_part : p.stats;
      table[((long long)split * 2 + which) * p.N + col] = s;
      for (int r = split + S; r < nrows; r += S) table[((long long)r * 2 + which) * p.N + col] = 0.f;
    }
  }
}

int mode() {
  static const int m = [] { const char* e = getenv("ICAMD_PW_RESIDENT"); return e ? atoi(e) : 1; }();
  return m;
}
int pw_nt() {   // non-temporal LDS-DMA for once-read streams (round 5): 1 = the activation tile of single-channel-tile launches, 2 (default)
                // = also the addend / BatchNorm-input / z patches of every launch (ResNet-50: 17.89 -> 17.87 ms with 1, 18.02-18.09 ->
                // 17.96 with 2 on another box; never worse); ICAMD_PW_NT=0: default cache policy
  static const int m = [] { const char* e = getenv("ICAMD_PW_NT"); return e ? atoi(e) : 2; }();
  return m;
}
int xcd_order() {   // ICAMD_PW_XCD=0: the consecutive numbering of rounds 2-3 (A/B runs); off when the device does not report 8 XCDs
  static const int m = [] { const char* e = getenv("ICAMD_PW_XCD"); return (e ? atoi(e) : 1) && icamd_num_xccs() == 8; }();
  return m;
}

struct Config { int ks, nf, mf, wn; };

// the instantiated shapes: (K, channels per workgroup)
bool pick(int N, int K, Config* c) {
  if (K == 64 && N % 256 == 0) { *c = {2, 4, 4, 4}; return true; }
  if (K == 64 && N == 64) { *c = {2, 4, 1, 1}; return true; }
  if (K == 128 && N % 256 == 0) { *c = {4, 4, 4, 4}; return true; }
  if (K == 256 && N % 256 == 0) { *c = {8, 4, 2, 4}; return true; }
  // K = 256, N = 64: 2 x 2 waves of 32 rows x 32 channels (64 filter VGPRs): data gradient of 64 -> 256 at 56x56 97 -> 90 us,
  // forward of 256 -> 64 unchanged.  (Four waves of 16 rows x 64 channels with the whole filter in each tied conv_igemm in
  // isolation and lost in-model.)
  if (K == 256 && N == 64) { *c = {8, 2, 2, 2}; return true; }
  if (K == 256 && N % 128 == 0) { *c = {8, 2, 4, 4}; return true; }
  if (K == 512 && N % 128 == 0) { *c = {16, 2, 2, 4}; return true; }
  return false;
}

template <int KS, int NF, int MF, int WN>
int launch(const PwResidentParams& p, int grid, hipStream_t stream) {
  if (p.bias != nullptr || p.relu) {   // inference epilogue (never with statistics)
    if (p.stats != nullptr) return ICAMD_ERR_UNSUPPORTED;
    if (p.addend != nullptr) {
      if constexpr (KS == 8 && NF == 2 && WN == 4) return ICAMD_ERR_UNSUPPORTED;
      else hipLaunchKernelGGL((conv1x1_resident_kernel<KS, NF, MF, WN, true, false, false, true>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    } else
      hipLaunchKernelGGL((conv1x1_resident_kernel<KS, NF, MF, WN, false, false, false, true>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    return icamd_launch_status();
  }
  if (p.addend != nullptr) {
    // (K = 256 with 32-channel waves: activation buffers + addend patches exceed the 80 KB of two workgroups per CU)
    if constexpr (KS == 8 && NF == 2 && WN == 4) return ICAMD_ERR_UNSUPPORTED;
    else hipLaunchKernelGGL((conv1x1_resident_kernel<KS, NF, MF, WN, true>), dim3((unsigned)grid), dim3(256), 0, stream, p);
  } else {
    hipLaunchKernelGGL((conv1x1_resident_kernel<KS, NF, MF, WN, false>), dim3((unsigned)grid), dim3(256), 0, stream, p);
  }
  return icamd_launch_status();
}

}  // namespace

// The residual data gradients of ResNet-50's identity blocks: (K, N) = (planes, 4 * planes) for layer1..layer3.  Half the row
// fragments per tile of the plain kernel (MF = 2), so that the second patch fits the 80 KB of two workgroups per CU; layer4
// (K = 512: 64 KB of activation buffers) does not fit and keeps the two-pass BatchNorm backward.
bool icamd_pw_resident_bnred_wanted(long long M, int N, int K) {
  static const int on = [] { const char* e = getenv("ICAMD_BNRED"); return e ? atoi(e) : 1; }();
  if (!on || mode() == 0 || M < 8192 || M >= (1ll << 30) || N % 256 != 0) return false;
  return K == 64 || K == 128 || K == 256;
}

int icamd_pw_resident_bnred_launch(PwResidentParams& p, hipStream_t stream) {
  if (!icamd_pw_resident_bnred_wanted(p.M, p.N, p.K) || p.addend == nullptr || p.stats != nullptr ||
      (p.sub2_h > 0 && p.addend_bits != nullptr) ||
      p.bn_y == nullptr || p.bn_bits == nullptr || p.bn_part == nullptr)
    return ICAMD_ERR_UNSUPPORTED;
  constexpr int tm = 32;                         // WM = 1, MF = 2
  p.ntiles_n = p.N / 256;
  const int wgs = 2 * icamd_num_cus();
  int S = (wgs + p.ntiles_n - 1) / p.ntiles_n;
  const int cap_tiles = (p.M + tm - 1) / tm, cap_rows = (p.M + 127) / 128;
  if (S > cap_tiles) S = cap_tiles;
  if (S > cap_rows) S = cap_rows;                // one partial row per split in the [ceil(M/128)] table
  if (S < 1) S = 1;
  int rows = (p.M + S - 1) / S;
  rows = (rows + tm - 1) / tm * tm;
  p.rows_per_split = rows;
  S = (p.M + rows - 1) / rows;
  if (p.sub2_h > 0) {
    p.divHW = make_fastdiv((unsigned)(p.sub2_h * p.sub2_w));
    p.divW = make_fastdiv((unsigned)p.sub2_w);
  }
  p.xcd_groups = xcd_order();
  p.nt_loads = pw_nt() ? ((p.ntiles_n == 1 ? 1 : 0) | (pw_nt() >= 2 ? 2 : 0)) : 0;
  const dim3 grid((unsigned)(S * p.ntiles_n)), block(256);
  if (p.K == 64) hipLaunchKernelGGL((conv1x1_resident_kernel<2, 4, 2, 4, true, false, true>), grid, block, 0, stream, p);
  else if (p.K == 128) hipLaunchKernelGGL((conv1x1_resident_kernel<4, 4, 2, 4, true, false, true>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((conv1x1_resident_kernel<8, 4, 2, 4, true, false, true>), grid, block, 0, stream, p);
  return icamd_launch_status();
}

// ConvNeXt-T's dim-96 Linear layers (96 -> 384 forward, and the data gradient of 384 -> 96): K = 96 is not a multiple of the
// 64-wide stages of conv_igemm / gemm_nt and ran on conv_igemm's general-channel path at 391 us per launch at batch 256
// (M = 802 816; 0.77-1.4 GB of HBM traffic: 120-220 us).  Here it is four 32-wide k-steps over 256 B staged rows.
bool icamd_pw_resident_ext_wanted(long long M, int N, int K) {
  if (mode() == 0 || M < 8192 || M >= (1ll << 30)) return false;
  if (K == 96 && N % 128 == 0) return true;
  // round 4: K = 192 (ConvNeXt-T stage 1: 192 -> 768 forward + GELU, data gradient of 768 -> 192 + GELU') as eight k-steps over
  // 512 B staged rows, 256 channels per workgroup.  ICAMD_PW_EXT192=0: back on conv_igemm.
  static const int k192 = [] { const char* e = getenv("ICAMD_PW_EXT192"); return e ? atoi(e) : 1; }();
  return k192 && K == 192 && N % 256 == 0;
}

int icamd_pw_resident_ext_launch(PwResidentParams& p, hipStream_t stream) {
  if (!icamd_pw_resident_ext_wanted(p.M, p.N, p.K) || p.addend != nullptr || p.stats != nullptr) return ICAMD_ERR_UNSUPPORTED;
  const bool k96 = p.K == 96;
  // K = 96 forward forms (bias / GELU, no z to read): 128-row tiles.  After the latency fixes of round 4 these kernels are bound by
  // their per-tile chain (barrier -> LDS transposition -> GELU -> stores), so rows per barrier are what counts; the LDS the z patch
  // would take pays for the second half of the tile.  ICAMD_PW_EXT_TM128=0: 64-row tiles for every form.
  static const int tm128 = [] { const char* e = getenv("ICAMD_PW_EXT_TM128"); return e ? atoi(e) : 1; }();
  // (64-row tiles at THREE workgroups per CU -- 40 KB and 138 VGPRs each -- measured a wash in round 4 and are gone)
  const bool big = k96 && p.gelu_z == nullptr && tm128 == 1;
  p.lda = p.K; p.Ktrue = p.K; p.K = k96 ? 128 : 256;
  const int tm = k96 ? (big ? 128 : 64) : 32;    // <4, 2, 4|8, 4>: 64 | 128 rows x 128 channels per workgroup; <8, 4, 2, 4>: 32 x 256
  p.ntiles_n = p.N / (k96 ? 128 : 256);
  const int wgs = 2 * icamd_num_cus();
  int S = (wgs + p.ntiles_n - 1) / p.ntiles_n;
  const int cap_tiles = (p.M + tm - 1) / tm;
  if (S > cap_tiles) S = cap_tiles;
  if (S < 1) S = 1;
  int rows = (p.M + S - 1) / S;
  rows = (rows + tm - 1) / tm * tm;
  p.rows_per_split = rows;
  S = (p.M + rows - 1) / rows;
  p.xcd_groups = xcd_order();
  p.nt_loads = pw_nt() ? ((p.ntiles_n == 1 ? 1 : 0) | (pw_nt() >= 2 ? 2 : 0)) : 0;
  const dim3 grid((unsigned)(S * p.ntiles_n)), block(256);
  if (big) hipLaunchKernelGGL((conv1x1_resident_kernel<4, 2, 8, 4, false, 2>), grid, block, 0, stream, p);
  else if (k96) hipLaunchKernelGGL((conv1x1_resident_kernel<4, 2, 4, 4, false, 1>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((conv1x1_resident_kernel<8, 4, 2, 4, false, 1>), grid, block, 0, stream, p);
  return icamd_launch_status();
}

bool icamd_pw_resident_epi_wanted() {
  static const int on = [] { const char* e = getenv("ICAMD_PW_RESIDENT_EPI"); return e ? atoi(e) : 1; }();
  return on != 0 && mode() != 0;
}

bool icamd_pw_resident_wanted(long long M, int N, int K, bool with_addend) {
  Config c;
  if (mode() == 0 || M <= 0 || M >= (1ll << 30) || !pick(N, K, &c)) return false;
  if (with_addend && c.ks == 8 && c.nf == 2 && c.wn == 4) return false;
  // persistent workgroups need rows to amortise the filter load: at least ~8 tiles per workgroup at 512 workgroups
  const int tm = (4 / c.wn) * c.mf * 16;
  const int ntn = N / (c.wn * c.nf * 16);
  if (mode() == 2) return true;
  // K = 512 leaves room for 32 filter rows per wave only: measured on MI355X (batch 256) it wins on the 7x7 layers and on
  // 512 -> 256 at 28x28, ties at N = 128 and LOSES at N = 1024, M = 50176 (64 -> 83 us), so that one stays on conv_igemm
  if (K == 512 && !(M < 32768 || N == 256)) return false;
  // K = 1024, N = 256 with one 256 x 256 tile per CU: the big-tile GEMM is faster (gemm_nt.hip, icamd_gemm_nt_wanted)
  if (K == 1024 && N == 256 && !with_addend && icamd_gemm_nt_wanted(M, N, K)) return false;
  return M / tm * ntn >= 8 * 512;
}

int icamd_pw_resident_launch(PwResidentParams& p, hipStream_t stream) {
  Config c;
  if (!pick(p.N, p.K, &c) || (p.stats != nullptr && p.addend != nullptr)) return ICAMD_ERR_UNSUPPORTED;
  const int tm = (4 / c.wn) * c.mf * 16;
  p.ntiles_n = p.N / (c.wn * c.nf * 16);
  // two workgroups per CU; the statistics table has ceil(M/128) rows, one per split at most
  const int wgs = 2 * icamd_num_cus();
  int S = (wgs + p.ntiles_n - 1) / p.ntiles_n;
  const int cap_tiles = (p.M + tm - 1) / tm, cap_rows = (p.M + 127) / 128;
  if (S > cap_tiles) S = cap_tiles;
  if (p.stats != nullptr && S > cap_rows) S = cap_rows;
  if (S < 1) S = 1;
  int rows = (p.M + S - 1) / S;
  rows = (rows + tm - 1) / tm * tm;
  p.rows_per_split = rows;
  S = (p.M + rows - 1) / rows;
  if (p.sub2_h > 0) {
    p.divHW = make_fastdiv((unsigned)(p.sub2_h * p.sub2_w));
    p.divW = make_fastdiv((unsigned)p.sub2_w);
  }
  if (p.gat_ow > 0) {
    if (p.addend != nullptr) return ICAMD_ERR_UNSUPPORTED;
    p.gdivHW = make_fastdiv((unsigned)(p.gat_oh * p.gat_ow));
    p.gdivW = make_fastdiv((unsigned)p.gat_ow);
  }
  const int grid = S * p.ntiles_n;
  p.xcd_groups = xcd_order();
  p.nt_loads = pw_nt() ? ((p.ntiles_n == 1 ? 1 : 0) | (pw_nt() >= 2 ? 2 : 0)) : 0;
  if (c.ks == 2 && c.wn == 4) return launch<2, 4, 4, 4>(p, grid, stream);
  if (c.ks == 2) return launch<2, 4, 1, 1>(p, grid, stream);
  if (c.ks == 4) return launch<4, 4, 4, 4>(p, grid, stream);
  if (c.ks == 8 && c.wn == 2) return launch<8, 2, 2, 2>(p, grid, stream);
  if (c.ks == 8 && c.nf == 4) return launch<8, 4, 2, 4>(p, grid, stream);
  if (c.ks == 8) return launch<8, 2, 4, 4>(p, grid, stream);
  return launch<16, 2, 2, 4>(p, grid, stream);
}
