// Fused backward of "pointwise convolution -> BatchNorm" at the END of a ResNet bottleneck block (conv3 + bn3) for gfx950:
//   BatchNorm-backward apply   dy[m][co] = scale[co] * (g[m][co] - c1[co] - xhat[m][co] * c2[co])
//   data gradient              dx[m][ci] = sum_co dy[m][co] * w[co][ci]
//   weight gradient            dw[co][ci] = sum_m dy[m][co] * x[m][ci]
// in ONE pass over g and y (reference: the autograd backward of timm Bottleneck.conv3 / bn3 under loss.backward(),
// /root/reference/engine.py:64,72).
//
// Why (round 5): as three launches (bn_bwd_apply, conv1x1_resident data gradient, conv_wgrad) the 4*planes-wide tensors of this
// pair cross HBM five times -- g and y are read, dy is written, then read by the data gradient and again by the weight gradient:
// 10 B per element of the widest tensor of the block (profiles/r04_bench_n1.json: BatchNorm backward 4.8 ms + these two
// convolution classes 8.4 ms of a 19.5 ms step, all HBM-shaped).  Here dy exists only in LDS: 4 B per element.
//
// One persistent workgroup of 8 waves per CU walks a range of pixel rows in tiles of 32:
//   * LDS-DMA (global_load_lds, 16 B per lane) stages g[32][CO], y[32][CO] and x[32][64] of the tiles ahead into a ring of NBUF
//     buffers (CO = 256: four buffers, three tiles = 108 KB in flight per CU; CO = 512: two);  16 B chunks permuted on the
//     SOURCE side so that the one LDS image serves row reads (ds_read_b128, data gradient) and transposed reads
//     (ds_read_b64_tr_b16, weight gradient) without bank conflicts: chunk' = chunk ^ 2 * key(row), key = row[1:0] | row[3] << 2;
//   * all eight waves turn (g, y) into dy IN PLACE of g (8 channels per lane and vector, the per-channel constants of a lane never
//     change; the arithmetic is bn_bwd_apply_kernel's expression, so dy is the tensor the three-launch path would have stored);
//   * waves 0-3 = data gradient: each keeps its 32 filter rows of w_t for ALL of CO in registers (64 / 128 VGPRs, loaded once),
//     16 / 32 MFMAs per tile into 16 rows x 32 channels; the bf16 tile goes through a double-buffered LDS patch and leaves as
//     whole 128 B rows ONE BARRIER LATER (no extra synchronisation for the transposition);
//   * waves 4-7 = weight gradient: each owns 64 input channels x CO / 4 output channels of the filter gradient for the life of the
//     kernel (64 / 128 fp32 accumulator VGPRs), both operands by transposed LDS reads, 16 / 32 MFMAs per tile; at the end one fp32
//     slab per workgroup, folded in fixed order by slab_reduce_kernel (bitwise reproducible, as every weight gradient here);
//   * a wave waits for its OWN loads of the tile with a counted s_waitcnt vmcnt that leaves the younger tiles' loads and the
//     deferred row stores in flight; two s_barriers per tile.
// Input channels beyond 64 (planes = 128: CO = 512) are cut into 64-channel slices, one workgroup each; the slices of one row range
// are placed 8 block ids apart (same XCD, same time) so that g and y cross HBM once and the second slice reads them from L2.
#include "common.h"
#include "icamd_internal.h"
#include <cstdlib>

// Measurement builds only (tools/fused_ablate/): bit 0 = no BatchNorm arithmetic (dy = g), bit 1 = no data-gradient MFMAs, bit 2 = no
// weight-gradient MFMAs, bit 3 = no transform at all.  The product is always built with 0; results of the others are garbage.
#ifndef ICAMD_FUSED_ABLATE
#define ICAMD_FUSED_ABLATE 0
#endif

namespace {

constexpr int FTM = 32;          // rows per tile = one 32-deep MFMA step of the weight gradient
constexpr int FCI = 64;          // input channels per workgroup

__device__ __forceinline__ int dy_key(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }          // 32 B blocks of >= 512 B rows
__device__ __forceinline__ int x_key(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }    // 32 B blocks of 128 B rows

__device__ __forceinline__ u32x4 lds_load16(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void lds_store16(unsigned addr, const u32x4 v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_store8(unsigned addr, const u32x2 v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ bf16x8 tr_pair(unsigned a0, unsigned a1) {
  bf16x4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1) : "memory");
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// wait until at most n of this wave's vector-memory operations (LDS-DMA and stores, in issue order) are outstanding
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
  }
}

template <int CO>
__global__ __launch_bounds__(512, 2) void conv1x1_bn_bwd_fused_kernel(const FusedBwdParams p) {
  // A UNIT = 32 rows x 256 output channels of g and y (+ the rows' 64 input channels of x): CO = 256 has one unit per row tile,
  // CO = 512 two (h = 0, 1), staged, transformed and multiplied one after the other through the same ring of four 36 KB buffers --
  // three units (108 KB) in flight per CU for both widths.  (Round 5, first form of CO = 512: whole 32 x 512 tiles, two 68 KB buffers,
  // one tile in flight: 171 us for 0.51 GB = 3.0 TB/s, with the transform and the MFMA phase -- 35-40 us each by ablation,
  // tools/fused_ablate/ -- exposed behind the loads.)
  constexpr int NH = CO / 256;                    // units per row tile
  constexpr int NBUF = 4;
  constexpr int ROWB = 512;                       // bytes per staged g / y / dy row (256 channels)
  constexpr int G_BYTES = FTM * ROWB;             // 16 KB
  constexpr int X_BYTES = FTM * FCI * 2;          // 4 KB
  constexpr int BUF_BYTES = 2 * G_BYTES + X_BYTES;
  constexpr int PATCH_BYTES = FTM * FCI * 2;      // dx tile [32][64] bf16
  constexpr int NPATCH = NH == 1 ? 2 : 1;         // one unit per tile: the patch of tile t is stored while tile t + 1's is written
  constexpr int PATCH0 = NBUF * BUF_BYTES;
  constexpr int CPR = 32;                         // 16 B chunks per staged row
  constexpr int KSU = 8;                          // k-steps of the data gradient per unit
  constexpr int KS = CO / 32;                     // ... per row tile
  constexpr int CR = 4;                           // 16-channel fragments of dy per weight-gradient wave and unit
  // The lane's BatchNorm constants (8 channels per unit).  CO = 256: 40 VGPRs of a 176-register kernel.  CO = 512: an LDS table read
  // back per unit -- beside the 128 VGPRs of the resident filter / the accumulators the kernel spilled with them in registers.
  constexpr bool CREG = CO == 256;
  constexpr int CONST0 = PATCH0 + NPATCH * PATCH_BYTES;   // [5][CO] floats: mean, invstd, scale, c1, c2
  constexpr int CONST_BYTES = CREG ? 0 : 5 * CO * 4;
  static_assert(CONST0 + CONST_BYTES <= 160 * 1024, "LDS");
  static_assert(CO == 256 || CO == 512, "instantiated widths");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[CONST0 + CONST_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const unsigned lds_base = (unsigned)(uintptr_t)LPTR(smem);

  // workgroup -> (row range, input-channel slice): the slices of one range 8 block ids apart (one XCD under round-robin dispatch)
  int split, slice;
  {
    const int ns = p.nslices, b = (int)blockIdx.x;
    if (p.xcd_pairs) { slice = (b >> 3) % ns; split = ((b >> 3) / ns) * 8 + (b & 7); }
    else { slice = b % ns; split = b / ns; }
  }
  const int ci0 = slice * FCI;
  const int m_begin = split * p.rows_per_split;
  const int m_end = (p.M < m_begin + p.rows_per_split) ? p.M : m_begin + p.rows_per_split;
  const int ntiles = (m_end - m_begin + FTM - 1) / FTM;
  const int nunits = ntiles * NH;
  const bf16_t* zero = (const bf16_t*)icamd_zero_page;

  // ---- staging of unit u = tile * NH + h: instruction q = j * 8 + wave of the g / y part covers LDS bytes [q * 1024, + 1024) = two
  // rows; the lane's row and source chunk are recomputed per call (a handful of VALU operations per 1 KiB instruction) rather than
  // held in registers across the loop.  x travels with EVERY unit of its tile (the buffer of unit h = 0 is refilled while h = 1
  // runs; the second read of the 4 KB comes from L2).
  auto stage = [&](int u, int buf) {
    const int t = u / NH, h = u - t * NH;
    const int m0 = m_begin + t * FTM;
    unsigned char* base = smem + buf * BUF_BYTES;
    const bf16_t* gp = p.g + (long long)m0 * CO + h * 256;
    const bf16_t* yp = p.y + (long long)m0 * CO + h * 256;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int byte = (j * 8 + wave) * 1024 + lane * 16;
      const int row = byte / ROWB, pc = (byte % ROWB) >> 4;
      const int off = row * CO + ((pc ^ (dy_key(row) << 1)) << 3);
      const bool ok = m0 + row < m_end;
      if (p.nt) {   // once-read streams: non-temporal LDS-DMA (MI355X_MICROARCH.md "nt-weights"; never with two slices per range)
        __builtin_amdgcn_global_load_lds(GPTR(ok ? gp + off : zero), LPTR(base + (j * 8 + wave) * 1024), 16, 0, 2);
        __builtin_amdgcn_global_load_lds(GPTR(ok ? yp + off : zero), LPTR(base + G_BYTES + (j * 8 + wave) * 1024), 16, 0, 2);
      } else {
        __builtin_amdgcn_global_load_lds(GPTR(ok ? gp + off : zero), LPTR(base + (j * 8 + wave) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GPTR(ok ? yp + off : zero), LPTR(base + G_BYTES + (j * 8 + wave) * 1024), 16, 0, 0);
      }
    }
    if (wave < 4) {
      const int byte = wave * 1024 + lane * 16;
      const int row = byte / 128, pc = (byte % 128) >> 4;
      const int off = row * p.CI + ci0 + ((pc ^ (x_key(row) << 1)) << 3);
      const bool ok = m0 + row < m_end;
      __builtin_amdgcn_global_load_lds(GPTR(ok ? p.x + (long long)m0 * p.CI + off : zero), LPTR(base + 2 * G_BYTES + wave * 1024), 16, 0, 0);
    }
  };

  // prologue: the first NBUF - 1 units in flight under the constant / filter loads
#pragma unroll
  for (int k = 0; k < NBUF - 1; ++k)
    if (k < nunits) stage(k, k);

  // ---- BatchNorm constants [5][CO] into LDS (CO = 512)
  const int lc = tid % CPR, rbase = tid / CPR;   // this thread transforms chunk lc of rows rbase and rbase + 16
  for (int idx = tid; !CREG && idx < 5 * CO / 4; idx += 512) {
    const int arr = idx / (CO / 4), c4 = idx - arr * (CO / 4);
    const float* src = arr == 0 ? p.mean : arr == 1 ? p.invstd : arr == 2 ? p.scale : arr == 3 ? p.c1 : p.c2;
    const f32x4 v = *(const f32x4*)(src + c4 * 4);
    // table layout [array][channel half hh of the 8-channel chunk][chunk]: the 16 B a lane reads sit next to its neighbours' (the
    // first layout, [array][channel], had the lanes 32 B apart: two-way bank conflicts on 10 of the transform's 16 LDS instructions)
    lds_store16(lds_base + (unsigned)(CONST0 + ((arr * 2 + (c4 & 1)) * (CO / 8) + (c4 >> 1)) * 16), __builtin_bit_cast(u32x4, v));
  }
  const unsigned caddr = lds_base + (unsigned)(CONST0 + lc * 16);

  // the constants are on their way to LDS, the prologue's staging loads to their buffers: everything of this wave has arrived (so the
  // counted waits of the loop may only over-wait, never under-wait); the first barrier of the loop publishes the constant table
  __builtin_amdgcn_s_waitcnt(0x0F70);

  // deferred row store of the dx tile of tile t (waves 0-3): 8 rows x 128 B per wave
  auto store_dx = [&](int t) {
    const int row = 8 * wave + (lane >> 3), c = lane & 7;
    const unsigned a = lds_base + (unsigned)(PATCH0 + (NPATCH == 2 ? (t & 1) : 0) * PATCH_BYTES + row * 128 + ((c ^ (row & 7)) << 4));
    u32x4 v = lds_load16(a);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)::"memory");
    const int m = m_begin + t * FTM + row;
    if (m < m_end) *(u32x4*)(p.dx + (long long)m * p.CI + ci0 + c * 8) = v;
  };

  // ---- the unit loop, once per ROLE (wave-uniform branch around the whole loop, not inside it: with one loop and the roles'
  // register sets overlaid the compiler kept two copies of the overlay, 2 x 128 VGPRs at CO = 512).  Both instances execute the
  // same barriers.  DG = data gradient (waves 0-3), else weight gradient (waves 4-7).
  auto run = [&](auto role) {
    constexpr bool DG = decltype(role)::value;
    constexpr int L = 4 + (DG ? 1 : 0);           // LDS-DMA instructions of this wave per unit
    // data gradient: wr = row half, wc = channel half; filter fragment (ks, j) = rows ci0 + 32 wc + 16 j + fr of w_t, all of CO
    const int wr = wave & 1, wc = (wave >> 1) & 1;
    bf16x8 wf[DG ? KS : 1][2];
    // weight gradient: wq = quarter of each unit's 256 output channels; [64 input channels][64 output channels] per unit
    const int wq = wave & 3;
    f32x4 wacc[DG ? 1 : NH][4][CR];
    // addresses inside a buffer.  Data gradient: ad[v] = row read of the k-steps with ks & 3 == v.  Weight gradient: ad[0] / ad[1] = x
    // reads of rows r0 / r1, ad[2] / ad[3] = dy reads, each for fragment 0 with the row's swizzle key in the (otherwise zero) 32 B-block
    // bits: fragment f is address ^ (f << 5) -- one v_xor per read instead of an address register per fragment
    unsigned ad[4];
    if constexpr (DG) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          wf[ks][j] = *(const bf16x8*)(p.wt + (long long)(ci0 + 32 * wc + 16 * j + fr) * CO + ks * 32 + fq * 8);
      const int row = 16 * wr + fr;
      const int kx = dy_key(row) << 1;
#pragma unroll
      for (int v = 0; v < 4; ++v) ad[v] = (unsigned)(row * ROWB + (((fq ^ (kx & 3)) | (((v ^ (kx >> 2)) & 3) << 2)) << 4));
    } else {
      const int g4 = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
      const int r0 = 8 * g4 + q, r1 = r0 + 4;
      // x rows are 128 B (bits 5-6 = block); dy rows 512 B with this wave's four blocks at 4 wq (bits 5-6 free)
      ad[0] = (unsigned)(2 * G_BYTES + r0 * 128 + (x_key(r0) << 5) + 8 * pq);
      ad[1] = (unsigned)(2 * G_BYTES + r1 * 128 + (x_key(r1) << 5) + 8 * pq);
      ad[2] = (unsigned)(r0 * ROWB + (((wq * CR) ^ dy_key(r0)) << 5) + 8 * pq);
      ad[3] = (unsigned)(r1 * ROWB + (((wq * CR) ^ dy_key(r1)) << 5) + 8 * pq);
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < CR; ++j) wacc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 creg[CREG ? 5 : 1][2];
    if constexpr (CREG) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        creg[0][hh] = *(const f32x4*)(p.mean + lc * 8 + hh * 4);
        creg[1][hh] = *(const f32x4*)(p.invstd + lc * 8 + hh * 4);
        creg[2][hh] = *(const f32x4*)(p.scale + lc * 8 + hh * 4);
        creg[3][hh] = *(const f32x4*)(p.c1 + lc * 8 + hh * 4);
        creg[4][hh] = *(const f32x4*)(p.c2 + lc * 8 + hh * 4);
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);           // the filter / the constants

    f32x4 dacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

    // (g, y) -> dy in place of g for the unit in buffer `tb` (h = its half of CO): chunk lc of rows rbase and rbase + 16.  CO = 512:
    // the lane's constants come from the LDS table in two halves of four channels through the same 20 registers.
    auto transform = [&](auto hc, int tb) {
      constexpr int h = decltype(hc)::value;
      if constexpr ((ICAMD_FUSED_ABLATE & 8) == 0) {
        const unsigned bb = lds_base + (unsigned)(tb * BUF_BYTES);
        u32x4 gv[2], yv[2], ov[2];
        unsigned addr[2];
        f32x4 cq[5];
        if constexpr (!CREG) {
#pragma unroll
          for (int a = 0; a < 5; ++a) cq[a] = __builtin_bit_cast(f32x4, lds_load16(caddr + (unsigned)(((a * 2 + 0) * (CO / 8) + h * 32) * 16)));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = rbase + 16 * i;
          addr[i] = bb + (unsigned)(row * ROWB + ((lc ^ (dy_key(row) << 1)) << 4));
          gv[i] = lds_load16(addr[i]);
          yv[i] = lds_load16(addr[i] + G_BYTES);
        }
        if constexpr (CREG) {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(gv[0]), "+v"(yv[0]), "+v"(gv[1]), "+v"(yv[1])::"memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(gv[0]), "+v"(yv[0]), "+v"(gv[1]), "+v"(yv[1]), "+v"(cq[0]), "+v"(cq[1]), "+v"(cq[2]),
                       "+v"(cq[3]), "+v"(cq[4])::"memory");
        }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          if constexpr (CREG) {
#pragma unroll
            for (int a = 0; a < 5; ++a) cq[a] = creg[a][hh];
          } else if (hh == 1) {
#pragma unroll
            for (int a = 0; a < 5; ++a) cq[a] = __builtin_bit_cast(f32x4, lds_load16(caddr + (unsigned)(((a * 2 + 1) * (CO / 8) + h * 32) * 16)));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cq[0]), "+v"(cq[1]), "+v"(cq[2]), "+v"(cq[3]), "+v"(cq[4])::"memory");
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned gw = gv[i][hh * 2 + (e >> 1)], yw = yv[i][hh * 2 + (e >> 1)];
              const float g = (e & 1) ? bf16_hi(gw) : bf16_lo(gw);
              const float yy = (e & 1) ? bf16_hi(yw) : bf16_lo(yw);
              if constexpr (ICAMD_FUSED_ABLATE & 1) o[e] = g + yy;
              else o[e] = cq[2][e] * (g - cq[3][e] - ((yy - cq[0][e]) * cq[1][e]) * cq[4][e]);     // bn_bwd_apply_kernel's expression
            }
            ov[i][hh * 2] = pack_bf16x2(o[0], o[1]);
            ov[i][hh * 2 + 1] = pack_bf16x2(o[2], o[3]);
          }
        }
        lds_store16(addr[0], ov[0]);
        lds_store16(addr[1], ov[1]);
      }
    };

    // the role's MFMAs on the dy of unit (t, h) in buffer `mb`
    auto multiply = [&](auto hc, int t, int mb) {
      constexpr int h = decltype(hc)::value;
      const unsigned bb = lds_base + (unsigned)(mb * BUF_BYTES);
      if constexpr (DG) {
        // ---- data gradient: [16 rows][32 channels] per wave, this unit's 256 of the CO reduction channels, four row fragments of
        // dy in flight
        if constexpr (h == 0) { dacc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; dacc[1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        static_for<0, (ICAMD_FUSED_ABLATE & 2) ? 0 : KSU / 4>([&](auto gc) {
          constexpr int g0 = decltype(gc)::value * 4;
          bf16x8 a[4];
          static_for<0, 4>([&](auto kc) {
            constexpr int ks = g0 + decltype(kc)::value;
            a[ks & 3] = lds_read128_off<(ks >> 2) * 256>(bb + ad[ks & 3]);
          });
          static_for<0, 4>([&](auto kc) {
            constexpr int k = decltype(kc)::value, ks = g0 + k;
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a[k]) : "n"(3 - k) : "memory");
#pragma unroll
            for (int j = 0; j < 2; ++j) dacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[h * KSU + ks][j], a[k], dacc[j], 0, 0, 0);
          });
        });
        if constexpr (h == NH - 1) {
          // lane: channels 32 wc + 16 j + 4 fq .. + 3 of row 16 wr + fr -> 8 B slot (8 wc + 4 j + fq) of the row in the patch
          const int row = 16 * wr + fr;
          const unsigned pa = lds_base + (unsigned)(PATCH0 + (NPATCH == 2 ? (t & 1) : 0) * PATCH_BYTES + row * 128);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int slot = 8 * wc + 4 * j + fq;
            u32x2 pk;
            pk[0] = pack_bf16x2(dacc[j][0], dacc[j][1]);
            pk[1] = pack_bf16x2(dacc[j][2], dacc[j][3]);
            lds_store8(pa + (unsigned)((((slot >> 1) ^ (row & 7)) << 4) + ((slot & 1) << 3)), pk);
          }
        }
      } else {
        // ---- weight gradient, one 32-row step: [64 input channels][64 output channels of this unit] per wave
        bf16x8 xf[4], yf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[i] = tr_pair(bb + (ad[0] ^ (unsigned)(i << 5)), bb + (ad[1] ^ (unsigned)(i << 5)));
#pragma unroll
        for (int j = 0; j < 4; ++j) yf[j] = tr_pair(bb + (ad[2] ^ (unsigned)(j << 5)), bb + (ad[3] ^ (unsigned)(j << 5)));
        static_for<0, (ICAMD_FUSED_ABLATE & 4) ? 0 : 4>([&](auto jj) {
          constexpr int j = decltype(jj)::value;
          // reads return in issue order: dy fragment j is complete once at most 2 * (3 - j) reads are outstanding (the x fragments
          // were issued first)
          asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(yf[j]), "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]) : "n"(2 * (3 - j)) : "memory");
#pragma unroll
          for (int i = 0; i < 4; ++i) wacc[h][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], yf[j], wacc[h][i][j], 0, 0, 0);
        });
      }
    };

    // ---- software pipeline: iteration u multiplies the dy of unit u while unit u + 1 is transformed; ONE barrier per unit publishes
    // both (dy of unit u from the previous iteration's transform, the landed loads of unit u + 1) and frees the buffer of unit u - 1
    // for the loads of unit u + 3.  The two roles take the two halves of an iteration in OPPOSITE order -- every SIMD hosts one wave
    // of each role, so one wave's MFMAs and transposed reads run beside the other's BatchNorm arithmetic instead of both queueing for
    // the same pipe (first form: transform, barrier, multiply in lockstep -- CO = 512 ran 1.7 us per unit against 1.25 us of loads).
    if (nunits > 0) {
      wait_vmcnt_dyn(((nunits - 1 < NBUF - 2) ? nunits - 1 : NBUF - 2) * L);     // unit 0 (the prologue's units 1, 2 stay in flight)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      transform(std::integral_constant<int, 0>{}, 0);
    }
    int buf = 0;                           // buffer of unit u
    for (int t = 0; t < ntiles; ++t) {
      static_for<0, NH>([&](auto hc) {
        constexpr int h = decltype(hc)::value;
        constexpr int hn = (h + 1) % NH;   // half of unit u + 1
        const int u = t * NH + h;
        const int nb = buf == NBUF - 1 ? 0 : buf + 1;
        // ---- this wave's loads of unit u + 1 have landed.  Younger in its queue: the loads of unit u + 2 (one group of L) and, data
        // gradient, the row stores issued since the loads of unit u + 1 were, i.e. in iterations u - 2 and u - 1 (iteration j issues
        // one iff it is the first unit of a tile other than the first: j >= NH and j % NH == 0; sb(j) = such iterations <= j)
        if (u + 1 < nunits) {
          const int nl = u + 2 < nunits ? 1 : 0;
          int ns = 0;
          if constexpr (DG) {
            const int j1 = u - 1, j0 = u - 3;
            ns = (j1 < 0 ? 0 : j1 / NH) - (j0 < 0 ? 0 : j0 / NH);
          }
          wait_vmcnt_dyn(nl * L + ns);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (u + NBUF - 1 < nunits) stage(u + NBUF - 1, buf == 0 ? NBUF - 1 : buf - 1);
        if constexpr (DG) {
          if constexpr (h == 0) {
            if (t > 0) store_dx(t - 1);
          }
          multiply(hc, t, buf);
          if (u + 1 < nunits) transform(std::integral_constant<int, hn>{}, nb);
        } else {
          if (u + 1 < nunits) transform(std::integral_constant<int, hn>{}, nb);
          multiply(hc, t, buf);
        }
        buf = nb;
      });
    }

    // ---- tail: the last dx tile, the filter-gradient slab
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (DG) {
      if (ntiles > 0) store_dx(ntiles - 1);
    } else {
      // wacc[h][i][j]: rows = input channels 16 i + 4 (lane >> 4) .. + 3, column = output channel 256 h + (4 wq + j) 16 + (lane & 15)
      float* slab = p.slab + (long long)split * CO * p.CI;
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < CR; ++j) {
            const int co = h * 256 + (wq * CR + j) * 16 + (lane & 15);
            const int ci = ci0 + 16 * i + 4 * (lane >> 4);
            *(f32x4*)(slab + (long long)co * p.CI + ci) = wacc[h][i][j];
          }
    }
  };
  if (wave < 4) run(std::true_type{});
  else run(std::false_type{});
}

int fused_mode() {
  static const int m = [] { const char* e = getenv("ICAMD_FUSED_CONV_BN_BWD"); return e ? atoi(e) : 1; }();
  return m;
}

}  // namespace

// (Cin, Cout) = (64, 256) and (128, 512): conv3 of ResNet-50's layer1 / layer2 bottlenecks (the widest tensors of the network;
// at Cout = 1024 the g / y tiles of 32 rows no longer fit LDS beside a ring).  ICAMD_FUSED_CONV_BN_BWD=0 switches it off (A/B),
// 2 lifts the size floor (tests).
bool icamd_conv1x1_bn_bwd_fused_wanted(long long M, int Cin, int Cout) {
  if (fused_mode() == 0 || M <= 0 || M >= (1ll << 30)) return false;
  if (!((Cin == 64 && Cout == 256) || (Cin == 128 && Cout == 512))) return false;
  return fused_mode() == 2 || M >= 16384;
}

void icamd_conv1x1_bn_bwd_fused_plan(int M, int Cin, int* S, int* rows_per_split) {
  const int nslices = Cin / FCI;
  int s = icamd_num_cus() / nslices;
  const int cap = (M + FTM - 1) / FTM;
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  int rows = (M + s - 1) / s;
  rows = (rows + FTM - 1) / FTM * FTM;
  *rows_per_split = rows;
  *S = (M + rows - 1) / rows;
}

int icamd_conv1x1_bn_bwd_fused_launch(FusedBwdParams& p, hipStream_t stream) {
  if (!icamd_conv1x1_bn_bwd_fused_wanted(p.M, p.CI, p.CO)) return ICAMD_ERR_UNSUPPORTED;
  p.nslices = p.CI / FCI;
  icamd_conv1x1_bn_bwd_fused_plan(p.M, p.CI, &p.S, &p.rows_per_split);
  p.xcd_pairs = (p.S % 8 == 0 && icamd_num_xccs() == 8) ? 1 : 0;
  // g, y and x are read once when one workgroup owns a row range: non-temporal LDS-DMA (64 -> 256 at 56 x 56: 217 -> 180 us =
  // 5.7 TB/s; with two input-channel slices the second one reads them from L2 and the default policy stays).  ICAMD_FUSED_NT=0: off.
  static const int nt = [] { const char* e = getenv("ICAMD_FUSED_NT"); return e ? atoi(e) : 1; }();
  p.nt = (nt && p.nslices == 1) ? 1 : 0;
  const dim3 grid((unsigned)(p.S * p.nslices)), block(512);
  if (p.CO == 256) hipLaunchKernelGGL((conv1x1_bn_bwd_fused_kernel<256>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((conv1x1_bn_bwd_fused_kernel<512>), grid, block, 0, stream, p);
  return icamd_launch_status();
}
