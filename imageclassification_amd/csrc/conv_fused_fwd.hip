// Fused forward across a ResNet bottleneck boundary for gfx950: the END of block b and the START of block b + 1 in one pass,
//   out[m][k]  = relu(y3[m][k] * scale[k] + shift[k] + residual[m][k])        (bn3-apply + shortcut + ReLU of block b, + 1-bit mask)
//   y1[m][n]   = sum_k out[m][k] * w1[n][k]                                   (conv1 of block b + 1, 1x1 / stride 1)
//   stats[n]   = per-channel sum / sum of squares of the rounded y1             (statistics of block b + 1's bn1)
// (reference: timm Bottleneck.forward -- bn3, shortcut add, act3 of one block and conv1, bn1 of the next -- under model(samples),
// /root/reference/engine.py:48,51).
//
// Why (round 5): as two launches (icamd_bn_apply + icamd_conv2d_fwd) the 4*planes-wide block output is written by the apply pass
// and read again by the next block's first convolution: 8.1 B per element of the widest tensor of the block.  Here the convolution
// takes the tile from LDS while it is on its way out: 6.1 B per element (y3, residual read; out written; y1 is planes-wide).
//
// Structure = conv_fused_bwd.hip's: one persistent 8-wave workgroup per CU walks its rows in UNITS of 32 rows x 256 channels of y3 and
// the residual through a ring of four 32 KB LDS buffers (LDS-DMA, three units in flight), all waves turn a unit into `out` IN PLACE of
// y3 (icamd_bn_apply's own expression, 8 channels per lane and vector: `out` and the mask bits are the bytes the apply kernel stores,
// and they leave for HBM straight from the registers that computed them), then the unit is the A operand of the convolution: every wave
// keeps its slice of w1 for ALL of K in registers (32 .. 128 VGPRs) and reads out rows with ds_read_b128 -- 16 B chunks permuted on the
// DMA source side by chunk ^ 2 * key(row) so those reads are conflict-free.  One barrier per unit (multiply unit u / transform unit
// u + 1, the two wave groups in opposite order so that each SIMD has one wave on the MFMA pipe and one on the VALU).  The y1 tile goes
// through an LDS patch and leaves as whole rows one barrier later; the lanes of that store pass keep the BatchNorm partial sums in
// registers across tiles (one partial row per workgroup, icamd_conv2d_fwd's table layout).
#include "common.h"
#include "icamd_internal.h"
#include <cstdlib>

namespace {

constexpr int GTM = 32;          // rows per tile

__device__ __forceinline__ int f_key(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

__device__ __forceinline__ u32x4 f_lds_load16(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void f_lds_store16(unsigned addr, const u32x4 v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void f_lds_store8(unsigned addr, const u32x2 v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// wait until at most n of this wave's vector-memory operations (LDS-DMA and stores, in issue order) are outstanding
__device__ __forceinline__ void f_wait_vmcnt(int n) {
  switch (n < 0 ? 0 : (n > 20 ? 20 : n)) {
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
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
  }
}

template <int K, int N>
__global__ __launch_bounds__(512, 2) void bn_apply_conv1x1_fused_kernel(const FusedFwdParams p) {
  constexpr int NH = K / 256;                     // units per row tile
  constexpr int NBUF = 4;
  constexpr int ROWB = 512;                       // bytes per staged row (256 channels)
  constexpr int T_BYTES = GTM * ROWB;             // 16 KB: one y3 (or residual) tile
  constexpr int BUF_BYTES = 2 * T_BYTES;
  constexpr int WC = N == 64 ? 4 : 8;             // waves across the output channels (N = 128: 16 channels per wave, 32 rows:
                                                  // K = 512 with 32 channels per wave kept 128 filter VGPRs and spilled)
  constexpr int WR = 8 / WC;                      // ... across the rows
  constexpr int RF = GTM / WR / 16;               // 16-row fragments per wave
  constexpr int NFW = N / WC / 16;                // 16-channel fragments per wave
  constexpr int KSU = 8, KS = K / 32;
  constexpr int PROWB = N * 2;                    // bytes per y1 patch row
  constexpr int PATCH_BYTES = GTM * PROWB;
  constexpr int NPATCH = NH == 1 ? 2 : 1;
  constexpr int PATCH0 = NBUF * BUF_BYTES;
  constexpr int CONST0 = PATCH0 + NPATCH * PATCH_BYTES;    // [4 arrays][2 channel halves][K / 8 chunks] float4
  constexpr int CONST_BYTES = 4 * K * 4;
  constexpr int CPN = N / 8;                      // 16 B chunks per y1 row
  constexpr int RPI = 64 / CPN;                   // y1 rows per store instruction
  constexpr int YI = PATCH_BYTES / 1024;          // store instructions per y1 tile (4 / 8 / 16)
  static_assert(CONST0 + CONST_BYTES <= 160 * 1024, "LDS");
  static_assert((K == 256 || K == 512) && (N == 64 || N == 128) && RF * WR * 16 == GTM && NFW >= 1, "shapes");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[CONST0 + CONST_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const unsigned lds_base = (unsigned)(uintptr_t)LPTR(smem);
  const int split = (int)blockIdx.x;
  const int m_begin = split * p.rows_per_split;
  const int m_end = (p.M < m_begin + p.rows_per_split) ? p.M : m_begin + p.rows_per_split;
  const int ntiles = (m_end - m_begin + GTM - 1) / GTM;
  const int nunits = ntiles * NH;
  const bf16_t* zero = (const bf16_t*)icamd_zero_page;

  // ---- staging of unit u = tile * NH + h: instruction q = j * 8 + wave covers LDS bytes [q * 1024, + 1024) = two rows of a tile
  auto stage = [&](int u, int buf) {
    const int t = u / NH, h = u - t * NH;
    const int m0 = m_begin + t * GTM;
    unsigned char* base = smem + buf * BUF_BYTES;
    const bf16_t* yp = p.y + (long long)m0 * K + h * 256;
    const bf16_t* rp = p.res + (long long)m0 * K + h * 256;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int byte = (j * 8 + wave) * 1024 + lane * 16;
      const int row = byte / ROWB, pc = (byte % ROWB) >> 4;
      const int off = row * K + ((pc ^ (f_key(row) << 1)) << 3);
      const bool ok = m0 + row < m_end;
      if (p.nt) {   // once-read streams: non-temporal LDS-DMA (MI355X_MICROARCH.md "nt-weights")
        __builtin_amdgcn_global_load_lds(GPTR(ok ? yp + off : zero), LPTR(base + (j * 8 + wave) * 1024), 16, 0, 2);
        __builtin_amdgcn_global_load_lds(GPTR(ok ? rp + off : zero), LPTR(base + T_BYTES + (j * 8 + wave) * 1024), 16, 0, 2);
      } else {
        __builtin_amdgcn_global_load_lds(GPTR(ok ? yp + off : zero), LPTR(base + (j * 8 + wave) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GPTR(ok ? rp + off : zero), LPTR(base + T_BYTES + (j * 8 + wave) * 1024), 16, 0, 0);
      }
    }
  };
  constexpr int L = 4;                            // LDS-DMA instructions per wave and unit

#pragma unroll
  for (int k = 0; k < NBUF - 1; ++k)
    if (k < nunits) stage(k, k);

  // ---- BatchNorm constants into LDS: arrays scale, shift, res_scale, res_shift; entry (array, channel half hh of a chunk, chunk)
  const int lc = tid % 32, rbase = tid / 32;      // this thread transforms chunk lc of rows rbase and rbase + 16
  for (int idx = tid; idx < 4 * K / 4; idx += 512) {
    const int arr = idx / (K / 4), c4 = idx - arr * (K / 4);
    const float* src = arr == 0 ? p.scale : arr == 1 ? p.shift : arr == 2 ? p.res_scale : p.res_shift;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (src != nullptr) v = *(const f32x4*)(src + c4 * 4);
    f_lds_store16(lds_base + (unsigned)(CONST0 + ((arr * 2 + (c4 & 1)) * (K / 8) + (c4 >> 1)) * 16), __builtin_bit_cast(u32x4, v));
  }
  const unsigned caddr = lds_base + (unsigned)(CONST0 + lc * 16);
  const bool res_bn = p.res_scale != nullptr;     // the residual is a RAW shortcut convolution output: its BatchNorm applied here

  // ---- the filter: fragment (ks, j) = rows n = (wc * NFW + j) * 16 + fr of w1, 8 input channels at ks * 32 + fq * 8
  const int wc = wave % WC, wr = wave / WC;
  bf16x8 wf[KS][NFW];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int j = 0; j < NFW; ++j) wf[ks][j] = *(const bf16x8*)(p.w + (long long)((wc * NFW + j) * 16 + fr) * K + ks * 32 + fq * 8);
  // row-read addresses of the wave's first row fragment inside a buffer, by ks & 3 (fragment i: + 16 rows = immediate)
  unsigned ad[4];
  {
    const int row = 16 * (wr * RF) + fr;
    const int kx = f_key(row) << 1;
#pragma unroll
    for (int v = 0; v < 4; ++v) ad[v] = (unsigned)(row * ROWB + (((fq ^ (kx & 3)) | (((v ^ (kx >> 2)) & 3) << 2)) << 4));
  }
  // everything of this wave has arrived (the counted waits below may then only over-wait)
  __builtin_amdgcn_s_waitcnt(0x0F70);

  // BatchNorm statistics of the stored y1 rows: lane = chunk (lane % CPN) of the rows it stores
  f32x2 s1[4], s2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { s1[e] = f32x2{0.f, 0.f}; s2[e] = f32x2{0.f, 0.f}; }
  const bool want_stats = p.stats != nullptr;
  // y1 store instructions of this wave per tile: instruction qi = wave + 8 * k, k < SY
  constexpr int SYMAX = (YI + 7) / 8;
  const int SY = (YI >= 8) ? YI / 8 : (wave < YI ? 1 : 0);

  auto store_y1 = [&](int t) {
#pragma unroll
    for (int k = 0; k < SYMAX; ++k) {
      const int qi = wave + 8 * k;
      if (qi < YI) {
        const int row = qi * RPI + lane / CPN, c = lane % CPN;
        const unsigned a = lds_base + (unsigned)(PATCH0 + (NPATCH == 2 ? (t & 1) : 0) * PATCH_BYTES + row * PROWB + ((c ^ (row & 7)) << 4));
        u32x4 v = f_lds_load16(a);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)::"memory");
        const int m = m_begin + t * GTM + row;
        if (m < m_end) {
          *(u32x4*)(p.y1 + (long long)m * N + c * 8) = v;
          if (want_stats) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const f32x2 x = {bf16_lo(v[e]), bf16_hi(v[e])};
              s1[e] += x;
              s2[e] = __builtin_elementwise_fma(x, x, s2[e]);
            }
          }
        }
      }
    }
  };

  // (y3, residual) -> out in place of y3 for the unit in buffer tb (h = its half of K), tile t: chunk lc of rows rbase, rbase + 16;
  // out and its mask byte leave for HBM from the registers.  icamd_bn_apply's expression.
  auto transform = [&](auto hc, int t, int tb) {
    constexpr int h = decltype(hc)::value;
    const unsigned bb = lds_base + (unsigned)(tb * BUF_BYTES);
    u32x4 yv[2], rv[2], ov[2];
    unsigned addr[2], bits[2] = {0u, 0u};
    f32x4 cq[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) cq[a] = __builtin_bit_cast(f32x4, f_lds_load16(caddr + (unsigned)(((a * 2 + 0) * (K / 8) + h * 32) * 16)));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = rbase + 16 * i;
      addr[i] = bb + (unsigned)(row * ROWB + ((lc ^ (f_key(row) << 1)) << 4));
      yv[i] = f_lds_load16(addr[i]);
      rv[i] = f_lds_load16(addr[i] + T_BYTES);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(yv[0]), "+v"(rv[0]), "+v"(yv[1]), "+v"(rv[1]), "+v"(cq[0]), "+v"(cq[1]), "+v"(cq[2]), "+v"(cq[3])::"memory");
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      if (hh == 1) {
#pragma unroll
        for (int a = 0; a < 4; ++a) cq[a] = __builtin_bit_cast(f32x4, f_lds_load16(caddr + (unsigned)(((a * 2 + 1) * (K / 8) + h * 32) * 16)));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cq[0]), "+v"(cq[1]), "+v"(cq[2]), "+v"(cq[3])::"memory");
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float f[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned yw = yv[i][hh * 2 + (e >> 1)], rw = rv[i][hh * 2 + (e >> 1)];
          const float yy = (e & 1) ? bf16_hi(yw) : bf16_lo(yw);
          const float rr = (e & 1) ? bf16_hi(rw) : bf16_lo(rw);
          float v = fmaf(yy, cq[0][e], cq[1][e]);
          if (res_bn) v += bf16_to_f32(f32_to_bf16(fmaf(rr, cq[2][e], cq[3][e])));
          else v += rr;
          f[e] = (v < 0.f) ? 0.f : v;     // NaN stays NaN, as torch.relu
          bits[i] |= (f[e] > 0.f ? 1u : 0u) << (hh * 4 + e);
        }
        ov[i][hh * 2] = pack_bf16x2(f[0], f[1]);
        ov[i][hh * 2 + 1] = pack_bf16x2(f[2], f[3]);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f_lds_store16(addr[i], ov[i]);
      const int m = m_begin + t * GTM + rbase + 16 * i;
      if (m < m_end) {
        const long long off = (long long)m * K + h * 256 + lc * 8;
        *(u32x4*)(p.out + off) = ov[i];
        if (p.maskbits != nullptr) p.maskbits[off >> 3] = (unsigned char)bits[i];
      }
    }
  };

  f32x4 acc[NFW][RF];
  // the convolution's MFMAs on the `out` unit (t, h) in buffer mb
  auto multiply = [&](auto hc, int t, int mb) {
    constexpr int h = decltype(hc)::value;
    const unsigned bb = lds_base + (unsigned)(mb * BUF_BYTES);
    if constexpr (h == 0) {
#pragma unroll
      for (int j = 0; j < NFW; ++j)
#pragma unroll
        for (int i = 0; i < RF; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // row fragments of `out` KG k-steps at a time (KG * RF reads in flight; fewer with the 128-register filters of K = 512)
    constexpr int KG = (K == 512) ? 2 : 4;
    static_for<0, KSU / KG>([&](auto gc) {
      constexpr int g0 = decltype(gc)::value * KG;
      bf16x8 a[KG][RF];
      static_for<0, KG>([&](auto kc) {
        constexpr int k = decltype(kc)::value, ks = g0 + k;
        static_for<0, RF>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          a[k][i] = lds_read128_off<(ks >> 2) * 256 + i * 16 * ROWB>(bb + ad[ks & 3]);
        });
      });
      static_for<0, KG>([&](auto kc) {
        constexpr int k = decltype(kc)::value, ks = g0 + k;
        if constexpr (RF == 1) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a[k][0]) : "n"(KG - 1 - k) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a[k][0]), "+v"(a[k][RF - 1]) : "n"(RF * (KG - 1 - k)) : "memory");
#pragma unroll
        for (int j = 0; j < NFW; ++j)
#pragma unroll
          for (int i = 0; i < RF; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[h * KSU + ks][j], a[k][i], acc[j][i], 0, 0, 0);
      });
    });
    if constexpr (h == NH - 1) {
      // lane: channels (wc NFW + j) 16 + 4 fq .. + 3 of row (wr RF + i) 16 + fr -> 8 B slot of the row in the patch
#pragma unroll
      for (int i = 0; i < RF; ++i) {
        const int row = (wr * RF + i) * 16 + fr;
        const unsigned pa = lds_base + (unsigned)(PATCH0 + (NPATCH == 2 ? (t & 1) : 0) * PATCH_BYTES + row * PROWB);
#pragma unroll
        for (int j = 0; j < NFW; ++j) {
          const int slot = (wc * NFW + j) * 4 + fq;
          u32x2 pk;
          pk[0] = pack_bf16x2(acc[j][i][0], acc[j][i][1]);
          pk[1] = pack_bf16x2(acc[j][i][2], acc[j][i][3]);
          f_lds_store8(pa + (unsigned)((((slot >> 1) ^ (row & 7)) << 4) + ((slot & 1) << 3)), pk);
        }
      }
    }
  };

  // ---- software pipeline (conv_fused_bwd.hip): iteration u multiplies unit u while unit u + 1 is transformed, one barrier per unit
  if (nunits > 0) {
    f_wait_vmcnt(((nunits - 1 < NBUF - 2) ? nunits - 1 : NBUF - 2) * L);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    transform(std::integral_constant<int, 0>{}, 0, 0);
  }
  int buf = 0;
  for (int t = 0; t < ntiles; ++t) {
    static_for<0, NH>([&](auto hc) {
      constexpr int h = decltype(hc)::value;
      constexpr int hn = (h + 1) % NH;
      const int u = t * NH + h;
      const int nb = buf == NBUF - 1 ? 0 : buf + 1;
      // ---- this wave's loads of unit u + 1 have landed.  Younger in its queue: the loads of unit u + 2; the stores of the transforms
      // of units u - 1 and u (4 each: two `out` vectors, two mask bytes); the y1 row stores of iterations u - 2 and u - 1 (iteration j
      // issues SY iff j >= NH and j % NH == 0).  The last two tiles wait for everything: a ragged tile's store instructions may be
      // skipped by whole waves, and an over-estimate of what is in the queue would under-wait.
      if (u + 1 < nunits) {
        int n = 0;
        if (t + 2 < ntiles) {
          const int j1 = u - 1, j0 = u - 3;
          n = (u + 2 < nunits ? L : 0) + 4 * (u >= 1 ? 2 : 1) + SY * ((j1 < 0 ? 0 : j1 / NH) - (j0 < 0 ? 0 : j0 / NH));
        }
        f_wait_vmcnt(n);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (u + NBUF - 1 < nunits) stage(u + NBUF - 1, buf == 0 ? NBUF - 1 : buf - 1);
      if constexpr (h == 0) {
        if (t > 0) store_y1(t - 1);
      }
      const int tn = (h == NH - 1) ? t + 1 : t;     // tile of unit u + 1
      if (wave < 4) {
        multiply(hc, t, buf);
        if (u + 1 < nunits) transform(std::integral_constant<int, hn>{}, tn, nb);
      } else {
        if (u + 1 < nunits) transform(std::integral_constant<int, hn>{}, tn, nb);
        multiply(hc, t, buf);
      }
      buf = nb;
    });
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (ntiles > 0) store_y1(ntiles - 1);

  if (want_stats) {
    // one partial row per workgroup (row `split` of the [ceil(M / 128)] table); rows no workgroup owns are zero.  Lanes that share a
    // chunk (same lane % CPN, any wave) are folded through LDS in a fixed order.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    float* red = (float*)smem;                       // [512 threads][16]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[tid * 16 + 2 * e] = s1[e][0]; red[tid * 16 + 2 * e + 1] = s1[e][1];
      red[tid * 16 + 8 + 2 * e] = s2[e][0]; red[tid * 16 + 8 + 2 * e + 1] = s2[e][1];
    }
    __syncthreads();
    const int nrows = (p.M + 127) / 128, S = (int)gridDim.x;
    for (int idx = tid; idx < 2 * N; idx += 512) {
      const int which = idx / N, col = idx - which * N;
      const int c = col >> 3, e = col & 7;
      float s = 0.f;
      for (int k = c; k < 512; k += CPN) s += red[k * 16 + which * 8 + e];     // threads k with k % CPN == c (64 % CPN == 0)
      p.stats[((long long)split * 2 + which) * N + col] = s;
      for (int r = split + S; r < nrows; r += S) p.stats[((long long)r * 2 + which) * N + col] = 0.f;
    }
  }
}

int ffwd_mode() {
  static const int m = [] { const char* e = getenv("ICAMD_FUSED_APPLY_CONV"); return e ? atoi(e) : 1; }();
  return m;
}

}  // namespace

// (K, N) = the block boundaries of ResNet-50's layer1 / layer2: 256 -> 64, 256 -> 128, 512 -> 128 (512 -> 256, the one boundary into
// layer3, would need 128 filter VGPRs per wave beside the transform's working set: it keeps the two launches).
// ICAMD_FUSED_APPLY_CONV=0 switches it off (A/B), 2 lifts the size floor (tests).
bool icamd_bn_apply_conv1x1_fused_wanted(long long M, int K, int N) {
  if (ffwd_mode() == 0 || M <= 0 || M >= (1ll << 30)) return false;
  if (!((K == 256 && (N == 64 || N == 128)) || (K == 512 && N == 128))) return false;
  return ffwd_mode() == 2 || M >= 16384;
}

int icamd_bn_apply_conv1x1_fused_launch(FusedFwdParams& p, hipStream_t stream) {
  if (!icamd_bn_apply_conv1x1_fused_wanted(p.M, p.K, p.N)) return ICAMD_ERR_UNSUPPORTED;
  int s = icamd_num_cus();
  const int cap = (p.M + GTM - 1) / GTM, cap_rows = (p.M + 127) / 128;
  if (s > cap) s = cap;
  if (s > cap_rows) s = cap_rows;       // one partial row per split in the [ceil(M / 128)] statistics table
  if (s < 1) s = 1;
  int rows = (p.M + s - 1) / s;
  rows = (rows + GTM - 1) / GTM * GTM;
  p.rows_per_split = rows;
  p.S = (p.M + rows - 1) / rows;
  // y3 and the residual are read once: non-temporal LDS-DMA by default (ResNet-50 18.08 -> 17.89 ms together with the fused
  // backward's, two A/B pairs on one box; ICAMD_FUSED_NT=0: default policy)
  static const int nt = [] { const char* e = getenv("ICAMD_FUSED_NT"); return e ? atoi(e) : 1; }();
  p.nt = nt;
  const dim3 grid((unsigned)p.S), block(512);
  if (p.K == 256 && p.N == 64) hipLaunchKernelGGL((bn_apply_conv1x1_fused_kernel<256, 64>), grid, block, 0, stream, p);
  else if (p.K == 256) hipLaunchKernelGGL((bn_apply_conv1x1_fused_kernel<256, 128>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((bn_apply_conv1x1_fused_kernel<512, 128>), grid, block, 0, stream, p);
  return icamd_launch_status();
}
