// Multi-head self-attention for short sequences (ViT: T = 197, head dim 64) on gfx950: forward and backward, everything of
// one (image, head) resident in LDS, bf16 operands, fp32 MFMA accumulation (v_mfma_f32_16x16x32_bf16), softmax in registers.
//
// Replaces what ATen runs for timm's Attention module (softmax(q k^T / sqrt(d)) v) under `model(samples)` and
// `loss.backward()` of the reference step (/root/reference/engine.py:48,51,64,72) for vit_base_patch16_224.
//
// Layout: qkv is the [B*T][3*H*64] output of the fused QKV projection (columns q | k | v, each [head][64]);
// out / dout are [B*T][H*64]; dqkv mirrors qkv; lse and delta are fp32 [B][H][T].
//
// Execution (round 3): ONE persistent workgroup of 16 waves per CU walks the (image, head) pairs; waves [0, 13) are
// consumers (wave w owns the w-th block of 16 queries or keys), the other three are producers that stage the next head's
// two matrices into the second LDS buffer by LDS-DMA while the consumers work (see "Persistent form" below).  Measured at
// batch 256 (3072 heads), round 2 -> round 3: forward 124 -> 85 us, backward (both kernels) 322 -> 307 us.
//
// The score tile is always computed TRANSPOSED relative to the operand that will consume it, so that an accumulator
// tile is directly the next MFMA's operand (k order permuted identically on both operands) and nothing ever crosses
// LDS between two products:
//   forward        : S^T = K Q^T (query on the lane) for ALL key tiles -> exact softmax per lane column -> O = P V with P
//                    straight from the accumulators and V^T fragments by ds_read_b64_tr_b16;
//   backward (dQ)  : S^T, dP^T = V dO^T (query on the lane) -> dS^T -> dQ^T = K^T dS^T (K^T by transposed reads);
//   backward (dK,dV): S = Q K^T, dP = dO V^T (key on the lane) -> dV^T = dO^T P, dK^T = Q^T dS (transposed reads of
//                    dO and Q); each wave owns a whole key block, so no cross-wave reduction and no atomics.
// The two backward kernels recompute S and dP independently (7 products instead of 5): attention is 4 % of ViT-B's
// FLOPs, and this keeps every sum in a fixed order (bitwise reproducible).  Their pair loop is what bounds them now: a
// chain ds_read -> MFMA -> exp -> pack -> transposed read -> MFMA per pair of 16-row blocks with three or four waves per
// SIMD to hide it and no registers left (126 of 128) to prefetch the next pair.  An eight-wave form of the dK / dV kernel (7
// consumers x TWO key blocks, 240 registers, half the LDS traffic per key, pair loop rolled or fully unrolled) measured the
// same 300 us: per head MFMA (3.0 us), other instructions (4.3 us) and LDS reads (3.6 us) add up to what is measured --
// nothing overlaps inside a wave as hipcc schedules it.  Ablations of the two backward kernels as they are (us for both, 302
// as is): without the exponentials 294, without the dK / dV products 275, without the S / dP products 293, without all
// three 286; without the producers' staging 272, without the consumers' output stores 286, WITHOUT THE CONSUMERS'
// FRAGMENT LOADS OF THE NEXT HEAD 233, without any global traffic 184; an LDS-only barrier (no vmcnt drain at the hand-over)
// 302.  I.e. the kernels move their 225 KB per head at 3.1 TB/s -- every CU asks for its next head at the same moment and
// then computes -- and arithmetic is not what bounds them: the next step is ONE backward kernel (q, k, v, dO, O read once:
// 125 KB per head), not a faster pair loop.
#ifndef ICAMD_ATTN_NT
#define ICAMD_ATTN_NT 0   // cache policy of the once-read LDS-DMA streams of this unit: 0 default, 2 non-temporal (round 5 A/B)
#endif
#include "common.h"
#include "icamd_internal.h"

namespace {

constexpr int HD = 64;          // head dimension
constexpr int ROWB = HD * 2;    // bytes per LDS row

// LDS image of a [rows][64] bf16 matrix (128 B rows): the 32 B column block is XOR-ed with (row>>1)&3.  ONE image
// serves both access patterns without bank conflicts: 4-row x 16-column blocks read transposed (ds_read_b64_tr_b16; a
// 32-lane half touches 8 rows x 32 B = 2 row parities x 4 block keys) and 16 B chunks of one row per lane
// (ds_read_b128; its 16-lane groups {0-3, 12-15, 20-27}, ... hold rows of four different keys for the even chunk and
// of four for the odd one).  Keeping a single image per matrix is what lets two workgroups share a CU's 160 KB.
__device__ __forceinline__ int tr_img(int row, int chunk) {
  return row * ROWB + ((((chunk >> 1) ^ ((row >> 1) & 3))) << 5) + ((chunk & 1) << 4);
}
__device__ __forceinline__ int row_img(int row, int chunk) { return tr_img(row, chunk); }

__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* img, int row0, int row1, int dblk, int lane) {
  const int c = lane & 15, q = c >> 2, pq = c & 3;
  const int ra = row0 + q, rb = row1 + q;
  bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (bf16x4 __attribute__((address_space(3)))*)(img + ra * ROWB + ((dblk ^ ((ra >> 1) & 3)) << 5) + 8 * pq));
  bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (bf16x4 __attribute__((address_space(3)))*)(img + rb * ROWB + ((dblk ^ ((rb >> 1) & 3)) << 5) + 8 * pq));
  return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

__device__ __forceinline__ bf16x8 pack_acc2(const f32x4& lo, const f32x4& hi) {   // four v_cvt_pk_bf16_f32
  u32x4 r;
  r[0] = pack_bf16x2(lo[0], lo[1]);
  r[1] = pack_bf16x2(lo[2], lo[3]);
  r[2] = pack_bf16x2(hi[0], hi[1]);
  r[3] = pack_bf16x2(hi[2], hi[3]);
  return __builtin_bit_cast(bf16x8, r);
}

// B-operand fragments of a row-major [row][64] matrix straight from global memory: lane (c, g) takes row `row`,
// columns 8g..8g+7 (+32 for the second k-step)
__device__ __forceinline__ void load_rowfrag(const bf16_t* __restrict__ base, long long ld, int row, int T, int g,
                                             bf16x8* f) {
  // rows past T read the zero page: a select AFTER the loads (round 1-4: `if (row >= T) f = 0`) made every caller wait for the
  // loads on the spot -- the "prefetch" of the next head's fragments at the top of a head was followed by s_waitcnt vmcnt(0)
  // before the first MFMA (round 5, found in the ISA)
  const bf16_t* p = row < T ? base + (long long)row * ld + 8 * g : (const bf16_t*)icamd_zero_page;
  f[0] = *(const bf16x8*)p;
  f[1] = *(const bf16x8*)(p + (row < T ? 32 : 0));
}

// The same two loads issued behind hipcc's back (round 5): its s_waitcnt pass put vmcnt(3..0) in front of the first MFMAs of a head
// for loads whose results are only read after the head (the next head's fragments) -- seen in the ISA of all three kernels,
// whatever the source did about selects and stores.  The results must not be touched before prefetch_wait() has named them.
__device__ __forceinline__ void load_rowfrag_async(const bf16_t* __restrict__ base, long long ld, int row, int T, int g,
                                                   bf16x8* f) {
  const bf16_t* p = row < T ? base + (long long)row * ld + 8 * g : (const bf16_t*)icamd_zero_page;
  const bf16_t* p1 = p + (row < T ? 32 : 0);
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(f[0]) : "v"(p) : "memory");
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(f[1]) : "v"(p1) : "memory");
}

// v_max3_f32 as is (fmaxf adds a canonicalising v_max per operand that comes out of an MFMA)
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float group_max(float v) {   // across the 4 lane groups that share a column
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// ----------------------------------------------------------------------------------------------------------------
// Persistent form shared by the three kernels: ONE workgroup of 16 waves per CU walks the (image, head) pairs h = blockIdx.x,
// + gridDim.x, ...  Waves [0, NKB) are CONSUMERS (wave w owns block w of 16 queries / keys), the other 16 - NKB waves are
// PRODUCERS: while the consumers work on head i out of LDS buffer i & 1, the producers stage the two matrices of head
// i + 1 into the other buffer; one __syncthreads per head hands the buffers over.  With one workgroup per head and two
// per CU (rounds 1-2) every workgroup of the chip staged, then computed, in lockstep: the forward ran 110 us, of which
// 42 us were the staging nobody overlapped (compute on unstaged LDS: 68 us).
// ----------------------------------------------------------------------------------------------------------------
constexpr int PTHREADS = 1024, PWAVES = PTHREADS / 64;

// producer side: `nthr` threads (index t) stage the [T][64] slices of two matrices, eight chunks of each in flight
template <int RP>
__device__ __forceinline__ void stage_two_by(const bf16_t* __restrict__ src0, long long ld0, unsigned char* img0,
                                             const bf16_t* __restrict__ src1, long long ld1, unsigned char* img1, int T,
                                             int t, int nthr) {
  constexpr int CH = RP * 8, BATCH = 4;
  for (int base = t; base < CH; base += nthr * BATCH) {
    u32x4 v0[BATCH], v1[BATCH];
#pragma unroll
    for (int k = 0; k < BATCH; ++k) {
      const int i = base + k * nthr;
      const int r = i >> 3, ch = i & 7;
      const int rr = (i < CH && r < T) ? r : 0;
      v0[k] = *(const u32x4*)(src0 + (long long)rr * ld0 + ch * 8);
      v1[k] = *(const u32x4*)(src1 + (long long)rr * ld1 + ch * 8);
    }
#pragma unroll
    for (int k = 0; k < BATCH; ++k) {
      const int i = base + k * nthr;
      const int r = i >> 3, ch = i & 7;
      if (i < CH) {
        const u32x4 z = {0u, 0u, 0u, 0u};
        *(u32x4*)(img0 + tr_img(r, ch)) = r < T ? v0[k] : z;
        *(u32x4*)(img1 + tr_img(r, ch)) = r < T ? v1[k] : z;
      }
    }
  }
}

// producer side, LDS-DMA form: wave pw of npw walks the 8-row slabs of both images; one wave-instruction fills 8 rows x 128 B
// contiguously, so the chunk permutation of the image is applied on the SOURCE side (the XOR key (row >> 1) & 3 of a
// lane's row does not depend on the slab); rows past T read the zero page.  Nothing passes through registers: all of a
// wave's ~19 instructions are in flight at once (the register form above needed five trips to HBM per head with three
// producer waves: the producers, not the consumers, set the pace).  The caller waits vmcnt(0) before the hand-over barrier.
template <int RP>
__device__ __forceinline__ void stage_two_dma(const bf16_t* __restrict__ src0, long long ld0, unsigned char* img0,
                                              const bf16_t* __restrict__ src1, long long ld1, unsigned char* img1, int T,
                                              int pw, int npw, int lane) {
  const int lr = lane >> 3, cpos = lane & 7;
  const int ch = ((((cpos >> 1) ^ ((lr >> 1) & 3))) << 1) | (cpos & 1);
  const unsigned char* zero = (const unsigned char*)icamd_zero_page;
  for (int j = pw; j < RP / 8; j += npw) {
    const int r = j * 8 + lr;
    const bool ok = r < T;
    const void* a0 = ok ? (const void*)(src0 + (long long)r * ld0 + ch * 8) : (const void*)zero;
    const void* a1 = ok ? (const void*)(src1 + (long long)r * ld1 + ch * 8) : (const void*)zero;
    __builtin_amdgcn_global_load_lds(GPTR(a0), LPTR(img0 + j * 1024), 16, 0, ICAMD_ATTN_NT);
    __builtin_amdgcn_global_load_lds(GPTR(a1), LPTR(img1 + j * 1024), 16, 0, ICAMD_ATTN_NT);
  }
}

// ----------------------------------------------------------------------------------------------------------------
// forward
// ----------------------------------------------------------------------------------------------------------------
template <int NKB, bool LASTONLY>   // LASTONLY: T > (NKB - 1) * 16, i.e. only the last key tile reaches past T
__global__ __launch_bounds__(PTHREADS, 4) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                float* __restrict__ lse, int T, int H, float scale, int nheads) {
  constexpr int NPAIR = (NKB + 1) / 2;
  constexpr int RP = NPAIR * 32;   // padded key rows (zeros beyond T)
  constexpr int IMG = RP * ROWB;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][IMG];   // [buffer][K rows | V (read transposed)]
  const long long ld = 3ll * H * HD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const float c1 = scale * 1.4426950408889634f;   // exp(scale * x) = 2^(c1 * x)
  auto head_base = [&](int hd) { const int b = hd / H, h = hd - b * H; return qkv + (long long)b * T * ld + h * HD; };
  int hd = blockIdx.x;
  if (hd >= nheads) return;
  {
    const bf16_t* base = head_base(hd);
    stage_two_by<RP>(base + (long long)H * HD, ld, lds[0][0], base + 2ll * H * HD, ld, lds[0][1], T, threadIdx.x, PTHREADS);
  }
  bf16x8 qf[2];
  if (wave < NKB) load_rowfrag(head_base(hd), ld, wave * 16 + c, T, g, qf);
  __syncthreads();
  for (int it = 0; hd < nheads; hd += gridDim.x, ++it) {
    const int nxt = hd + gridDim.x;
    const unsigned char* Kr = lds[it & 1][0];
    const unsigned char* Vt = lds[it & 1][1];
    if (wave >= NKB) {
      if (nxt < nheads) {
        const bf16_t* base = head_base(nxt);
        stage_two_dma<RP>(base + (long long)H * HD, ld, lds[(it & 1) ^ 1][0], base + 2ll * H * HD, ld, lds[(it & 1) ^ 1][1], T,
                          wave - NKB, PWAVES - NKB, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    } else {
      const int b = hd / H, h = hd - b * H;
      const int qblk = wave;
      const int qrow = qblk * 16 + c;
      // the next head's query fragment is requested now and used after the barrier
      bf16x8 qn[2];
      if (nxt < nheads) load_rowfrag(head_base(nxt), ld, qrow, T, g, qn);
      // Two passes over the WHOLE key range instead of an online softmax (T <= 208: the 13 score tiles of a query block are
      // 52 registers): all S^T tiles first -- 2 NKB independent MFMA chains --, one exact maximum and one sum per query (two
      // shuffles each per block instead of six per pair of key blocks), every exponential independent of the others, then
      // O = P V.  Lane (c, g) holds, for query c, the keys kb*16 + 4g + r of tile kb.
      f32x4 sv[NKB];
      static_for<0, NKB>([&](auto kc) {
        constexpr int kb = decltype(kc)::value;
        const bf16x8 k0 = *(const bf16x8*)(Kr + row_img(kb * 16 + c, g));
        const bf16x8 k1 = *(const bf16x8*)(Kr + row_img(kb * 16 + c, 4 + g));
        sv[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);   // S^T[key][query]
        sv[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[1], sv[kb], 0, 0, 0);
        // (fragments of at most three tiles in flight: left alone, hipcc hoists all 2 NKB reads and spills)
        if constexpr (kb % 3 == 2) __builtin_amdgcn_sched_barrier(0);
      });
      float m = -INFINITY;
      static_for<0, NKB>([&](auto kc) {
        constexpr int kb = decltype(kc)::value;
        if ((!LASTONLY || kb == NKB - 1) && kb * 16 + 16 > T) {   // the tile(s) that reach past T
#pragma unroll
          for (int r = 0; r < 4; ++r) sv[kb][r] = (kb * 16 + 4 * g + r < T) ? sv[kb][r] : -INFINITY;
        }
        m = max3_raw(max3_raw(m, sv[kb][0], sv[kb][1]), sv[kb][2], sv[kb][3]);
      });
      m = group_max(m);             // finite: key 0 exists
      const float mc = m * c1;
      float l = 0.f;
      static_for<0, NKB>([&](auto kc) {
        constexpr int kb = decltype(kc)::value;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sv[kb][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[kb][r], c1, -mc));   // masked keys: 2^-inf = 0
          l += sv[kb][r];
        }
      });
      m *= scale;                   // the maximum of the scaled scores, for the log-sum-exp below
      f32x4 o[4];
#pragma unroll
      for (int db = 0; db < 4; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
      static_for<0, NPAIR>([&](auto pc) {
        constexpr int pp = decltype(pc)::value;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        const bf16x8 pf = pack_acc2(sv[2 * pp], 2 * pp + 1 < NKB ? sv[2 * pp + 1 < NKB ? 2 * pp + 1 : 0] : zero4);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const bf16x8 vf = tr_pair(Vt, 32 * pp + 4 * g, 32 * pp + 16 + 4 * g, db, lane);
          o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, vf, o[db], 0, 0, 0);   // D[query 4g+r][d = db*16 + c]
        }
        if constexpr (pp % 2 == 1) __builtin_amdgcn_sched_barrier(0);
      });
      l = group_sum(l);
      if (g == 0 && qrow < T) lse[((long long)b * H + h) * T + qrow] = m + __logf(l);
      const float inv_l = 1.f / l;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float il = __shfl(inv_l, 4 * g + r, 64);   // 1/l of query 4g+r (held by the lanes with c == 4g+r)
        const int qo = qblk * 16 + 4 * g + r;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const float v = o[db][r] * il;
          const float vn = __shfl_xor(v, 1, 64);
          if ((c & 1) == 0 && qo < T)
            *(unsigned int*)(out + ((long long)b * T + qo) * (H * HD) + h * HD + db * 16 + c) = pack_bf16x2(v, vn);
        }
      }
      qf[0] = qn[0]; qf[1] = qn[1];
    }
    __syncthreads();
  }
}

// ----------------------------------------------------------------------------------------------------------------
// backward, part 1: dQ (query on the lane) and delta = rowsum(dO * O).  Persistent form (see above): K and V of the next
// head are staged by the producer waves.  No masks in the loop: a query row past T gets lse = +inf (p = 2^-inf = 0), a key
// past T is a zero row of K, so whatever its dS is it adds nothing to dQ = K^T dS.  The softmax scale is applied once to
// the dQ accumulators (dS = p (dP - delta) here, without the factor).
// ----------------------------------------------------------------------------------------------------------------
template <int NKB>
__global__ __launch_bounds__(PTHREADS, 4) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                                   const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                                   float* __restrict__ delta, bf16_t* __restrict__ dqkv, int T,
                                                                   int H, float scale, int nheads) {
  constexpr int NPAIR = (NKB + 1) / 2;
  constexpr int RP = NPAIR * 32;
  constexpr int IMG = RP * ROWB;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][IMG];   // [buffer][K | V]; K read by rows and transposed
  const long long ld = 3ll * H * HD, ldo = (long long)H * HD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const float c1 = scale * 1.4426950408889634f;
  auto head_base = [&](int hd) { const int b = hd / H, h = hd - b * H; return qkv + (long long)b * T * ld + h * HD; };
  auto head_obase = [&](int hd) { const int b = hd / H, h = hd - b * H; return (long long)b * T * ldo + h * HD; };
  int hd = blockIdx.x;
  if (hd >= nheads) return;
  {
    const bf16_t* base = head_base(hd);
    stage_two_by<RP>(base + (long long)H * HD, ld, lds[0][0], base + 2ll * H * HD, ld, lds[0][1], T, threadIdx.x, PTHREADS);
  }
  const int qrow = wave * 16 + c;          // (consumers) this lane's query
  bf16x8 qf[2], dof[2], of[2];
  float lq = 0.f;
  if (wave < NKB) {
    load_rowfrag(head_base(hd), ld, qrow, T, g, qf);
    load_rowfrag(dout + head_obase(hd), ldo, qrow, T, g, dof);
    load_rowfrag(out + head_obase(hd), ldo, qrow, T, g, of);
    lq = qrow < T ? lse[(long long)hd * T + qrow] : 0.f;
  }
  __syncthreads();
  for (int it = 0; hd < nheads; hd += gridDim.x, ++it) {
    const int nxt = hd + gridDim.x;
    const unsigned char* Kr = lds[it & 1][0];
    const unsigned char* Vr = lds[it & 1][1];
    const unsigned char* Kt = Kr;
    if (wave >= NKB) {
      if (nxt < nheads) {
        const bf16_t* base = head_base(nxt);
        stage_two_dma<RP>(base + (long long)H * HD, ld, lds[(it & 1) ^ 1][0], base + 2ll * H * HD, ld, lds[(it & 1) ^ 1][1], T,
                          wave - NKB, PWAVES - NKB, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    } else {
      // the next head's fragments are requested now and used after the barrier
      bf16x8 qn[2] = {bf16x8{0, 0, 0, 0, 0, 0, 0, 0}, bf16x8{0, 0, 0, 0, 0, 0, 0, 0}}, don[2] = {bf16x8{0, 0, 0, 0, 0, 0, 0, 0}, bf16x8{0, 0, 0, 0, 0, 0, 0, 0}}, on[2] = {bf16x8{0, 0, 0, 0, 0, 0, 0, 0}, bf16x8{0, 0, 0, 0, 0, 0, 0, 0}};
      float lqn = 0.f;
      if (nxt < nheads) {
        load_rowfrag_async(head_base(nxt), ld, qrow, T, g, qn);
        load_rowfrag_async(dout + head_obase(nxt), ldo, qrow, T, g, don);
        load_rowfrag_async(out + head_obase(nxt), ldo, qrow, T, g, on);
        const float* lp = qrow < T ? lse + ((long long)nxt * T + qrow) : (const float*)icamd_zero_page;
        asm volatile("global_load_dword %0, %1, off" : "=v"(lqn) : "v"(lp) : "memory");
      }
      float dl = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) dl += bf16_to_f32((bf16_t)dof[ks][j]) * bf16_to_f32((bf16_t)of[ks][j]);
      dl = group_sum(dl);
      if (g == 0 && qrow < T) delta[(long long)hd * T + qrow] = dl;
      const float lq2 = qrow < T ? lq * 1.4426950408889634f : INFINITY;
      // per pair of key blocks: S^T and dP^T (key on the accumulator rows, query on the lane) -> dS^T -> straight into
      // dQ^T[d][query] += K^T[d][key] dS^T[key][query]; nothing but dq[] lives across iterations (rolled: unrolled, hipcc
      // hoists the fragment reads of every pair and spills)
      f32x4 dq[4];
#pragma unroll
      for (int db = 0; db < 4; ++db) dq[db] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int pp = 0; pp < NPAIR; ++pp) {
        f32x4 ds2[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int kb = 2 * pp + u;   // rows beyond NKB*16 are zero in the images
          f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 kf = *(const bf16x8*)(Kr + row_img(kb * 16 + c, ks * 4 + g));
            const bf16x8 vf = *(const bf16x8*)(Vr + row_img(kb * 16 + c, ks * 4 + g));
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s, 0, 0, 0);      // S^T[key][query]
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[ks], dp, 0, 0, 0);   // dP^T[key][query]
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c1, -lq2));
            ds2[u][r] = p * (dp[r] - dl);
          }
        }
        const bf16x8 dsf = pack_acc2(ds2[0], ds2[1]);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const bf16x8 ktf = tr_pair(Kt, 32 * pp + 4 * g, 32 * pp + 16 + 4 * g, db, lane);   // A[d = db*16 + c][keys]
          dq[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf, dq[db], 0, 0, 0);       // D[d 4g+r][query c]
        }
      }
      if (qrow < T) {
        const int b = hd / H, h = hd - b * H;
        bf16_t* dst = dqkv + ((long long)b * T + qrow) * ld + h * HD;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          u32x2 pk;
          pk[0] = pack_bf16x2(dq[db][0] * scale, dq[db][1] * scale);
          pk[1] = pack_bf16x2(dq[db][2] * scale, dq[db][3] * scale);
          *(u32x2*)(dst + db * 16 + 4 * g) = pk;
        }
      }
      // Everything this wave has in flight retires HERE, in front of the hand-over barrier (round 5): with the output stores of this
      // head still counted in vmcnt at the top of the next one, hipcc could not tell them from the fragment loads it issues there
      // and put s_waitcnt vmcnt(0 / 1) in front of the first use of the CURRENT fragments -- the prefetch was waited for on the spot.
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(qn[0]), "+v"(qn[1]), "+v"(don[0]), "+v"(don[1]), "+v"(on[0]), "+v"(on[1]), "+v"(lqn) :: "memory");
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { qf[ks] = qn[ks]; dof[ks] = don[ks]; of[ks] = on[ks]; }
      lq = lqn;
    }
    __syncthreads();
  }
}

// ----------------------------------------------------------------------------------------------------------------
// backward, part 2: dK and dV (key on the lane; a consumer wave owns one key block and walks all queries).  Persistent form:
// the producers stage Q and dO of the next head and its lse (as log2, +inf past T: those queries get p = 0) and delta rows.
// No masks in the loop (a key lane past T computes values nobody stores); the scale is applied once to dK.
// ----------------------------------------------------------------------------------------------------------------
template <int NKB>
__global__ __launch_bounds__(PTHREADS, 4) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                                    const float* __restrict__ lse, const float* __restrict__ delta,
                                                                    bf16_t* __restrict__ dqkv, int T, int H, float scale, int nheads) {
  constexpr int NPAIR = (NKB + 1) / 2;
  constexpr int RP = NPAIR * 32;
  constexpr int IMG = RP * ROWB;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][IMG];   // [buffer][Q | dO]: each read by rows (S, dP) and
  __shared__ __attribute__((aligned(16))) float s_lse2[2][RP], s_dl[2][RP];   // transposed (dK^T, dV^T)
  const long long ld = 3ll * H * HD, ldo = (long long)H * HD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const float c1 = scale * 1.4426950408889634f;
  auto head_base = [&](int hd) { const int b = hd / H, h = hd - b * H; return qkv + (long long)b * T * ld + h * HD; };
  auto head_obase = [&](int hd) { const int b = hd / H, h = hd - b * H; return (long long)b * T * ldo + h * HD; };
  auto stage_rows = [&](int hd, int buf, int t, int nthr) {
    for (int i = t; i < RP; i += nthr) {
      s_lse2[buf][i] = i < T ? lse[(long long)hd * T + i] * 1.4426950408889634f : INFINITY;
      s_dl[buf][i] = i < T ? delta[(long long)hd * T + i] : 0.f;
    }
  };
  int hd = blockIdx.x;
  if (hd >= nheads) return;
  stage_two_by<RP>(head_base(hd), ld, lds[0][0], dout + head_obase(hd), ldo, lds[0][1], T, threadIdx.x, PTHREADS);
  stage_rows(hd, 0, threadIdx.x, PTHREADS);
  const int krow = wave * 16 + c;          // (consumers) this lane's key
  bf16x8 kf[2], vf[2];
  if (wave < NKB) {
    load_rowfrag(head_base(hd) + (long long)H * HD, ld, krow, T, g, kf);
    load_rowfrag(head_base(hd) + 2ll * H * HD, ld, krow, T, g, vf);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the first head's fragments are here before the loop (hipcc re-waited for them in it)
  __syncthreads();
  for (int it = 0; hd < nheads; hd += gridDim.x, ++it) {
    const int nxt = hd + gridDim.x;
    const int cur = it & 1;
    const unsigned char* Qr = lds[cur][0];
    const unsigned char* Dr = lds[cur][1];
    const unsigned char* Qt = Qr;
    const unsigned char* Dt = Dr;
    if (wave >= NKB) {
      if (nxt < nheads) {
        stage_two_dma<RP>(head_base(nxt), ld, lds[cur ^ 1][0], dout + head_obase(nxt), ldo, lds[cur ^ 1][1], T, wave - NKB,
                          PWAVES - NKB, lane);
        if (wave == PWAVES - 1) stage_rows(nxt, cur ^ 1, lane, 64);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    } else {
      bf16x8 kn[2] = {bf16x8{0, 0, 0, 0, 0, 0, 0, 0}, bf16x8{0, 0, 0, 0, 0, 0, 0, 0}}, vn[2] = {bf16x8{0, 0, 0, 0, 0, 0, 0, 0}, bf16x8{0, 0, 0, 0, 0, 0, 0, 0}};
      if (nxt < nheads) {
        load_rowfrag_async(head_base(nxt) + (long long)H * HD, ld, krow, T, g, kn);
        load_rowfrag_async(head_base(nxt) + 2ll * H * HD, ld, krow, T, g, vn);
      }
      f32x4 dk[4], dv[4];
#pragma unroll
      for (int db = 0; db < 4; ++db) { dk[db] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[db] = dk[db]; }
#pragma unroll 1
      for (int pp = 0; pp < NPAIR; ++pp) {
        f32x4 p2[2], ds2[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int qblk = 2 * pp + u;
          f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 qf = *(const bf16x8*)(Qr + row_img(qblk * 16 + c, ks * 4 + g));
            const bf16x8 df = *(const bf16x8*)(Dr + row_img(qblk * 16 + c, ks * 4 + g));
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf[ks], s, 0, 0, 0);     // S[query 4g+r][key c]
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, vf[ks], dp, 0, 0, 0);   // dP[query][key]
          }
          const f32x4 l4 = *(const f32x4*)&s_lse2[cur][qblk * 16 + 4 * g];          // queries qblk*16 + 4g + r (< RP always)
          const f32x4 d4 = *(const f32x4*)&s_dl[cur][qblk * 16 + 4 * g];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c1, -l4[r]));
            p2[u][r] = p;
            ds2[u][r] = p * (dp[r] - d4[r]);
          }
        }
        const bf16x8 pf = pack_acc2(p2[0], p2[1]);     // B[k = queries 32pp + 4g + r | 32pp + 16 + 4g + r][col = key c]
        const bf16x8 dsf = pack_acc2(ds2[0], ds2[1]);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const bf16x8 dotf = tr_pair(Dt, 32 * pp + 4 * g, 32 * pp + 16 + 4 * g, db, lane);   // A[d][queries] = dO^T
          const bf16x8 qtf = tr_pair(Qt, 32 * pp + 4 * g, 32 * pp + 16 + 4 * g, db, lane);    // A[d][queries] = Q^T
          dv[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dotf, pf, dv[db], 0, 0, 0);        // dV^T[d 4g+r][key c]
          dk[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf, dsf, dk[db], 0, 0, 0);        // dK^T[d 4g+r][key c]
        }
      }
      if (krow < T) {
        const int b = hd / H, h = hd - b * H;
        bf16_t* dstk = dqkv + ((long long)b * T + krow) * ld + (long long)H * HD + h * HD;
        bf16_t* dstv = dqkv + ((long long)b * T + krow) * ld + 2ll * H * HD + h * HD;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          u32x2 pk;
          pk[0] = pack_bf16x2(dk[db][0] * scale, dk[db][1] * scale); pk[1] = pack_bf16x2(dk[db][2] * scale, dk[db][3] * scale);
          *(u32x2*)(dstk + db * 16 + 4 * g) = pk;
          pk[0] = pack_bf16x2(dv[db][0], dv[db][1]); pk[1] = pack_bf16x2(dv[db][2], dv[db][3]);
          *(u32x2*)(dstv + db * 16 + 4 * g) = pk;
        }
      }
      // Everything this wave has in flight retires HERE, in front of the hand-over barrier (round 5): with the output stores of this
      // head still counted in vmcnt at the top of the next one, hipcc could not tell them from the fragment loads it issues there
      // and put s_waitcnt vmcnt(0 / 1) in front of the first use of the CURRENT fragments -- the prefetch was waited for on the spot.
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(kn[0]), "+v"(kn[1]), "+v"(vn[0]), "+v"(vn[1]) :: "memory");
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { kf[ks] = kn[ks]; vf[ks] = vn[ks]; }
    }
    __syncthreads();
  }
}

}  // namespace

int icamd_attention_fwd_launch(const bf16_t* qkv, bf16_t* out, float* lse, int B, int T, int H, float scale, hipStream_t s) {
  const int nheads = B * H;
  const dim3 grid((unsigned)(nheads < icamd_num_cus() ? nheads : icamd_num_cus()));   // one persistent workgroup per CU
  if (T <= 64) hipLaunchKernelGGL((attn_fwd_kernel<4, false>), grid, dim3(PTHREADS), 0, s, qkv, out, lse, T, H, scale, nheads);
  else if (T > 192 && T <= 208) hipLaunchKernelGGL((attn_fwd_kernel<13, true>), grid, dim3(PTHREADS), 0, s, qkv, out, lse, T, H, scale, nheads);
  else if (T <= 208) hipLaunchKernelGGL((attn_fwd_kernel<13, false>), grid, dim3(PTHREADS), 0, s, qkv, out, lse, T, H, scale, nheads);
  else return ICAMD_ERR_UNSUPPORTED;
  return icamd_launch_status();
}

int icamd_attention_bwd_launch(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* lse, float* delta,
                               bf16_t* dqkv, int B, int T, int H, float scale, hipStream_t s) {
  const int nheads = B * H;
  const dim3 grid((unsigned)(nheads < icamd_num_cus() ? nheads : icamd_num_cus()));   // one persistent workgroup per CU
  if (T <= 64) {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<4>, grid, dim3(PTHREADS), 0, s, qkv, out, dout, lse, delta, dqkv, T, H, scale, nheads);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<4>, grid, dim3(PTHREADS), 0, s, qkv, dout, lse, delta, dqkv, T, H, scale, nheads);
  } else if (T <= 208) {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<13>, grid, dim3(PTHREADS), 0, s, qkv, out, dout, lse, delta, dqkv, T, H, scale, nheads);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<13>, grid, dim3(PTHREADS), 0, s, qkv, dout, lse, delta, dqkv, T, H, scale, nheads);
  } else {
    return ICAMD_ERR_UNSUPPORTED;
  }
  return icamd_launch_status();
}
