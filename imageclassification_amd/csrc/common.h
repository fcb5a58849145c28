// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the training-step hot path.
// Wave = 64 lanes everywhere; bf16 values travel as raw 16-bit patterns (unsigned short).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

#define ICAMD_OK 0
#define ICAMD_ERR_BAD_ARG 1
#define ICAMD_ERR_UNSUPPORTED 2
#define ICAMD_ERR_WORKSPACE 3
#define ICAMD_ERR_LAUNCH 4

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define GPTR(p) ((const void __attribute__((address_space(1)))*)(p))
#define LPTR(p) ((void __attribute__((address_space(3)))*)(p))

// Streaming loads of once-read tensors (round 5).  ICAMD_STREAM_NT (per translation unit, compile time): 1 = non-temporal.  The
// BatchNorm passes gained 1.4 % of a ResNet-50 step from it (norm_pool.hip's own switch); tools/r5_stream_nt_variants.sh builds the
// A/B variants of the other elementwise translation units.
#ifndef ICAMD_STREAM_NT
#define ICAMD_STREAM_NT 0
#endif
template <class T>
__device__ __forceinline__ T ld_stream(const T* p) {
  if constexpr (ICAMD_STREAM_NT != 0) return __builtin_nontemporal_load(p);
  else return *p;
}

// 256 B of zeros in device memory: the source of every out-of-image / out-of-range LDS-DMA lane.
// (one zero-initialised copy per translation unit: no relocatable device code needed)
static __device__ __attribute__((aligned(256))) unsigned int icamd_zero_page[64];

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((unsigned int)v) << 16);
}
// Round-to-nearest-even f32 -> bf16 through the hardware convert (keeps NaN a NaN).
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
// Two conversions in ONE v_cvt_pk_bf16_f32 (the scalar form costs two converts and an OR per pair); same rounding.
typedef __bf16 icamd_bf16x2_hw __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
  typedef float icamd_f32x2_t __attribute__((ext_vector_type(2)));
  const icamd_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, icamd_bf16x2_hw));
}
__device__ __forceinline__ float bf16_lo(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Exact unsigned division by a runtime constant (n < 2^31, d >= 1): q = (n * mul) >> 32 >> sh.
struct FastDiv {
  unsigned int mul, sh, d, pad;
};
__device__ __forceinline__ unsigned int fdiv(unsigned int n, const FastDiv& f) {
  return f.d == 1 ? n : (__umulhi(n, f.mul) >> f.sh);
}
static inline FastDiv make_fastdiv(unsigned int d) {
  FastDiv f; f.d = d; f.pad = 0;
  if (d == 1) { f.mul = 0; f.sh = 0; return f; }
  unsigned int l = 0; while ((1ull << l) < d) ++l;          // ceil(log2 d)
  unsigned long long m = ((1ull << (32 + l - 1)) + d - 1) / d;  // round-up magic for n < 2^31
  f.mul = (unsigned int)m; f.sh = l - 1;
  return f;
}

// GELU (erf form, as torch.nn.GELU() / timm's Mlp use) and its derivative: gelu(z) = z Phi(z), gelu'(z) = Phi(z) + z phi(z).
// The epilogues that apply it are VALU-bound (ConvNeXt-T evaluates 6.6 G of them per step -- fc1 forward twice under mixup,
// fc2 data gradient --, ViT-B/16 3.7 G), so the cost per element is what counts: the device library's erff is ~50
// instructions; Abramowitz & Stegun 7.1.26 (round 3, first form) one v_rcp_f32 + one v_exp_f32 + 14 full-rate ones.  This
// form: the lower tail as ONE exponential of a polynomial, Phi(-u) = 2^P(u), u = min(|z|, 9.5), P = the degree-8 fit of
// log2 Phi(-u) on [0, 9.5] (max |error| 3e-5 in fp32 Horner, i.e. 2e-5 RELATIVE in Phi(-u) down to Phi = 1e-21: the error
// sits in the exponent, so the deep negative side needs no special case).  Two elements per
// lane: the Horner steps are v_pk_fma_f32 (two multiply-adds per lane and issue slot), the one transcendental is v_exp_f32.
// Against fp64 over all bf16 inputs in [-9.5, 9.5]: |gelu error| <= 1.2e-6 and <= 2.1e-5 relative, two orders below the
// bf16 rounding of the result.  The derivative adds phi(z) = 2^(-z^2 log2(e)/2 - log2 sqrt(2 pi)): a second v_exp_f32.
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
struct GeluTail { f32x2 u, q; };   // u = min(|z|, 9.5), q = Phi(-u)
__device__ __forceinline__ GeluTail gelu_tail2(const f32x2 z) {
  // (v_med3_f32 with the |z| modifier: one instruction; fminf(fabsf()) adds a canonicalising v_max)
  const f32x2 u = {__builtin_amdgcn_fmed3f(__builtin_fabsf(z[0]), 0.f, 9.5f), __builtin_amdgcn_fmed3f(__builtin_fabsf(z[1]), 0.f, 9.5f)};
  f32x2 p = __builtin_elementwise_fma(splat2(6.770644489506594e-08f), u, splat2(-3.415887022129027e-06f));
  p = __builtin_elementwise_fma(p, u, splat2(7.608014857396483e-05f));
  p = __builtin_elementwise_fma(p, u, splat2(-0.000995745649561286f));
  p = __builtin_elementwise_fma(p, u, splat2(0.00865430012345314f));
  p = __builtin_elementwise_fma(p, u, splat2(-0.05409912392497063f));
  p = __builtin_elementwise_fma(p, u, splat2(-0.45845112204551697f));
  p = __builtin_elementwise_fma(p, u, splat2(-1.1512187719345093f));
  p = __builtin_elementwise_fma(p, u, splat2(-0.9999991059303284f));
  return {u, f32x2{__builtin_amdgcn_exp2f(p[0]), __builtin_amdgcn_exp2f(p[1])}};
}
// gelu(z) = z Phi(z) = max(z, 0) - u Phi(-u) on both sides of zero (z >= 0: z - z q; z < 0: z q = -u q): no select, and
// past the clamp u q ~ 1e-20.  max(z, 0) as (z + |z|) / 2: exact, two instructions like the canonicalising fmaxf, and a NaN
// stays a NaN.
__device__ __forceinline__ f32x2 gelu2(const f32x2 z) {
  const GeluTail t = gelu_tail2(z);
  const f32x2 h = z * splat2(0.5f);
  const f32x2 relu = {__builtin_fmaf(__builtin_fabsf(z[0]), 0.5f, h[0]), __builtin_fmaf(__builtin_fabsf(z[1]), 0.5f, h[1])};
  return __builtin_elementwise_fma(-t.u, t.q, relu);
}
__device__ __forceinline__ f32x2 gelu_grad2(const f32x2 z) {
  const GeluTail t = gelu_tail2(z);
  const f32x2 r = splat2(1.0f) - t.q;
  const f32x2 cdf = {z[0] < 0.f ? t.q[0] : r[0], z[1] < 0.f ? t.q[1] : r[1]};
  const f32x2 e = __builtin_elementwise_fma(z * z, splat2(-0.72134752044448170f), splat2(-1.3257480647361593f));
  const f32x2 phi = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
  return __builtin_elementwise_fma(z, phi, cdf);
}
__device__ __forceinline__ float gelu_f(float z) { return gelu2(f32x2{z, z})[0]; }
__device__ __forceinline__ float gelu_grad_f(float z) { return gelu_grad2(f32x2{z, z})[0]; }
// 8 bf16 values at once: a = gelu(z);  dz = da * gelu'(z)
__device__ __forceinline__ u32x4 gelu8(const u32x4 z) {
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const f32x2 g = gelu2(f32x2{bf16_lo(z[e]), bf16_hi(z[e])});
    o[e] = pack_bf16x2(g[0], g[1]);
  }
  return o;
}
__device__ __forceinline__ u32x4 gelu_bwd8(const u32x4 da, const u32x4 z) {
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const f32x2 g = f32x2{bf16_lo(da[e]), bf16_hi(da[e])} * gelu_grad2(f32x2{bf16_lo(z[e]), bf16_hi(z[e])});
    o[e] = pack_bf16x2(g[0], g[1]);
  }
  return o;
}

// ---- helpers of the persistent "operand resident in registers" kernels (conv3x3_halo.hip, conv1x1_resident.hip,
// conv_stem.hip, conv_wgrad.hip) ----
// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N) -- the index is usable as an asm immediate
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
// LDS reads through inline asm with an IMMEDIATE offset.  hipcc's wait-count pass puts s_waitcnt vmcnt(0) in front of C++
// (and builtin) LDS loads while an LDS-DMA is in flight, because it cannot tell the buffers apart; the asm form is invisible
// to it and its completion is waited for by hand (counted s_waitcnt lgkmcnt naming the destination as an in/out operand).
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read128_off(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset is a 16-bit field");
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ bf16x8 tr_read_pair_off(unsigned a0, unsigned a1) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset is a 16-bit field");
  bf16x4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a0), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a1), "n"(OFF));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// CUs of the current device (persistent grids are sized from it); 256 if the query fails
static inline int icamd_num_cus() {
  static const int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 256;
    return cus;
  }();
  return n;
}

// XCDs (XCCs) of the current device as the runtime reports them: 8 on an MI355X in its default (SPX) mode.  The XCD-aware
// workgroup orders assume round-robin dispatch over exactly 8 XCDs; they are bijections on any device (correctness never depends
// on them) and are switched off when the count differs (CPX / partitioned modes, other parts).  0 if the query fails.
static inline int icamd_num_xccs() {
  static const int n = [] {
    int dev = 0, x = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&x, hipDeviceAttributeNumberOfXccs, dev) != hipSuccess || x <= 0) return 0;
    return x;
  }();
  return n;
}

static inline int icamd_launch_status() {
  return hipGetLastError() == hipSuccess ? ICAMD_OK : ICAMD_ERR_LAUNCH;
}
