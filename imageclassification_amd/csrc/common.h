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

// GELU (erf form, as torch.nn.GELU() / timm's Mlp use) and its derivative.  gelu(z) = z * Phi(z), 2 Phi(z) = erfc(-z / sqrt 2).
// erfc through Abramowitz & Stegun 7.1.26 (erfc(x) = (a1 t + .. + a5 t^5) e^{-x^2}, t = 1 / (1 + p x), x >= 0; |error| <= 1.5e-7):
// one v_rcp_f32, one v_exp_f32 and ten multiply-adds, against the ~50 instructions of the device library's erff -- which made
// the GELU epilogues VALU-bound: ConvNeXt-T evaluates 6.6 G of them per step (fc1 forward twice under mixup, fc2 data
// gradient), ViT-B/16 3.7 G; the 96 -> 384 forward at batch 256 ran 527 us against 220 us of HBM time.  The negative side uses
// erfc directly (no 1 - erf cancellation); measured against fp64 over all bf16 inputs in [-9, 9]: |gelu error| <= 4.6e-7,
// |gelu' error| <= 3.1e-7 -- three orders below the bf16 rounding of the result.
struct GeluParts { float cdf2, e; };   // 2 Phi(z), exp(-z^2 / 2)
__device__ __forceinline__ GeluParts gelu_parts(float z) {
  const float x = fabsf(z) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __expf(-x * x);
  const float q = p * t * e;                       // erfc(|z| / sqrt 2)
  return {z < 0.f ? q : 2.f - q, e};
}
__device__ __forceinline__ float gelu_f(float z) { return 0.5f * z * gelu_parts(z).cdf2; }
__device__ __forceinline__ float gelu_grad_f(float z) {
  const GeluParts g = gelu_parts(z);
  return fmaf(z * 0.39894228040143268f, g.e, 0.5f * g.cdf2);
}
// 8 bf16 values at once: a = gelu(z);  dz = da * gelu'(z)
__device__ __forceinline__ u32x4 gelu8(const u32x4 z) {
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(gelu_f(bf16_lo(z[e])), gelu_f(bf16_hi(z[e])));
  return o;
}
__device__ __forceinline__ u32x4 gelu_bwd8(const u32x4 da, const u32x4 z) {
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    o[e] = pack_bf16x2(bf16_lo(da[e]) * gelu_grad_f(bf16_lo(z[e])), bf16_hi(da[e]) * gelu_grad_f(bf16_hi(z[e])));
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

static inline int icamd_launch_status() {
  return hipGetLastError() == hipSuccess ? ICAMD_OK : ICAMD_ERR_LAUNCH;
}
