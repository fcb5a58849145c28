// Internal launch-parameter blocks shared between the C-ABI layer (capi.hip) and the kernels.
#pragma once
#include "common.h"

#define ICAMD_MAX_TAPS 56

// One implicit-GEMM problem: out[n, p*ostr+ooff_h, q*ostr+ooff_w, :] = sum over active taps t, ci of
//   in[n, p*istr+dh[t], q*istr+dw[t], ci] * wt[co][wtap[t]][ci]        (+ bias[co] + addend[same pixel])
struct IgemmParams {
  const bf16_t* in;
  const bf16_t* wt;
  bf16_t* out;
  const bf16_t* addend;  // optional, laid out like out
  const unsigned char* addend_bits;  // optional 1 bit per addend element: the addend counts only where its bit is set
  int addend_sub2;       // 1: addend is [N][ceil(OH/2)][ceil(OW/2)][Cout], added at even (oh, ow) only; others get none
  const float* bias;     // optional [Cout]
  float* stats;          // optional [ceil(M/128)][2][Cout] partial sum / sum-of-squares of rounded outputs
  // fused BatchNorm-backward pass 1 (data-gradient launches only; enabled by bnb_y != nullptr)
  const bf16_t* bnb_y;     // BN input (conv output) of the layer whose output gradient this launch produces
  const bf16_t* bnb_mask;  // post-activation tensor for the ReLU mask, or nullptr: mask = bnb_y*scale+shift > 0
  const float *bnb_mean, *bnb_invstd, *bnb_scale, *bnb_shift;
  int bnb_relu;
  int relu;              // EPI 0 only: clamp the sum at zero before rounding (inference epilogue)
  bf16_t* gelu_out;      // EPI 0, optional second output laid out like out: gelu(rounded out)   (Mlp fc1 forward)
  int gelu_inplace;      // EPI 0: out itself receives gelu(rounded result) (forward passes that keep nothing for backward)
  const bf16_t* gelu_z;  // EPI 0, optional, laid out like out: out = rounded result * gelu'(gelu_z)  (Mlp fc2 data gradient)
  int N, IH, IW, Cin;
  int OH, OW, Cout;
  int P, Q, M;           // output sub-grid and row count N*P*Q
  int ostr, ooff_h, ooff_w, istr;
  int ntaps, Ktot;       // active taps; filter row length (elements)
  int KW, pad;           // regular-tap rule of the general (Cin % 64 != 0) path: dh = tap_sign*(r - pad)
  int tap_sign, regular_taps;
  int stem7;             // ResNet stem on the rgb4 layout (conv_igemm.hip KMODE 3)
  int ksteps, ntiles_n;  // filled by the launcher
  FastDiv divPQ, divQ, divCin, divKW;   // filled by the launcher
  short dh[ICAMD_MAX_TAPS], dw[ICAMD_MAX_TAPS], wtap[ICAMD_MAX_TAPS];
};
int icamd_igemm_launch(IgemmParams& p, hipStream_t stream);
int icamd_igemm_pick_bn(int Cout);

// 3x3 / stride 1 / pad 1 convolution, input tile staged once per 64-channel slice (conv3x3_halo.hip)
struct Halo3x3Params {
  const bf16_t* in;      // [N][H][W][C]
  const bf16_t* wt;      // [Cout][3][3][C]
  bf16_t* out;           // [N][H][W][Cout]
  const float* bias;     // optional [Cout]
  float* stats;          // optional [ceil(M/128)][2][Cout]
  int relu;
  int flip;              // 1: mirrored taps (data gradient with the transposed filter)
  int N, H, W, C, Cout;
  int M, ntiles_n;       // filled by the launcher
  FastDiv divHW, divW;
};
bool icamd_halo3x3_wanted(int N, int H, int W, int C, int Cout);
int icamd_halo3x3_launch(Halo3x3Params& p, hipStream_t stream);

// Dense NT GEMM for big pointwise problems: out[m][n] = sum_k A[m][k] * B[n][k] (+ bias[n]) (+ addend[m][n])
struct GemmNtParams {
  const bf16_t* A;       // [M][K]
  const bf16_t* B;       // [N][K]
  bf16_t* out;           // [M][N]
  const bf16_t* addend;  // optional [M][N]
  const unsigned char* addend_bits;   // optional, 1 bit per addend element (full-size addend only): counted where set
  const float* bias;     // optional [N]
  float* stats;          // optional [ceil(M/128)][2][N]: per-channel sum / sum of squares of the rounded outputs, one row
                         // per 256-row tile at row tile_m, zeros in the rows no tile owns
  int M, N, K;
  int sub2_h, sub2_w;    // > 0: rows are pixels of [.][sub2_h][sub2_w]; addend is [.][ceil(h/2)][ceil(w/2)][N], added at even (h, w)
  FastDiv divHW, divW;   // filled by the launcher when sub2_h > 0
  int relu;              // clamp at zero before rounding
  bf16_t* gelu_out;      // optional second output [M][N]: gelu(rounded out)
  int gelu_inplace;      // out itself receives gelu(rounded result)
  const bf16_t* gelu_z;  // optional [M][N]: out = rounded result * gelu'(gelu_z)
  int ntiles_n;          // filled by the launcher
  int group_n;           // 8-phase kernel: n-tiles per group of its tile order (filled by the launcher)
  int out_policy;        // cache policy of the output stores: 0 plain, 1 sc1 (the line leaves the XCD's L2), 2 nt (filled by the launcher)
};
int icamd_gemm_nt_launch(GemmNtParams& p, hipStream_t stream);
// true when the 256x256-tile kernel is expected to beat the 128x128 implicit-GEMM kernel for this problem
bool icamd_gemm_nt_wanted(long long M, int N, int K);

// Pointwise convolution with the filter resident in registers (conv1x1_resident.hip): out[m][n] = sum_k A[m][k] * B[n][k]
struct PwResidentParams {
  const bf16_t* A;       // [M][K]
  const bf16_t* B;       // [N][K]
  bf16_t* out;           // [M][N]
  const bf16_t* addend;  // optional [M][N] (or the even-grid form, see sub2_h)
  const unsigned char* addend_bits;   // optional 1 bit per addend element
  float* stats;          // optional [ceil(M/128)][2][N]; one partial row per workgroup, the other rows zero
  int M, N, K;
  int sub2_h, sub2_w;    // > 0: addend is [.][ceil(h/2)][ceil(w/2)][N], added at even (h, w)
  FastDiv divHW, divW;   // filled by the launcher
  int rows_per_split, ntiles_n;   // filled by the launcher
  int xcd_groups;                 // filled by the launcher: XCD-aware workgroup -> (row range, channel tile) order
  int nt_loads;                   // filled by the launcher: non-temporal LDS-DMA for the activation streams
  // "ext" launches (ConvNeXt's dim-96 Linear layers: K = 96 runs as four 32-wide k-steps, the last one against zero filter
  // columns): A rows are lda elements apart and only Ktrue of the K = 128 staged columns are real
  // "bnred" launches (residual data gradient whose output is the output gradient of the PREVIOUS block's last BatchNorm):
  // the epilogue gates the result with that block's ReLU mask bits, stores g and accumulates sum g and sum g * bn_y per
  // channel into one partial row per workgroup of bn_part[ceil(M/128)][2][N] (other rows zero); the finalize forms
  // sum g * xhat = invstd * (sum g*y - mean * sum g) in fp64
  const bf16_t* bn_y;             // [M][N] raw conv output the BatchNorm normalised
  const unsigned char* bn_bits;   // 1 bit per element: the block output was > 0
  float* bn_part;
  int lda, Ktrue;                 // 0: K
  const float* bias;              // optional [N]
  int relu;                       // inference epilogue (EPI instantiations): out = relu(acc + bias + addend)
  bf16_t* gelu_out;               // optional second output: gelu(rounded out)
  int gelu_inplace;               // out itself receives gelu(rounded result)
  const bf16_t* gelu_z;           // optional [M][N]: out = rounded result * gelu'(gelu_z)
  // round 5, stride-2 pointwise forward (ResNet's projection shortcuts): output row m = (n, oh, ow) of [.][gat_oh][gat_ow] reads
  // input row (n, 2 oh, 2 ow) of [.][gat_ih][gat_iw]; 0: rows are read in order
  int gat_oh, gat_ow, gat_ih, gat_iw;
  FastDiv gdivHW, gdivW;          // filled by the launcher when gat_ow > 0
};
bool icamd_pw_resident_wanted(long long M, int N, int K, bool with_addend = false);
bool icamd_pw_resident_epi_wanted();   // ICAMD_PW_RESIDENT_EPI=0: evaluate()'s pointwise layers stay on conv_igemm (A/B, tests)
// the ext form: plain pointwise problems with K = 96 (bias / GELU epilogues allowed, no addend / statistics)
bool icamd_pw_resident_ext_wanted(long long M, int N, int K);
int icamd_pw_resident_launch(PwResidentParams& p, hipStream_t stream);
int icamd_pw_resident_ext_launch(PwResidentParams& p, hipStream_t stream);
// the bnred form: full-size addend (optionally gated by its own mask bits), (K, N) in {(64, 256), (128, 512), (256, 1024)}
bool icamd_pw_resident_bnred_wanted(long long M, int N, int K);
int icamd_pw_resident_bnred_launch(PwResidentParams& p, hipStream_t stream);

// ResNet stem forward with the filter resident in registers (conv_stem.hip)
struct StemParams {
  const bf16_t* x;   // [N][H][W+8][4]
  const bf16_t* w;   // [64][8][8][4]
  bf16_t* y;         // [N][OH][OW][64]
  float* stats;      // optional [ceil(M/128)][2][64]
  int N, H, W, OH, OW;
  const float* bias; // optional [64]: inference epilogue (folded BatchNorm shift)
  int relu;
};
struct StemWgradParams {
  const bf16_t* x;   // [N][H][W+8][4]
  const bf16_t* dy;  // [N][OH][OW][64]
  float* slab;       // [S][64][256]
  int N, H, W, OH, OW;
};
bool icamd_stem_resident_wanted(int N, int H, int W, int Cout);
int icamd_stem_wgrad_resident_splits(int N, int H);
int icamd_stem_wgrad_resident_launch(StemWgradParams& p, int S, hipStream_t stream);
int icamd_stem_resident_launch(StemParams& p, hipStream_t stream);

// Fused backward of "pointwise convolution -> BatchNorm" (conv_fused_bwd.hip): BatchNorm-backward apply + data gradient + weight
// gradient of the convolution in one pass over g and y
struct FusedBwdParams {
  const bf16_t* g;      // [M][CO] masked output gradient of the BatchNorm
  const bf16_t* y;      // [M][CO] BatchNorm input (raw convolution output)
  const bf16_t* x;      // [M][CI] convolution input
  const bf16_t* wt;     // [CI][CO] transposed filter
  bf16_t* dx;           // [M][CI]
  float* slab;          // [S][CO][CI] partial filter gradients
  const float *mean, *invstd, *scale, *c1, *c2;   // [CO]: batch statistics, gamma * invstd, mean g, mean g * xhat
  int M, CI, CO;
  int S, rows_per_split, nslices, xcd_pairs, nt;  // filled by the launcher
};
bool icamd_conv1x1_bn_bwd_fused_wanted(long long M, int Cin, int Cout);
void icamd_conv1x1_bn_bwd_fused_plan(int M, int Cin, int* S, int* rows_per_split);
int icamd_conv1x1_bn_bwd_fused_launch(FusedBwdParams& p, hipStream_t stream);

// Fused forward across a bottleneck boundary (conv_fused_fwd.hip): out = relu(y * scale + shift + residual) with its mask bits, and
// y1 = out * w^T (the next block's 1x1 convolution) with that layer's BatchNorm statistics, in one pass
struct FusedFwdParams {
  const bf16_t* y;        // [M][K] raw convolution output the BatchNorm normalises
  const bf16_t* res;      // [M][K] residual: an activation, or (res_scale != nullptr) the raw shortcut convolution output
  const float *scale, *shift, *res_scale, *res_shift;   // [K]
  bf16_t* out;            // [M][K]
  unsigned char* maskbits;   // [M * K / 8]: bit = [out > 0]
  const bf16_t* w;        // [N][K] filter of the next block's conv1
  bf16_t* y1;             // [M][N]
  float* stats;           // optional [ceil(M / 128)][2][N]
  int M, K, N;
  int S, rows_per_split, nt;  // filled by the launcher
};
bool icamd_bn_apply_conv1x1_fused_wanted(long long M, int K, int N);
int icamd_bn_apply_conv1x1_fused_launch(FusedFwdParams& p, hipStream_t stream);

// Weight-gradient problem: dw[co][t][ci] = sum_m dy[m][co] * x[n, p*stride+r-pad, q*stride+s-pad, ci]
struct WgradParams {
  const bf16_t* x;    // [N, IH, IW, Cin]
  const bf16_t* dy;   // [N, OH, OW, Cout]
  float* slab;        // [S][Cout][Ktot] partial sums
  float* bias_slab;   // optional [S][Cout] partial column sums of dy (bias gradient)
  int N, IH, IW, Cin, OH, OW, Cout;
  int KH, KW, stride, pad;
  int M, Ktot;        // N*OH*OW ; KH*KW*Cin
  int S, rows_per_split;  // split of the m reduction
  int ntiles_k, ntiles_c;
  int pointwise;      // 1x1, stride 1, pad 0: the gather is the identity
  int xcd_chunk;      // workgroup order: 1 = contiguous chunks of the split-major list per XCD (conv_wgrad.hip wgrad_block_order)
  int stem7;          // ResNet stem on the rgb4 layout: k = row*32 + pixel*4 + channel, IW = padded pitch, Cin = 4
  FastDiv divHW, divW, divCin, divKW;
};
int icamd_wgrad_launch(WgradParams& p, hipStream_t stream);
void icamd_wgrad_plan(int M, int Cout, int Ktot, int* S, int* rows_per_split);
void icamd_wgrad_tile(long long M, int Ktot, int Cout, int* bmk, int* bnc);
// halo-staged 3x3 / stride 1 weight gradient (conv_wgrad.hip): own pixel split; slab layout as the other kernels
bool icamd_wgrad_halo_wanted(int KH, int KW, int stride, int pad, int H, int W, int Cin, int Cout, long long M);
void icamd_wgrad_halo_plan(int M, int Cin, int Cout, int* S, int* rows_per_split);
int icamd_wgrad_halo_launch(WgradParams& p, hipStream_t stream);   // output-tile sides the launcher will use
// out[i] = (accumulate ? out[i] : 0) + sum over S slabs of slab[s][i], fixed order; n % 4 == 0
int icamd_slab_reduce_launch(const float* slab, float* out, long long n, int S, int accumulate, hipStream_t stream,
                             int stem7_mask = 0);
