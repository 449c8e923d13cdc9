// Shared declarations of the Winograd F(m x m, 3 x 3) path (conv_wino.hip: transforms, GEMMs, layer calls, weight gradient;
// conv_wino_chain.hip: runs of frozen layers whose output transform feeds the next input transform without the tensor
// in between).  gfx950 only.
#pragma once
#include <stdlib.h>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

// --------------------------------------------------------------------------------------------
// transform matrices
template <int M> struct WinoMat;
template <> struct WinoMat<2> {
  static constexpr int A = 4;
  __host__ __device__ static constexpr float BT(int i, int j) {
    constexpr float t[4][4] = {{1, 0, -1, 0}, {0, 1, 1, 0}, {0, -1, 1, 0}, {0, 1, 0, -1}};
    return t[i][j];
  }
  __host__ __device__ static constexpr float G(int i, int j) {
    constexpr float t[4][3] = {{1, 0, 0}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0, 0, 1}};
    return t[i][j];
  }
  __host__ __device__ static constexpr float AT(int i, int j) {
    constexpr float t[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};
    return t[i][j];
  }
};
// F(4x4, 3x3): Toom-Cook on the points {0, +-5/8, +-3/2, inf} [r4] instead of the textbook {0, +-1, +-2, inf}.  Same sparsity
// pattern (symmetric pairs, 0 and infinity), hence the same instruction count in every transform kernel -- they are all driven by
// these tables -- but 2.2x less rounding error per layer on the forward / data-gradient pass and on the Winograd-form weight gradient
// (tools/wino_points.py: fp32 pipeline vs fp64 over 30 symmetric candidates; rms 0.66e-6 vs 1.45e-6 of the output rms, max 1.0e-6
// vs 3.6e-6 of the output max at 256 - 512 channels): the large powers 4, 8, 16 in A^T / B^T of the textbook points amplify the
// cancellation in the transforms.  All entries of B^T and A^T are dyadic (exact in fp32); the rows of B^T are scaled by powers of two to
// max |entry| in [1, 2) (V stays at the activations' magnitude), the inverse factors live in G.
template <> struct WinoMat<4> {
  static constexpr int A = 6;
  __host__ __device__ static constexpr float BT(int i, int j) {
    constexpr float t[6][6] = {{225.f / 512, 0, -169.f / 128, 0, 1.f / 2, 0},   {0, -45.f / 64, -9.f / 8, 5.f / 16, 1.f / 2, 0},
                               {0, 45.f / 64, -9.f / 8, -5.f / 16, 1.f / 2, 0}, {0, -75.f / 128, -25.f / 64, 3.f / 2, 1, 0},
                               {0, 75.f / 128, -25.f / 64, -3.f / 2, 1, 0},     {0, 225.f / 512, 0, -169.f / 128, 0, 1.f / 2}};
    return t[i][j];
  }
  __host__ __device__ static constexpr float G(int i, int j) {
    constexpr float t[6][3] = {{512.f / 225, 0, 0},
                               {-4096.f / 2975, -512.f / 595, -64.f / 119},
                               {-4096.f / 2975, 512.f / 595, -64.f / 119},
                               {128.f / 1071, 64.f / 357, 32.f / 119},
                               {128.f / 1071, -64.f / 357, 32.f / 119},
                               {0, 0, 2}};
    return t[i][j];
  }
  __host__ __device__ static constexpr float AT(int i, int j) {
    constexpr float t[4][6] = {{1, 1, 1, 1, 1, 0},
                               {0, 5.f / 8, -5.f / 8, 3.f / 2, -3.f / 2, 0},
                               {0, 25.f / 64, 25.f / 64, 9.f / 4, 9.f / 4, 0},
                               {0, 125.f / 512, -125.f / 512, 27.f / 8, -27.f / 8, 1}};
    return t[i][j];
  }
};

// Virtual channel concatenation (the U-Net decoder's cat([branch-1 skip, branch-2 skip, upsampled], dim=1)): up to three
// (N, c[i], H, W) tensors stand for ONE (N, sum c[i], H, W) operand, c[i] % 32 == 0.  n == 0: plain single tensor.
struct WinoCat {
  const float* p[3];
  int c[3];
  int n;
};
// tensor and its channel count holding concatenated channel ch; ch becomes the channel inside that tensor
__device__ __forceinline__ const float* wino_cat_pick(const WinoCat& k, int& ch, int& chans) {
  int s = 0;
  if (k.n > 1 && ch >= k.c[0]) { ch -= k.c[0]; s = 1; if (k.n > 2 && ch >= k.c[1]) { ch -= k.c[1]; s = 2; } }
  chans = k.c[s];
  return k.p[s];
}

struct WinoInArgs {
  const float* x;             // SRC 0/1: (N, C, H, W); SRC 2: pooled gradient (N, C, Hp, Wp)
  const float* mask;          // SRC 1: ReLU output, same shape as x ...
  const unsigned short* mbits;  // ... or [r3] its sign as 16 bits per (n, c, 4 x 4 tile), bit 4 i + j = [y(4 ty + i, 4 tx + j) > 0],
                              // written by the forward pass's output transform (1 / 32 of the mask traffic)
  const unsigned char* code;  // SRC 2: argmax code of the pooled tensor
  float* V;                   // [xi][Q][T][32]
  int N, C, H, W, Hp, Wp, TH, TW, Q;
  long long T;
  int exp;                    // diagnostics (FCD_WINO_IN_EXP): 2 = no V stores, 4 = no source loads
  int xcd;                    // 1: blocks renumbered so that each XCD (own L2) walks a contiguous range
  WinoCat cat;                // plain source of the rolling kernel only: x = cat(cat.p[...]) (cat.n > 0)
  // SRC 3 (rolling kernel only): x is the INPUT of a train-mode BatchNorm + ReLU whose only consumer is this convolution
  // (reference Module.py:25-31: conv -> BN -> ReLU -> conv): the strip is normalised and rectified on its way into LDS,
  // relu(x * aff_scale[(n / aff_ng) C + c] + aff_shift[...]) -- the arithmetic of bn_act_apply_kernel -- and the activation
  // never exists as a tensor.  Padding stays zero (the affine map is applied to in-range elements only).
  const float* aff_scale;
  const float* aff_shift;
  int aff_ng;
};

// Workgroups are handed to the 8 XCDs round-robin in launch order, so neighbouring strips of one plane --
// which share their halo rows -- would land on 8 different L2s and fetch the shared rows from HBM again.
// Renumber: XCD j (launch ids j, j+8, ...) takes the contiguous range of blocks [start_j, start_j + count_j).
__device__ __forceinline__ unsigned xcd_contiguous_id(unsigned lin, unsigned total) {
  const unsigned j = lin & 7u, i = lin >> 3, q8 = total >> 3, r8 = total & 7u;
  return j * q8 + (j < r8 ? j : r8) + i;
}

struct WinoGemmArgs {
  const float* A;   // row m of batch b at A + b * a_batch + m * a_ld, stage q at + q * 32 floats
  const float* B;   // row n of batch b at B + b * b_batch + n * b_ld, stage q at + q * b_adv floats
  float* C;         // [split][batch][M][N]
  int M, N, Kc, m_tiles, n_tiles, xcd_remap;
  long long a_ld, a_batch, b_ld, b_adv, b_batch;
  long long a_adv;  // split kernel <2, 2>: elements from stage q to stage q + 1 of an A row (0: 32, rows contiguous along the reduction)
  int stages_per_split;   // blockIdx.z = split of the reduction: stages [z * sps, min((z + 1) * sps, Kc / 32))
  const unsigned short* As;   // split kernel: bf16 planes (high, middle, low part) of A, plane p at As + p * as_plane,
  long long as_plane;         // each laid out like A (a_ld, a_batch in elements)
  int batches, xb;        // blockIdx.y = group of xb consecutive batches (transform positions) run by ONE workgroup as a
                          // single software pipeline: the first slabs of batch b + 1 are in flight while batch b's last
                          // MFMAs run and its C tile is stored -- no pipeline refill per batch
  int c_blk;              // split kernels: C in MFMA-native 32 x 32 blocks [xi][M/32][c_tblk][half*4 + r/4][32 cols][r%4] (one
  int c_mblk, c_tblk;     // dwordx4 store per four accumulator registers: 16 / 32 stores per lane and batch instead of 64 / 128);
  long long c_batch;      // c_batch = floats per xi.  Read back by wino_output_blk_kernel.
  int bt;                 // split kernel, weight gradient: B is the FORWARD pass's V [xi][N / 32][bt_T][32] (GEMM row n = channel,
  long long bt_T;         // reduction = tile index): b_batch = elements per xi, the reduction runs to bt_T (rows clamped), b_ld / b_adv unused
  unsigned long long* tbuf;   // YG_TIME builds: [workgroup][wave][8] cycle sums
};

// Workgroup -> (tile v, transform-position group by, reduction split bz) of the split GEMM kernels (WinoGemmArgs.xcd_remap):
//   0  the launch geometry as it is;
//   1  many tiles per position group: each XCD (own L2; workgroups go to the 8 XCDs round-robin in launch order) walks a CONTIGUOUS
//      range of the tiles of a group, so the workgroups resident on it share their operand panels' neighbours;
//   2  [r5] few tiles per group (the Segmentor's deep layers: 8 tiles, filters of 450 MB; every split-K weight-gradient GEMM: 1 - 8 tiles):
//      ALL tiles of a group go to ONE XCD -- groups are dealt to the XCDs eight at a time, the tiles of a group sit 8 apart in launch
//      order -- so that the operand panels the tiles share (A: one per row tile, read by every column tile and vice versa) come
//      from HBM once per group instead of once per tile: with 4 x 2 tiles that is 0.6 instead of 1.5 GB for the 2048 -> 1024 layer.
__device__ __forceinline__ void wino_gemm_block(int mode, unsigned& v, unsigned& by, unsigned& bz) {
  const unsigned gx = gridDim.x, gy = gridDim.y;
  if (mode == 2) {
    const unsigned L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned P = gy * gridDim.z, P8 = P & ~7u;
    unsigned p, t;
    if (L < P8 * gx) {
      const unsigned r = L >> 3;
      t = r % gx;
      p = (r / gx) * 8 + (L & 7u);
    } else {                      // the last P % 8 groups: plain order
      const unsigned Lt = L - P8 * gx, rem = P - P8;
      p = P8 + Lt % rem;
      t = Lt / rem;
    }
    v = t; by = p % gy; bz = p / gy;
    return;
  }
  by = blockIdx.y; bz = blockIdx.z;
  const unsigned b = blockIdx.x;
  if (mode == 1) {
    const unsigned q8 = gx >> 3, r8 = gx & 7u, xcd = b & 7u;
    v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
  } else {
    v = b;
  }
}

struct WinoOutArgs {
  const float* Mb;      // [xi][K][T]
  const float* bias;
  float* y;             // (N, K, P, Q) or NULL when pooling
  float* pool_y;        // (N, K, P/2, Q/2)
  unsigned char* code;
  int K, P, Q, TH, TW, relu;
  long long T;
  WinoCat cat;          // data gradient of a virtually concatenated input: channel k of dx goes to its own tensor (cat.n > 0)
  int tblk;             // wino_output_blk_kernel: Mb in the split GEMM's 32 x 32 blocks (WinoGemmArgs.c_blk): blocks per
  long long xs_blk;     // row block, floats per xi
  unsigned short* bits; // [r3] m = 4, relu, no pooling: sign of the outputs, 16 bits per (n, k, tile) (WinoInArgs.mbits)
  const unsigned short* gate;  // [r5] m = 4, data gradient: sign bits (same layout) of the forward activation this gradient belongs to --
                        // outputs are zeroed where it was <= 0 (the ReLU in front of the layer, applied here for a consumer that cannot)
  double* bn_part;      // [r3] wino_output_blk_kernel: per-workgroup {sum y, sum y^2} of each channel for the BatchNorm that follows
  int bn_bpg;           //      (reference Module.py:25-31: Conv2d -> BatchNorm2d): part[((g K + k) bn_bpg + block in group) 3 + {0, 1}],
                        //      bn_bpg = workgroups per sample group (a workgroup's 256 tiles never straddle two groups: host-checked)
};

// fused output -> input transform between two F(4x4) layers (conv_wino_chain.hip)
struct WinoOiArgs {
  const float* Mb;              // products of the producing layer in the split GEMM's 32 x 32 blocks (WinoOutArgs.tblk / xs_blk)
  long long xs_blk;
  int tblk;
  const float* bias;            // forward: bias of the producing layer (NULL: none)
  int relu;                     // forward: ReLU between the layers
  unsigned short* bits_out;     // forward: sign bits of the activation between the layers, 16 per (n, k, tile) (NULL: not kept)
  const unsigned short* gate;   // data gradient: sign bits of the forward activation at this boundary; the gradient is zeroed where it was <= 0
  float* V;                     // [xi][K / 32][T][32]: transformed input of the consuming layer
  int N, K, H, W, TH, TW, Q;    // the tensor between the layers: (N, K, H, W), 4 x 4 tiles, Q = K / 32
  long long T;
};
int wino_oi_ok(int K, int H, int W);
void wino_oi_launch(const WinoOiArgs& a, hipStream_t st);

struct WinoPlan {
  int m, A2, rows, red, Kc, Q, TH, TW;
  long long T;
  size_t v_bytes, m_bytes;
};


// --------------------------------------------------------------------------------------------
// tile transforms shared by the stand-alone transform kernels and the fused output -> input kernel of conv_wino_chain.hip.
// ONE definition with a PINNED operation order: contraction is switched off inside these functions and every fused multiply-add
// is written out (first non-zero term a product, coefficient +-1 an add / subtract, every other term one fmaf), so a tile
// comes out bit-identical from every kernel that inlines them -- left to -ffp-contract=fast the same source gave different
// mul / add / fma mixes in different kernels (12 of the 192 fmas of a B^T d B became mul + sub in the one-strip input kernel).
#define WINO_TERM(S, FIRST, C, X)                                        \
  {                                                                      \
    const float c_ = (C);                                                \
    if (c_ != 0.f) {                                                     \
      if (FIRST) { S = (c_ == 1.f) ? (X) : ((c_ == -1.f) ? -(X) : c_ * (X)); FIRST = false; } \
      else if (c_ == 1.f) S = S + (X);                                   \
      else if (c_ == -1.f) S = S - (X);                                  \
      else S = __builtin_fmaf(c_, (X), S);                               \
    }                                                                    \
  }
// o = A^T M A + b, optional ReLU
template <int MM>
__device__ __forceinline__ void wino_out_tile(const float (&mv)[WinoMat<MM>::A][WinoMat<MM>::A], float b, int relu,
                                              float (&o)[MM][MM]) {
#pragma clang fp contract(off)
  constexpr int A = WinoMat<MM>::A;
  float t1[MM][A];   // A^T M
#pragma unroll
  for (int i = 0; i < MM; ++i)
#pragma unroll
    for (int j = 0; j < A; ++j) {
      float s = 0.f;
      bool first = true;
#pragma unroll
      for (int q = 0; q < A; ++q) WINO_TERM(s, first, WinoMat<MM>::AT(i, q), mv[q][j])
      t1[i][j] = s;
    }
#pragma unroll
  for (int i = 0; i < MM; ++i)
#pragma unroll
    for (int j = 0; j < MM; ++j) {
      float s = 0.f;
      bool first = true;
#pragma unroll
      for (int q = 0; q < A; ++q) WINO_TERM(s, first, WinoMat<MM>::AT(j, q), t1[i][q])
      s = s + b;
      if (relu) s = s > 0.f ? s : 0.f;
      o[i][j] = s;
    }
}
// first half of V = B^T d B: t1 = B^T d; the callers finish with wino_in_col (their stores interleaved with the sums)
template <int MM>
__device__ __forceinline__ void wino_in_rows(const float (&d)[WinoMat<MM>::A][WinoMat<MM>::A],
                                             float (&t1)[WinoMat<MM>::A][WinoMat<MM>::A]) {
#pragma clang fp contract(off)
  constexpr int A = WinoMat<MM>::A;
#pragma unroll
  for (int i = 0; i < A; ++i)
#pragma unroll
    for (int j = 0; j < A; ++j) {
      float s = 0.f;
      bool first = true;
#pragma unroll
      for (int k = 0; k < A; ++k) WINO_TERM(s, first, WinoMat<MM>::BT(i, k), d[k][j])
      t1[i][j] = s;
    }
}
template <int MM>
__device__ __forceinline__ float wino_in_col(const float (&t1)[WinoMat<MM>::A][WinoMat<MM>::A], int i, int j) {
#pragma clang fp contract(off)
  constexpr int A = WinoMat<MM>::A;
  float s = 0.f;
  bool first = true;
#pragma unroll
  for (int k = 0; k < A; ++k) WINO_TERM(s, first, WinoMat<MM>::BT(j, k), t1[i][k])
  return s;
}
