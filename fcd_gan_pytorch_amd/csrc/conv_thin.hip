// Data gradient of a 3x3 / stride-1 / pad-1 convolution whose INPUT has only 1..4
// channels (the first VGG layer, Loss.py:25: 64 -> 3 in the backward direction).
// On the MFMA path this layer pads 3 output channels to a 32-row tile (>90 % of the
// matrix work wasted) -- it is purely HBM-bound (it reads the 64-channel gradient and the
// ReLU mask once), so it gets a VALU kernel instead:
//   dx[n,c,h,w] = sum_k sum_{r',s'} dy'[n,k,h+r'-1,w+s'-1] * wf[(k,r',s')][c]
// with dy' = dy * [relu_out > 0] (optional) and wf the mode-1 packed (flipped) filter.
// Workgroup = 16 x 64 output pixels; dy' is staged through LDS in chunks of 8 channels
// (18 x 66 halo tile each), every thread owns a 1x4 pixel strip and keeps its 6-wide
// input window in registers across the three taps of a row; filter taps are wave-uniform
// scalar loads.
#include "common.h"
#include <stdlib.h>
#include <algorithm>

#define TH_ROWS 16
#define TH_COLS 64
#define TH_KC 8
#define TH_PH (TH_ROWS + 2)
#define TH_PW (TH_COLS + 2)
#define TH_PWP (TH_PW + 1)

template <int CS>
__global__ __launch_bounds__(256) void conv3x3_dgrad_thin_kernel(const float* __restrict__ dy,
                                                                 const float* __restrict__ mask,
                                                                 const float* __restrict__ wf, float* __restrict__ dx,
                                                                 int K, int H, int W, int Cpad, int tiles_w) {
  __shared__ float tile[TH_KC * TH_PH * TH_PWP];
  const int tid = threadIdx.x;
  const int n = blockIdx.y;
  const int th = blockIdx.x / tiles_w, tw = blockIdx.x % tiles_w;
  const int h0 = th * TH_ROWS, w0 = tw * TH_COLS;
  const int row = tid >> 4, col = (tid & 15) * 4;     // 16 rows x 16 strips of 4 pixels
  float acc[CS][4];
#pragma unroll
  for (int c = 0; c < CS; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[c][j] = 0.f;

  const size_t img = (size_t)n * K * H * W;
  for (int k0 = 0; k0 < K; k0 += TH_KC) {
    __syncthreads();
    for (int i = tid; i < TH_KC * TH_PH * TH_PW; i += 256) {
      const int kk = i / (TH_PH * TH_PW), rem = i % (TH_PH * TH_PW);
      const int ph = rem / TH_PW, pw = rem % TH_PW;
      const int h = h0 + ph - 1, w = w0 + pw - 1, k = k0 + kk;
      float v = 0.f;
      if (k < K && h >= 0 && h < H && w >= 0 && w < W) {
        const size_t off = img + ((size_t)k * H + h) * W + w;
        v = dy[off];
        if (mask && !(mask[off] > 0.f)) v = 0.f;
      }
      tile[(kk * TH_PH + ph) * TH_PWP + pw] = v;
    }
    __syncthreads();
    // one channel at a time (not unrolled: keeps only 9*CS filter scalars live in SGPRs); rows of
    // the packed filter beyond K are zero (packed height is a multiple of 8), no guard needed
#pragma unroll 1
    for (int kk = 0; kk < TH_KC; ++kk) {
      const float* wk = wf + (size_t)(k0 + kk) * 9 * Cpad;     // wave-uniform -> scalar loads
      float wv[9][CS];
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
        for (int c = 0; c < CS; ++c) wv[t9][c] = wk[t9 * Cpad + c];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float* t = tile + (kk * TH_PH + row + r) * TH_PWP + col;
        float win[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) win[j] = t[j];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
          for (int c = 0; c < CS; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[c][j] = fmaf(win[j + s], wv[r * 3 + s][c], acc[c][j]);
      }
    }
  }
  const int h = h0 + row;
  if (h < H) {
#pragma unroll
    for (int c = 0; c < CS; ++c) {
      float* o = dx + (((size_t)n * CS + c) * H + h) * W + w0 + col;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (w0 + col + j < W) o[j] = acc[c][j];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Same computation, W % 4 == 0: the 64-wide interior of every halo row is fetched with float4
// loads (9 per thread and chunk instead of 37 dword loads with index arithmetic), and the loads
// of chunk i+1 stay in flight while chunk i is consumed from LDS (values are only touched when
// they are stored to LDS after the compute block).
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// BITS: the ReLU mask arrives as one byte per 1 x 4 pixel strip (bit j = [relu_out > 0] of pixel 4 s + j, written
// by conv3x3_fwd_thin_kernel) instead of the fp32 activation: 1/16 of the mask traffic.
template <int CS, bool BITS>
__global__ __launch_bounds__(256) void conv3x3_dgrad_thin_v4_kernel(const float* __restrict__ dy,
                                                                    const float* __restrict__ mask,
                                                                    const float* __restrict__ wf, float* __restrict__ dx,
                                                                    int K, int H, int W, int Cpad, int tiles_w) {
  __shared__ float tile[TH_KC * TH_PH * TH_PWP];
  const int tid = threadIdx.x;
  const int n = blockIdx.y;
  const int th = blockIdx.x / tiles_w, tw = blockIdx.x % tiles_w;
  const int h0 = th * TH_ROWS, w0 = tw * TH_COLS;
  const int row = tid >> 4, col = (tid & 15) * 4;
  const bool has_mask = mask != nullptr;
  float acc[CS][4];
#pragma unroll
  for (int c = 0; c < CS; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[c][j] = 0.f;

  // interior items: 8 channels x 18 rows x 16 float4 = 2304 = 9 per thread
  constexpr int NV = TH_KC * TH_PH * (TH_COLS / 4) / 256;
  static_assert(NV * 256 == TH_KC * TH_PH * (TH_COLS / 4), "interior split");
  int v_goff[NV], v_loff[NV], v_kk[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = tid + i * 256;
    const int rowid = idx >> 4, v4 = idx & 15;
    const int kk = rowid / TH_PH, ph = rowid % TH_PH;
    const int h = h0 + ph - 1, w = w0 + v4 * 4;
    const bool ok = h >= 0 && h < H && w < W;
    v_kk[i] = ok ? kk : (1 << 20);                       // never < remaining channels
    v_goff[i] = ok ? (kk * H + h) * W + w : 0;
    v_loff[i] = (kk * TH_PH + ph) * TH_PWP + 1 + v4 * 4;
  }
  // halo columns: 8 x 18 rows x 2 = 288 scalars (threads 0..255 one, threads 0..31 a second)
  int s_goff[2], s_loff[2], s_kk[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + i * 256;
    const int rowid = idx >> 1, side = idx & 1;
    const int kk = rowid / TH_PH, ph = rowid % TH_PH;
    const int h = h0 + ph - 1, w = side ? w0 + TH_COLS : w0 - 1;
    const bool ok = idx < TH_KC * TH_PH * 2 && h >= 0 && h < H && w >= 0 && w < W;
    s_kk[i] = ok ? kk : (1 << 20);
    s_goff[i] = ok ? (kk * H + h) * W + w : 0;
    s_loff[i] = idx < TH_KC * TH_PH * 2 ? (kk * TH_PH + ph) * TH_PWP + (side ? TH_COLS + 1 : 0) : -1;
  }

  const size_t img = (size_t)n * K * H * W;
  const size_t plane_chunk = (size_t)TH_KC * H * W;
  f32x4_t dv[NV], mv[NV];
  unsigned mb[NV];                 // BITS: mask byte of the strip
  float ds_[2], ms_[2];
  const unsigned char* mbits = (const unsigned char*)mask;

#define THIN_LOAD(K0)                                                                     \
  {                                                                                       \
    const float* dsrc = dy + img + (size_t)((K0) / TH_KC) * plane_chunk;                  \
    const float* msrc = (has_mask ? mask : dy) + img + (size_t)((K0) / TH_KC) * plane_chunk; \
    const int kleft = K - (K0);                                                           \
    _Pragma("unroll") for (int i = 0; i < NV; ++i) {                                      \
      const int off = v_kk[i] < kleft ? v_goff[i] : 0;                                    \
      dv[i] = *(const f32x4_t*)(dsrc + off);                                              \
      if (BITS) mb[i] = mbits[(img + (size_t)((K0) / TH_KC) * plane_chunk + off) >> 2];   \
      else if (has_mask) mv[i] = *(const f32x4_t*)(msrc + off);                           \
    }                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                       \
      const int off = s_kk[i] < kleft ? s_goff[i] : 0;                                    \
      ds_[i] = dsrc[off];                                                                 \
      if (BITS) {                                                                         \
        const size_t e = img + (size_t)((K0) / TH_KC) * plane_chunk + off;                \
        ms_[i] = ((mbits[e >> 2] >> (e & 3)) & 1u) ? 1.f : 0.f;                           \
      } else if (has_mask) ms_[i] = msrc[off];                                            \
    }                                                                                     \
  }
#define THIN_STORE(K0)                                                                    \
  {                                                                                       \
    const int kleft = K - (K0);                                                           \
    _Pragma("unroll") for (int i = 0; i < NV; ++i) {                                      \
      const bool ok = v_kk[i] < kleft;                                                    \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                       \
        tile[v_loff[i] + j] = (ok && (BITS ? ((mb[i] >> j) & 1u) != 0u : (!has_mask || mv[i][j] > 0.f))) ? dv[i][j] : 0.f; \
    }                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                         \
      if (s_loff[i] >= 0)                                                                 \
        tile[s_loff[i]] = (s_kk[i] < kleft && (!has_mask || ms_[i] > 0.f)) ? ds_[i] : 0.f; \
  }

  THIN_LOAD(0)
  THIN_STORE(0)
  __syncthreads();
  for (int k0 = 0; k0 < K; k0 += TH_KC) {
    const bool have_next = k0 + TH_KC < K;
    if (have_next) THIN_LOAD(k0 + TH_KC)
#pragma unroll 1
    for (int kk = 0; kk < TH_KC; ++kk) {
      const float* wk = wf + (size_t)(k0 + kk) * 9 * Cpad;     // wave-uniform -> scalar loads
      float wv[9][CS];
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
        for (int c = 0; c < CS; ++c) wv[t9][c] = wk[t9 * Cpad + c];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float* t = tile + (kk * TH_PH + row + r) * TH_PWP + col;
        float win[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) win[j] = t[j];
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2)
#pragma unroll
          for (int c = 0; c < CS; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[c][j] = fmaf(win[j + s2], wv[r * 3 + s2][c], acc[c][j]);
      }
    }
    __syncthreads();
    if (have_next) THIN_STORE(k0 + TH_KC)
    __syncthreads();
  }
#undef THIN_LOAD
#undef THIN_STORE
  const int h = h0 + row;
  if (h < H && w0 + col < W) {
#pragma unroll
    for (int c = 0; c < CS; ++c) {
      f32x4_t o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = acc[c][j];
      *(f32x4_t*)(dx + (((size_t)n * CS + c) * H + h) * W + w0 + col) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Forward of a 3x3 / stride-1 / pad-1 convolution with 1..4 INPUT channels and > 32 output
// channels (VGG conv1_1 on single bands, Loss.py:25,52-58; Segmentor.inc on 3/4-band tiles):
//   y[n,k,h,w] = act(b[k] + sum_{c,r,s} w[k,c,r,s] * x[n,c,h+r-1,w+s-1])
// One pass over x, one over y: bound by the HBM write of y.  Every thread keeps the 3x6 input
// window of its 1x4 pixel strip in registers for all output channels; filter taps are
// wave-uniform scalar loads from the T-layout pack (conv_igemm.hip):
//   w(k, c, tap) = wp[((c >> 1) * Mpad + k) * 20 + (c & 1) * 9 + tap].
template <int CS>
__global__ __launch_bounds__(256) void conv3x3_fwd_thin_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               int K, int Mpad, int H, int W, int tiles_w, int relu,
                                                               unsigned char* __restrict__ bits) {
  __shared__ float tile[CS * TH_PH * TH_PWP];
  const int tid = threadIdx.x;
  const int n = blockIdx.y;
  const int th = blockIdx.x / tiles_w, tw = blockIdx.x % tiles_w;
  const int h0 = th * TH_ROWS, w0 = tw * TH_COLS;
  const int row = tid >> 4, col = (tid & 15) * 4;
  for (int i = tid; i < CS * TH_PH * TH_PW; i += 256) {
    const int c = i / (TH_PH * TH_PW), rem = i % (TH_PH * TH_PW);
    const int ph = rem / TH_PW, pw = rem % TH_PW;
    const int h = h0 + ph - 1, w = w0 + pw - 1;
    float v = 0.f;
    if (h >= 0 && h < H && w >= 0 && w < W) v = x[(((size_t)n * CS + c) * H + h) * W + w];
    tile[(c * TH_PH + ph) * TH_PWP + pw] = v;
  }
  __syncthreads();
  float win[CS][3][6];
#pragma unroll
  for (int c = 0; c < CS; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int j = 0; j < 6; ++j) win[c][r][j] = tile[(c * TH_PH + row + r) * TH_PWP + col + j];
  const int h = h0 + row;
  const bool row_ok = h < H && w0 + col < W;
  const bool vec_ok = (W & 3) == 0;
  float* yrow = y + ((size_t)n * K * H + h) * W + w0 + col;
#pragma unroll 2
  for (int k = 0; k < K; ++k) {
    const float b = bias ? bias[k] : 0.f;
    float a4[4] = {b, b, b, b};
#pragma unroll
    for (int c = 0; c < CS; ++c) {
      const float* wk = wp + ((size_t)(c >> 1) * Mpad + k) * 20 + (c & 1) * 9;      // wave-uniform
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2) {
          const float wv = wk[r * 3 + s2];
#pragma unroll
          for (int j = 0; j < 4; ++j) a4[j] = fmaf(win[c][r][j + s2], wv, a4[j]);
        }
    }
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) a4[j] = a4[j] > 0.f ? a4[j] : 0.f;
    }
    if (row_ok) {
      float* o = yrow + (size_t)k * H * W;
      if (bits != nullptr)        // W % 4 == 0: element index / 4 = strip index
        bits[(((size_t)n * K + k) * H + h) * (W >> 2) + ((w0 + col) >> 2)] =
            (unsigned char)((a4[0] > 0.f) | ((a4[1] > 0.f) << 1) | ((a4[2] > 0.f) << 2) | ((a4[3] > 0.f) << 3));
      if (vec_ok) {
        f32x4_t v = {a4[0], a4[1], a4[2], a4[3]};
        *(f32x4_t*)o = v;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (w0 + col + j < W) o[j] = a4[j];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// [r3] ONE input channel, 64 filters (VGG conv1_1 on single bands: 208 images of 256 x 256 per step), matrix-core form.
// The VALU kernel above spends 576 FMAs per pixel and ran at 3.2 TB/s of the gradient it reads.  Here the channel
// reduction of every tap is an MFMA and only the 3 x 3 shift-and-add stays on the VALU:
//   res_t[y][x] = sum_k wf[k][t] dy'[k][y][x]          v_mfma_f32_16x16x4_f32: rows = 9 taps (of 16), 4 channels per step
//   dx[h][w]    = sum_{r,s} res_{3r+s}[h + r - 1][w + s - 1]
// A wave owns a 64-pixel row segment: lane (l15, kq) loads float4 dy[4 i + kq][y][4 l15 ..] (+ the strip's mask byte) for
// i = 0 .. 15 -- 256 contiguous bytes per channel row -- and runs 16 x 4 MFMAs (B operand = element j of the float4: MFMA j
// covers pixels 4 l + j), after which lane (l15, kq) holds taps 4 kq .. 4 kq + 3 of its four pixels.  The workgroup (one wave
// per 64 columns, the whole image width) walks TH + 2 gradient rows; the tap planes of the last rows sit in a 4-slot LDS
// ring (one barrier per row), from which thread x sums the nine shifted values of output pixel (y - 1, x).
template <bool BITS>
__global__ __launch_bounds__(256, 2) void conv3x3_dgrad_c1_mfma_kernel(const float* __restrict__ dy,
                                                                       const unsigned char* __restrict__ bits,
                                                                       const float* __restrict__ wf, float* __restrict__ dx,
                                                                       int H, int W, int Cpad, int TH, int blocks_h) {
  constexpr int K = 64, KQ = K / 4, PITCH = 256 + 8;
  __shared__ __attribute__((aligned(16))) float res[4][9][PITCH];
  const int tid = threadIdx.x, lane = tid & 63, seg = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const int n = blockIdx.x / blocks_h, h0 = (blockIdx.x % blocks_h) * TH, h1 = min(h0 + TH, H);
  float a[KQ];
#pragma unroll
  for (int i = 0; i < KQ; ++i) {
    const float wv_ = wf[(size_t)((4 * i + kq) * 9 + min(l15, 8)) * Cpad];
    a[i] = l15 < 9 ? wv_ : 0.f;
  }
  for (int i = tid; i < 4 * 9 * 8; i += blockDim.x) {          // the zero columns left and right of the image
    const int e = i & 7;
    res[i / 72][(i >> 3) % 9][e < 4 ? e : W + e] = 0.f;
  }
  const size_t plane = (size_t)H * W;
  const float* src = dy + ((size_t)n * K + kq) * plane + seg * 64 + 4 * l15;
  // the gradient row after the current one is in flight (registers) while the current one is multiplied
  f32x4_t dv[KQ], dn[KQ];
  unsigned mb[KQ], mn[KQ];
#define C1_LOAD(Y, DV, MB)          /* rows outside the image: any valid row, the products are zeroed below */ \
  {                                                                                             \
    const int yc_ = min(max((Y), 0), H - 1);                                                    \
    _Pragma("unroll") for (int i = 0; i < KQ; ++i) {                                            \
      const size_t off = (size_t)(4 * i) * plane + (size_t)yc_ * W;                             \
      DV[i] = __builtin_nontemporal_load((const f32x4_t*)(src + off));                          \
      MB[i] = BITS ? bits[(size_t)((src - dy) + off) >> 2] : 0xFu;                              \
    }                                                                                           \
  }
#define C1_ROW(Y, DV, MB, DN, MN)                                                               \
  {                                                                                             \
    const int yy_ = (Y);                                                                          \
    C1_LOAD(yy_ + 1, DN, MN)                                                                      \
    f32x4_t acc[4];                                                                             \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};         \
    const unsigned rowm_ = (yy_ >= 0 && yy_ < H) ? 0xFu : 0u;                                       \
    _Pragma("unroll") for (int i = 0; i < KQ; ++i)                                              \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                           \
        const float b = ((MB[i] & rowm_) >> j) & 1u ? DV[i][j] : 0.f;                           \
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b, acc[j], 0, 0, 0);                \
      }                                                                                         \
    const int slot = (yy_ + 1) & 3;                                                               \
    _Pragma("unroll") for (int r = 0; r < 4; ++r)                                               \
      if (4 * kq + r < 9)                                                                       \
        *(f32x4_t*)&res[slot][4 * kq + r][4 + seg * 64 + 4 * l15] = f32x4_t{acc[0][r], acc[1][r], acc[2][r], acc[3][r]}; \
    __syncthreads();                                                                            \
    const int h = yy_ - 1;                                                                        \
    if (h >= h0 && h < h1) {                                                                    \
      float sum = 0.f;                                                                          \
      _Pragma("unroll") for (int r = 0; r < 3; ++r)                                             \
        _Pragma("unroll") for (int s2 = 0; s2 < 3; ++s2) sum += res[(h + r) & 3][r * 3 + s2][4 + tid + s2 - 1]; \
      dx[((size_t)n * H + h) * W + tid] = sum;                                                  \
    }                                                                                           \
  }
  C1_LOAD(h0 - 1, dv, mb)
  for (int y = h0 - 1; y <= h1; y += 2) {
    C1_ROW(y, dv, mb, dn, mn)
    if (y + 1 <= h1) C1_ROW(y + 1, dn, mn, dv, mb)
  }
#undef C1_ROW
#undef C1_LOAD
}

// FCD_THIN_MFMA=0: VALU kernel; =<n>: output rows per workgroup.  conv1_1 data gradient of the headline step (N = 208, bit
// mask): VALU 1.14 ms; MFMA 8 rows 0.84, 16 rows 0.76, 32 rows 0.74 (4.7 TB/s of the gradient it reads).  The same idea for
// the FORWARD of this layer (taps as the MFMA's k dimension, 48 MFMAs per 64-pixel row segment) lost: 1.03 vs 0.85 ms --
// that kernel is bound by its 64 scattered output planes, not by the 576 FMAs per pixel
static int thin_mfma_rows() { return fcd_sw(FCD_SW_THIN_MFMA); }

// returns 0 when handled, 1 when the shape is not a thin-channel case
int fcd_try_dgrad_thin(const fcd_conv_desc* d, const float* dy, const float* relu_out, const float* wp_bwd, float* dx,
                       hipStream_t st, int mask_is_bits) {
  if (!(d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1 && d->C >= 1 && d->C <= 4)) return 1;
  const int tiles_w = cdiv(d->W, TH_COLS), tiles_h = cdiv(d->H, TH_ROWS);
  const int Cpad = round_up(d->C, 128);
  dim3 grid((unsigned)(tiles_w * tiles_h), (unsigned)d->N);
#define THIN_LAUNCH(KERNEL, CS_)                                                                              \
  hipLaunchKernelGGL(KERNEL<CS_>, grid, dim3(256), 0, st, dy, relu_out, wp_bwd, dx, d->K, d->H, d->W, Cpad, tiles_w)
  if (d->C == 1 && d->K == 64 && (d->W & 63) == 0 && d->W <= 256 && thin_mfma_rows() > 0 && (mask_is_bits || !relu_out) &&
      (long long)d->N * cdiv(d->H, thin_mfma_rows()) < (1LL << 31)) {
    const int TH = thin_mfma_rows(), blocks_h = cdiv(d->H, TH);
    const dim3 g((unsigned)(d->N * blocks_h));
    if (mask_is_bits)
      hipLaunchKernelGGL(conv3x3_dgrad_c1_mfma_kernel<true>, g, dim3(d->W), 0, st, dy, (const unsigned char*)relu_out, wp_bwd, dx,
                         d->H, d->W, Cpad, TH, blocks_h);
    else
      hipLaunchKernelGGL(conv3x3_dgrad_c1_mfma_kernel<false>, g, dim3(d->W), 0, st, dy, (const unsigned char*)nullptr, wp_bwd, dx,
                         d->H, d->W, Cpad, TH, blocks_h);
    return 0;
  }
  if ((d->W & 3) == 0 && (d->K % TH_KC) == 0) {     // float4 rows, whole 8-channel chunks
#define THIN_V4(CS_, B_) \
  hipLaunchKernelGGL((conv3x3_dgrad_thin_v4_kernel<CS_, B_>), grid, dim3(256), 0, st, dy, relu_out, wp_bwd, dx, d->K, d->H, d->W, Cpad, tiles_w)
    switch (d->C) {
      case 1: if (mask_is_bits) THIN_V4(1, true); else THIN_V4(1, false); break;
      case 2: if (mask_is_bits) THIN_V4(2, true); else THIN_V4(2, false); break;
      case 3: if (mask_is_bits) THIN_V4(3, true); else THIN_V4(3, false); break;
      default: if (mask_is_bits) THIN_V4(4, true); else THIN_V4(4, false); break;
    }
#undef THIN_V4
    return 0;
  }
  if (mask_is_bits) return 1;      // bit masks only exist for the float4 configuration
  switch (d->C) {
    case 1: THIN_LAUNCH(conv3x3_dgrad_thin_kernel, 1); break;
    case 2: THIN_LAUNCH(conv3x3_dgrad_thin_kernel, 2); break;
    case 3: THIN_LAUNCH(conv3x3_dgrad_thin_kernel, 3); break;
    default: THIN_LAUNCH(conv3x3_dgrad_thin_kernel, 4); break;
  }
#undef THIN_LAUNCH
  return 0;
}

// returns 0 when handled, 1 when the shape / epilogue is not covered by the thin forward kernel
// bits != NULL: additionally write the ReLU bit mask (needs W % 4 == 0 and K % 8 == 0, the configuration the
// float4 data-gradient kernel consumes)
int fcd_try_fwd_thin(const fcd_conv_desc* d, const float* x, const float* wp, const float* bias, float* y, int relu,
                     hipStream_t st, unsigned char* bits) {
  if (bits && ((d->W & 3) != 0 || (d->K % TH_KC) != 0)) return 1;
  if (!(d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1 && d->C >= 1 && d->C <= 4 && d->K > 32)) return 1;
  const int tiles_w = cdiv(d->W, TH_COLS), tiles_h = cdiv(d->H, TH_ROWS);
  const int Mpad = round_up(d->K, 128);
  dim3 grid((unsigned)(tiles_w * tiles_h), (unsigned)d->N);
#define THIN_LAUNCH(CS_)                                                                                       \
  hipLaunchKernelGGL(conv3x3_fwd_thin_kernel<CS_>, grid, dim3(256), 0, st, x, wp, bias, y, d->K, Mpad, d->H, d->W, \
                     tiles_w, relu, bits)
  switch (d->C) {
    case 1: THIN_LAUNCH(1); break;
    case 2: THIN_LAUNCH(2); break;
    case 3: THIN_LAUNCH(3); break;
    default: THIN_LAUNCH(4); break;
  }
#undef THIN_LAUNCH
  return 0;
}
