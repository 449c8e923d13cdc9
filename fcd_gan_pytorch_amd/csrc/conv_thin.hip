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

// returns 0 when handled, 1 when the shape is not a thin-channel case
int fcd_try_dgrad_thin(const fcd_conv_desc* d, const float* dy, const float* relu_out, const float* wp_bwd, float* dx,
                       hipStream_t st) {
  if (!(d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1 && d->C >= 1 && d->C <= 4)) return 1;
  const int tiles_w = cdiv(d->W, TH_COLS), tiles_h = cdiv(d->H, TH_ROWS);
  const int Cpad = round_up(d->C, 128);
  dim3 grid((unsigned)(tiles_w * tiles_h), (unsigned)d->N);
  switch (d->C) {
    case 1: hipLaunchKernelGGL(conv3x3_dgrad_thin_kernel<1>, grid, dim3(256), 0, st, dy, relu_out, wp_bwd, dx, d->K, d->H, d->W, Cpad, tiles_w); break;
    case 2: hipLaunchKernelGGL(conv3x3_dgrad_thin_kernel<2>, grid, dim3(256), 0, st, dy, relu_out, wp_bwd, dx, d->K, d->H, d->W, Cpad, tiles_w); break;
    case 3: hipLaunchKernelGGL(conv3x3_dgrad_thin_kernel<3>, grid, dim3(256), 0, st, dy, relu_out, wp_bwd, dx, d->K, d->H, d->W, Cpad, tiles_w); break;
    default: hipLaunchKernelGGL(conv3x3_dgrad_thin_kernel<4>, grid, dim3(256), 0, st, dy, relu_out, wp_bwd, dx, d->K, d->H, d->W, Cpad, tiles_w); break;
  }
  return 0;
}
