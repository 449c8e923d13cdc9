// Weight (+ bias) gradient of the 9x9 / stride-1 / pad-4 layers that have <= 4 channels on ONE side: the Generator's first
// (bands -> 64, Module.py:150) and last (64 -> bands, Module.py:165) convolutions on 3- / 4-band data (Demo_USSS, Demo_WSSS).
//
//   dW[k][c][r][s] = sum_{n,p,q} dY[n,k,p,q] * X[n,c,p+r-4,q+s-4]
//
// The general kernel (conv_wgrad.hip) pads the thin side to 64 channels and walks the 81 taps in nine row groups: 16x the
// MFMA work and 16x the re-laid-out operand (7.4 ms per layer and step at 16 x 4 x 256 x 256, 0.04 of the fp32 MFMA peak --
// 18 % of the USSS Generator step).  Here the problem is reduction-contiguous like conv_wgrad_thin.hip and reads NCHW as it
// lies:
//   GEMM rows = the 64-channel side ("wide": dY for the first layer, X for the last), GEMM columns = (thin channel, tap)
//   flattened (<= 4 x 81 = 324) + ONE column of ones (bias gradient, first-layer form only), reduction = pixels, two per
//   v_mfma_f32_32x32x2_f32.  A workgroup walks tiles of 2 x 64 pixels: the wide tile [64][128 px] and the thin patch
//   [<= 4][10][72] are staged in LDS; a lane's A operand is wide[row = lane][px], its B operand the patch element under
//   the tap of its column (compile-time pixel offsets).  The 12 column blocks are dealt to the four waves (3 each, 96
//   accumulator registers), every wave walks all 128 pixels of the tile, so no cross-wave sum is needed: per-workgroup
//   partials [64][384] go to the workspace, a second kernel adds them in a fixed order.
// Last-layer form (wide = X, thin = dY): sum_{p,q} dY[k,p,q] X[c,p+r-4,q+s-4] = sum_{p',q'} X[c,p',q'] dY[k,p'+(8-r)-4,q'+(8-s)-4]:
// the same kernel with the operands swapped yields G[c][k][r'][s'] = dW[k][c][8-r'][8-s']; the finishing kernel un-flips.
#include "common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int T9_TH = 2, T9_TW = 64;
constexpr int T9_WP = T9_TH * T9_TW + 4;        // wide-tile row pitch (floats)
constexpr int T9_ROWS = 64, T9_COLS = 384, T9_NB = 3;
constexpr int T9_XR = T9_TH + 8, T9_XC = T9_TW + 8, T9_XP = T9_XC + 1;     // patch rows / columns / pitch (73: odd)
constexpr int T9_CMAX = 4;
constexpr int T9_MAX_PARTS = 512;

struct Wg9Args {
  const float* wide;       // (N, Cw, H, W): the 64-channel operand
  const float* thin;       // (N, Ct, H, W): the <= 4-channel operand (read with a 4-pixel halo)
  const float* wmask;      // optional gate on `wide` (same shape): read as wide * [mask > 0]
  const float* tmask;      // optional gate on `thin`
  float* part;             // [gridDim.x][64][384]
  int N, Cw, Ct, H, W;
  int tiles_p, tiles_q, total_tiles, bias_col;     // bias_col: 1 = column Ct * 81 is a column of ones
};

__global__ __launch_bounds__(256, 2) void conv_wgrad_thin9_kernel(Wg9Args a) {
  __shared__ __attribute__((aligned(16))) float sw[T9_ROWS * T9_WP];
  __shared__ __attribute__((aligned(16))) float sx[T9_CMAX * T9_XR * T9_XP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int ncol = a.Ct * 81;

  // B operand of this lane's column in each of the wave's three column blocks (blocks wave, wave + 4, wave + 8)
  int boff[T9_NB];
  float bconst[T9_NB];
  bool blds[T9_NB];
#pragma unroll
  for (int b = 0; b < T9_NB; ++b) {
    const int col = 32 * (wave + 4 * b) + l31;
    blds[b] = col < ncol;
    bconst[b] = (a.bias_col && col == ncol) ? 1.f : 0.f;
    const int c = blds[b] ? col / 81 : 0, tap = blds[b] ? col % 81 : 0;
    boff[b] = (c * T9_XR + tap / 9) * T9_XP + tap % 9 + half;
  }
  int aoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) aoff[i] = (32 * i + l31) * T9_WP + half;

  f32x16 acc[2][T9_NB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int b = 0; b < T9_NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][b][r] = 0.f;

  const size_t plane = (size_t)a.H * a.W;
  constexpr int W_PER_T = (T9_ROWS * T9_TH * T9_TW / 4) / 256;        // 8 float4
  const bool vec = (a.W & 3) == 0;
  const int xtotal = a.Ct * T9_XR * T9_XC;
  f32x4 wr[W_PER_T];
  // the next tile's wide slab is requested before the MFMA block of the current tile and committed behind it
  auto issue = [&](int t) {
    const int tq = t % a.tiles_q;
    const int tp = (t / a.tiles_q) % a.tiles_p;
    const int n = t / (a.tiles_q * a.tiles_p);
    const int p0 = tp * T9_TH, q0 = tq * T9_TW;
    const float* wn = a.wide + (size_t)n * a.Cw * plane;
    const float* mn = a.wmask ? a.wmask + (size_t)n * a.Cw * plane : nullptr;
#pragma unroll
    for (int j = 0; j < W_PER_T; ++j) {
      const int idx = tid + 256 * j;                     // (row, tile row, 16 float4)
      const int c4 = idx & 15, row = (idx >> 4) & (T9_TH - 1), k = idx >> 5;
      const int p = p0 + row, q = q0 + 4 * c4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (k < a.Cw && p < a.H && q < a.W) {
        const size_t off = (size_t)k * plane + (size_t)p * a.W + q;
        if (vec) {
          v = *(const f32x4*)(wn + off);
          if (mn) {
            const f32x4 m = *(const f32x4*)(mn + off);
#pragma unroll
            for (int e = 0; e < 4; ++e) if (!(m[e] > 0.f)) v[e] = 0.f;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (q + e < a.W) {
              float sv = wn[off + e];
              if (mn && !(mn[off + e] > 0.f)) sv = 0.f;
              v[e] = sv;
            }
        }
      }
      wr[j] = v;
    }
  };
  auto commit = [&](int t) {
#pragma unroll
    for (int j = 0; j < W_PER_T; ++j) {
      const int idx = tid + 256 * j;
      const int c4 = idx & 15, row = (idx >> 4) & (T9_TH - 1), k = idx >> 5;
      *(f32x4*)(sw + k * T9_WP + row * T9_TW + 4 * c4) = wr[j];
    }
    const int tq = t % a.tiles_q;
    const int tp = (t / a.tiles_q) % a.tiles_p;
    const int n = t / (a.tiles_q * a.tiles_p);
    const float* xn = a.thin + (size_t)n * a.Ct * plane;
    const float* tm = a.tmask ? a.tmask + (size_t)n * a.Ct * plane : nullptr;
    const int ih0 = tp * T9_TH - 4, iw0 = tq * T9_TW - 4;
    for (int idx = tid; idx < xtotal; idx += 256) {
      const int jj = idx % T9_XC, rr = (idx / T9_XC) % T9_XR, c = idx / (T9_XC * T9_XR);
      const int ih = ih0 + rr, iw = iw0 + jj;
      float v = 0.f;
      if (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) {
        const size_t off = (size_t)c * plane + (size_t)ih * a.W + iw;
        v = xn[off];
        if (tm && !(tm[off] > 0.f)) v = 0.f;
      }
      sx[(c * T9_XR + rr) * T9_XP + jj] = v;
    }
  };

  if ((int)blockIdx.x < a.total_tiles) issue(blockIdx.x);
  for (int t = blockIdx.x; t < a.total_tiles; t += gridDim.x) {
    __syncthreads();                                       // previous tile fully consumed
    commit(t);
    __syncthreads();
    if (t + (int)gridDim.x < a.total_tiles) issue(t + gridDim.x);
    // ---- 64 pixel pairs (2 tile rows x 32) x 6 MFMAs
#pragma unroll
    for (int tr = 0; tr < T9_TH; ++tr) {
#pragma unroll 8
      for (int j = 0; j < T9_TW / 2; ++j) {
        float av[2], bv[T9_NB];
#pragma unroll
        for (int i = 0; i < 2; ++i) av[i] = sw[aoff[i] + tr * T9_TW + 2 * j];
#pragma unroll
        for (int b = 0; b < T9_NB; ++b) {
          const float l = sx[boff[b] + tr * T9_XP + 2 * j];
          bv[b] = blds[b] ? l : bconst[b];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int b = 0; b < T9_NB; ++b)
            acc[i][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[b], acc[i][b], 0, 0, 0);
      }
    }
  }

  // ---- every wave owns its column blocks: straight to the workgroup's partial, lanes along the columns
  float* out = a.part + (size_t)blockIdx.x * (T9_ROWS * T9_COLS);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int b = 0; b < T9_NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half, col = 32 * (wave + 4 * b) + l31;
        out[k * T9_COLS + col] = acc[i][b][r];
      }
}

// sum over the workgroups' partials in a fixed order; first-layer form: dw[row][col] (row = filter k, col = c * 81 + tap), bias
// column -> db; last-layer form (swapped): row = input channel c, col = k * 81 + tap': dw[k][c][80 - tap']
__global__ __launch_bounds__(256) void conv_wgrad_thin9_finish_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                      float* __restrict__ db, int nparts, int rows, int thin_ch,
                                                                      int swapped) {
  const int o = blockIdx.x * 256 + threadIdx.x;            // (row, col)
  const int row = o / T9_COLS, col = o % T9_COLS;
  const int ncol = thin_ch * 81;
  if (row >= rows || col > ncol) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int p = 0;
  for (; p + 3 < nparts; p += 4) {
    s0 += part[(size_t)p * (T9_ROWS * T9_COLS) + o];
    s1 += part[(size_t)(p + 1) * (T9_ROWS * T9_COLS) + o];
    s2 += part[(size_t)(p + 2) * (T9_ROWS * T9_COLS) + o];
    s3 += part[(size_t)(p + 3) * (T9_ROWS * T9_COLS) + o];
  }
  for (; p < nparts; ++p) s0 += part[(size_t)p * (T9_ROWS * T9_COLS) + o];
  const float s = (s0 + s1) + (s2 + s3);
  if (col < ncol) {
    if (!swapped) dw[(size_t)row * ncol + col] = s;
    else dw[((size_t)(col / 81) * rows + row) * 81 + (80 - col % 81)] = s;
  } else if (db && !swapped) {
    db[row] = s;
  }
}

// db[k] = sum_{n,p,q} dY[n,k,p,q] (* [mask > 0]) for the last-layer form (<= 4 filters): grid (K, T9_BSPLIT) fp64 partials over
// (sample, pixel) chunks, then one thread per filter adds them in a fixed order
constexpr int T9_BSPLIT = 128;
__global__ __launch_bounds__(256) void thin9_bias_part_kernel(const float* __restrict__ dy, const float* __restrict__ mask,
                                                              double* __restrict__ part, int N, int K, int HW) {
  __shared__ double red[256];
  const int k = blockIdx.x, sp = blockIdx.y;
  const long long total = (long long)N * HW;
  const long long chunk = (total + T9_BSPLIT - 1) / T9_BSPLIT;
  const long long beg = sp * chunk, end = min(beg + chunk, total);
  double s = 0.0;
  for (long long e = beg + threadIdx.x; e < end; e += 256) {
    const int n = (int)(e / HW), i = (int)(e % HW);
    const size_t off = ((size_t)n * K + k) * HW + i;
    float v = dy[off];
    if (mask && !(mask[off] > 0.f)) v = 0.f;
    s += (double)v;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[(size_t)k * T9_BSPLIT + sp] = red[0];
}

__global__ void thin9_bias_fin_kernel(const double* __restrict__ part, float* __restrict__ db, int K) {
  const int k = threadIdx.x;
  if (k >= K) return;
  double s = 0.0;
  for (int i = 0; i < T9_BSPLIT; ++i) s += part[(size_t)k * T9_BSPLIT + i];
  db[k] = (float)s;
}

int t9_env() { return fcd_sw(FCD_SW_WGRAD_THIN9); }
}  // namespace

// 1: first-layer form (<= 4 input channels, <= 64 filters), 2: last-layer form (<= 64 input channels, <= 4 filters), 0: not this kernel
int fcd_wgrad_thin9_plan(const fcd_conv_desc* d) {
  if (!d || !t9_env()) return 0;
  if (!(d->R == 9 && d->S == 9 && d->pad == 4 && d->stride == 1)) return 0;
  if ((long long)d->N * d->P * d->Q < 4096) return 0;
  if (d->C <= T9_CMAX && d->K <= T9_ROWS) return 1;
  if (d->K <= T9_CMAX && d->C <= T9_ROWS) return 2;
  return 0;
}

size_t fcd_wgrad_thin9_ws_bytes(const fcd_conv_desc* d) {
  return fcd_wgrad_thin9_plan(d) ? (size_t)T9_MAX_PARTS * T9_ROWS * T9_COLS * sizeof(float) + T9_CMAX * T9_BSPLIT * sizeof(double) : 0;
}

int fcd_wgrad_thin9_run(const fcd_conv_desc* d, const float* x, const float* dy, const float* relu_out, float* dw, float* db,
                        void* ws, hipStream_t st) {
  const int form = fcd_wgrad_thin9_plan(d);
  if (!form) return -1;
  Wg9Args a;
  memset(&a, 0, sizeof(a));
  a.part = (float*)ws;
  a.N = d->N; a.H = d->H; a.W = d->W;
  if (form == 1) {
    a.wide = dy; a.wmask = relu_out; a.Cw = d->K;
    a.thin = x; a.tmask = nullptr; a.Ct = d->C;
    a.bias_col = db ? 1 : 0;
  } else {
    a.wide = x; a.wmask = nullptr; a.Cw = d->C;
    a.thin = dy; a.tmask = relu_out; a.Ct = d->K;
    a.bias_col = 0;
  }
  a.tiles_p = cdiv(d->H, T9_TH);
  a.tiles_q = cdiv(d->W, T9_TW);
  a.total_tiles = d->N * a.tiles_p * a.tiles_q;
  const int nparts = std::min(a.total_tiles, T9_MAX_PARTS);
  hipLaunchKernelGGL(conv_wgrad_thin9_kernel, dim3(nparts), dim3(256), 0, st, a);
  hipLaunchKernelGGL(conv_wgrad_thin9_finish_kernel, dim3(T9_ROWS * T9_COLS / 256), dim3(256), 0, st, (const float*)ws, dw, db,
                     nparts, a.Cw, a.Ct, form == 2 ? 1 : 0);
  if (form == 2 && db) {
    double* bp = (double*)((char*)ws + (size_t)T9_MAX_PARTS * T9_ROWS * T9_COLS * sizeof(float));
    hipLaunchKernelGGL(thin9_bias_part_kernel, dim3(d->K, T9_BSPLIT), dim3(256), 0, st, dy, relu_out, bp, d->N, d->K, d->P * d->Q);
    hipLaunchKernelGGL(thin9_bias_fin_kernel, dim3(1), dim3(64), 0, st, (const double*)bp, db, d->K);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
