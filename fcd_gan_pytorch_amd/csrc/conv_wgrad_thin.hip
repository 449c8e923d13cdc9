// Weight (+ bias) gradient of the 3x3 FIRST layers: few input channels (C x 9 < 128: 3, 4, 13 bands), <= 64 filters,
// stride 1 or 2, pad 1 -- Segmentor inc.double_conv.0 (Module.py:25, 13 -> 64 at 256 x 256) and Discriminator net.0
// (Module.py:195, 13 -> 64 stride 2).
//
//   dW[k][c][r][s] = sum_{n,p,q} dY[n,k,p,q] * X[n,c,p*stride+r-1,q*stride+s-1]        db[k] = sum dY[n,k,p,q]
//
// The general kernel (conv_wgrad.hip) re-lays both operands out channel-minor with the channels padded to 64: for 13
// bands it moves 4.9x the input and executes 4.9x the MFMA work (0.81 + 0.57 ms per step, 0.09 - 0.12 of the fp32 MFMA
// peak).  This layer is HBM-bound (dY read once: 268 MB; 15.7 GFLOP), so the kernel here is reduction-contiguous and
// reads NCHW as it lies:
//   GEMM rows = k (<= 64), GEMM columns = (c, r, s) flattened (C x 9 <= 127) + ONE column of ones whose "weight
//   gradient" is the bias gradient, reduction = output pixels, two per v_mfma_f32_32x32x2_f32 (lane half = pixel parity).
//   A workgroup walks tiles of 2 output rows x 64 columns: dY tile [64 k][128 px] and the input patch [C][rows][cols]
//   are staged in LDS by coalesced row loads; a lane's A operand is dY[k = lane][px], its B operand the patch element
//   under tap (r, s) of channel c = its column -- plain ds_read_b32 with compile-time pixel offsets.  Each of the four
//   waves owns 32 pixels of the tile and all 2 x 4 accumulator blocks (128 VGPRs); the waves' accumulators are summed
//   through LDS once per workgroup, per-workgroup partials [64][128] go to the workspace and a second kernel adds them
//   in a fixed order (deterministic, no atomics).
#include "common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int WT_TH = 2, WT_TW = 64;            // output pixels per tile
constexpr int WT_DYP = WT_TH * WT_TW + 4;       // dY row pitch (floats); 16-B aligned rows for the float4 stores (column reads: 4-way, off the critical path)
constexpr int WT_COLS = 128, WT_ROWS = 64;

struct WgThinArgs {
  const float* x;
  const float* dy;
  const float* mask;     // optional: dY is read as dY * [mask > 0]
  float* part;           // [gridDim.x][64][128]
  int N, C, H, W, K, P, Q;
  int tiles_p, tiles_q, total_tiles;
};

template <int STRIDE>
__global__ __launch_bounds__(256, 2) void conv_wgrad_thin_kernel(WgThinArgs a) {
  constexpr int XR = (WT_TH - 1) * STRIDE + 3;            // patch rows
  constexpr int XC = (WT_TW - 1) * STRIDE + 3;            // patch columns
  constexpr int XP = XC + (STRIDE == 1 ? 3 : 4);          // row pitch: 69 (odd) / 133 (odd)
  constexpr int CMAX = 14;
  __shared__ __attribute__((aligned(16))) float sdy[WT_ROWS * WT_DYP];
  __shared__ __attribute__((aligned(16))) float sx[CMAX * XR * XP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int ncol = a.C * 9;                               // column `ncol` is the bias column

  // B operand: LDS offset of (channel, tap) of this lane's column in each of the four column blocks
  int boff[4];
  float bconst[4];        // value used instead of LDS for the bias column (1) and the padding columns (0)
  bool blds[4];
  const int tr = wave >> 1, tc0 = 32 * (wave & 1);        // this wave's 32 pixels: tile row tr, columns tc0 .. tc0 + 31
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int col = 32 * b + l31;
    blds[b] = col < ncol;
    bconst[b] = col == ncol ? 1.f : 0.f;
    const int c = blds[b] ? col / 9 : 0, tap = blds[b] ? col % 9 : 0;
    boff[b] = (c * XR + tr * STRIDE + tap / 3) * XP + (tc0 + half) * STRIDE + tap % 3;
  }
  int aoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) aoff[i] = (32 * i + l31) * WT_DYP + tr * WT_TW + tc0 + half;

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][b][r] = 0.f;

  const size_t x_plane = (size_t)a.H * a.W, y_plane = (size_t)a.P * a.Q;
  // [r4] The next tile's dY slab and input patch are requested into registers BEFORE the MFMA block of the current tile and
  // committed to LDS behind it (the kernel used to load, wait, store and only then multiply: two workgroups per CU were all that
  // hid the loads).
  constexpr int DY_PER_T = (WT_ROWS * WT_TH * WT_TW / 4) / 256;        // 8 float4
  constexpr bool XPRE = false;      // (staging the input patch through registers too: stride 1 spills at 14 per thread, 0.38 vs 0.30 ms; stride 2 needs 36)
  constexpr int X_PER_T = XPRE ? (CMAX * XR * XC + 255) / 256 : 1;
  const bool vec = (a.Q & 3) == 0;
  const int xtotal = a.C * XR * XC;
  f32x4 dyr[DY_PER_T];
  float xr[X_PER_T];
  auto issue = [&](int t) {
    const int tq = t % a.tiles_q;
    const int tp = (t / a.tiles_q) % a.tiles_p;
    const int n = t / (a.tiles_q * a.tiles_p);
    const int p0 = tp * WT_TH, q0 = tq * WT_TW;
    const float* dyn = a.dy + (size_t)n * a.K * y_plane;
    const float* mn = a.mask ? a.mask + (size_t)n * a.K * y_plane : nullptr;
#pragma unroll
    for (int j = 0; j < DY_PER_T; ++j) {
      const int idx = tid + 256 * j;                     // (k, row, 16 float4)
      const int c4 = idx & 15, row = (idx >> 4) & (WT_TH - 1), k = idx >> 5;
      const int p = p0 + row, q = q0 + 4 * c4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (k < a.K && p < a.P && q < a.Q) {
        const size_t off = (size_t)k * y_plane + (size_t)p * a.Q + q;
        if (vec) {
          v = *(const f32x4*)(dyn + off);
          if (mn) {
            const f32x4 m = *(const f32x4*)(mn + off);
#pragma unroll
            for (int e = 0; e < 4; ++e) if (!(m[e] > 0.f)) v[e] = 0.f;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (q + e < a.Q) {
              float sv = dyn[off + e];
              if (mn && !(mn[off + e] > 0.f)) sv = 0.f;
              v[e] = sv;
            }
        }
      }
      dyr[j] = v;
    }
    if (XPRE) {
      const float* xn = a.x + (size_t)n * a.C * x_plane;
      const int ih0 = p0 * STRIDE - 1, iw0 = q0 * STRIDE - 1;
#pragma unroll
      for (int j = 0; j < X_PER_T; ++j) {
        const int idx = tid + 256 * j;
        float v = 0.f;
        if (idx < xtotal) {
          const int jj = idx % XC, rr = (idx / XC) % XR, c = idx / (XC * XR);
          const int ih = ih0 + rr, iw = iw0 + jj;
          if (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) v = xn[(size_t)c * x_plane + (size_t)ih * a.W + iw];
        }
        xr[j] = v;
      }
    }
  };
  auto commit = [&](int t) {
#pragma unroll
    for (int j = 0; j < DY_PER_T; ++j) {
      const int idx = tid + 256 * j;
      const int c4 = idx & 15, row = (idx >> 4) & (WT_TH - 1), k = idx >> 5;
      *(f32x4*)(sdy + k * WT_DYP + row * WT_TW + 4 * c4) = dyr[j];
    }
    if (XPRE) {
#pragma unroll
      for (int j = 0; j < X_PER_T; ++j) {
        const int idx = tid + 256 * j;
        if (idx < xtotal) {
          const int jj = idx % XC, rr = (idx / XC) % XR, c = idx / (XC * XR);
          sx[(c * XR + rr) * XP + jj] = xr[j];
        }
      }
    } else {
      const int tq = t % a.tiles_q;
      const int tp = (t / a.tiles_q) % a.tiles_p;
      const int n = t / (a.tiles_q * a.tiles_p);
      const float* xn = a.x + (size_t)n * a.C * x_plane;
      const int ih0 = tp * WT_TH * STRIDE - 1, iw0 = tq * WT_TW * STRIDE - 1;
      for (int idx = tid; idx < xtotal; idx += 256) {
        const int jj = idx % XC, rr = (idx / XC) % XR, c = idx / (XC * XR);
        const int ih = ih0 + rr, iw = iw0 + jj;
        float v = 0.f;
        if (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) v = xn[(size_t)c * x_plane + (size_t)ih * a.W + iw];
        sx[(c * XR + rr) * XP + jj] = v;
      }
    }
  };
  if ((int)blockIdx.x < a.total_tiles) issue(blockIdx.x);
  for (int t = blockIdx.x; t < a.total_tiles; t += gridDim.x) {
    __syncthreads();                                       // previous tile fully consumed
    commit(t);
    __syncthreads();
    if (t + (int)gridDim.x < a.total_tiles) issue(t + gridDim.x);
    // ---- 16 pixel pairs of this wave x 8 MFMAs
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float av[2], bv[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) av[i] = sdy[aoff[i] + 2 * j];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float l = sx[boff[b] + 2 * j * STRIDE];
        bv[b] = blds[b] ? l : bconst[b];                   // bias column: 1, padding columns: 0
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[i][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[b], acc[i][b], 0, 0, 0);
    }
  }

  // ---- sum the four waves' accumulators through LDS (reusing the dY tile), one 32 x 32 block at a time
  float* red = sdy;                                        // 4 waves x 16 x 64 floats = 16 KiB
  float* out = a.part + (size_t)blockIdx.x * (WT_ROWS * WT_COLS);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[i][b][r];
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int o = tid + 256 * u, r = o >> 6, ln = o & 63;
        const float s = (red[(0 * 16 + r) * 64 + ln] + red[(1 * 16 + r) * 64 + ln]) +
                        (red[(2 * 16 + r) * 64 + ln] + red[(3 * 16 + r) * 64 + ln]);
        const int k = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), col = 32 * b + (ln & 31);
        out[k * WT_COLS + col] = s;
      }
    }
}

// dw[k][c*9 + tap] / db[k] = sum over the workgroups' partials, fixed order
__global__ __launch_bounds__(256) void conv_wgrad_thin_finish_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                     float* __restrict__ db, int nparts, int K, int ncol) {
  const int o = blockIdx.x * 256 + threadIdx.x;            // (k, col)
  const int k = o / WT_COLS, col = o % WT_COLS;
  if (k >= K || col > ncol) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int p = 0;
  for (; p + 3 < nparts; p += 4) {
    s0 += part[(size_t)p * (WT_ROWS * WT_COLS) + o];
    s1 += part[(size_t)(p + 1) * (WT_ROWS * WT_COLS) + o];
    s2 += part[(size_t)(p + 2) * (WT_ROWS * WT_COLS) + o];
    s3 += part[(size_t)(p + 3) * (WT_ROWS * WT_COLS) + o];
  }
  for (; p < nparts; ++p) s0 += part[(size_t)p * (WT_ROWS * WT_COLS) + o];
  const float s = (s0 + s1) + (s2 + s3);
  if (col < ncol) dw[(size_t)k * ncol + col] = s;
  else if (db) db[k] = s;
}

int wt_env() { return fcd_sw(FCD_SW_WGRAD_THIN); }
constexpr int WT_MAX_PARTS = 512;
}  // namespace

// 1 when the layer's weight gradient runs on the reduction-contiguous first-layer kernel
int fcd_wgrad_thin_plan(const fcd_conv_desc* d) {
  if (!d || !wt_env()) return 0;
  if (!(d->R == 3 && d->S == 3 && d->pad == 1 && (d->stride == 1 || d->stride == 2))) return 0;
  if (d->C * 9 >= WT_COLS || d->K > WT_ROWS) return 0;
  if ((long long)d->N * d->P * d->Q < 4096) return 0;      // tiny maps: the general kernel's split logic is fine
  return 1;
}

size_t fcd_wgrad_thin_ws_bytes(const fcd_conv_desc* d) {
  return fcd_wgrad_thin_plan(d) ? (size_t)WT_MAX_PARTS * WT_ROWS * WT_COLS * sizeof(float) : 0;
}

int fcd_wgrad_thin_run(const fcd_conv_desc* d, const float* x, const float* dy, const float* relu_out, float* dw, float* db,
                       void* ws, hipStream_t st) {
  WgThinArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.dy = dy; a.mask = relu_out; a.part = (float*)ws;
  a.N = d->N; a.C = d->C; a.H = d->H; a.W = d->W; a.K = d->K; a.P = d->P; a.Q = d->Q;
  a.tiles_p = cdiv(d->P, WT_TH);
  a.tiles_q = cdiv(d->Q, WT_TW);
  a.total_tiles = d->N * a.tiles_p * a.tiles_q;
  const int nparts = std::min(a.total_tiles, WT_MAX_PARTS);
  if (d->stride == 1) hipLaunchKernelGGL(conv_wgrad_thin_kernel<1>, dim3(nparts), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(conv_wgrad_thin_kernel<2>, dim3(nparts), dim3(256), 0, st, a);
  hipLaunchKernelGGL(conv_wgrad_thin_finish_kernel, dim3(WT_ROWS * WT_COLS / 256), dim3(256), 0, st, (const float*)ws, dw, db,
                     nparts, d->K, d->C * 9);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
