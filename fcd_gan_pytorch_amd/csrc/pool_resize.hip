// MaxPool2d(2), bilinear x2 (align_corners=True) and the padded 2x2 average pool
// of MS-SSIM, forward + backward.  All HBM-bound gather kernels: one thread per
// OUTPUT element of the pass (so no atomics and fully written outputs),
// consecutive lanes on consecutive addresses of the wider tensor.
#include "common.h"

// ---- MaxPool2d(kernel 2, stride 2, floor) -------------------------------------
__global__ void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int NC, int H, int W,
                                    int P, int Q) {
  const long long total = (long long)NC * P * Q;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i % Q);
    const long long t = i / Q;
    const int p = (int)(t % P);
    const long long pl = t / P;
    const float* s = x + (pl * H + 2 * p) * W + 2 * q;
    const float2 a = *reinterpret_cast<const float2*>(s);   // W may be odd: 2q+1 <= W-1 always holds
    const float2 b = *reinterpret_cast<const float2*>(s + W);
    float m = a.x;
    // NaN-propagating max in window order (PyTorch: val > max || isnan(val))
    if (a.y > m || a.y != a.y) m = a.y;
    if (b.x > m || b.x != b.x) m = b.x;
    if (b.y > m || b.y != b.y) m = b.y;
    y[i] = m;
  }
}

__global__ void maxpool2_fwd_scalar_kernel(const float* __restrict__ x, float* __restrict__ y, int NC, int H, int W,
                                           int P, int Q) {
  const long long total = (long long)NC * P * Q;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i % Q);
    const long long t = i / Q;
    const int p = (int)(t % P);
    const long long pl = t / P;
    const float* s = x + (pl * H + 2 * p) * W + 2 * q;
    float m = s[0];
    const float v1 = s[1], v2 = s[W], v3 = s[W + 1];
    if (v1 > m || v1 != v1) m = v1;
    if (v2 > m || v2 != v2) m = v2;
    if (v3 > m || v3 != v3) m = v3;
    y[i] = m;
  }
}

// one thread per 2x2 input window (plus edge cells): recompute the argmax, route dy
// `add` != NULL: dx = add + routed gradient -- the gradient a SECOND consumer of x (the decoder's skip connection, reference
// Module.py:116-132) has already produced, summed here instead of by a three-pass add of its own
__global__ void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                    float* __restrict__ dx, int NC, int H, int W, int P, int Q,
                                    const float* __restrict__ add = nullptr) {
  const int PH = (H + 1) / 2, PW = (W + 1) / 2;  // cover trailing odd row/col with zero windows
  const long long total = (long long)NC * PH * PW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i % PW);
    const long long t = i / PW;
    const int p = (int)(t % PH);
    const long long pl = t / PH;
    const long long base = (pl * H + 2 * p) * W + 2 * q;
    if (p < P && q < Q) {
      const float v00 = x[base], v01 = x[base + 1], v10 = x[base + W], v11 = x[base + W + 1];
      int arg = 0;
      float m = v00;
      if (v01 > m || v01 != v01) { m = v01; arg = 1; }
      if (v10 > m || v10 != v10) { m = v10; arg = 2; }
      if (v11 > m || v11 != v11) { m = v11; arg = 3; }
      const float g = dy[(pl * P + p) * Q + q];
      if (add) {
        const float a00 = add[base], a01 = add[base + 1], a10 = add[base + W], a11 = add[base + W + 1];
        dx[base] = a00 + (arg == 0 ? g : 0.f);
        dx[base + 1] = a01 + (arg == 1 ? g : 0.f);
        dx[base + W] = a10 + (arg == 2 ? g : 0.f);
        dx[base + W + 1] = a11 + (arg == 3 ? g : 0.f);
      } else {
        dx[base] = arg == 0 ? g : 0.f;
        dx[base + 1] = arg == 1 ? g : 0.f;
        dx[base + W] = arg == 2 ? g : 0.f;
        dx[base + W + 1] = arg == 3 ? g : 0.f;
      }
    } else {
      // trailing odd row / column never pooled: zero gradient
      const int h = 2 * p, w = 2 * q;
      for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
          if (h + a < H && w + b < W) {
            const long long o = (pl * H + h + a) * W + w + b;
            dx[o] = add ? add[o] + 0.f : 0.f;
          }
    }
  }
}

static inline int ew_grid(long long total) { return (int)std::min<long long>(cdiv64(total, 256), 8192); }

extern "C" int fcd_maxpool2_fwd(const float* x, float* y, int NC, int H, int W, void* stream) {
  FCD_CHECK_ARG(x && y && NC > 0 && H >= 2 && W >= 2, "fcd_maxpool2_fwd: bad arguments");
  const int P = H / 2, Q = W / 2;
  const long long total = (long long)NC * P * Q;
  FcdProfScope prof(FCD_K_POOL, (hipStream_t)stream, 0.0, 4.0 * NC * ((double)H * W + (double)P * Q));
  if ((W & 1) == 0) {
    hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, y, NC, H, W,
                       P, Q);
  } else {
    // odd width: rows are not 8-byte aligned -> scalar variant through the bwd-style reader
    hipLaunchKernelGGL(maxpool2_fwd_scalar_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, y, NC,
                       H, W, P, Q);
  }
  FCD_LAUNCH_CHECK("maxpool2_fwd");
  return FCD_OK;
}

extern "C" int fcd_maxpool2_bwd(const float* x, const float* dy, float* dx, int NC, int H, int W, void* stream) {
  FCD_CHECK_ARG(x && dy && dx && NC > 0 && H >= 2 && W >= 2, "fcd_maxpool2_bwd: bad arguments");
  const int P = H / 2, Q = W / 2;
  const long long total = (long long)NC * ((H + 1) / 2) * ((W + 1) / 2);
  FcdProfScope prof(FCD_K_POOL, (hipStream_t)stream, 0.0, 4.0 * NC * (2.0 * H * W + (double)P * Q));
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, NC, H,
                     W, P, Q);
  FCD_LAUNCH_CHECK("maxpool2_bwd");
  return FCD_OK;
}

// dx = add + maxpool2 backward(x, dy): the pooled path's gradient summed onto the gradient `add` that the other consumer of x
// produced (U-Net skip connection: x feeds MaxPool2d and, concatenated, the decoder -- reference Module.py:116-132).  One pass
// instead of the routed tensor + a separate three-pass add; add + routed in fp32, the value torch's accumulation gives.  dx may be add.
extern "C" int fcd_maxpool2_bwd_add(const float* x, const float* dy, const float* add, float* dx, int NC, int H, int W, void* stream) {
  FCD_CHECK_ARG(x && dy && add && dx && NC > 0 && H >= 2 && W >= 2, "fcd_maxpool2_bwd_add: bad arguments");
  const int P = H / 2, Q = W / 2;
  const long long total = (long long)NC * ((H + 1) / 2) * ((W + 1) / 2);
  FcdProfScope prof(FCD_K_POOL, (hipStream_t)stream, 0.0, 4.0 * NC * (3.0 * H * W + (double)P * Q));
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, NC, H,
                     W, P, Q, add);
  FCD_LAUNCH_CHECK("maxpool2_bwd_add");
  return FCD_OK;
}

// ---- bilinear x2, align_corners=True ------------------------------------------
// src = dst * (in-1)/(out-1);  i0 = (int)src, i1 = i0 + (i0 < in-1), l1 = src - i0
__device__ __forceinline__ void ac_src(int dst, float scale, int in, int* i0, int* i1, float* l1) {
  // keep the product ROUNDED (ATen does): an fma-contracted src - floor(src) changes
  // lambda by ~src*2^-24 and the output by ~3e-6 relative.
#pragma clang fp contract(off)
  const float src = scale * (float)dst;
  int a = (int)src;
  if (a > in - 1) a = in - 1;
  *i0 = a;
  *i1 = a + (a < in - 1 ? 1 : 0);
  *l1 = src - (float)a;
}

__global__ void upsample2x_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int NC, int H, int W,
                                      float sh, float sw) {
  const int OH = 2 * H, OW = 2 * W;
  const long long total = (long long)NC * OH * OW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int ow = (int)(i % OW);
    const long long t = i / OW;
    const int oh = (int)(t % OH);
    const long long pl = t / OH;
    int h0, h1, w0, w1;
    float lh, lw;
    ac_src(oh, sh, H, &h0, &h1, &lh);
    ac_src(ow, sw, W, &w0, &w1, &lw);
    const float* s = x + pl * H * W;
    const float h0l = 1.f - lh, w0l = 1.f - lw;
    y[i] = h0l * (w0l * s[h0 * W + w0] + lw * s[h0 * W + w1]) + lh * (w0l * s[h1 * W + w0] + lw * s[h1 * W + w1]);
  }
}

// gather form of the adjoint: each input pixel sums the (<= ~3x3.. 5x5) outputs that read it
__global__ void upsample2x_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int NC, int H, int W,
                                      float sh, float sw) {
  const int OH = 2 * H, OW = 2 * W;
  const long long total = (long long)NC * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    const long long t = i / W;
    const int h = (int)(t % H);
    const long long pl = t / H;
    // candidate output rows: those whose src lies in (h-1, h+1)
    int oh_lo = (sh > 0.f) ? (int)floorf((float)(h - 1) / sh) : 0;
    int oh_hi = (sh > 0.f) ? (int)ceilf((float)(h + 1) / sh) : OH - 1;
    int ow_lo = (sw > 0.f) ? (int)floorf((float)(w - 1) / sw) : 0;
    int ow_hi = (sw > 0.f) ? (int)ceilf((float)(w + 1) / sw) : OW - 1;
    oh_lo = max(oh_lo - 1, 0); oh_hi = min(oh_hi + 1, OH - 1);
    ow_lo = max(ow_lo - 1, 0); ow_hi = min(ow_hi + 1, OW - 1);
    const float* g = dy + pl * OH * OW;
    // column weights once per pixel (they do not depend on the output row): <= 8 candidate columns for a x2 grid
    constexpr int MAXC = 8;
    float ww[MAXC];
    const int ncol = min(ow_hi - ow_lo + 1, MAXC);
#pragma unroll
    for (int j = 0; j < MAXC; ++j) {
      float v = 0.f;
      if (j < ncol) {
        int w0, w1; float lw;
        ac_src(ow_lo + j, sw, W, &w0, &w1, &lw);
        if (w0 == w) v += 1.f - lw;
        if (w1 == w) v += lw;
      }
      ww[j] = v;
    }
    float acc = 0.f;
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
      int h0, h1; float lh;
      ac_src(oh, sh, H, &h0, &h1, &lh);
      float wh = 0.f;
      if (h0 == h) wh += 1.f - lh;
      if (h1 == h) wh += lh;
      if (wh == 0.f) continue;
      float rowacc = 0.f;
#pragma unroll
      for (int j = 0; j < MAXC; ++j)
        if (j < ncol && ww[j] != 0.f) rowacc += ww[j] * g[oh * OW + ow_lo + j];
      acc += wh * rowacc;
    }
    dx[i] = acc;
  }
}

// Same arithmetic, four output columns per thread (one float4 store, 32-bit index math, the two source rows' few distinct
// columns served by L1): the element-per-thread kernel above ran at 1.5 TB/s on the decoder's maps.  W % 2 == 0 => OW % 4 == 0.
__global__ __launch_bounds__(256) void upsample2x_fwd_v4_kernel(const float* __restrict__ x, float* __restrict__ y, int NC, int H,
                                                                int W, float sh, float sw) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  const int OH = 2 * H, OW = 2 * W, Q4 = OW >> 2;
  const long long total = (long long)NC * OH * Q4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int q4 = (int)(i % Q4);
    const long long t = i / Q4;
    const int oh = (int)(t % OH);
    const long long pl = t / OH;
    int h0, h1;
    float lh;
    ac_src(oh, sh, H, &h0, &h1, &lh);
    const float h0l = 1.f - lh;
    const float* s0 = x + pl * H * W + (long long)h0 * W;
    const float* s1 = x + pl * H * W + (long long)h1 * W;
    // [r5] src(o) = o (W - 1) / (2 W - 1) lies in [o / 2 - 1 / 2, o / 2]: the four outputs 4 q4 .. 4 q4 + 3 read source columns
    // 2 q4 - 1 .. 2 q4 + 2 only.  That window is fetched once per row as scalar + aligned pair + scalar (6 loads per thread) and the
    // taps are picked from registers, instead of 16 four-byte gathers: the kernel was bound by its load instructions (2.0 TB/s),
    // not by bytes.  Same values, same expression.  (x 8-B aligned and W even: the pair at column 2 q4 is aligned; else gathers.)
    const int cw = 2 * q4 - 1;
    const bool win_ok = (((size_t)x) & 7) == 0;
    float r0[4], r1[4];
    if (win_ok) {
      typedef float v2 __attribute__((ext_vector_type(2)));
      const v2 m0 = *(const v2*)(s0 + cw + 1), m1 = *(const v2*)(s1 + cw + 1);       // columns 2 q4, 2 q4 + 1 < W (W even)
      r0[1] = m0[0]; r0[2] = m0[1]; r1[1] = m1[0]; r1[2] = m1[1];
      r0[0] = cw >= 0 ? s0[cw] : 0.f;
      r1[0] = cw >= 0 ? s1[cw] : 0.f;
      r0[3] = cw + 3 < W ? s0[cw + 3] : 0.f;
      r1[3] = cw + 3 < W ? s1[cw + 3] : 0.f;
    }
    v4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int w0, w1;
      float lw;
      ac_src(4 * q4 + e, sw, W, &w0, &w1, &lw);
      const float w0l = 1.f - lw;
      float a0, a1, b0, b1;
      const int i0 = w0 - cw, i1 = w1 - cw;
      if (win_ok && i0 >= 0 && i1 <= 3) {
        a0 = i0 == 0 ? r0[0] : (i0 == 1 ? r0[1] : (i0 == 2 ? r0[2] : r0[3]));
        a1 = i1 == 0 ? r0[0] : (i1 == 1 ? r0[1] : (i1 == 2 ? r0[2] : r0[3]));
        b0 = i0 == 0 ? r1[0] : (i0 == 1 ? r1[1] : (i0 == 2 ? r1[2] : r1[3]));
        b1 = i1 == 0 ? r1[0] : (i1 == 1 ? r1[1] : (i1 == 2 ? r1[2] : r1[3]));
      } else {
        a0 = s0[w0]; a1 = s0[w1]; b0 = s1[w0]; b1 = s1[w1];
      }
      o[e] = h0l * (w0l * a0 + lw * a1) + lh * (w0l * b0 + lw * b1);
    }
    *(v4*)(y + (pl * OH + oh) * OW + 4 * q4) = o;
  }
}

// Adjoint, gather form, same weights and summation order as upsample2x_bwd_kernel -- but the candidate window is the
// one a x2 align_corners grid can actually produce: src(o) = o (in - 1) / (2 in - 1) lies in (o/2 - 1/2, o/2], so
// input index i is read by outputs 2i - 1 .. 2i + 2 only (a 4 x 4 window; 6 x 6 is scanned to stay clear of rounding at
// the borders), and the row / column weights come from ONE ac_src each instead of floorf / ceilf window arithmetic.
__global__ __launch_bounds__(256) void upsample2x_bwd_win_kernel(const float* __restrict__ dy, float* __restrict__ dx, int NC,
                                                                 int H, int W, float sh, float sw) {
  const int OH = 2 * H, OW = 2 * W;
  const long long total = (long long)NC * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    const long long t = i / W;
    const int h = (int)(t % H);
    const long long pl = t / H;
    const int ow_lo = max(2 * w - 2, 0), oh_lo = max(2 * h - 2, 0);
    const int ow_hi = min(2 * w + 3, OW - 1), oh_hi = min(2 * h + 3, OH - 1);
    float ww[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float v = 0.f;
      if (ow_lo + j <= ow_hi) {
        int w0, w1; float lw;
        ac_src(ow_lo + j, sw, W, &w0, &w1, &lw);
        if (w0 == w) v += 1.f - lw;
        if (w1 == w) v += lw;
      }
      ww[j] = v;
    }
    const float* g = dy + pl * OH * OW;
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int oh = oh_lo + r;
      if (oh > oh_hi) break;
      int h0, h1; float lh;
      ac_src(oh, sh, H, &h0, &h1, &lh);
      float wh = 0.f;
      if (h0 == h) wh += 1.f - lh;
      if (h1 == h) wh += lh;
      if (wh == 0.f) continue;
      float rowacc = 0.f;
#pragma unroll
      for (int j = 0; j < 6; ++j)
        if (ow_lo + j <= ow_hi && ww[j] != 0.f) rowacc += ww[j] * g[(long long)oh * OW + ow_lo + j];
      acc += wh * rowacc;
    }
    dx[i] = acc;
  }
}

// [r4] The same gather for FOUR consecutive input columns per thread (W % 4 == 0): their 4 x 6 candidate columns 2w - 2 .. 2w + 9 lie
// in the 16 columns 2w - 4 .. 2w + 11 = four aligned float4 of a gradient row (each entirely inside or outside the row), whose
// source indices / weights are computed once per thread instead of once per input pixel.  Weights, skipped terms and summation
// order per input pixel are those of upsample2x_bwd_win_kernel: bit-identical results at a quarter of the load instructions
// (the one-pixel kernel ran at 1.1 TB/s on the 128 -> 256 map: instruction-bound, not HBM-bound).
__global__ __launch_bounds__(256) void upsample2x_bwd_win4_kernel(const float* __restrict__ dy, float* __restrict__ dx, int NC,
                                                                  int H, int W, float sh, float sw) {
  const int OH = 2 * H, OW = 2 * W, W4 = W >> 2;
  const long long total = (long long)NC * H * W4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int w = 4 * (int)(i % W4);
    const long long t = i / W4;
    const int h = (int)(t % H);
    const long long pl = t / H;
    const int cb = 2 * w - 4;                      // first of the 16 gradient columns
    int c0[16], c1[16];
    float cl[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      c0[k] = -1; c1[k] = -1; cl[k] = 0.f;
      if (k >= 2 && k < 14 && cb + k >= 0 && cb + k < OW) ac_src(cb + k, sw, W, &c0[k], &c1[k], &cl[k]);
    }
    const bool q_ok[4] = {cb >= 0, true, true, cb + 12 < OW};
    const int oh_lo = max(2 * h - 2, 0), oh_hi = min(2 * h + 3, OH - 1);
    const float* g = dy + pl * OH * OW;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int oh = oh_lo + r;
      if (oh > oh_hi) break;
      int h0, h1; float lh;
      ac_src(oh, sh, H, &h0, &h1, &lh);
      float wh = 0.f;
      if (h0 == h) wh += 1.f - lh;
      if (h1 == h) wh += lh;
      if (wh == 0.f) continue;
      float gv[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (q_ok[q]) v = *reinterpret_cast<const float4*>(g + (long long)oh * OW + cb + 4 * q);
        gv[4 * q] = v.x; gv[4 * q + 1] = v.y; gv[4 * q + 2] = v.z; gv[4 * q + 3] = v.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float rowacc = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int k = 2 * e + 2 + j;               // gradient column 2 (w + e) - 2 + j
          float wv = 0.f;
          if (c0[k] == w + e) wv += 1.f - cl[k];
          if (c1[k] == w + e) wv += cl[k];
          if (wv != 0.f) rowacc += wv * gv[k];
        }
        acc[e] += wh * rowacc;
      }
    }
    *reinterpret_cast<float4*>(dx + (pl * H + h) * W + w) = float4{acc[0], acc[1], acc[2], acc[3]};
  }
}

// [r5] The adjoint kernel above spends ~1000 instructions per 16 bytes stored -- 64-bit index divisions and the COLUMN source indices /
// weights recomputed for every row -- and ran at 2.4 - 3.0 TB/s.  Here a thread owns its four columns and walks `rpb` consecutive rows
// of the (plane, row) space: column weights once per thread, 32-bit index math, the same terms in the same order (bit-identical,
// tests/test_gpu_ops.py).  Block (tx, ty): tx = column group, ty = row lane.  Same box, us: 128 -> 256 maps 112 -> 90, 64 -> 128: 59 -> 39,
// 32 -> 64: 35 -> 31; 16 -> 32: 25 -> 28 (stays on the kernel above).  The same restructuring of the FORWARD kernel bought nothing
// (164 -> 181 us on the largest map): it is bound by its 16 four-byte gathers per thread, not by index arithmetic.
__global__ __launch_bounds__(256) void upsample2x_bwd_rows_kernel(const float* __restrict__ dy, float* __restrict__ dx, int rows_total,
                                                                  int H, int W, float sh, float sw, int rpb) {
  const int OH = 2 * H, OW = 2 * W, W4 = W >> 2;
  const int wq = blockIdx.x * blockDim.x + threadIdx.x;
  if (wq >= W4) return;
  const int w = 4 * wq;
  const int cb = 2 * w - 4;                      // first of the 16 gradient columns (upsample2x_bwd_win4_kernel)
  float wv[4][6];
  {
    int c0[16], c1[16];
    float cl[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      c0[k] = -1; c1[k] = -1; cl[k] = 0.f;
      if (k >= 2 && k < 14 && cb + k >= 0 && cb + k < OW) ac_src(cb + k, sw, W, &c0[k], &c1[k], &cl[k]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int k = 2 * e + 2 + j;
        float v = 0.f;
        if (c0[k] == w + e) v += 1.f - cl[k];
        if (c1[k] == w + e) v += cl[k];
        wv[e][j] = v;
      }
  }
  const bool q_ok[4] = {cb >= 0, true, true, cb + 12 < OW};
  const int r0 = (blockIdx.y * blockDim.y + threadIdx.y) * rpb;
  if (r0 >= rows_total) return;
  int pl = r0 / H, h = r0 % H;
  const int r1 = min(r0 + rpb, rows_total);
  for (int r = r0; r < r1; ++r) {
    const int oh_lo = max(2 * h - 2, 0), oh_hi = min(2 * h + 3, OH - 1);
    const float* g = dy + (size_t)pl * OH * OW;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) {
      const int oh = oh_lo + rr;
      if (oh > oh_hi) break;
      int h0, h1; float lh;
      ac_src(oh, sh, H, &h0, &h1, &lh);
      float wh = 0.f;
      if (h0 == h) wh += 1.f - lh;
      if (h1 == h) wh += lh;
      if (wh == 0.f) continue;
      float gv[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (q_ok[q]) v = *reinterpret_cast<const float4*>(g + (size_t)oh * OW + cb + 4 * q);
        gv[4 * q] = v.x; gv[4 * q + 1] = v.y; gv[4 * q + 2] = v.z; gv[4 * q + 3] = v.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float rowacc = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j)
          if (wv[e][j] != 0.f) rowacc += wv[e][j] * gv[2 * e + 2 + j];
        acc[e] += wh * rowacc;
      }
    }
    *reinterpret_cast<float4*>(dx + ((size_t)pl * H + h) * W + w) = float4{acc[0], acc[1], acc[2], acc[3]};
    if (++h == H) { h = 0; ++pl; }
  }
}

// launch geometry of the row-walking kernel: tx column groups (<= 256), 256 / tx row lanes, rpb rows per lane
static bool rows_geom(int groups, long long rows_total, dim3* grid, dim3* block, int* rpb) {
  if (!fcd_sw(FCD_SW_UPSAMPLE_ROWS) || rows_total >= (1ll << 31) || groups < 1) return false;
  int tx = 1;
  while (tx < groups && tx < 256) tx <<= 1;
  const int ty = 256 / tx;
  *rpb = 8;
  const long long by = (rows_total + (long long)ty * *rpb - 1) / ((long long)ty * *rpb);
  if (by > 65535) return false;
  *block = dim3((unsigned)tx, (unsigned)ty);
  *grid = dim3((unsigned)((groups + tx - 1) / tx), (unsigned)by);
  return true;
}

static inline float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

extern "C" int fcd_upsample2x_fwd(const float* x, float* y, int NC, int H, int W, void* stream) {
  FCD_CHECK_ARG(x && y && NC > 0 && H > 0 && W > 0, "fcd_upsample2x_fwd: bad arguments");
  const long long total = (long long)NC * 4 * H * W;
  FcdProfScope prof(FCD_K_POOL, (hipStream_t)stream, 0.0, 4.0 * NC * 5.0 * H * W);
  if ((W & 1) == 0 && (((size_t)y) & 15) == 0)
    hipLaunchKernelGGL(upsample2x_fwd_v4_kernel, dim3(ew_grid(total / 4)), dim3(256), 0, (hipStream_t)stream, x, y, NC, H, W,
                       ac_scale(H, 2 * H), ac_scale(W, 2 * W));
  else
    hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, y, NC, H, W,
                       ac_scale(H, 2 * H), ac_scale(W, 2 * W));
  FCD_LAUNCH_CHECK("upsample2x_fwd");
  return FCD_OK;
}

extern "C" int fcd_upsample2x_bwd(const float* dy, float* dx, int NC, int H, int W, void* stream) {
  FCD_CHECK_ARG(dy && dx && NC > 0 && H > 0 && W > 0, "fcd_upsample2x_bwd: bad arguments");
  const long long total = (long long)NC * H * W;
  FcdProfScope prof(FCD_K_POOL, (hipStream_t)stream, 0.0, 4.0 * NC * 5.0 * H * W);
  dim3 rg, rb;
  int rpb = 0;
  const bool v4ok = H >= 2 && W >= 4 && (W & 3) == 0 && (((size_t)dy | (size_t)dx) & 15) == 0;
  if (v4ok && H >= 32 && rows_geom(W >> 2, (long long)NC * H, &rg, &rb, &rpb))
    hipLaunchKernelGGL(upsample2x_bwd_rows_kernel, rg, rb, 0, (hipStream_t)stream, dy, dx, NC * H, H, W, ac_scale(H, 2 * H),
                       ac_scale(W, 2 * W), rpb);
  else if (v4ok)
    hipLaunchKernelGGL(upsample2x_bwd_win4_kernel, dim3(ew_grid(total / 4)), dim3(256), 0, (hipStream_t)stream, dy, dx, NC, H, W,
                       ac_scale(H, 2 * H), ac_scale(W, 2 * W));
  else if (H >= 2 && W >= 2)
    hipLaunchKernelGGL(upsample2x_bwd_win_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, dy, dx, NC, H, W,
                       ac_scale(H, 2 * H), ac_scale(W, 2 * W));
  else
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, dy, dx, NC, H, W,
                       ac_scale(H, 2 * H), ac_scale(W, 2 * W));
  FCD_LAUNCH_CHECK("upsample2x_bwd");
  return FCD_OK;
}

// ---- F.avg_pool2d(kernel 2, stride 2, padding = size % 2, count_include_pad) ----
// output size: floor((S + 2p - 2)/2) + 1 ; window of output o covers inputs 2o-p, 2o-p+1
__global__ void avgpool2_pad_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int NC, int H, int W,
                                        int P, int Q, int ph, int pw) {
  const long long total = (long long)NC * P * Q;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i % Q);
    const long long t = i / Q;
    const int p = (int)(t % P);
    const long long pl = t / P;
    const float* s = x + pl * H * W;
    float acc = 0.f;
    for (int a = 0; a < 2; ++a) {
      const int h = 2 * p - ph + a;
      if (h < 0 || h >= H) continue;
      for (int b = 0; b < 2; ++b) {
        const int w = 2 * q - pw + b;
        if (w < 0 || w >= W) continue;
        acc += s[h * W + w];
      }
    }
    y[i] = acc / 4.f;
  }
}

__global__ void avgpool2_pad_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int NC, int H, int W,
                                        int P, int Q, int ph, int pw) {
  const long long total = (long long)NC * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    const long long t = i / W;
    const int h = (int)(t % H);
    const long long pl = t / H;
    const int p = (h + ph) >> 1, q = (w + pw) >> 1;
    dx[i] = (p < P && q < Q) ? dy[(pl * P + p) * Q + q] * 0.25f : 0.f;
  }
}

extern "C" int fcd_avgpool2_pad_fwd(const float* x, float* y, int NC, int H, int W, void* stream) {
  FCD_CHECK_ARG(x && y && NC > 0 && H > 0 && W > 0, "fcd_avgpool2_pad_fwd: bad arguments");
  const int ph = H & 1, pw = W & 1;
  const int P = (H + 2 * ph - 2) / 2 + 1, Q = (W + 2 * pw - 2) / 2 + 1;
  const long long total = (long long)NC * P * Q;
  FcdProfScope prof(FCD_K_POOL, (hipStream_t)stream, 0.0, 4.0 * NC * 1.25 * H * W);
  hipLaunchKernelGGL(avgpool2_pad_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, y, NC, H, W,
                     P, Q, ph, pw);
  FCD_LAUNCH_CHECK("avgpool2_pad_fwd");
  return FCD_OK;
}

extern "C" int fcd_avgpool2_pad_bwd(const float* dy, float* dx, int NC, int H, int W, void* stream) {
  FCD_CHECK_ARG(dy && dx && NC > 0 && H > 0 && W > 0, "fcd_avgpool2_pad_bwd: bad arguments");
  const int ph = H & 1, pw = W & 1;
  const int P = (H + 2 * ph - 2) / 2 + 1, Q = (W + 2 * pw - 2) / 2 + 1;
  const long long total = (long long)NC * H * W;
  FcdProfScope prof(FCD_K_POOL, (hipStream_t)stream, 0.0, 4.0 * NC * 1.25 * H * W);
  hipLaunchKernelGGL(avgpool2_pad_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, dy, dx, NC, H,
                     W, P, Q, ph, pw);
  FCD_LAUNCH_CHECK("avgpool2_pad_bwd");
  return FCD_OK;
}

// ---------------------------------------------------------------------------
// Per-band normalisation of raw tiles (NORMALIZE, CommonFunc.py:199-224, applied by the reference on the HOST to the
// float64 read block before it is embedded in the zero patch, data_utils.py:106-116): out = float((double(x) - mean_c)
// / std_c) inside the valid window, 0 outside.  fp64 arithmetic per element => bit-identical to the reference's
// float32 tensors; one pass, HBM-bound.
__global__ void normalize_tiles_kernel(const float* __restrict__ x, const float* __restrict__ valid,
                                       const double* __restrict__ mean, const double* __restrict__ stdv,
                                       float* __restrict__ out, int C, long long HW, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long pl = i / HW;
    const int c = (int)(pl % C);
    const long long n = pl / C;
    const float v = valid ? valid[n * HW + (i - pl * HW)] : 1.f;
    out[i] = v != 0.f ? (float)(((double)x[i] - mean[c]) / stdv[c]) : 0.f;
  }
}

extern "C" int fcd_normalize_tiles(const float* x, const float* valid, const double* mean, const double* stdv,
                                   float* out, int N, int C, int HW, void* stream) {
  FCD_CHECK_ARG(x && mean && stdv && out && N > 0 && C > 0 && HW > 0, "fcd_normalize_tiles: bad arguments");
  const long long total = (long long)N * C * HW;
  FcdProfScope prof(FCD_K_POOL, (hipStream_t)stream, 0.0, 8.0 * total);
  hipLaunchKernelGGL(normalize_tiles_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, valid, mean,
                     stdv, out, C, (long long)HW, total);
  FCD_LAUNCH_CHECK("normalize_tiles");
  return FCD_OK;
}


// ---- pooled pair difference: AdaptiveAvgPool2d(1)(f_x - f_y) of the Discriminator (reference Module.py:211,222-223) ----
// f holds 2 * pairs groups of n samples: pair i = (group 2i, group 2i + 1).  d[i * n + s][c] = mean_p(f_x[s][c][p] - f_y[s][c][p]).
// The element-wise difference is taken FIRST, in the reference's order (f_x ~ f_y on nearly unchanged pairs: the fp32 difference of
// two close numbers is exact, the difference of two rounded means is not), and accumulated in fp64; one wave per (pair sample, channel).
__global__ void pair_gap_diff_fwd_kernel(const float* __restrict__ f, float* __restrict__ d, int pairs, int n, int C, int HW) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const int total = pairs * n * C;
  if (wave >= total) return;
  const int c = wave % C, s = (wave / C) % n, i = wave / (C * n);
  const float* fx = f + ((size_t)((2 * i) * n + s) * C + c) * HW;
  const float* fy = f + ((size_t)((2 * i + 1) * n + s) * C + c) * HW;
  double acc = 0.0;
  if ((HW & 3) == 0 && ((size_t)f & 15) == 0) {      // 16-B loads need an aligned base too (a view with a storage offset is contiguous, not aligned)
    for (int p = 4 * lane; p < HW; p += 256) {
      const float4 a = *reinterpret_cast<const float4*>(fx + p), b = *reinterpret_cast<const float4*>(fy + p);
      acc += (double)(a.x - b.x) + (double)(a.y - b.y) + (double)(a.z - b.z) + (double)(a.w - b.w);
    }
  } else {
    for (int p = lane; p < HW; p += 64) acc += (double)(fx[p] - fy[p]);
  }
  acc = wave_sum_d(acc);
  if (lane == 0) d[wave] = (float)(acc / (double)HW);
}

// adjoint: df_x = g / HW, df_y = -g / HW (every element of f written)
__global__ void pair_gap_diff_bwd_kernel(const float* __restrict__ g, float* __restrict__ df, int pairs, int n, int C, int HW) {
  const long long total = (long long)2 * pairs * n * C * HW;
  const float inv = 1.0f / (float)HW;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long pl = e / HW;                     // (group, sample, channel) plane
    const int c = (int)(pl % C);
    const long long t = pl / C;
    const int s = (int)(t % n), grp = (int)(t / n);
    const float v = g[((size_t)(grp >> 1) * n + s) * C + c] * inv;
    df[e] = (grp & 1) ? -v : v;
  }
}

extern "C" int fcd_pair_gap_diff_fwd(const float* f, float* d, int pairs, int n, int C, int HW, void* stream) {
  FCD_CHECK_ARG(f && d && pairs > 0 && n > 0 && C > 0 && HW > 0, "fcd_pair_gap_diff_fwd: bad arguments");
  const long long waves = (long long)pairs * n * C;
  FcdProfScope prof(FCD_K_POOL, (hipStream_t)stream, 0.0, 4.0 * (2.0 * waves * HW + waves));
  const int block = 256;
  hipLaunchKernelGGL(pair_gap_diff_fwd_kernel, dim3((unsigned)cdiv64(waves * 64, block)), dim3(block), 0, (hipStream_t)stream,
                     f, d, pairs, n, C, HW);
  FCD_LAUNCH_CHECK("pair_gap_diff_fwd");
  return FCD_OK;
}

extern "C" int fcd_pair_gap_diff_bwd(const float* g, float* df, int pairs, int n, int C, int HW, void* stream) {
  FCD_CHECK_ARG(g && df && pairs > 0 && n > 0 && C > 0 && HW > 0, "fcd_pair_gap_diff_bwd: bad arguments");
  const long long total = (long long)2 * pairs * n * C * HW;
  FcdProfScope prof(FCD_K_POOL, (hipStream_t)stream, 0.0, 4.0 * (double)total);
  const int block = 256;
  const long long blocks = std::min<long long>(cdiv64(total, block), 256 * 64);
  hipLaunchKernelGGL(pair_gap_diff_bwd_kernel, dim3((unsigned)blocks), dim3(block), 0, (hipStream_t)stream, g, df, pairs, n, C, HW);
  FCD_LAUNCH_CHECK("pair_gap_diff_bwd");
  return FCD_OK;
}

// ------------------------------------------------------------------------------------------------
// masked stack [r4]: z[i * N + n][c][p] = src_i[n][c][p] * (1 - cmask[n][p]),  i < k <= 4 -- the reference's
// `x * (1 - cmask).repeat(1, C, 1, 1)` of every tensor that goes into the Discriminator / the perception VGG / SSIM
// (Demo_RSSS.py:290-300, Demo_WSSS.py:264-277, Loss.py:78-79,111-112), written straight into the batch the consumer reads:
// one pass instead of rsub + k broadcast multiplies + cat, and in the backward pass one kernel instead of 2 k multiplies,
// k channel reductions, the adds joining them and a negation.  Forward values are bit-identical to the ATen sequence
// (one rounding in 1 - cmask, one in the product); the mask gradient is accumulated in fp64.
template <int VEC>
__global__ void masked_stack_fwd_kernel(const float* __restrict__ s0, const float* __restrict__ s1, const float* __restrict__ s2,
                                        const float* __restrict__ s3, int k, const float* __restrict__ cmask, float* __restrict__ z,
                                        int N, int C, int HW) {
  const long long per = (long long)N * C * (HW / VEC), total = per * k;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(e / per);
    const long long r = e % per;
    const int pv = (int)(r % (HW / VEC));
    const long long nc = r / (HW / VEC);
    const int n = (int)(nc / C);
    const float* src = i == 0 ? s0 : (i == 1 ? s1 : (i == 2 ? s2 : s3));
    if (VEC == 4) {
      const float4 m = *reinterpret_cast<const float4*>(cmask + (size_t)n * HW + 4 * pv);
      const float4 v = *reinterpret_cast<const float4*>(src + (size_t)nc * HW + 4 * pv);
      float4 o;
      o.x = v.x * (1.0f - m.x); o.y = v.y * (1.0f - m.y); o.z = v.z * (1.0f - m.z); o.w = v.w * (1.0f - m.w);
      *reinterpret_cast<float4*>(z + ((size_t)i * N * C + nc) * HW + 4 * pv) = o;
    } else {
      z[((size_t)i * N * C + nc) * HW + pv] = src[(size_t)nc * HW + pv] * (1.0f - cmask[(size_t)n * HW + pv]);
    }
  }
}

// one thread per (sample, VEC pixels): walks the k tensors' C channels; dcmask = -sum_i sum_c dz_i * src_i (fp64), d_i = dz_i * (1 - cmask)
template <int VEC>
__global__ void masked_stack_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ s0, const float* __restrict__ s1,
                                        const float* __restrict__ s2, const float* __restrict__ s3, int k,
                                        const float* __restrict__ cmask, float* __restrict__ dcmask, float* __restrict__ d0,
                                        float* __restrict__ d1, float* __restrict__ d2, float* __restrict__ d3, int N, int C, int HW) {
  const int per = HW / VEC;
  const long long total = (long long)N * per;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(e / per), pv = (int)(e % per);
    float keep[VEC];
    double acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { keep[j] = 1.0f - cmask[(size_t)n * HW + VEC * pv + j]; acc[j] = 0.0; }
    for (int i = 0; i < k; ++i) {
      const float* src = i == 0 ? s0 : (i == 1 ? s1 : (i == 2 ? s2 : s3));
      float* dst = i == 0 ? d0 : (i == 1 ? d1 : (i == 2 ? d2 : d3));
      for (int c = 0; c < C; ++c) {
        const size_t so = ((size_t)n * C + c) * HW + (size_t)VEC * pv;
        const size_t zo = (((size_t)i * N + n) * C + c) * HW + (size_t)VEC * pv;
        float g[VEC], v[VEC];
        if (VEC == 4) {
          const float4 g4 = *reinterpret_cast<const float4*>(dz + zo);
          g[0] = g4.x; g[1] = g4.y; g[2] = g4.z; g[3] = g4.w;
          if (dcmask) {
            const float4 v4 = *reinterpret_cast<const float4*>(src + so);
            v[0] = v4.x; v[1] = v4.y; v[2] = v4.z; v[3] = v4.w;
          }
        } else {
          g[0] = dz[zo];
          if (dcmask) v[0] = src[so];
        }
        if (dcmask) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) acc[j] += (double)g[j] * (double)v[j];
        }
        if (dst) {
          if (VEC == 4) {
            float4 o; o.x = g[0] * keep[0]; o.y = g[1] * keep[1]; o.z = g[2] * keep[2]; o.w = g[3] * keep[3];
            *reinterpret_cast<float4*>(dst + so) = o;
          } else {
            dst[so] = g[0] * keep[0];
          }
        }
      }
    }
    if (dcmask) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) dcmask[(size_t)n * HW + VEC * pv + j] = (float)(-acc[j]);
    }
  }
}

extern "C" int fcd_masked_stack_fwd(const float* s0, const float* s1, const float* s2, const float* s3, int k, const float* cmask,
                                    float* z, int N, int C, int HW, void* stream) {
  FCD_CHECK_ARG(s0 && cmask && z && k >= 1 && k <= 4 && N > 0 && C > 0 && HW > 0, "fcd_masked_stack_fwd: bad arguments");
  FCD_CHECK_ARG((k < 2 || s1) && (k < 3 || s2) && (k < 4 || s3), "fcd_masked_stack_fwd: fewer sources than k");
  const double elems = (double)k * N * C * HW;
  FcdProfScope prof(FCD_K_POOL, (hipStream_t)stream, 0.0, 4.0 * 2.0 * elems);
  const int block = 256;
  const size_t al_f = (size_t)s0 | (size_t)s1 | (size_t)s2 | (size_t)s3 | (size_t)cmask | (size_t)z;      // (NULL sources: 0)
  if ((HW & 3) == 0 && (al_f & 15) == 0) {
    const long long blocks = std::min<long long>(cdiv64((long long)(elems / 4), block), 256 * 32);
    hipLaunchKernelGGL(masked_stack_fwd_kernel<4>, dim3((unsigned)blocks), dim3(block), 0, (hipStream_t)stream, s0, s1, s2, s3, k, cmask, z, N, C, HW);
  } else {
    const long long blocks = std::min<long long>(cdiv64((long long)elems, block), 256 * 32);
    hipLaunchKernelGGL(masked_stack_fwd_kernel<1>, dim3((unsigned)blocks), dim3(block), 0, (hipStream_t)stream, s0, s1, s2, s3, k, cmask, z, N, C, HW);
  }
  FCD_LAUNCH_CHECK("masked_stack_fwd");
  return FCD_OK;
}

extern "C" int fcd_masked_stack_bwd(const float* dz, const float* s0, const float* s1, const float* s2, const float* s3, int k,
                                    const float* cmask, float* dcmask, float* d0, float* d1, float* d2, float* d3,
                                    int N, int C, int HW, void* stream) {
  FCD_CHECK_ARG(dz && cmask && k >= 1 && k <= 4 && N > 0 && C > 0 && HW > 0, "fcd_masked_stack_bwd: bad arguments");
  FCD_CHECK_ARG(!dcmask || (s0 && (k < 2 || s1) && (k < 3 || s2) && (k < 4 || s3)), "fcd_masked_stack_bwd: the mask gradient needs every source");
  FCD_CHECK_ARG(dcmask || d0 || d1 || d2 || d3, "fcd_masked_stack_bwd: nothing to compute");
  const double elems = (double)k * N * C * HW;
  FcdProfScope prof(FCD_K_POOL, (hipStream_t)stream, 0.0, 4.0 * 2.0 * elems);
  const int block = 128;
  const size_t al_b = (size_t)dz | (size_t)s0 | (size_t)s1 | (size_t)s2 | (size_t)s3 | (size_t)cmask | (size_t)dcmask | (size_t)d0 |
                      (size_t)d1 | (size_t)d2 | (size_t)d3;
  if ((HW & 3) == 0 && (al_b & 15) == 0) {
    hipLaunchKernelGGL(masked_stack_bwd_kernel<4>, dim3((unsigned)cdiv64((long long)N * (HW / 4), block)), dim3(block), 0, (hipStream_t)stream,
                       dz, s0, s1, s2, s3, k, cmask, dcmask, d0, d1, d2, d3, N, C, HW);
  } else {
    hipLaunchKernelGGL(masked_stack_bwd_kernel<1>, dim3((unsigned)cdiv64((long long)N * HW, block)), dim3(block), 0, (hipStream_t)stream,
                       dz, s0, s1, s2, s3, k, cmask, dcmask, d0, d1, d2, d3, N, C, HW);
  }
  FCD_LAUNCH_CHECK("masked_stack_bwd");
  return FCD_OK;
}
