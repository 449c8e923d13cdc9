// 1x1 convolution to ONE output channel with the sigmoid in its epilogue: the Segmentor's change-density head
// (reference Module.py:82-90 `OutConv`: Conv2d(128, 1, 1) + Sigmoid on 256 x 256 maps).  As a GEMM this layer has one
// row: on the implicit-GEMM MFMA tiles it used 1 of 32 rows and ran at 1.1 (weight gradient) .. 3.3 TB/s (forward) of
// its only real cost, ONE pass over the (N, C, HW) activation.  Three streaming kernels instead [r3]:
//   forward   y[n, p]     = sigmoid(b + sum_c w[c] x[n, c, p])            reads x once, 4 pixels per thread
//   data grad dx[n, c, p] = w[c] g[n, p],  g = dy * y (1 - y)             writes dx once
//   weight    dw[c]       = sum_{n, p} g[n, p] x[n, c, p],  db = sum g    one block per (c, n) plane -> fp64 partial,
//                                                                          summed over n in a fixed order (deterministic)
#include "common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }

// grid (HW / 4 / 256 rounded up, N)
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int C, int HW4,
                                                        int sigmoid) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= HW4) return;
  const int n = blockIdx.y;
  const f32x4* xp = (const f32x4*)(x + (size_t)n * C * HW4 * 4) + i;
  const float b = bias ? bias[0] : 0.f;
  f32x4 acc = {b, b, b, b};
  int c = 0;
  for (; c + 8 <= C; c += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(xp + (size_t)(c + u) * HW4);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u] * w[c + u];
  }
  for (; c < C; ++c) acc += __builtin_nontemporal_load(xp + (size_t)c * HW4) * w[c];
  if (sigmoid) acc = f32x4{sigmoidf_(acc[0]), sigmoidf_(acc[1]), sigmoidf_(acc[2]), sigmoidf_(acc[3])};
  ((f32x4*)(y + (size_t)n * HW4 * 4))[i] = acc;
}

__device__ __forceinline__ f32x4 head_grad(const float* __restrict__ dy, const float* __restrict__ ys, size_t off4) {
  f32x4 g = ((const f32x4*)dy)[off4];
  if (ys) {
    const f32x4 s = ((const f32x4*)ys)[off4];
    g = g * s * (1.f - s);
  }
  return g;
}

// grid (HW / 4 / 256 rounded up, N)
__global__ __launch_bounds__(256) void head_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ ys,
                                                          const float* __restrict__ w, float* __restrict__ dx, int C, int HW4) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= HW4) return;
  const int n = blockIdx.y;
  const f32x4 g = head_grad(dy, ys, (size_t)n * HW4 + i);
  f32x4* dp = (f32x4*)(dx + (size_t)n * C * HW4 * 4) + i;
  for (int c = 0; c < C; ++c) dp[(size_t)c * HW4] = g * w[c];
}

// grid (C + 1, N): block (c, n) -> part[c * N + n] = sum_p g[n, p] x[n, c, p]; row c == C: sum_p g[n, p]
__global__ __launch_bounds__(256) void head_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const float* __restrict__ ys, double* __restrict__ part, int C,
                                                          int HW4) {
  const int c = blockIdx.x, n = blockIdx.y, N = gridDim.y;
  const f32x4* xp = (c < C) ? (const f32x4*)(x + ((size_t)n * C + c) * HW4 * 4) : nullptr;
  double s = 0.0;
  for (int i0 = threadIdx.x; i0 < HW4; i0 += 256 * 8) {      // fp32 over <= 32 products, then fp64
    float a = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 256;
      if (i < HW4) {
        const f32x4 g = head_grad(dy, ys, (size_t)n * HW4 + i);
        if (xp) {
          const f32x4 v = __builtin_nontemporal_load(xp + i);
          a += g[0] * v[0] + g[1] * v[1] + g[2] * v[2] + g[3] * v[3];
        } else {
          a += g[0] + g[1] + g[2] + g[3];
        }
      }
    }
    s += (double)a;
  }
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
#pragma unroll
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[(size_t)c * N + n] = red[0];
}

__global__ void head_wgrad_final_kernel(const double* __restrict__ part, float* __restrict__ dw, float* __restrict__ db, int C,
                                        int N) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > C) return;
  double s = 0.0;
  for (int n = 0; n < N; ++n) s += part[(size_t)c * N + n];
  if (c < C) { if (dw) dw[c] = (float)s; }
  else if (db) db[0] = (float)s;
}

// ---------------------------------------------------------------------------------------------------------------------
// [r5] The head behind a train-mode BatchNorm + ReLU (reference Module.py:25-31 -> :82-90: the last DoubleConv of the decoder feeds
// OutConv and nothing else).  The 128-channel activation a = relu(z * scale + shift) is the largest tensor of the Segmentor
// (268 MB at 8 x 256 x 256) and every pass over it is pure HBM time.  Unfused, the tail costs ten passes: BatchNorm apply (read z,
// write a), head forward (read a); backward: head data gradient (write da), head weight gradient (read a), BatchNorm reduce (read
// da, z), BatchNorm apply (read da, z, write dz).  The head has ONE output channel, so da[c, p] = w[c] g[p] with g = dy y (1 - y) a
// single-channel map that stays in L2: a, da never need to exist.  Four passes:
//   forward   y = sigmoid(b + sum_c w[c] relu(z[c] scale[c] + shift[c]))                                          read z
//   reduce    per (c, n): Sa = sum g a (dw), S1 = sum g [a > 0], S2 = sum g [a > 0] xhat, xhat = (z - mean) invstd      read z
//   apply     dz = scale (w[c] g [a > 0] - k1 - xhat k2), k1 = w[c] S1 / count, k2 = w[c] S2 / count              read z, write dz
// dgamma = w[c] S2, dbeta = w[c] S1, db = sum g.  Forward values are bit-identical to the unfused kernels (same products in the same
// order); the gradients differ from them by the rounding of w[c] * (sum) against sum of (w[c] * term): fp64 sums, ~1e-7 relative.
__global__ __launch_bounds__(256) void head_fwd_bn_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int Ng, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int C, int HW4,
                                                           int sigmoid) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= HW4) return;
  const int n = blockIdx.y;
  const f32x4* xp = (const f32x4*)(z + (size_t)n * C * HW4 * 4) + i;
  const float* sc = scale + (size_t)(n / Ng) * C;
  const float* sh = shift + (size_t)(n / Ng) * C;
  const float b = bias ? bias[0] : 0.f;
  f32x4 acc = {b, b, b, b};
  auto act = [](f32x4 v, float s, float t) {
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float yv = fmaf(v[e], s, t);
      r[e] = yv > 0.f ? yv : 0.f;
    }
    return r;
  };
  int c = 0;
  for (; c + 8 <= C; c += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(xp + (size_t)(c + u) * HW4);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += act(v[u], sc[c + u], sh[c + u]) * w[c + u];
  }
  for (; c < C; ++c) acc += act(__builtin_nontemporal_load(xp + (size_t)c * HW4), sc[c], sh[c]) * w[c];
  if (sigmoid) acc = f32x4{sigmoidf_(acc[0]), sigmoidf_(acc[1]), sigmoidf_(acc[2]), sigmoidf_(acc[3])};
  ((f32x4*)(y + (size_t)n * HW4 * 4))[i] = acc;
}

// grid (C + 1, N): block (c, n) -> part[(c N + n) 3 + {0, 1, 2}] = {Sa, S1, S2}; row c == C: {sum g, -, -}
__global__ __launch_bounds__(256) void head_bn_reduce_kernel(const float* __restrict__ z, const float* __restrict__ dy,
                                                              const float* __restrict__ ys, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, int Ng,
                                                              double* __restrict__ part, int C, int HW4) {
  const int c = blockIdx.x, n = blockIdx.y, N = gridDim.y;
  const f32x4* xp = (c < C) ? (const f32x4*)(z + ((size_t)n * C + c) * HW4 * 4) : nullptr;
  const int sidx = (n / Ng) * C + min(c, C - 1);
  const float sc = scale[sidx], sh = shift[sidx], mu = mean[sidx], is = invstd[sidx];
  double sa = 0.0, s1 = 0.0, s2 = 0.0;
  for (int i0 = threadIdx.x; i0 < HW4; i0 += 256 * 4) {
    f32x4 v[4], g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 256;
      v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      g[u] = v[u];
      if (i < HW4) {
        g[u] = head_grad(dy, ys, (size_t)n * HW4 + i);
        if (xp) v[u] = __builtin_nontemporal_load(xp + i);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (xp) {
          const float yv = fmaf(v[u][e], sc, sh);
          const float gp = yv > 0.f ? g[u][e] : 0.f;
          sa += (double)gp * (double)yv;
          s1 += (double)gp;
          s2 += (double)gp * (double)((v[u][e] - mu) * is);
        } else {
          sa += (double)g[u][e];
        }
      }
  }
  __shared__ double red[16];
  sa = block_sum_d(sa, red);
  s1 = block_sum_d(s1, red);
  s2 = block_sum_d(s2, red);
  if (threadIdx.x == 0) {
    double* o = part + ((size_t)c * N + n) * 3;
    o[0] = sa;
    o[1] = s1;
    o[2] = s2;
  }
}

// one thread per channel (+ one for the bias): sums over the samples in a fixed order; coef[(g C + c) 2 + {0, 1}] = {k1, k2}
__global__ void head_bn_final_kernel(const double* __restrict__ part, const float* __restrict__ w, float* __restrict__ coef,
                                     float* __restrict__ dw, float* __restrict__ db, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta, int C, int N, int Ng, double inv_count) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > C) return;
  if (c == C) {
    double s = 0.0;
    for (int n = 0; n < N; ++n) s += part[((size_t)c * N + n) * 3];
    if (db) db[0] = (float)s;
    return;
  }
  const double wc = (double)w[c];
  double ta = 0.0, t1 = 0.0, t2 = 0.0;
  for (int g = 0; g < N / Ng; ++g) {
    double a1 = 0.0, a2 = 0.0;
    for (int j = 0; j < Ng; ++j) {
      const double* p = part + ((size_t)c * N + g * Ng + j) * 3;
      ta += p[0];
      a1 += p[1];
      a2 += p[2];
    }
    coef[((size_t)g * C + c) * 2 + 0] = (float)(wc * a1 * inv_count);
    coef[((size_t)g * C + c) * 2 + 1] = (float)(wc * a2 * inv_count);
    t1 += a1;
    t2 += a2;
  }
  if (dw) dw[c] = (float)ta;
  if (dgamma) dgamma[c] = (float)(wc * t2);
  if (dbeta) dbeta[c] = (float)(wc * t1);
}

// grid (HW / 4 / 256 rounded up, N)
__global__ __launch_bounds__(256) void head_bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ dy,
                                                             const float* __restrict__ ys, const float* __restrict__ w,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ coef, int Ng, float* __restrict__ dz, int C,
                                                             int HW4) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= HW4) return;
  const int n = blockIdx.y;
  const f32x4 g = head_grad(dy, ys, (size_t)n * HW4 + i);
  const f32x4* xp = (const f32x4*)(z + (size_t)n * C * HW4 * 4) + i;
  f32x4* dp = (f32x4*)(dz + (size_t)n * C * HW4 * 4) + i;
  const size_t so = (size_t)(n / Ng) * C;
  auto one = [&](int c, f32x4 v) {
    const float sc = scale[so + c], sh = shift[so + c], mu = mean[so + c], is = invstd[so + c];
    const float k1 = coef[(so + c) * 2], k2 = coef[(so + c) * 2 + 1], wc = w[c];
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float yv = fmaf(v[e], sc, sh);
      const float dyv = (g[e] * wc) * (yv > 0.f ? 1.f : 0.f);      // head data gradient, then the ReLU gate: the unfused kernels' values
      r[e] = sc * (dyv - k1 - (v[e] - mu) * is * k2);
    }
    dp[(size_t)c * HW4] = r;
  };
  int c = 0;
  for (; c + 4 <= C; c += 4) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(xp + (size_t)(c + u) * HW4);
#pragma unroll
    for (int u = 0; u < 4; ++u) one(c + u, v[u]);
  }
  for (; c < C; ++c) one(c, __builtin_nontemporal_load(xp + (size_t)c * HW4));
}

}  // namespace

extern "C" int fcd_conv1x1_head_plan(int N, int C, int HW, int K) {
  return (fcd_sw(FCD_SW_CONV_HEAD) && K == 1 && N > 0 && N <= 65535 && C >= 8 && C <= 65534 && HW >= 1024 && (HW & 3) == 0) ? 1 : 0;
}

extern "C" int fcd_conv1x1_head_fwd(const float* x, const float* w, const float* bias, float* y, int N, int C, int HW, int sigmoid,
                                    void* stream) {
  FCD_CHECK_ARG(x && w && y, "fcd_conv1x1_head_fwd: null pointer");
  FCD_CHECK_ARG(fcd_conv1x1_head_plan(N, C, HW, 1), "fcd_conv1x1_head_fwd: unsupported shape N=%d C=%d HW=%d", N, C, HW);
  const int HW4 = HW >> 2;
  FcdProfScope prof(FCD_K_CONV_FWD, (hipStream_t)stream, 2.0 * N * C * (double)HW, 4.0 * N * (C + 1.0) * HW,
                    fcd_prof_tagf("head_fwd N=%d C=%d HW=%d", N, C, HW));
  hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)cdiv(HW4, 256), (unsigned)N), dim3(256), 0, (hipStream_t)stream, x, w, bias, y,
                     C, HW4, sigmoid);
  FCD_LAUNCH_CHECK("conv1x1_head_fwd");
  return FCD_OK;
}

extern "C" size_t fcd_conv1x1_head_bwd_ws_bytes(int N, int C) { return (size_t)(C + 1) * (size_t)N * sizeof(double); }

// g = dy * y_sig (1 - y_sig) when y_sig != NULL (the forward's sigmoid output), else g = dy.  Any of dx / (dw, db) may be NULL.
extern "C" int fcd_conv1x1_head_bwd(const float* x, const float* w, const float* dy, const float* y_sig, float* dx, float* dw,
                                    float* db, int N, int C, int HW, void* ws, size_t ws_bytes, void* stream) {
  FCD_CHECK_ARG(dy && (dx || dw || db), "fcd_conv1x1_head_bwd: null pointer");
  FCD_CHECK_ARG(fcd_conv1x1_head_plan(N, C, HW, 1), "fcd_conv1x1_head_bwd: unsupported shape N=%d C=%d HW=%d", N, C, HW);
  hipStream_t st = (hipStream_t)stream;
  const int HW4 = HW >> 2;
  if (dx) {
    FCD_CHECK_ARG(w, "fcd_conv1x1_head_bwd: dx needs the filter");
    FcdProfScope prof(FCD_K_CONV_DGRAD, st, 2.0 * N * C * (double)HW, 4.0 * N * (C + 1.0) * HW,
                      fcd_prof_tagf("head_dgrad N=%d C=%d HW=%d", N, C, HW));
    hipLaunchKernelGGL(head_dgrad_kernel, dim3((unsigned)cdiv(HW4, 256), (unsigned)N), dim3(256), 0, st, dy, y_sig, w, dx, C, HW4);
  }
  if (dw || db) {
    FCD_CHECK_ARG(x, "fcd_conv1x1_head_bwd: dw needs the input");
    if (!ws || ws_bytes < fcd_conv1x1_head_bwd_ws_bytes(N, C)) {
      fcd_set_error("fcd_conv1x1_head_bwd: workspace too small");
      return FCD_ERR_WORKSPACE;
    }
    FcdProfScope prof(FCD_K_CONV_WGRAD, st, 2.0 * N * C * (double)HW, 4.0 * N * (C + 1.0) * HW,
                      fcd_prof_tagf("head_wgrad N=%d C=%d HW=%d", N, C, HW));
    hipLaunchKernelGGL(head_wgrad_kernel, dim3((unsigned)(C + 1), (unsigned)N), dim3(256), 0, st, x, dy, y_sig, (double*)ws, C, HW4);
    hipLaunchKernelGGL(head_wgrad_final_kernel, dim3((unsigned)cdiv(C + 1, 128)), dim3(128), 0, st, (const double*)ws, dw, db, C, N);
  }
  FCD_LAUNCH_CHECK("conv1x1_head_bwd");
  return FCD_OK;
}

// ---- [r5] head behind a train-mode BatchNorm + ReLU (kernels above).  scale / shift / mean / invstd: [groups][C] as fcd_bn_train_stats
// leaves them (scale, shift) and saves them (mean, invstd); groups | N.
extern "C" int fcd_conv1x1_head_bn_fwd(const float* z, const float* scale, const float* shift, int groups, const float* w,
                                       const float* bias, float* y, int N, int C, int HW, int sigmoid, void* stream) {
  FCD_CHECK_ARG(z && scale && shift && w && y && groups > 0 && N % groups == 0, "fcd_conv1x1_head_bn_fwd: bad arguments");
  FCD_CHECK_ARG(fcd_conv1x1_head_plan(N, C, HW, 1), "fcd_conv1x1_head_bn_fwd: unsupported shape N=%d C=%d HW=%d", N, C, HW);
  const int HW4 = HW >> 2;
  FcdProfScope prof(FCD_K_CONV_FWD, (hipStream_t)stream, 2.0 * N * C * (double)HW, 4.0 * N * (C + 1.0) * HW,
                    fcd_prof_tagf("head_bn_fwd N=%d C=%d HW=%d", N, C, HW));
  hipLaunchKernelGGL(head_fwd_bn_kernel, dim3((unsigned)cdiv(HW4, 256), (unsigned)N), dim3(256), 0, (hipStream_t)stream, z, scale,
                     shift, N / groups, w, bias, y, C, HW4, sigmoid);
  FCD_LAUNCH_CHECK("conv1x1_head_bn_fwd");
  return FCD_OK;
}

extern "C" size_t fcd_conv1x1_head_bn_bwd_ws_bytes(int N, int C, int groups) {
  return (size_t)(C + 1) * (size_t)N * 3 * sizeof(double) + (size_t)groups * C * 2 * sizeof(float) + 256;
}

// dz, dw (C floats), db (1), dgamma, dbeta (C each; any of the four parameter gradients may be NULL) of
// y = [sigmoid](b + sum_c w[c] relu(BatchNorm_train(z)[c])) given dy; y_sig as in fcd_conv1x1_head_bwd
extern "C" int fcd_conv1x1_head_bn_bwd(const float* z, const float* w, const float* dy, const float* y_sig, const float* scale,
                                       const float* shift, const float* mean, const float* invstd, int groups, float* dz, float* dw,
                                       float* db, float* dgamma, float* dbeta, int N, int C, int HW, void* ws, size_t ws_bytes,
                                       void* stream) {
  FCD_CHECK_ARG(z && w && dy && scale && shift && mean && invstd && dz && groups > 0 && N % groups == 0,
                "fcd_conv1x1_head_bn_bwd: bad arguments");
  FCD_CHECK_ARG(fcd_conv1x1_head_plan(N, C, HW, 1), "fcd_conv1x1_head_bn_bwd: unsupported shape N=%d C=%d HW=%d", N, C, HW);
  if (!ws || ws_bytes < fcd_conv1x1_head_bn_bwd_ws_bytes(N, C, groups)) {
    fcd_set_error("fcd_conv1x1_head_bn_bwd: workspace too small");
    return FCD_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int HW4 = HW >> 2, Ng = N / groups;
  double* part = (double*)ws;
  float* coef = (float*)(part + (size_t)(C + 1) * N * 3);
  {
    FcdProfScope prof(FCD_K_CONV_WGRAD, st, 2.0 * N * C * (double)HW, 4.0 * N * (C + 1.0) * HW,
                      fcd_prof_tagf("head_bn_reduce N=%d C=%d HW=%d", N, C, HW));
    hipLaunchKernelGGL(head_bn_reduce_kernel, dim3((unsigned)(C + 1), (unsigned)N), dim3(256), 0, st, z, dy, y_sig, scale, shift, mean,
                       invstd, Ng, part, C, HW4);
    hipLaunchKernelGGL(head_bn_final_kernel, dim3((unsigned)cdiv(C + 1, 128)), dim3(128), 0, st, (const double*)part, w, coef, dw, db,
                       dgamma, dbeta, C, N, Ng, 1.0 / ((double)Ng * HW));
  }
  {
    FcdProfScope prof(FCD_K_CONV_DGRAD, st, 2.0 * N * C * (double)HW, 4.0 * N * (2.0 * C + 1.0) * HW,
                      fcd_prof_tagf("head_bn_dgrad N=%d C=%d HW=%d", N, C, HW));
    hipLaunchKernelGGL(head_bn_apply_kernel, dim3((unsigned)cdiv(HW4, 256), (unsigned)N), dim3(256), 0, st, z, dy, y_sig, w, scale,
                       shift, mean, invstd, (const float*)coef, Ng, dz, C, HW4);
  }
  FCD_LAUNCH_CHECK("conv1x1_head_bn_bwd");
  return FCD_OK;
}
