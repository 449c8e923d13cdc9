// BatchNorm2d (train / eval) fused with ReLU / LeakyReLU / PReLU, forward and
// backward, with "sample groups" so that the Siamese / shared-net double
// invocation of one BN layer (Module.py:114-131, :220-221) runs as ONE batched
// launch while keeping per-call batch statistics and the ordered running-stat
// updates of the reference.  HBM-bound: statistics are wavefront-reduced in
// fp64, the normalise+activate pass is one read + one write with float4 access.
#include "common.h"

#define BN_MAX_SPLIT 64
// [r5] backward kernels: dz and the BatchNorm input are each read once by the reduce pass and once by the apply pass, tensors of up
// to 268 MB that no cache holds between the two: non-temporal (streaming) loads.  Measured on the Segmentor's layer shapes
// (tools/bn_probe.py under rocprofv3): reduce 81 -> 62 us, apply 90 -> 96 us per call on average (the apply pass lived on what
// the reduce pass left in the Infinity Cache; with both streaming the pair is 8 % faster).  The forward statistics / apply kernels
// keep the default policy: their input was written by the convolution's output transform just before.
typedef float bn_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 NTLD(const float* p) {
  const bn_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const bn_f32x4*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}

struct BnWs {
  double* part;   // [G*C*BN_MAX_SPLIT*3]
  double* fin;    // [G*C*3]
  float* scale;   // [G*C]
  float* shift;   // [G*C]
  float* mean_e;  // [G*C] eval-mode mean / invstd (from running stats)
  float* invstd_e;
};

static BnWs carve(void* ws, int C, int G) {
  BnWs w;
  char* p = (char*)ws;
  w.part = (double*)p; p += sizeof(double) * (size_t)G * C * BN_MAX_SPLIT * 3;
  w.fin = (double*)p;  p += sizeof(double) * (size_t)G * C * 3;
  w.scale = (float*)p; p += sizeof(float) * (size_t)G * C;
  w.shift = (float*)p; p += sizeof(float) * (size_t)G * C;
  w.mean_e = (float*)p; p += sizeof(float) * (size_t)G * C;
  w.invstd_e = (float*)p; p += sizeof(float) * (size_t)G * C;
  return w;
}

extern "C" size_t fcd_bn_act_ws_bytes(int C, int groups) {
  return (size_t)groups * C * (sizeof(double) * (BN_MAX_SPLIT * 3 + 3) + 4 * sizeof(float)) + 256;
}

static int pick_split(int C, int G, long long per_group_elems) {
  int split = cdiv(2048, C * G);
  const long long maxs = std::max<long long>(1, per_group_elems / 2048);
  if (split > maxs) split = (int)maxs;
  if (split > BN_MAX_SPLIT) split = BN_MAX_SPLIT;
  if (split < 1) split = 1;
  return split;
}

// ---- forward statistics ------------------------------------------------------
// grid (C, G, split); part[((g*C+c)*split+sp)*3 + {0,1}] = {sum x, sum x^2}
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, double* __restrict__ part,
                                                       int C, int HW, int Ng, int split) {
  __shared__ double red[16];
  const int c = blockIdx.x, g = blockIdx.y, sp = blockIdx.z;
  double s1 = 0.0, s2 = 0.0;
  if ((HW & 3) == 0 && ((size_t)x & 15) == 0) {
    const int hw4 = HW >> 2;
    const long long total = (long long)Ng * hw4;
    const long long chunk = (total + split - 1) / split;
    const long long beg = sp * chunk, end = min(beg + chunk, total);
    for (long long e = beg + threadIdx.x; e < end; e += 256) {
      const int n = (int)(e / hw4), i = (int)(e % hw4);
      const float4 v = reinterpret_cast<const float4*>(x + ((size_t)(g * Ng + n) * C + c) * HW)[i];
      s1 += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
      s2 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
  } else {
    const long long total = (long long)Ng * HW;
    const long long chunk = (total + split - 1) / split;
    const long long beg = sp * chunk, end = min(beg + chunk, total);
    for (long long e = beg + threadIdx.x; e < end; e += 256) {
      const int n = (int)(e / HW), i = (int)(e % HW);
      const float v = x[((size_t)(g * Ng + n) * C + c) * HW + i];
      s1 += (double)v;
      s2 += (double)v * v;
    }
  }
  s1 = block_sum_d(s1, red);
  s2 = block_sum_d(s2, red);
  if (threadIdx.x == 0) {
    double* o = part + ((size_t)(g * C + c) * split + sp) * 3;
    o[0] = s1;
    o[1] = s2;
  }
}

// one thread per channel: finish stats for every group in order, update running stats.
// [r5] norder > 0: the running statistics take the groups' updates in the order the 4-bit entries of order_code list (entry i =
// bits 4 i .. 4 i + 3; a group may appear more than once) instead of 0 .. G - 1 once each: a net that was called twice on the SAME
// samples in train mode (the Discriminator's shared first argument, Demo_RSSS.py:293,302) normalises them once here and still leaves
// the running statistics the two calls would have left.
__global__ void bn_finalize_kernel(const double* __restrict__ part, int C, int G, int split, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float momentum, float eps, float* __restrict__ save_mean,
                                   float* __restrict__ save_invstd, float* __restrict__ scale,
                                   float* __restrict__ shift, unsigned long long order_code = 0ull, int norder = 0) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 1.f;
  float gm[16], gu[16];          // replay only (G <= 16 there)
  for (int g = 0; g < G; ++g) {
    double s1 = 0.0, s2 = 0.0;
    const double* p = part + (size_t)(g * C + c) * split * 3;
    for (int s = 0; s < split; ++s) {
      s1 += p[s * 3 + 0];
      s2 += p[s * 3 + 1];
    }
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float meanf = (float)mean;
    save_mean[g * C + c] = meanf;
    save_invstd[g * C + c] = invstd;
    const float sc = gamma[c] * invstd;
    scale[g * C + c] = sc;
    shift[g * C + c] = beta[c] - meanf * sc;
    const float unbiased = (float)(count > 1.0 ? var * count / (count - 1.0) : var);
    if (norder > 0) {
      gm[g & 15] = meanf;
      gu[g & 15] = unbiased;
    } else {
      rm = (1.f - momentum) * rm + momentum * meanf;
      rv = (1.f - momentum) * rv + momentum * unbiased;
    }
  }
  for (int i = 0; i < norder; ++i) {
    const int g = (int)((order_code >> (4 * i)) & 15ull);
    rm = (1.f - momentum) * rm + momentum * gm[g];
    rv = (1.f - momentum) * rv + momentum * gu[g];
  }
  if (running_mean) running_mean[c] = rm;
  if (running_var) running_var[c] = rv;
}

__global__ void bn_eval_prep_kernel(int C, int G, const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                    float eps, float* __restrict__ scale, float* __restrict__ shift,
                                    float* __restrict__ mean_e, float* __restrict__ invstd_e) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.f / sqrtf(running_var[c] + eps);
  const float sc = gamma[c] * invstd;
  const float sh = beta[c] - running_mean[c] * sc;
  for (int g = 0; g < G; ++g) {
    scale[g * C + c] = sc;
    shift[g * C + c] = sh;
    mean_e[g * C + c] = running_mean[c];
    invstd_e[g * C + c] = invstd;
  }
}

// ---- forward apply: y = act(x*scale + shift) ------------------------------------
__global__ __launch_bounds__(256) void bn_act_apply_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           int C, int HW, int Ng, int affine,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int act,
                                                           const float* __restrict__ slope_ptr, float slope_imm) {
  const int plane = blockIdx.x;
  const int n = plane / C, c = plane % C;
  const int sidx = (n / Ng) * C + c;
  const float sc = affine ? scale[sidx] : 1.f;
  const float sh = affine ? shift[sidx] : 0.f;
  const float slope = slope_ptr ? slope_ptr[0] : slope_imm;
  const float* xp = x + (size_t)plane * HW;
  float* yp = y + (size_t)plane * HW;
  if ((HW & 3) == 0 && ((((size_t)x | (size_t)y)) & 15) == 0) {
    const int hw4 = HW >> 2;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < hw4; i += gridDim.y * 256) {
      float4 v = reinterpret_cast<const float4*>(xp)[i];
      v.x = act_apply(fmaf(v.x, sc, sh), act, slope);
      v.y = act_apply(fmaf(v.y, sc, sh), act, slope);
      v.z = act_apply(fmaf(v.z, sc, sh), act, slope);
      v.w = act_apply(fmaf(v.w, sc, sh), act, slope);
      reinterpret_cast<float4*>(yp)[i] = v;
    }
  } else {
    for (int i = blockIdx.y * 256 + threadIdx.x; i < HW; i += gridDim.y * 256)
      yp[i] = act_apply(fmaf(xp[i], sc, sh), act, slope);
  }
}

static dim3 plane_grid(int planes, int HW) {
  int chunks = cdiv(HW, 256 * 4 * 4);
  if (chunks > 64) chunks = 64;
  if (chunks < 1) chunks = 1;
  return dim3((unsigned)planes, (unsigned)chunks);
}

static int bn_act_fwd_impl(const float* x, float* y, int N, int C, int HW, int groups, int has_bn,
                           const float* gamma, const float* beta, float* running_mean, float* running_var,
                           float momentum, float eps, int training, float* save_mean, float* save_invstd,
                           int act, const float* slope, float slope_imm, void* ws, size_t ws_bytes,
                           void* stream, unsigned long long order_code, int norder) {
  FCD_CHECK_ARG(x && y && N > 0 && C > 0 && HW > 0 && groups > 0 && N % groups == 0,
                "fcd_bn_act_fwd: bad geometry N=%d C=%d HW=%d groups=%d", N, C, HW, groups);
  hipStream_t st = (hipStream_t)stream;
  const int Ng = N / groups;
  BnWs w{};
  FcdProfScope prof(FCD_K_NORM, st, 0.0, 4.0 * N * C * (double)HW * (has_bn && training ? 3.0 : 2.0));
  if (has_bn) {
    FCD_CHECK_ARG(gamma && beta, "fcd_bn_act_fwd: BN needs gamma/beta");
    if (ws == nullptr || ws_bytes < fcd_bn_act_ws_bytes(C, groups)) {
      fcd_set_error("fcd_bn_act_fwd: workspace too small");
      return FCD_ERR_WORKSPACE;
    }
    w = carve(ws, C, groups);
    if (training) {
      FCD_CHECK_ARG(save_mean && save_invstd, "fcd_bn_act_fwd: training needs save buffers");
      const int split = pick_split(C, groups, (long long)Ng * HW);
      hipLaunchKernelGGL(bn_stats_kernel, dim3(C, groups, split), dim3(256), 0, st, x, w.part, C, HW, Ng, split);
      hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, (const double*)w.part, C, groups,
                         split, (double)Ng * HW, gamma, beta, running_mean, running_var, momentum, eps, save_mean,
                         save_invstd, w.scale, w.shift, order_code, norder);
    } else {
      FCD_CHECK_ARG(running_mean && running_var, "fcd_bn_act_fwd: eval needs running stats");
      hipLaunchKernelGGL(bn_eval_prep_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, C, groups, gamma, beta,
                         (const float*)running_mean, (const float*)running_var, eps, w.scale, w.shift, w.mean_e,
                         w.invstd_e);
    }
  }
  hipLaunchKernelGGL(bn_act_apply_kernel, plane_grid(N * C, HW), dim3(256), 0, st, x, y, C, HW, Ng, has_bn,
                     (const float*)w.scale, (const float*)w.shift, act, slope, slope_imm);
  FCD_LAUNCH_CHECK("bn_act_fwd");
  return FCD_OK;
}

extern "C" int fcd_bn_act_fwd(const float* x, float* y, int N, int C, int HW, int groups, int has_bn,
                              const float* gamma, const float* beta, float* running_mean, float* running_var,
                              float momentum, float eps, int training, float* save_mean, float* save_invstd,
                              int act, const float* slope, float slope_imm, void* ws, size_t ws_bytes,
                              void* stream) {
  return bn_act_fwd_impl(x, y, N, C, HW, groups, has_bn, gamma, beta, running_mean, running_var, momentum, eps, training,
                         save_mean, save_invstd, act, slope, slope_imm, ws, ws_bytes, stream, 0ull, 0);
}

// train-mode BatchNorm + activation whose running statistics replay the groups' updates in the order `order[0 .. norder)` (host
// array of group indices, repeats allowed, <= 16 entries, groups <= 16)
extern "C" int fcd_bn_act_fwd_replay(const float* x, float* y, int N, int C, int HW, int groups, const int* order, int norder,
                                     const float* gamma, const float* beta, float* running_mean, float* running_var,
                                     float momentum, float eps, float* save_mean, float* save_invstd, int act,
                                     const float* slope, float slope_imm, void* ws, size_t ws_bytes, void* stream) {
  FCD_CHECK_ARG(order && norder > 0 && norder <= 16 && groups > 0 && groups <= 16, "fcd_bn_act_fwd_replay: 1 .. 16 groups / order entries");
  unsigned long long code = 0ull;
  for (int i = 0; i < norder; ++i) {
    FCD_CHECK_ARG(order[i] >= 0 && order[i] < groups, "fcd_bn_act_fwd_replay: order[%d] = %d is not a group", i, order[i]);
    code |= (unsigned long long)order[i] << (4 * i);
  }
  return bn_act_fwd_impl(x, y, N, C, HW, groups, 1, gamma, beta, running_mean, running_var, momentum, eps, 1, save_mean,
                         save_invstd, act, slope, slope_imm, ws, ws_bytes, stream, code, norder);
}

__global__ void bn_train_prep_kernel(int C, int G, const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                     float* __restrict__ scale, float* __restrict__ shift) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * G) return;
  const int c = i % C;
  const float sc = gamma[c] * invstd[i];
  scale[i] = sc;
  shift[i] = beta[c] - mean[i] * sc;
}

// ---- backward ----------------------------------------------------------------
// reduce: per (g,c): s1 = sum dy', s2 = sum dy'*xhat, s3 = sum dz*y*[y<=0] (PReLU slope grad)
// where y = x*scale+shift (pre-activation), dy' = dz*act'(y), xhat = (x-mean)*invstd
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const float* __restrict__ dz, const float* __restrict__ x, double* __restrict__ part, int C, int HW, int Ng,
    int split, int affine, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ mean, const float* __restrict__ invstd, int act,
    const float* __restrict__ slope_ptr, float slope_imm, const float* __restrict__ gamma = nullptr,
    const float* __restrict__ beta = nullptr) {
  __shared__ double red[16];
  const int c = blockIdx.x, g = blockIdx.y, sp = blockIdx.z;
  const int sidx = g * C + c;
  const float mu = affine ? mean[sidx] : 0.f, is = affine ? invstd[sidx] : 1.f;
  float sc = 1.f, sh = 0.f;
  if (affine && gamma) {            // [r4] bn_train_prep_kernel's two lines, here instead of in a launch of their own
    sc = gamma[c] * is;
    sh = beta[c] - mu * sc;
  } else if (affine) {
    sc = scale[sidx];
    sh = shift[sidx];
  }
  const float slope = slope_ptr ? slope_ptr[0] : slope_imm;
  double s1 = 0.0, s2 = 0.0, s3 = 0.0;
  auto elem = [&](float xv, float dzv) {
    const float yv = fmaf(xv, sc, sh);
    const float dyv = dzv * act_grad(yv, act, slope);
    s1 += (double)dyv;
    s2 += (double)dyv * (double)((xv - mu) * is);
    if (act == FCD_ACT_PRELU && yv <= 0.f) s3 += (double)dzv * (double)yv;
  };
  if ((HW & 3) == 0 && ((((size_t)x | (size_t)dz)) & 15) == 0) {
    // [r4] 16-B accesses, two units of each stream requested before the first is consumed (the scalar loop had one 4-B load per
    // stream in flight per thread and a 64-bit division per element: 4.2 TB/s where a two-stream read reaches > 5.5)
    const int hw4 = HW >> 2;
    const long long total = (long long)Ng * hw4;
    const long long chunk = (total + split - 1) / split;
    const long long beg = sp * chunk, end = min(beg + chunk, total);
    for (long long e = beg + threadIdx.x; e < end; e += 512) {
      const long long e1 = e + 256;
      const bool two = e1 < end;
      const int n0 = (int)(e / hw4), i0 = (int)(e % hw4);
      const size_t o0 = ((size_t)(g * Ng + n0) * C + c) * HW + 4 * (size_t)i0;
      const float4 x0 = NTLD(x + o0), d0 = NTLD(dz + o0);
      float4 x1 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
      if (two) {
        const int n1 = (int)(e1 / hw4), i1 = (int)(e1 % hw4);
        const size_t o1 = ((size_t)(g * Ng + n1) * C + c) * HW + 4 * (size_t)i1;
        x1 = NTLD(x + o1);
        d1 = NTLD(dz + o1);
      }
      elem(x0.x, d0.x); elem(x0.y, d0.y); elem(x0.z, d0.z); elem(x0.w, d0.w);
      if (two) { elem(x1.x, d1.x); elem(x1.y, d1.y); elem(x1.z, d1.z); elem(x1.w, d1.w); }
    }
  } else {
    const long long total = (long long)Ng * HW;
    const long long chunk = (total + split - 1) / split;
    const long long beg = sp * chunk, end = min(beg + chunk, total);
    for (long long e = beg + threadIdx.x; e < end; e += 256) {
      const int n = (int)(e / HW), i = (int)(e % HW);
      const size_t off = ((size_t)(g * Ng + n) * C + c) * HW + i;
      elem(x[off], dz[off]);
    }
  }
  s1 = block_sum_d(s1, red);
  s2 = block_sum_d(s2, red);
  s3 = block_sum_d(s3, red);
  if (threadIdx.x == 0) {
    double* o = part + ((size_t)sidx * split + sp) * 3;
    o[0] = s1;
    o[1] = s2;
    o[2] = s3;
  }
}

__global__ void bn_bwd_finalize_kernel(const double* __restrict__ part, double* __restrict__ fin, int C, int G,
                                       int split, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double tg = 0.0, tb = 0.0;
  for (int g = 0; g < G; ++g) {
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const double* p = part + (size_t)(g * C + c) * split * 3;
    for (int s = 0; s < split; ++s) {
      s1 += p[s * 3];
      s2 += p[s * 3 + 1];
      s3 += p[s * 3 + 2];
    }
    double* f = fin + (size_t)(g * C + c) * 3;
    f[0] = s1;
    f[1] = s2;
    f[2] = s3;
    tb += s1;
    tg += s2;
  }
  if (dgamma) dgamma[c] = (float)tg;
  if (dbeta) dbeta[c] = (float)tb;
}

__global__ void slope_grad_kernel(const double* __restrict__ fin, int GC, float* __restrict__ dslope) {
  __shared__ double red[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < GC; i += blockDim.x) s += fin[(size_t)i * 3 + 2];
  s = block_sum_d(s, red);
  if (threadIdx.x == 0) dslope[0] = (float)s;
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const float* __restrict__ dz, const float* __restrict__ x, float* __restrict__ dx, int C, int HW, int Ng,
    int affine, int training, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ mean, const float* __restrict__ invstd, const double* __restrict__ fin,
    double inv_count, int act, const float* __restrict__ slope_ptr, float slope_imm,
    const double* __restrict__ part = nullptr, int split = 0, int G = 0, const float* __restrict__ gamma = nullptr,
    const float* __restrict__ beta = nullptr, float* __restrict__ dgamma = nullptr, float* __restrict__ dbeta = nullptr) {
  const int plane = blockIdx.x;
  const int n = plane / C, c = plane % C;
  const int sidx = (n / Ng) * C + c;
  const float slope = slope_ptr ? slope_ptr[0] : slope_imm;
  float sc = 1.f, sh = 0.f, mu = 0.f, is = 1.f, k1 = 0.f, k2 = 0.f;
  if (affine && training && part) {
    // [r4] the work of bn_train_prep_kernel and bn_bwd_finalize_kernel, per block instead of in two launches of a few
    // microseconds each: the block sums the (<= 64) partial sums of its channel in the order the finalize kernel did
    mu = mean[sidx];
    is = invstd[sidx];
    sc = gamma[c] * is;
    sh = beta[c] - mu * sc;
    double s1 = 0.0, s2 = 0.0;
    const double* p = part + (size_t)sidx * split * 3;
    for (int s = 0; s < split; ++s) {
      s1 += p[s * 3];
      s2 += p[s * 3 + 1];
    }
    k1 = (float)(s1 * inv_count);
    k2 = (float)(s2 * inv_count);
    if (n == 0 && blockIdx.y == 0 && threadIdx.x == 0 && (dgamma || dbeta)) {      // one block per channel: the parameter gradients
      double tg = 0.0, tb = 0.0;
      for (int g = 0; g < G; ++g) {
        double a1 = 0.0, a2 = 0.0;
        const double* q = part + (size_t)(g * C + c) * split * 3;
        for (int s = 0; s < split; ++s) {
          a1 += q[s * 3];
          a2 += q[s * 3 + 1];
        }
        tb += a1;
        tg += a2;
      }
      if (dgamma) dgamma[c] = (float)tg;
      if (dbeta) dbeta[c] = (float)tb;
    }
  } else {
    if (affine) {
      sc = scale[sidx];
      sh = shift[sidx];
    }
    if (affine && training) {
      mu = mean[sidx];
      is = invstd[sidx];
      k1 = (float)(fin[(size_t)sidx * 3 + 0] * inv_count);
      k2 = (float)(fin[(size_t)sidx * 3 + 1] * inv_count);
    }
  }
  const float* xp = x + (size_t)plane * HW;
  const float* dp = dz + (size_t)plane * HW;
  float* op = dx + (size_t)plane * HW;
  auto elem = [&](float xv, float dzv) -> float {
    const float yv = fmaf(xv, sc, sh);
    const float dyv = dzv * act_grad(yv, act, slope);
    if (!affine) return dyv;
    if (training) return sc * (dyv - k1 - (xv - mu) * is * k2);
    return dyv * sc;
  };
  if ((HW & 3) == 0 && ((((size_t)x | (size_t)dz | (size_t)dx)) & 15) == 0) {
    const int hw4 = HW >> 2;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < hw4; i += gridDim.y * 256) {
      const float4 xv = NTLD(xp + 4 * (size_t)i), dv = NTLD(dp + 4 * (size_t)i);
      float4 r;
      r.x = elem(xv.x, dv.x); r.y = elem(xv.y, dv.y); r.z = elem(xv.z, dv.z); r.w = elem(xv.w, dv.w);
      reinterpret_cast<float4*>(op)[i] = r;
    }
  } else {
    for (int i = blockIdx.y * 256 + threadIdx.x; i < HW; i += gridDim.y * 256) op[i] = elem(xp[i], dp[i]);
  }
}

extern "C" int fcd_bn_act_bwd(const float* dz, const float* x, float* dx, int N, int C, int HW, int groups,
                              int has_bn, const float* gamma, const float* beta, const float* running_mean,
                              const float* running_var, float eps, int training, const float* save_mean,
                              const float* save_invstd, int act, const float* slope, float slope_imm,
                              float* dgamma, float* dbeta, float* dslope, void* ws, size_t ws_bytes,
                              void* stream) {
  FCD_CHECK_ARG(dz && x && dx && N > 0 && C > 0 && HW > 0 && groups > 0 && N % groups == 0,
                "fcd_bn_act_bwd: bad geometry");
  if (ws == nullptr || ws_bytes < fcd_bn_act_ws_bytes(C, groups)) {
    fcd_set_error("fcd_bn_act_bwd: workspace too small");
    return FCD_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int Ng = N / groups;
  BnWs w = carve(ws, C, groups);
  FcdProfScope prof(FCD_K_NORM, st, 0.0, 4.0 * N * C * (double)HW * 5.0);
  const float* mean = save_mean;
  const float* invstd = save_invstd;
  if (has_bn) {
    FCD_CHECK_ARG(gamma && beta, "fcd_bn_act_bwd: BN needs gamma/beta");
    if (training) {
      FCD_CHECK_ARG(save_mean && save_invstd, "fcd_bn_act_bwd: training needs saved stats");
      // rebuild scale/shift from the saved statistics
      // (scale = gamma*invstd, shift = beta - mean*scale)
    } else {
      FCD_CHECK_ARG(running_mean && running_var, "fcd_bn_act_bwd: eval needs running stats");
    }
  }
  // [r4] training-mode BatchNorm without a PReLU slope gradient: two launches (reduce, apply) instead of four
  if (has_bn && training && !(act == FCD_ACT_PRELU && dslope)) {
    const int split = pick_split(C, groups, (long long)Ng * HW);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C, groups, split), dim3(256), 0, st, dz, x, w.part, C, HW, Ng, split, 1,
                       (const float*)nullptr, (const float*)nullptr, mean, invstd, act, slope, slope_imm, gamma, beta);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, plane_grid(N * C, HW), dim3(256), 0, st, dz, x, dx, C, HW, Ng, 1, 1,
                       (const float*)nullptr, (const float*)nullptr, mean, invstd, (const double*)nullptr,
                       1.0 / ((double)Ng * HW), act, slope, slope_imm, (const double*)w.part, split, groups, gamma, beta, dgamma,
                       dbeta);
    FCD_LAUNCH_CHECK("bn_act_bwd");
    return FCD_OK;
  }
  // scale/shift
  if (has_bn) {
    if (training) {
      // reuse eval-prep style kernel through a lambda-free path: small kernel below
      hipLaunchKernelGGL(bn_train_prep_kernel, dim3(cdiv(C * groups, 128)), dim3(128), 0, st, C, groups, gamma, beta,
                         save_mean, save_invstd, w.scale, w.shift);
    } else {
      hipLaunchKernelGGL(bn_eval_prep_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, C, groups, gamma, beta,
                         running_mean, running_var, eps, w.scale, w.shift, w.mean_e, w.invstd_e);
      mean = w.mean_e;
      invstd = w.invstd_e;
    }
  }
  const bool need_reduce = (has_bn && (training || dgamma || dbeta)) || (act == FCD_ACT_PRELU && dslope);
  if (need_reduce) {
    const int split = pick_split(C, groups, (long long)Ng * HW);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C, groups, split), dim3(256), 0, st, dz, x, w.part, C, HW, Ng,
                       split, has_bn, (const float*)w.scale, (const float*)w.shift, mean, invstd, act, slope,
                       slope_imm);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, (const double*)w.part, w.fin, C,
                       groups, split, has_bn ? dgamma : nullptr, has_bn ? dbeta : nullptr);
    if (act == FCD_ACT_PRELU && dslope)
      hipLaunchKernelGGL(slope_grad_kernel, dim3(1), dim3(256), 0, st, (const double*)w.fin, C * groups, dslope);
  }
  hipLaunchKernelGGL(bn_bwd_apply_kernel, plane_grid(N * C, HW), dim3(256), 0, st, dz, x, dx, C, HW, Ng, has_bn,
                     training, (const float*)w.scale, (const float*)w.shift, mean, invstd, (const double*)w.fin,
                     1.0 / ((double)Ng * HW), act, slope, slope_imm);
  FCD_LAUNCH_CHECK("bn_act_bwd");
  return FCD_OK;
}


// ---------------------------------------------------------------------------
// Synchronised BatchNorm (optional): the same kernels split at the point where the
// per-channel sums exist, so the host can all-reduce them over RCCL between the two halves.
//   fwd:  fcd_bn_partial_stats -> all-reduce(sum) -> fcd_bn_act_fwd_from_stats
//   bwd:  fcd_bn_bwd_partial   -> all-reduce(sum) -> fcd_bn_bwd_from_sums
__global__ void bn_sum_parts_kernel(const double* __restrict__ part, double* __restrict__ out, int GC, int split,
                                    int nvals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= GC) return;
  for (int v = 0; v < nvals; ++v) {
    double s = 0.0;
    for (int k = 0; k < split; ++k) s += part[((size_t)i * split + k) * 3 + v];
    out[(size_t)i * nvals + v] = s;
  }
}

__global__ void bn_spread_sums_kernel(const double* __restrict__ in, double* __restrict__ part3, int GC, int nvals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= GC) return;
  for (int v = 0; v < 3; ++v) part3[(size_t)i * 3 + v] = v < nvals ? in[(size_t)i * nvals + v] : 0.0;
}

extern "C" int fcd_bn_partial_stats(const float* x, double* out, int N, int C, int HW, int groups, void* ws,
                                    size_t ws_bytes, void* stream) {
  FCD_CHECK_ARG(x && out && N > 0 && C > 0 && HW > 0 && groups > 0 && N % groups == 0, "fcd_bn_partial_stats: bad geometry");
  if (!ws || ws_bytes < fcd_bn_act_ws_bytes(C, groups)) {
    fcd_set_error("fcd_bn_partial_stats: workspace too small");
    return FCD_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  BnWs w = carve(ws, C, groups);
  const int Ng = N / groups;
  const int split = pick_split(C, groups, (long long)Ng * HW);
  FcdProfScope prof(FCD_K_NORM, st, 0.0, 4.0 * N * C * (double)HW);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(C, groups, split), dim3(256), 0, st, x, w.part, C, HW, Ng, split);
  hipLaunchKernelGGL(bn_sum_parts_kernel, dim3(cdiv(C * groups, 128)), dim3(128), 0, st, (const double*)w.part, out,
                     C * groups, split, 2);
  FCD_LAUNCH_CHECK("bn_partial_stats");
  return FCD_OK;
}

extern "C" int fcd_bn_act_fwd_from_stats(const float* x, float* y, int N, int C, int HW, int groups,
                                         const double* sums, double count, const float* gamma, const float* beta,
                                         float* running_mean, float* running_var, float momentum, float eps,
                                         float* save_mean, float* save_invstd, int act, const float* slope,
                                         float slope_imm, void* ws, size_t ws_bytes, void* stream) {
  FCD_CHECK_ARG(x && y && sums && gamma && beta && save_mean && save_invstd && count > 0 && N % groups == 0,
                "fcd_bn_act_fwd_from_stats: bad arguments");
  if (!ws || ws_bytes < fcd_bn_act_ws_bytes(C, groups)) {
    fcd_set_error("fcd_bn_act_fwd_from_stats: workspace too small");
    return FCD_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  BnWs w = carve(ws, C, groups);
  FcdProfScope prof(FCD_K_NORM, st, 0.0, 8.0 * N * C * (double)HW);
  hipLaunchKernelGGL(bn_spread_sums_kernel, dim3(cdiv(C * groups, 128)), dim3(128), 0, st, sums, w.part, C * groups, 2);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, (const double*)w.part, C, groups, 1,
                     count, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_invstd, w.scale,
                     w.shift);
  hipLaunchKernelGGL(bn_act_apply_kernel, plane_grid(N * C, HW), dim3(256), 0, st, x, y, C, HW, N / groups, 1,
                     (const float*)w.scale, (const float*)w.shift, act, slope, slope_imm);
  FCD_LAUNCH_CHECK("bn_act_fwd_from_stats");
  return FCD_OK;
}

// [r3] train-mode y = act(BN(x)) from per-workgroup partial sums the PRODUCING convolution left behind
// (fcd_conv2d_fwd_wino_x: part[(g C + c) split + s][3], {sum, sum of squares, -}): the statistics pass over x is gone.
extern "C" int fcd_bn_act_fwd_parts(const float* x, float* y, int N, int C, int HW, int groups, const double* part, int split,
                                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                                    float momentum, float eps, float* save_mean, float* save_invstd, int act,
                                    const float* slope, float slope_imm, void* ws, size_t ws_bytes, void* stream) {
  FCD_CHECK_ARG(x && y && part && split > 0 && gamma && beta && save_mean && save_invstd && groups > 0 && N % groups == 0,
                "fcd_bn_act_fwd_parts: bad arguments");
  if (!ws || ws_bytes < fcd_bn_act_ws_bytes(C, groups)) {
    fcd_set_error("fcd_bn_act_fwd_parts: workspace too small");
    return FCD_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  BnWs w = carve(ws, C, groups);
  const int Ng = N / groups;
  FcdProfScope prof(FCD_K_NORM, st, 0.0, 8.0 * N * C * (double)HW);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, part, C, groups, split, (double)Ng * HW, gamma,
                     beta, running_mean, running_var, momentum, eps, save_mean, save_invstd, w.scale, w.shift);
  hipLaunchKernelGGL(bn_act_apply_kernel, plane_grid(N * C, HW), dim3(256), 0, st, x, y, C, HW, Ng, 1,
                     (const float*)w.scale, (const float*)w.shift, act, slope, slope_imm);
  FCD_LAUNCH_CHECK("bn_act_fwd_parts");
  return FCD_OK;
}

// [r5] train-mode statistics of a BatchNorm whose normalise + activate pass is done by its CONSUMER's loader (the F(4x4) input
// transform of the next convolution, fcd_wino_fwd_extras.in_scale / in_shift; reference Module.py:25-31): finalises mean / invstd, updates
// the running statistics and leaves scale = gamma * invstd, shift = beta - mean * scale per (group, channel) in caller-owned buffers --
// everything fcd_bn_act_fwd does except the pass over x.  `part` / `split`: the producing convolution's partial sums as in
// fcd_bn_act_fwd_parts, or NULL / 0 (the statistics kernel reads x).
extern "C" int fcd_bn_train_stats(const float* x, int N, int C, int HW, int groups, const double* part, int split, const float* gamma,
                                  const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                  float* save_mean, float* save_invstd, float* scale, float* shift, void* ws, size_t ws_bytes,
                                  void* stream) {
  FCD_CHECK_ARG((x || part) && N > 0 && C > 0 && HW > 0 && groups > 0 && N % groups == 0 && gamma && beta && save_mean && save_invstd &&
                    scale && shift && (!part || split > 0),
                "fcd_bn_train_stats: bad arguments");
  if (!ws || ws_bytes < fcd_bn_act_ws_bytes(C, groups)) {
    fcd_set_error("fcd_bn_train_stats: workspace too small");
    return FCD_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  BnWs w = carve(ws, C, groups);
  const int Ng = N / groups;
  FcdProfScope prof(FCD_K_NORM, st, 0.0, part ? 0.0 : 4.0 * N * C * (double)HW);
  if (!part) {
    split = pick_split(C, groups, (long long)Ng * HW);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(C, groups, split), dim3(256), 0, st, x, w.part, C, HW, Ng, split);
    part = w.part;
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, part, C, groups, split, (double)Ng * HW, gamma,
                     beta, running_mean, running_var, momentum, eps, save_mean, save_invstd, scale, shift);
  FCD_LAUNCH_CHECK("bn_train_stats");
  return FCD_OK;
}

extern "C" int fcd_bn_bwd_partial(const float* dz, const float* x, double* out, int N, int C, int HW, int groups,
                                  const float* gamma, const float* beta, const float* save_mean,
                                  const float* save_invstd, int act, const float* slope, float slope_imm, void* ws,
                                  size_t ws_bytes, void* stream) {
  FCD_CHECK_ARG(dz && x && out && gamma && beta && save_mean && save_invstd && N % groups == 0,
                "fcd_bn_bwd_partial: bad arguments");
  if (!ws || ws_bytes < fcd_bn_act_ws_bytes(C, groups)) {
    fcd_set_error("fcd_bn_bwd_partial: workspace too small");
    return FCD_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  BnWs w = carve(ws, C, groups);
  const int Ng = N / groups;
  const int split = pick_split(C, groups, (long long)Ng * HW);
  FcdProfScope prof(FCD_K_NORM, st, 0.0, 8.0 * N * C * (double)HW);
  hipLaunchKernelGGL(bn_train_prep_kernel, dim3(cdiv(C * groups, 128)), dim3(128), 0, st, C, groups, gamma, beta,
                     save_mean, save_invstd, w.scale, w.shift);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C, groups, split), dim3(256), 0, st, dz, x, w.part, C, HW, Ng, split, 1,
                     (const float*)w.scale, (const float*)w.shift, save_mean, save_invstd, act, slope, slope_imm);
  hipLaunchKernelGGL(bn_sum_parts_kernel, dim3(cdiv(C * groups, 128)), dim3(128), 0, st, (const double*)w.part, out,
                     C * groups, split, 3);
  FCD_LAUNCH_CHECK("bn_bwd_partial");
  return FCD_OK;
}

extern "C" int fcd_bn_bwd_from_sums(const float* dz, const float* x, float* dx, int N, int C, int HW, int groups,
                                    const double* sums, double count, const float* gamma, const float* beta,
                                    const float* save_mean, const float* save_invstd, int act, const float* slope,
                                    float slope_imm, void* ws, size_t ws_bytes, void* stream) {
  FCD_CHECK_ARG(dz && x && dx && sums && gamma && beta && save_mean && save_invstd && count > 0 && N % groups == 0,
                "fcd_bn_bwd_from_sums: bad arguments");
  if (!ws || ws_bytes < fcd_bn_act_ws_bytes(C, groups)) {
    fcd_set_error("fcd_bn_bwd_from_sums: workspace too small");
    return FCD_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  BnWs w = carve(ws, C, groups);
  FcdProfScope prof(FCD_K_NORM, st, 0.0, 12.0 * N * C * (double)HW);
  hipLaunchKernelGGL(bn_train_prep_kernel, dim3(cdiv(C * groups, 128)), dim3(128), 0, st, C, groups, gamma, beta,
                     save_mean, save_invstd, w.scale, w.shift);
  hipLaunchKernelGGL(bn_spread_sums_kernel, dim3(cdiv(C * groups, 128)), dim3(128), 0, st, sums, w.fin, C * groups, 3);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, plane_grid(N * C, HW), dim3(256), 0, st, dz, x, dx, C, HW, N / groups, 1, 1,
                     (const float*)w.scale, (const float*)w.shift, save_mean, save_invstd, (const double*)w.fin,
                     1.0 / count, act, slope, slope_imm);
  FCD_LAUNCH_CHECK("bn_bwd_from_sums");
  return FCD_OK;
}

// =====================================================================================================================
// [r5] Encoder tail: train-mode BatchNorm + ReLU whose result feeds MaxPool2d(2) AND a skip connection (reference Module.py:30-31 ->
// :43-44 and :116-132).  Unfused, the tail of an encoder level is: apply (read z, write a), max-pool (read a, write p); backward:
// max-pool routing + skip sum (read a, skip gradient; write dz), BatchNorm reduce (read dz, z), BatchNorm apply (read dz, z; write).
// a = relu(z * scale + shift) is one fma + compare away from z, and the pooling argmax is a function of a: the backward pass needs
// neither a nor dz as tensors.  Here:
//   forward   one kernel writes a (the skip tensor) and p from z                                  (the max-pool's read of a is gone)
//   reduce    reads z + skip gradient (+ the pooled gradient, a quarter): recomputes a, the argmax of each 2 x 2 window, dz = skip
//             gradient + routed pooled gradient, the ReLU gate, and sums dy', dy' xhat per channel in fp64
//   apply     the same reads, writes the BatchNorm input gradient
// 5 passes over the level's activation instead of 8 in the backward tail.  Values: a, p, argmax, dz per element are the unfused
// kernels' expressions (same fma, same NaN-propagating window order); the fp64 sums run in a different order.
// One thread = 2 rows x 4 columns = two pooling windows (H, W even, W % 4 == 0, 16-B aligned rows: host-checked).
typedef float bnp_f4 __attribute__((ext_vector_type(4)));
typedef float bnp_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int pool_arg(float v00, float v01, float v10, float v11, float* mx) {
  int arg = 0;
  float m = v00;
  if (v01 > m || v01 != v01) { m = v01; arg = 1; }      // maxpool2_fwd / _bwd_kernel's order (PyTorch: val > max || isnan(val))
  if (v10 > m || v10 != v10) { m = v10; arg = 2; }
  if (v11 > m || v11 != v11) { m = v11; arg = 3; }
  *mx = m;
  return arg;
}

// grid (planes, chunks): items of a plane = (H / 2) x (W / 4)
__global__ __launch_bounds__(256) void bn_relu_pool_fwd_kernel(const float* __restrict__ z, float* __restrict__ a, float* __restrict__ p,
                                                               int C, int H, int W, int Ng, const float* __restrict__ scale,
                                                               const float* __restrict__ shift) {
  const int plane = blockIdx.x;
  const int n = plane / C, c = plane % C;
  const float sc = scale[(n / Ng) * C + c], sh = shift[(n / Ng) * C + c];
  const int W4 = W >> 2, items = (H >> 1) * W4, Wp = W >> 1;
  const float* zp = z + (size_t)plane * H * W;
  float* ap = a + (size_t)plane * H * W;
  float* pp = p + (size_t)plane * (H >> 1) * Wp;
  for (int i = blockIdx.y * 256 + threadIdx.x; i < items; i += gridDim.y * 256) {
    const int r2 = i / W4, j = i % W4;
    const size_t o = (size_t)(2 * r2) * W + 4 * j;
    const bnp_f4 z0 = *(const bnp_f4*)(zp + o), z1 = *(const bnp_f4*)(zp + o + W);
    bnp_f4 a0, a1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float y0 = fmaf(z0[e], sc, sh), y1 = fmaf(z1[e], sc, sh);
      a0[e] = y0 > 0.f ? y0 : 0.f;
      a1[e] = y1 > 0.f ? y1 : 0.f;
    }
    *(bnp_f4*)(ap + o) = a0;
    *(bnp_f4*)(ap + o + W) = a1;
    bnp_f2 m;
    float t;
    pool_arg(a0[0], a0[1], a1[0], a1[1], &t); m[0] = t;
    pool_arg(a0[2], a0[3], a1[2], a1[3], &t); m[1] = t;
    *(bnp_f2*)(pp + (size_t)r2 * Wp + 2 * j) = m;
  }
}

// what one thread of the two backward kernels sees of its 2 x 4 patch: z, dy' = (skip gradient + routed pooled gradient) gated by
// the ReLU, per element
struct BnPoolPatch { bnp_f4 z0, z1, d0, d1; };
__device__ __forceinline__ BnPoolPatch bn_pool_patch(const float* __restrict__ zp, const float* __restrict__ sp,
                                                     const float* __restrict__ gp, int W, int Wp, int r2, int j, float sc, float sh) {
  BnPoolPatch q;
  const size_t o = (size_t)(2 * r2) * W + 4 * j;
  q.z0 = __builtin_nontemporal_load((const bnp_f4*)(zp + o));
  q.z1 = __builtin_nontemporal_load((const bnp_f4*)(zp + o + W));
  const bnp_f4 s0 = __builtin_nontemporal_load((const bnp_f4*)(sp + o)), s1 = __builtin_nontemporal_load((const bnp_f4*)(sp + o + W));
  const bnp_f2 g = *(const bnp_f2*)(gp + (size_t)r2 * Wp + 2 * j);
  bnp_f4 a0, a1;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float y0 = fmaf(q.z0[e], sc, sh), y1 = fmaf(q.z1[e], sc, sh);
    a0[e] = y0 > 0.f ? y0 : 0.f;
    a1[e] = y1 > 0.f ? y1 : 0.f;
  }
  float t;
  const int g0 = pool_arg(a0[0], a0[1], a1[0], a1[1], &t), g1 = pool_arg(a0[2], a0[3], a1[2], a1[3], &t);
  // dz = skip gradient + routed gradient (maxpool2_bwd_kernel with `add`), then the ReLU gate (bn_bwd_*: dz * [y > 0])
  q.d0[0] = (s0[0] + (g0 == 0 ? g[0] : 0.f)) * (a0[0] > 0.f ? 1.f : 0.f);
  q.d0[1] = (s0[1] + (g0 == 1 ? g[0] : 0.f)) * (a0[1] > 0.f ? 1.f : 0.f);
  q.d1[0] = (s1[0] + (g0 == 2 ? g[0] : 0.f)) * (a1[0] > 0.f ? 1.f : 0.f);
  q.d1[1] = (s1[1] + (g0 == 3 ? g[0] : 0.f)) * (a1[1] > 0.f ? 1.f : 0.f);
  q.d0[2] = (s0[2] + (g1 == 0 ? g[1] : 0.f)) * (a0[2] > 0.f ? 1.f : 0.f);
  q.d0[3] = (s0[3] + (g1 == 1 ? g[1] : 0.f)) * (a0[3] > 0.f ? 1.f : 0.f);
  q.d1[2] = (s1[2] + (g1 == 2 ? g[1] : 0.f)) * (a1[2] > 0.f ? 1.f : 0.f);
  q.d1[3] = (s1[3] + (g1 == 3 ? g[1] : 0.f)) * (a1[3] > 0.f ? 1.f : 0.f);
  return q;
}

// grid (C, G, split): part[((g C + c) split + sp) 3 + {0, 1}] = {sum dy', sum dy' xhat}
__global__ __launch_bounds__(256) void bn_pool_bwd_reduce_kernel(const float* __restrict__ z, const float* __restrict__ dskip,
                                                                 const float* __restrict__ dpool, double* __restrict__ part, int C,
                                                                 int H, int W, int Ng, int split, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd) {
  __shared__ double red[16];
  const int c = blockIdx.x, g = blockIdx.y, sp = blockIdx.z;
  const int sidx = g * C + c;
  const float mu = mean[sidx], is = invstd[sidx];
  const float sc = gamma[c] * is, sh = beta[c] - mu * sc;
  const int W4 = W >> 2, Wp = W >> 1, items = (H >> 1) * W4;
  const long long total = (long long)Ng * items;
  const long long chunk = (total + split - 1) / split;
  const long long beg = sp * chunk, end = min(beg + chunk, total);
  double s1 = 0.0, s2 = 0.0;
  for (long long e = beg + threadIdx.x; e < end; e += 256) {
    const int n = (int)(e / items), i = (int)(e % items);
    const size_t plane = (size_t)(g * Ng + n) * C + c;
    const BnPoolPatch q = bn_pool_patch(z + plane * H * W, dskip + plane * H * W, dpool + plane * (H >> 1) * Wp, W, Wp, i / W4, i % W4, sc, sh);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s1 += (double)q.d0[k];
      s2 += (double)q.d0[k] * (double)((q.z0[k] - mu) * is);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s1 += (double)q.d1[k];
      s2 += (double)q.d1[k] * (double)((q.z1[k] - mu) * is);
    }
  }
  s1 = block_sum_d(s1, red);
  s2 = block_sum_d(s2, red);
  if (threadIdx.x == 0) {
    double* o = part + ((size_t)sidx * split + sp) * 3;
    o[0] = s1;
    o[1] = s2;
    o[2] = 0.0;
  }
}

// grid (planes, chunks); the block sums its channel's partials itself (bn_bwd_apply_kernel's [r4] form) and block (n = 0, chunk 0) of
// a channel writes the parameter gradients
__global__ __launch_bounds__(256) void bn_pool_bwd_apply_kernel(const float* __restrict__ z, const float* __restrict__ dskip,
                                                                const float* __restrict__ dpool, float* __restrict__ dx, int C, int H,
                                                                int W, int Ng, int G, int split, const double* __restrict__ part,
                                                                double inv_count, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta) {
  const int plane = blockIdx.x;
  const int n = plane / C, c = plane % C;
  const int sidx = (n / Ng) * C + c;
  const float mu = mean[sidx], is = invstd[sidx];
  const float sc = gamma[c] * is, sh = beta[c] - mu * sc;
  double t1 = 0.0, t2 = 0.0;
  const double* p = part + (size_t)sidx * split * 3;
  for (int s = 0; s < split; ++s) {
    t1 += p[s * 3];
    t2 += p[s * 3 + 1];
  }
  const float k1 = (float)(t1 * inv_count), k2 = (float)(t2 * inv_count);
  if (n == 0 && blockIdx.y == 0 && threadIdx.x == 0 && (dgamma || dbeta)) {
    double tg = 0.0, tb = 0.0;
    for (int g = 0; g < G; ++g) {
      double a1 = 0.0, a2 = 0.0;
      const double* q = part + (size_t)(g * C + c) * split * 3;
      for (int s = 0; s < split; ++s) {
        a1 += q[s * 3];
        a2 += q[s * 3 + 1];
      }
      tb += a1;
      tg += a2;
    }
    if (dgamma) dgamma[c] = (float)tg;
    if (dbeta) dbeta[c] = (float)tb;
  }
  const int W4 = W >> 2, Wp = W >> 1, items = (H >> 1) * W4;
  const float* zp = z + (size_t)plane * H * W;
  const float* sp = dskip + (size_t)plane * H * W;
  const float* gp = dpool + (size_t)plane * (H >> 1) * Wp;
  float* op = dx + (size_t)plane * H * W;
  for (int i = blockIdx.y * 256 + threadIdx.x; i < items; i += gridDim.y * 256) {
    const int r2 = i / W4, j = i % W4;
    const BnPoolPatch q = bn_pool_patch(zp, sp, gp, W, Wp, r2, j, sc, sh);
    bnp_f4 o0, o1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      o0[k] = sc * (q.d0[k] - k1 - (q.z0[k] - mu) * is * k2);
      o1[k] = sc * (q.d1[k] - k1 - (q.z1[k] - mu) * is * k2);
    }
    const size_t o = (size_t)(2 * r2) * W + 4 * j;
    *(bnp_f4*)(op + o) = o0;
    *(bnp_f4*)(op + o + W) = o1;
  }
}

// 1: the shapes the three kernels take (even H and W, W % 4 == 0)
extern "C" int fcd_bn_relu_pool_plan(int N, int C, int H, int W, int groups) {
  return (fcd_sw(FCD_SW_BN_POOL) && N > 0 && C > 0 && groups > 0 && N % groups == 0 && H >= 2 && W >= 4 && (H & 1) == 0 && (W & 3) == 0 &&
          (long long)N * C < (1ll << 31)) ? 1 : 0;
}

// a = relu(z * scale + shift) (N, C, H, W) and p = maxpool2(a) (N, C, H / 2, W / 2) in one pass; scale / shift from fcd_bn_train_stats
extern "C" int fcd_bn_relu_pool_fwd(const float* z, float* a, float* p, int N, int C, int H, int W, int groups, const float* scale,
                                    const float* shift, void* stream) {
  FCD_CHECK_ARG(z && a && p && scale && shift && fcd_bn_relu_pool_plan(N, C, H, W, groups), "fcd_bn_relu_pool_fwd: bad arguments");
  FCD_CHECK_ARG(((((size_t)z | (size_t)a) & 15) == 0) && (((size_t)p & 7) == 0), "fcd_bn_relu_pool_fwd: 16-B aligned z / a, 8-B aligned p");
  hipStream_t st = (hipStream_t)stream;
  FcdProfScope prof(FCD_K_NORM, st, 0.0, 4.0 * N * C * (double)H * W * 2.25);
  hipLaunchKernelGGL(bn_relu_pool_fwd_kernel, plane_grid(N * C, H * W / 2), dim3(256), 0, st, z, a, p, C, H, W, N / groups, scale, shift);
  FCD_LAUNCH_CHECK("bn_relu_pool_fwd");
  return FCD_OK;
}

// backward of (a, p) = fcd_bn_relu_pool_fwd(BatchNorm_train statistics of z): dz from the gradient `dskip` of a (the skip consumer's;
// required) and `dpool` of p; dgamma / dbeta may be NULL.  Workspace fcd_bn_act_ws_bytes(C, groups).
extern "C" int fcd_bn_relu_pool_bwd(const float* z, const float* dskip, const float* dpool, float* dz, int N, int C, int H, int W,
                                    int groups, const float* gamma, const float* beta, const float* save_mean,
                                    const float* save_invstd, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream) {
  FCD_CHECK_ARG(z && dskip && dpool && dz && gamma && beta && save_mean && save_invstd && fcd_bn_relu_pool_plan(N, C, H, W, groups),
                "fcd_bn_relu_pool_bwd: bad arguments");
  FCD_CHECK_ARG(((((size_t)z | (size_t)dskip | (size_t)dz) & 15) == 0) && (((size_t)dpool & 7) == 0),
                "fcd_bn_relu_pool_bwd: 16-B aligned z / dskip / dz, 8-B aligned dpool");
  if (ws == nullptr || ws_bytes < fcd_bn_act_ws_bytes(C, groups)) {
    fcd_set_error("fcd_bn_relu_pool_bwd: workspace too small");
    return FCD_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int Ng = N / groups, HW = H * W;
  BnWs w = carve(ws, C, groups);
  FcdProfScope prof(FCD_K_NORM, st, 0.0, 4.0 * N * C * (double)HW * 5.5);
  const int split = pick_split(C, groups, (long long)Ng * HW);
  hipLaunchKernelGGL(bn_pool_bwd_reduce_kernel, dim3(C, groups, split), dim3(256), 0, st, z, dskip, dpool, w.part, C, H, W, Ng, split,
                     gamma, beta, save_mean, save_invstd);
  hipLaunchKernelGGL(bn_pool_bwd_apply_kernel, plane_grid(N * C, HW / 2), dim3(256), 0, st, z, dskip, dpool, dz, C, H, W, Ng, groups,
                     split, (const double*)w.part, 1.0 / ((double)Ng * HW), gamma, beta, save_mean, save_invstd, dgamma, dbeta);
  FCD_LAUNCH_CHECK("bn_relu_pool_bwd");
  return FCD_OK;
}
