// Loss-term kernels: masked per-sample reconstruction sums (Loss.py:76-84,
// :110-119, :127-141) and the fused SSIM level (ssim.py:26-92).
#include "common.h"

// ---- masked per-sample reduction ------------------------------------------------
// d = (a - b) * w,  w = complement ? (1 - m) : m,  m is (N,1,HW) broadcast over C
// num[n] = sum_{c,p} |d| (kind 0) or d^2 (kind 1);  wsum[n] = sum_p w
// grid (splits, N): partial sums in fp64 -> finalize
__global__ __launch_bounds__(256) void masked_recon_part_kernel(const float* __restrict__ a,
                                                                const float* __restrict__ b,
                                                                const float* __restrict__ m, double* __restrict__ part,
                                                                int C, int HW, int kind, int complement) {
  __shared__ double red[16];
  const int n = blockIdx.y, sp = blockIdx.x, nsp = gridDim.x;
  const float* mp = m + (size_t)n * HW;
  double s = 0.0, ws = 0.0;
  for (int p = sp * 256 + threadIdx.x; p < HW; p += nsp * 256) {
    float w = mp[p];
    if (complement) w = 1.f - w;
    ws += (double)w;
    float accp = 0.f;
    for (int c = 0; c < C; ++c) {
      const size_t off = ((size_t)n * C + c) * HW + p;
      const float d = (a[off] - (b ? b[off] : 0.f)) * w;
      accp += kind == 0 ? fabsf(d) : d * d;
    }
    s += (double)accp;
  }
  s = block_sum_d(s, red);
  ws = block_sum_d(ws, red);
  if (threadIdx.x == 0) {
    part[((size_t)n * nsp + sp) * 2] = s;
    part[((size_t)n * nsp + sp) * 2 + 1] = ws;
  }
}

__global__ void masked_recon_fin_kernel(const double* __restrict__ part, float* __restrict__ out2, int N, int nsp) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double s = 0.0, ws = 0.0;
  for (int i = 0; i < nsp; ++i) {
    s += part[((size_t)n * nsp + i) * 2];
    ws += part[((size_t)n * nsp + i) * 2 + 1];
  }
  out2[n] = (float)s;
  out2[N + n] = (float)ws;
}

#define MR_SPLITS 64

extern "C" size_t fcd_masked_recon_ws_bytes(int N) { return (size_t)N * MR_SPLITS * 2 * sizeof(double); }

extern "C" int fcd_masked_recon_fwd(const float* a, const float* b, const float* m, float* out2, int N, int C,
                                    int HW, int kind, int complement, void* ws, size_t ws_bytes, void* stream) {
  FCD_CHECK_ARG(a && m && out2 && N > 0 && C > 0 && HW > 0 && (kind == 0 || kind == 1),
                "fcd_masked_recon_fwd: bad arguments");
  if (!ws || ws_bytes < fcd_masked_recon_ws_bytes(N)) {
    fcd_set_error("fcd_masked_recon_fwd: workspace too small");
    return FCD_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  int nsp = std::min(MR_SPLITS, std::max(1, cdiv(HW, 256)));
  FcdProfScope prof(FCD_K_LOSS, st, 0.0, 4.0 * N * (double)HW * (2.0 * C + 1));
  hipLaunchKernelGGL(masked_recon_part_kernel, dim3(nsp, N), dim3(256), 0, st, a, b, m, (double*)ws, C, HW, kind,
                     complement);
  hipLaunchKernelGGL(masked_recon_fin_kernel, dim3(cdiv(N, 64)), dim3(64), 0, st, (const double*)ws, out2, N, nsp);
  FCD_LAUNCH_CHECK("masked_recon_fwd");
  return FCD_OK;
}

// mean_n( num[n] * scale / wsum[n] ) of the per-sample loops Loss.py:82-84,115-119,135-138 (skip_zero: samples with wsum == 0 are
// skipped -- the reference's `continue` -- but still counted in the mean), and its adjoint in the form fcd_masked_recon_bwd takes:
// coef[n] = dL/dnum[n], cw[n] = dL/dwsum[n].  One wave; replaces ~9 ATen launches forward and ~10 backward per loss term.
__global__ void ratio_mean_fwd_kernel(const float* __restrict__ out2, int N, float scale, int skip_zero, float* __restrict__ loss) {
  double s = 0.0;
  for (int n = threadIdx.x; n < N; n += 64) {
    const float num = out2[n], ws = out2[N + n];
    if (!skip_zero || ws != 0.f) s += (double)(num * scale / ws);
  }
  s = wave_sum_d(s);
  if (threadIdx.x == 0) loss[0] = (float)(s / (double)N);
}

__global__ void ratio_mean_bwd_kernel(const float* __restrict__ g, const float* __restrict__ out2, int N, float scale, int skip_zero,
                                      float* __restrict__ coef, float* __restrict__ cw) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float num = out2[n], ws = out2[N + n];
  const float gn = g[0] / (float)N;
  if (skip_zero && ws == 0.f) {
    coef[n] = 0.f;
    cw[n] = 0.f;
  } else {
    coef[n] = gn * scale / ws;
    cw[n] = -gn * (num * scale / ws) / ws;
  }
}

extern "C" int fcd_ratio_mean_fwd(const float* out2, int N, float scale, int skip_zero, float* loss, void* stream) {
  FCD_CHECK_ARG(out2 && loss && N > 0, "fcd_ratio_mean_fwd: bad arguments");
  hipLaunchKernelGGL(ratio_mean_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out2, N, scale, skip_zero, loss);
  FCD_LAUNCH_CHECK("ratio_mean_fwd");
  return FCD_OK;
}

extern "C" int fcd_ratio_mean_bwd(const float* g, const float* out2, int N, float scale, int skip_zero, float* coef, float* cw,
                                  void* stream) {
  FCD_CHECK_ARG(g && out2 && coef && cw && N > 0, "fcd_ratio_mean_bwd: bad arguments");
  hipLaunchKernelGGL(ratio_mean_bwd_kernel, dim3(cdiv(N, 64)), dim3(64), 0, (hipStream_t)stream, g, out2, N, scale, skip_zero, coef, cw);
  FCD_LAUNCH_CHECK("ratio_mean_bwd");
  return FCD_OK;
}

// L = sum_n coef[n]*num[n] + cw[n]*wsum[n]
// da = coef*f'(d)*w ; db = -da ; dm = +-( sum_c coef*f'(d)*(a-b) + cw )
__global__ __launch_bounds__(256) void masked_recon_bwd_kernel(const float* __restrict__ a,
                                                               const float* __restrict__ b,
                                                               const float* __restrict__ m,
                                                               const float* __restrict__ coef,
                                                               const float* __restrict__ cw, float* __restrict__ da,
                                                               float* __restrict__ db, float* __restrict__ dm, int C,
                                                               int HW, int kind, int complement) {
  const int n = blockIdx.y;
  const float cf = coef[n], cwn = cw[n];
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
    float w = m[(size_t)n * HW + p];
    if (complement) w = 1.f - w;
    float gw = 0.f;
    for (int c = 0; c < C; ++c) {
      const size_t off = ((size_t)n * C + c) * HW + p;
      const float diff = a[off] - (b ? b[off] : 0.f);
      const float d = diff * w;
      const float fp = kind == 0 ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 2.f * d;
      const float gd = cf * fp;
      if (da) da[off] = gd * w;
      if (db) db[off] = -gd * w;
      gw += gd * diff;
    }
    if (dm) {
      const float g = gw + cwn;
      dm[(size_t)n * HW + p] = complement ? -g : g;
    }
  }
}

extern "C" int fcd_masked_recon_bwd(const float* a, const float* b, const float* m, const float* coef,
                                    const float* cw, float* da, float* db, float* dm, int N, int C, int HW, int kind,
                                    int complement, void* stream) {
  FCD_CHECK_ARG(a && m && coef && cw && N > 0 && C > 0 && HW > 0, "fcd_masked_recon_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int nsp = std::min(256, std::max(1, cdiv(HW, 256)));
  FcdProfScope prof(FCD_K_LOSS, st, 0.0, 4.0 * N * (double)HW * (4.0 * C + 2));
  hipLaunchKernelGGL(masked_recon_bwd_kernel, dim3(nsp, N), dim3(256), 0, st, a, b, m, coef, cw, da, db, dm, C, HW,
                     kind, complement);
  FCD_LAUNCH_CHECK("masked_recon_bwd");
  return FCD_OK;
}
