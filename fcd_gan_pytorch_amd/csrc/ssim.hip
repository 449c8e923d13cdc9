// Fused SSIM level (ssim.py:26-92): one read of the X,Y tile into LDS, the five
// separable VALID Gaussian window statistics (mu1, mu2, E[x^2], E[y^2], E[xy])
// computed out of LDS (H pass then W pass, the reference's order), ssim/cs maps
// formed in registers and reduced per (n,c) -- 8 B/pixel of HBM traffic instead
// of the ~30 full-tensor passes of the op-by-op formulation.
// Backward: the same tile pipeline emits the four adjoint maps, a second kernel
// applies the transposed (full) window and the X/Y product rule.
#include "common.h"

#define ST_TY 16
#define ST_TX 32
#define ST_WMAX 11
#define ST_PH (ST_TY + ST_WMAX - 1)
#define ST_PW (ST_TX + ST_WMAX - 1)

struct SsimGeom {
  int H, W, OH, OW, ws, tiles_x, tiles_y;
  int wsh, wsw;   // taps applied along H / W: ws, or 1 (identity) when that dimension is shorter than the window -- gaussian_filter
                  // skips the smoothing along such a dimension (reference ssim.py:44-50)
  float C1, C2;
};

// MODE 0: reduce ssim/cs sums per block -> part[(nc*tiles + tile)*2 + {0,1}]
// MODE 1: write adjoint maps dmu1, dmu2, dB2, de12 (each [NC][OH][OW]) given per-(n,c) upstream grads
template <int MODE>
__global__ __launch_bounds__(256) void ssim_stats_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                         const float* __restrict__ win, SsimGeom g,
                                                         double* __restrict__ part, const float* __restrict__ g_ssim,
                                                         const float* __restrict__ g_cs, float* __restrict__ maps,
                                                         long long map_stride) {
  __shared__ float sx[ST_PH * ST_PW], sy[ST_PH * ST_PW];
  __shared__ float vb[5][ST_TY * ST_PW];
  __shared__ float wvh[ST_WMAX], wvw[ST_WMAX];
  __shared__ double red[16];
  const int tid = threadIdx.x;
  const int nc = blockIdx.z;
  const int oy0 = blockIdx.y * ST_TY, ox0 = blockIdx.x * ST_TX;
  const int wsh = g.wsh, wsw = g.wsw;
  const int ph = ST_TY + wsh - 1, pw = ST_TX + wsw - 1;
  if (tid < g.ws) {
    wvh[tid] = wsh == g.ws ? win[tid] : 1.f;      // (a skipped dimension uses the single tap 1)
    wvw[tid] = wsw == g.ws ? win[tid] : 1.f;
  }
  const float* xp = X + (size_t)nc * g.H * g.W;
  const float* yp = Y + (size_t)nc * g.H * g.W;
  for (int i = tid; i < ph * pw; i += 256) {
    const int r = i / pw, c = i % pw;
    const int iy = oy0 + r, ix = ox0 + c;
    float a = 0.f, b = 0.f;
    if (iy < g.H && ix < g.W) {
      a = xp[(size_t)iy * g.W + ix];
      b = yp[(size_t)iy * g.W + ix];
    }
    sx[r * ST_PW + c] = a;
    sy[r * ST_PW + c] = b;
  }
  __syncthreads();
  // vertical (H) pass
  for (int i = tid; i < ST_TY * pw; i += 256) {
    const int r = i / pw, c = i % pw;
    float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
    for (int k = 0; k < wsh; ++k) {
      const float w = wvh[k];
      const float a = sx[(r + k) * ST_PW + c], b = sy[(r + k) * ST_PW + c];
      m1 = fmaf(w, a, m1);
      m2 = fmaf(w, b, m2);
      e11 = fmaf(w, a * a, e11);
      e22 = fmaf(w, b * b, e22);
      e12 = fmaf(w, a * b, e12);
    }
    vb[0][r * ST_PW + c] = m1;
    vb[1][r * ST_PW + c] = m2;
    vb[2][r * ST_PW + c] = e11;
    vb[3][r * ST_PW + c] = e22;
    vb[4][r * ST_PW + c] = e12;
  }
  __syncthreads();
  double s_ssim = 0.0, s_cs = 0.0;
  float gs = 0.f, gc = 0.f;
  if (MODE == 1) {
    const float inv = 1.f / ((float)g.OH * (float)g.OW);
    gs = g_ssim[nc] * inv;
    gc = g_cs[nc] * inv;
  }
  for (int i = tid; i < ST_TY * ST_TX; i += 256) {
    const int r = i / ST_TX, c = i % ST_TX;
    const int oy = oy0 + r, ox = ox0 + c;
    if (oy >= g.OH || ox >= g.OW) continue;
    float q[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      float acc = 0.f;
      for (int k = 0; k < wsw; ++k) acc = fmaf(wvw[k], vb[j][r * ST_PW + c + k], acc);
      q[j] = acc;
    }
    const float mu1 = q[0], mu2 = q[1];
    const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
    const float s1 = q[2] - mu1s, s2 = q[3] - mu2s, s12 = q[4] - mu12;
    const float A1 = 2.f * mu12 + g.C1, B1 = mu1s + mu2s + g.C1;
    const float A2 = 2.f * s12 + g.C2, B2 = s1 + s2 + g.C2;
    const float cs = A2 / B2;
    const float l = A1 / B1;
    if (MODE == 0) {
      s_ssim += (double)(l * cs);
      s_cs += (double)cs;
    } else {
      const float G2 = gc + gs * l;   // dL/dcs
      const float Gl = gs * cs;       // dL/dl
      const float dA2 = G2 / B2, dB2 = -G2 * cs / B2;
      const float dA1 = Gl / B1, dB1 = -Gl * l / B1;
      const float de12 = 2.f * dA2;
      const float dmu1 = dA1 * 2.f * mu2 + dB1 * 2.f * mu1 - dB2 * 2.f * mu1 - de12 * mu2;
      const float dmu2 = dA1 * 2.f * mu1 + dB1 * 2.f * mu2 - dB2 * 2.f * mu2 - de12 * mu1;
      const size_t o = ((size_t)nc * g.OH + oy) * g.OW + ox;
      maps[o] = dmu1;
      maps[map_stride + o] = dmu2;
      maps[2 * map_stride + o] = dB2;
      maps[3 * map_stride + o] = de12;
    }
  }
  if (MODE == 0) {
    s_ssim = block_sum_d(s_ssim, red);
    s_cs = block_sum_d(s_cs, red);
    if (tid == 0) {
      const size_t t = (size_t)nc * g.tiles_x * g.tiles_y + blockIdx.y * g.tiles_x + blockIdx.x;
      part[t * 2] = s_ssim;
      part[t * 2 + 1] = s_cs;
    }
  }
}

__global__ void ssim_fin_kernel(const double* __restrict__ part, float* __restrict__ out, int NC, int tiles,
                                double inv_count) {
  const int nc = blockIdx.x * blockDim.x + threadIdx.x;
  if (nc >= NC) return;
  double a = 0.0, b = 0.0;
  for (int t = 0; t < tiles; ++t) {
    a += part[((size_t)nc * tiles + t) * 2];
    b += part[((size_t)nc * tiles + t) * 2 + 1];
  }
  out[nc] = (float)(a * inv_count);
  out[NC + nc] = (float)(b * inv_count);
}

// transposed window + product rule: one block per ST_TY x ST_TX tile of INPUT pixels
__global__ __launch_bounds__(256) void ssim_bwd_apply_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                             const float* __restrict__ win, SsimGeom g,
                                                             const float* __restrict__ maps, long long map_stride,
                                                             float* __restrict__ dX, float* __restrict__ dY) {
  __shared__ float m[4][ST_PH * ST_PW];
  __shared__ float vb[4][ST_TY * ST_PW];
  __shared__ float wvh[ST_WMAX], wvw[ST_WMAX];
  const int tid = threadIdx.x, nc = blockIdx.z;
  const int iy0 = blockIdx.y * ST_TY, ix0 = blockIdx.x * ST_TX;
  const int wsh = g.wsh, wsw = g.wsw, halo_h = wsh - 1, halo_w = wsw - 1;
  const int ph = ST_TY + halo_h, pw = ST_TX + halo_w;
  if (tid < g.ws) {                               // transposed taps
    wvh[tid] = wsh == g.ws ? win[g.ws - 1 - tid] : 1.f;
    wvw[tid] = wsw == g.ws ? win[g.ws - 1 - tid] : 1.f;
  }
  for (int i = tid; i < ph * pw; i += 256) {
    const int r = i / pw, c = i % pw;
    const int oy = iy0 - halo_h + r, ox = ix0 - halo_w + c;
    const bool ok = oy >= 0 && oy < g.OH && ox >= 0 && ox < g.OW;
    const size_t o = ok ? ((size_t)nc * g.OH + oy) * g.OW + ox : 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j][r * ST_PW + c] = ok ? maps[j * map_stride + o] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < ST_TY * pw; i += 256) {
    const int r = i / pw, c = i % pw;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float acc = 0.f;
      for (int k = 0; k < wsh; ++k) acc = fmaf(wvh[k], m[j][(r + k) * ST_PW + c], acc);
      vb[j][r * ST_PW + c] = acc;
    }
  }
  __syncthreads();
  for (int i = tid; i < ST_TY * ST_TX; i += 256) {
    const int r = i / ST_TX, c = i % ST_TX;
    const int iy = iy0 + r, ix = ix0 + c;
    if (iy >= g.H || ix >= g.W) continue;
    float t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float acc = 0.f;
      for (int k = 0; k < wsw; ++k) acc = fmaf(wvw[k], vb[j][r * ST_PW + c + k], acc);
      t[j] = acc;
    }
    const size_t o = ((size_t)nc * g.H + iy) * g.W + ix;
    const float x = X[o], y = Y[o];
    dX[o] = t[0] + 2.f * x * t[2] + y * t[3];
    dY[o] = t[1] + 2.f * y * t[2] + x * t[3];
  }
}

static int make_geom(int H, int W, int win_size, float C1, float C2, SsimGeom* g) {
  if (win_size < 1 || win_size > ST_WMAX || (win_size & 1) == 0) return -1;
  if (H < 1 || W < 1) return -1;
  g->H = H; g->W = W; g->ws = win_size;
  g->wsh = H >= win_size ? win_size : 1;      // gaussian_filter (ssim.py:44-50): no smoothing along a dimension shorter than the window
  g->wsw = W >= win_size ? win_size : 1;
  g->OH = H - g->wsh + 1; g->OW = W - g->wsw + 1;
  g->tiles_x = cdiv(g->OW, ST_TX); g->tiles_y = cdiv(g->OH, ST_TY);
  g->C1 = C1; g->C2 = C2;
  return 0;
}

extern "C" size_t fcd_ssim_ws_bytes(int NC, int H, int W) {
  // max(forward partials, backward adjoint maps)
  const size_t fwd = (size_t)NC * cdiv(W, ST_TX) * cdiv(H, ST_TY) * 2 * sizeof(double);
  const size_t bwd = (size_t)4 * NC * H * W * sizeof(float);
  return std::max(fwd, bwd);
}

extern "C" int fcd_ssim_level_fwd(const float* X, const float* Y, const float* win, int win_size, float* out, int NC,
                                  int H, int W, float C1, float C2, void* ws, size_t ws_bytes, void* stream) {
  FCD_CHECK_ARG(X && Y && win && out && NC > 0, "fcd_ssim_level_fwd: bad arguments");
  SsimGeom g;
  FCD_CHECK_ARG(make_geom(H, W, win_size, C1, C2, &g) == 0,
                "fcd_ssim_level_fwd: window %d unsupported for %dx%d (odd, <= 11)", win_size, H, W);
  const size_t need = (size_t)NC * g.tiles_x * g.tiles_y * 2 * sizeof(double);
  if (!ws || ws_bytes < need) {
    fcd_set_error("fcd_ssim_level_fwd: workspace too small");
    return FCD_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  FcdProfScope prof(FCD_K_LOSS, st, 0.0, 8.0 * NC * (double)H * W);
  hipLaunchKernelGGL(ssim_stats_kernel<0>, dim3(g.tiles_x, g.tiles_y, NC), dim3(256), 0, st, X, Y, win, g, (double*)ws,
                     (const float*)nullptr, (const float*)nullptr, (float*)nullptr, 0LL);
  hipLaunchKernelGGL(ssim_fin_kernel, dim3(cdiv(NC, 64)), dim3(64), 0, st, (const double*)ws, out, NC,
                     g.tiles_x * g.tiles_y, 1.0 / ((double)g.OH * g.OW));
  FCD_LAUNCH_CHECK("ssim_level_fwd");
  return FCD_OK;
}

extern "C" int fcd_ssim_level_bwd(const float* X, const float* Y, const float* win, int win_size, const float* g_ssim,
                                  const float* g_cs, float* dX, float* dY, int NC, int H, int W, float C1, float C2,
                                  void* ws, size_t ws_bytes, void* stream) {
  FCD_CHECK_ARG(X && Y && win && g_ssim && g_cs && dX && dY && NC > 0, "fcd_ssim_level_bwd: bad arguments");
  SsimGeom g;
  FCD_CHECK_ARG(make_geom(H, W, win_size, C1, C2, &g) == 0, "fcd_ssim_level_bwd: unsupported window/size");
  const long long map_stride = (long long)NC * g.OH * g.OW;
  if (!ws || ws_bytes < (size_t)4 * map_stride * sizeof(float)) {
    fcd_set_error("fcd_ssim_level_bwd: workspace too small");
    return FCD_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  FcdProfScope prof(FCD_K_LOSS, st, 0.0, 4.0 * NC * (double)H * W * 14.0);
  hipLaunchKernelGGL(ssim_stats_kernel<1>, dim3(g.tiles_x, g.tiles_y, NC), dim3(256), 0, st, X, Y, win, g,
                     (double*)nullptr, g_ssim, g_cs, (float*)ws, map_stride);
  hipLaunchKernelGGL(ssim_bwd_apply_kernel, dim3(cdiv(W, ST_TX), cdiv(H, ST_TY), NC), dim3(256), 0, st, X, Y, win, g,
                     (const float*)ws, map_stride, dX, dY);
  FCD_LAUNCH_CHECK("ssim_level_bwd");
  return FCD_OK;
}
