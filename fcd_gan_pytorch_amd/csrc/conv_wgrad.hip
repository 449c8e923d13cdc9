// Weight gradient of conv2d on the fp32 matrix cores.
//
//   dW[k][c][r][s] = sum_{n,p,q} dY[n,k,p,q] * X[n,c,p*stride+r-pad,q*stride+s-pad]
//
// GEMM view: M = K (64 per workgroup), N = C (64 per workgroup) x filter taps,
// reduction = output pixels.  A workgroup owns a 64x64x(RB*S taps) slab of dW
// and walks a range of pixel tiles (split-K over pixels across workgroups);
// each tile stages dY[64][TH*TW] and the input patch X[64][(TH-1)*st+RB][(TW-1)*st+S]
// in LDS once and feeds every tap's MFMA from the SAME staged patch (shifted
// reads), so X is read from HBM/L2 once per tile, not once per tap.  Wave
// layout 2x2: each wave accumulates 32(k) x 32(c) x taps in registers
// (v_mfma_f32_32x32x2_f32, two pixels per instruction).
// Partial slabs of different pixel splits are written to a workspace and summed
// by a second small kernel (deterministic; no atomics).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgradArgs {
  const float* x;
  const float* dy;
  float* out;
  int N, C, H, W, K, P, Q, pad;
  int tiles_p, tiles_q, total_tiles, tiles_per_split;
  int c_tiles;
  long long split_stride;
};

template <int R, int S, int RB, int STRIDE, int TH, int TW>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int PT = TH * TW;
  constexpr int PTP = PT + 1;
  constexpr int T = RB * S;
  constexpr int PH = (TH - 1) * STRIDE + RB;
  constexpr int PW = (TW - 1) * STRIDE + S;
  constexpr int PLANE = (PH * PW) | 1;
  static_assert(TW % 2 == 0, "pixel pairs must not straddle rows");
  __shared__ __attribute__((aligned(16))) float smem[64 * PTP + 64 * PLANE];
  float* dYs = smem;
  float* Xs = smem + 64 * PTP;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wc = wave & 1;
  const int ko0 = (blockIdx.y / a.c_tiles) * 64;
  const int c0 = (blockIdx.y % a.c_tiles) * 64;
  const int r0 = blockIdx.z * RB;
  const int split = blockIdx.x;

  f32x16 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const float* a_base = dYs + (wm * 32 + l31) * PTP + half;
  const float* b_base = Xs + (wc * 32 + l31) * PLANE + half * STRIDE;

  const int tile_beg = split * a.tiles_per_split;
  const int tile_end = min(tile_beg + a.tiles_per_split, a.total_tiles);
  for (int tile = tile_beg; tile < tile_end; ++tile) {
    int tt = tile;
    const int tq = tt % a.tiles_q;
    tt /= a.tiles_q;
    const int tp = tt % a.tiles_p;
    const int n = tt / a.tiles_p;
    const int p0 = tp * TH, q0 = tq * TW;
    __syncthreads();  // previous tile's reads done
    // stage dY tile
    for (int idx = tid; idx < 64 * PT; idx += 256) {
      const int ko = idx / PT, pix = idx % PT;
      const int p = p0 + pix / TW, q = q0 + pix % TW;
      float v = 0.f;
      if (ko0 + ko < a.K && p < a.P && q < a.Q) v = a.dy[(((size_t)n * a.K + ko0 + ko) * a.P + p) * a.Q + q];
      dYs[ko * PTP + pix] = v;
    }
    // stage input patch
    const int ih0 = p0 * STRIDE - a.pad + r0, iw0 = q0 * STRIDE - a.pad;
    for (int idx = tid; idx < 64 * PH * PW; idx += 256) {
      const int c = idx / (PH * PW), rem = idx % (PH * PW);
      const int ph = rem / PW, pw = rem % PW;
      const int ih = ih0 + ph, iw = iw0 + pw;
      float v = 0.f;
      if (c0 + c < a.C && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W)
        v = a.x[(((size_t)n * a.C + c0 + c) * a.H + ih) * a.W + iw];
      Xs[c * PLANE + ph * PW + pw] = v;
    }
    __syncthreads();
#pragma unroll
    for (int t2 = 0; t2 < PT / 2; ++t2) {
      constexpr int dummy = 0;
      (void)dummy;
      const int row = (2 * t2) / TW, col = (2 * t2) % TW;
      const float av = a_base[2 * t2];
#pragma unroll
      for (int rl = 0; rl < RB; ++rl)
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const float bv = b_base[(row * STRIDE + rl) * PW + col * STRIDE + s];
          acc[rl * S + s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[rl * S + s], 0, 0, 0);
        }
    }
  }

  float* out = a.out + (size_t)split * a.split_stride;
  const int c = c0 + wc * 32 + l31;
  if (c < a.C) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int r = r0 + t / S, s = t % S;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int ko = ko0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
        if (ko < a.K) out[(((size_t)ko * a.C + c) * R + r) * S + s] = acc[t][reg];
      }
    }
  }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, long long n,
                                    int splits) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += part[(long long)k * n + i];
    dw[i] = s;
  }
}

__global__ void channel_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int C,
                                   int HW) {
  __shared__ double red[16];
  const int c = blockIdx.x;
  double s = 0.0;
  for (int n = 0; n < N; ++n) {
    const float* p = x + ((size_t)n * C + c) * HW;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) s += (double)p[i];
  }
  s = block_sum_d(s, red);
  if (threadIdx.x == 0) out[c] = (float)s;
}

extern "C" int fcd_channel_sum(const float* x, float* out, int N, int C, int HW, void* stream) {
  FCD_CHECK_ARG(x && out && N > 0 && C > 0 && HW > 0, "fcd_channel_sum: bad arguments");
  FcdProfScope prof(FCD_K_MISC, (hipStream_t)stream, 0.0, 4.0 * N * C * (double)HW);
  hipLaunchKernelGGL(channel_sum_kernel, dim3(C), dim3(HW >= 1024 ? 512 : 128), 0, (hipStream_t)stream, x,
                     out, N, C, HW);
  FCD_LAUNCH_CHECK("channel_sum");
  return FCD_OK;
}

// ---------------------------------------------------------------------------
struct WgradPlan {
  int TH, TW, RB, tiles_p, tiles_q, total_tiles, tiles_per_split, splits, k_tiles, c_tiles, r_groups;
};

static bool wgrad_plan(const fcd_conv_desc* d, WgradPlan* pl) {
  const int R = d->R, S = d->S, st = d->stride;
  if (R == 3 && S == 3 && st == 1) { pl->TH = 2; pl->TW = 32; pl->RB = 3; }
  else if (R == 3 && S == 3 && st == 2) { pl->TH = 2; pl->TW = 16; pl->RB = 3; }
  else if (R == 9 && S == 9 && st == 1) { pl->TH = 2; pl->TW = 32; pl->RB = 1; }
  else if (R == 1 && S == 1 && st == 1) { pl->TH = 2; pl->TW = 32; pl->RB = 1; }
  else if (R == 2 && S == 2 && st == 2) { pl->TH = 2; pl->TW = 16; pl->RB = 2; }
  else return false;
  if (d->Q <= 16 && pl->TW == 32) { pl->TH = 4; pl->TW = 16; }
  pl->tiles_p = cdiv(d->P, pl->TH);
  pl->tiles_q = cdiv(d->Q, pl->TW);
  pl->total_tiles = d->N * pl->tiles_p * pl->tiles_q;
  pl->k_tiles = cdiv(d->K, 64);
  pl->c_tiles = cdiv(d->C, 64);
  pl->r_groups = R / pl->RB;
  const int base = pl->k_tiles * pl->c_tiles * pl->r_groups;
  int splits = cdiv(1024, base);
  const long long dw_bytes = 4LL * d->K * d->C * R * S;
  const long long cap = std::max<long long>(1, (512LL << 20) / dw_bytes);
  if (splits > cap) splits = (int)cap;
  if (splits > pl->total_tiles) splits = pl->total_tiles;
  if (splits < 1) splits = 1;
  pl->tiles_per_split = cdiv(pl->total_tiles, splits);
  pl->splits = cdiv(pl->total_tiles, pl->tiles_per_split);
  return true;
}

extern "C" size_t fcd_conv2d_bwd_weight_ws_bytes(const fcd_conv_desc* d) {
  WgradPlan pl;
  if (!d || !wgrad_plan(d, &pl)) return 0;
  if (pl.splits <= 1) return 0;
  return (size_t)pl.splits * d->K * d->C * d->R * d->S * sizeof(float);
}

template <int R, int S, int RB, int STRIDE, int TH, int TW>
static void launch_wgrad(const WgradArgs& a, const WgradPlan& pl, hipStream_t st) {
  dim3 grid((unsigned)pl.splits, (unsigned)(pl.k_tiles * pl.c_tiles), (unsigned)pl.r_groups);
  hipLaunchKernelGGL((conv_wgrad_kernel<R, S, RB, STRIDE, TH, TW>), grid, dim3(256), 0, st, a);
}

extern "C" int fcd_conv2d_bwd_weight(const fcd_conv_desc* d, const float* x, const float* dy, float* dw,
                                     void* ws, size_t ws_bytes, void* stream) {
  FCD_CHECK_ARG(d && x && dy && dw, "fcd_conv2d_bwd_weight: null pointer");
  WgradPlan pl;
  FCD_CHECK_ARG(wgrad_plan(d, &pl), "fcd_conv2d_bwd_weight: unsupported filter %dx%d stride %d", d->R, d->S,
                d->stride);
  const size_t need = fcd_conv2d_bwd_weight_ws_bytes(d);
  if (need > 0 && (ws == nullptr || ws_bytes < need)) {
    fcd_set_error("fcd_conv2d_bwd_weight: workspace %zu < %zu bytes", ws_bytes, need);
    return FCD_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  WgradArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.dy = dy;
  a.out = pl.splits > 1 ? (float*)ws : dw;
  a.N = d->N; a.C = d->C; a.H = d->H; a.W = d->W; a.K = d->K; a.P = d->P; a.Q = d->Q; a.pad = d->pad;
  a.tiles_p = pl.tiles_p; a.tiles_q = pl.tiles_q; a.total_tiles = pl.total_tiles;
  a.tiles_per_split = pl.tiles_per_split; a.c_tiles = pl.c_tiles;
  a.split_stride = (long long)d->K * d->C * d->R * d->S;
  const double flops = 2.0 * d->N * d->K * (double)d->P * d->Q * d->C * d->R * d->S;
  const double bytes = 4.0 * ((double)d->N * d->C * d->H * d->W + (double)d->N * d->K * d->P * d->Q +
                              (double)d->K * d->C * d->R * d->S);
  FcdProfScope prof(FCD_K_CONV_WGRAD, st, flops, bytes);
  const int R = d->R, S = d->S, sd = d->stride;
  const bool narrow = (pl.TW == 16 && pl.TH == 4);
  if (R == 3 && S == 3 && sd == 1) {
    if (narrow) launch_wgrad<3, 3, 3, 1, 4, 16>(a, pl, st); else launch_wgrad<3, 3, 3, 1, 2, 32>(a, pl, st);
  } else if (R == 3 && S == 3 && sd == 2) {
    launch_wgrad<3, 3, 3, 2, 2, 16>(a, pl, st);
  } else if (R == 9 && S == 9) {
    if (narrow) launch_wgrad<9, 9, 1, 1, 4, 16>(a, pl, st); else launch_wgrad<9, 9, 1, 1, 2, 32>(a, pl, st);
  } else if (R == 1 && S == 1) {
    if (narrow) launch_wgrad<1, 1, 1, 1, 4, 16>(a, pl, st); else launch_wgrad<1, 1, 1, 1, 2, 32>(a, pl, st);
  } else {
    launch_wgrad<2, 2, 2, 2, 2, 16>(a, pl, st);
  }
  FCD_LAUNCH_CHECK("conv2d_bwd_weight");
  if (pl.splits > 1) {
    const long long n = a.split_stride;
    const int grid = (int)std::min<long long>(cdiv64(n, 256), 2048);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid), dim3(256), 0, st, (const float*)ws, dw, n, pl.splits);
    FCD_LAUNCH_CHECK("wgrad_reduce");
  }
  return FCD_OK;
}
