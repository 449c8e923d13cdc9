// Implicit-GEMM fp32 convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
// GEMM view:  Y[k][pixel] = sum_{(c,r,s)} Wp[(c,r,s)][k] * Xpatch[c][pixel + (r,s)]
//   M = output channels (MFMA rows), N = output pixels of one image tile (MFMA
//   cols, contiguous in NCHW => coalesced stores), Kdim = C*R*S.
// Per workgroup (256 threads = 4 waves): BM x BN output tile; the K loop walks
// channel chunks of CB channels; each step stages the CB x (TH+R-1) x (TW+S-1)
// input patch and the matching CB*RCH*S x BM slab of pre-packed filter taps in
// LDS (global -> registers -> LDS, issued one step ahead so HBM/L2 latency hides
// under the MFMAs of the current step), then every wave reads its A/B operands
// with conflict-free ds_read_b32 (lanes 0-31 walk consecutive k / pixels, the
// two wave halves take the two k-slices of the 32x32x2 MFMA = two channels).
//
// The same kernel serves: forward (mode-0 packed weights), stride-1 data
// gradient (mode-1 packed = flipped/transposed taps, pad' = R-1-pad) and, with
// DIL=2, the data gradient of stride-2 convolutions (reads a zero-dilated dY).
#include "common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
int fcd_try_dgrad_thin(const fcd_conv_desc* d, const float* dy, const float* relu_out, const float* wp_bwd, float* dx,
                       hipStream_t st, int mask_is_bits);  // conv_thin.hip
int fcd_try_fwd_thin(const fcd_conv_desc* d, const float* x, const float* wp, const float* bias, float* y, int relu,
                     hipStream_t st, unsigned char* bits);    // conv_thin.hip

struct ConvArgs {
  const float* x;
  const float* wp;
  const float* bias;
  const float* mask;  // optional, same shape as x: x is read as x * (mask > 0)  (ReLU backward fused in)
  float* y;
  int relu;           // epilogue: y = max(y, 0)
  // generalised epilogue (inference fusions): slope activation y = y > 0 ? y : y*slope (PReLU /
  // LeakyReLU; slope read from device memory when slope_ptr != NULL) and a residual tensor of the
  // output's shape added AFTER the activation (ResidualBlock / block7 skip adds, Module.py:171,190)
  int act_slope;
  const float* slope_ptr;
  float slope_imm;
  const float* residual;
  // fused MaxPool2d(2) (v2 kernel only).  Forward: the epilogue writes the 2x2-pooled ReLU
  // output to pool_y and an argmax code byte (bits 0-1: window slot row*2+col, bit 2: max > 0)
  // to pool_code_out INSTEAD of y.  Data gradient: x is the POOLED gradient (N,C,H/2,W/2) and
  // pool_code_in its code bytes; element (h,w) reads x[h/2,w/2] iff its slot is the argmax of a
  // positive window (ReLU + max-pool backward folded into the patch loader).
  float* pool_y;
  unsigned char* pool_code_out;
  const unsigned char* pool_code_in;
  int Hp, Wp;         // pooled extents of the dgrad source
  int N, C, H, W;   // stored input tensor
  int K, Kpad;      // real / packed output channels
  int P, Q;         // output extent
  int pad;
  int nchunks;      // ceil(C / CB)
  int tiles_p, tiles_q;
  int k_tiles, xcd_remap;   // v2 kernel: 1-D grid with the XCD-aware (k-tile fastest) order
  // sub-pixel epilogue (register-staged kernel; data gradient of stride-2 convolutions): output row ko = cls * shuf_C + c of
  // the pseudo-convolution is pixel (2 p + (cls >> 1), 2 q + (cls & 1)) of channel c of a (N, shuf_C, shuf_H, shuf_W) tensor
  int shuf_C, shuf_H, shuf_W;
  unsigned long long* tbuf;   // IG_TIME builds
};

// IG_TIME: attribution build of the register-staged implicit-GEMM kernel (tools/igemm_segments.py): per-wave cycle sums {prologue, global
// loads of the next step issued, MFMA block, first barrier, LDS commit of the next step (waits for its loads), second barrier, epilogue,
// total} in a debug buffer.  Never defined in the product build.
#ifndef IG_TIME
#define IG_TIME 0
#endif
#if IG_TIME
#define IG_T(var) __builtin_amdgcn_sched_barrier(0); const unsigned long long var = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#define IG_TACC(slot, t1, t0) itacc[slot] += (t1) - (t0);
unsigned long long* g_ig_tbuf = nullptr;
extern "C" void fcd_igemm_time_buf(void* p) { g_ig_tbuf = (unsigned long long*)p; }
#else
#define IG_T(var)
#define IG_TACC(slot, t1, t0)
#endif

template <int R, int S, int RCH, int STRIDE, int DIL, int CB, int MI, int NI, int WM, int WN, int TH,
          int TW>
__global__ __launch_bounds__(256, 3) void conv_igemm_kernel(ConvArgs a) {
  constexpr int BM = 32 * MI * WM;
  constexpr int BN = 32 * NI * WN;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(TH * TW == BN, "pixel tile");
  static_assert(R % RCH == 0, "row chunks");
  constexpr int PH = (TH - 1) * STRIDE + R;
  constexpr int PW = (TW - 1) * STRIDE + S;
  constexpr int PWP = PW | 1;
  constexpr int PLANE = PH * PWP;
  constexpr int KC = CB * RCH * S;  // packed-weight rows per step
  constexpr int NR = R / RCH;       // steps per channel chunk
  constexpr int W_F4 = KC * BM / 4;
  constexpr int W_PER_T = (W_F4 + 255) / 256;
  constexpr int X_ELEMS = CB * PH * PW;
  constexpr int X_PER_T = (X_ELEMS + 255) / 256;

  __shared__ __attribute__((aligned(16))) float smem[KC * BM + CB * PLANE];
  float* Ws = smem;
  float* Xs = smem + KC * BM;

#if IG_TIME
  unsigned long long itacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  IG_T(i_begin)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave / WN, wn = wave % WN;

  int bx = blockIdx.x;
  const int tq = bx % a.tiles_q;
  bx /= a.tiles_q;
  const int tp = bx % a.tiles_p;
  const int n = bx / a.tiles_p;
  const int ko0 = blockIdx.y * BM;
  const int p0 = tp * TH, q0 = tq * TW;

  int xoff[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int pidx = wn * (32 * NI) + ni * 32 + l31;
    xoff[ni] = half * PLANE + (pidx / TW) * STRIDE * PWP + (pidx % TW) * STRIDE;
  }
  const int woff = half * (RCH * S) * BM + wm * (32 * MI) + l31;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // ---- staging registers and their (loop-invariant) addresses, computed once --------
  float4 wr[W_PER_T];
  float xr[X_PER_T], mr[X_PER_T];
  int w_goff[W_PER_T];      // packed-weight offset of this thread's i-th float4 within a step slab
  int x_goff[X_PER_T];      // input offset of this thread's i-th patch element within a channel chunk
  int x_loff[X_PER_T];      // ... and its LDS slot
  int x_cc[X_PER_T];        // ... and its channel within the chunk (-1: out of the image => always 0)
  const int ih0 = p0 * STRIDE - a.pad, iw0 = q0 * STRIDE - a.pad;
#pragma unroll
  for (int i = 0; i < W_PER_T; ++i) {
    const int idx = tid + i * 256;
    const int row = idx / (BM / 4), col4 = idx % (BM / 4);
    const int cc = row / (RCH * S), rem = row % (RCH * S);
    w_goff[i] = (cc * (R * S) + rem) * a.Kpad + ko0 + col4 * 4;
  }
#pragma unroll
  for (int i = 0; i < X_PER_T; ++i) {
    const int idx = tid + i * 256;
    const int cc = idx / (PH * PW), rem = idx % (PH * PW);
    const int ph = rem / PW, pw = rem % PW;
    int ih = ih0 + ph, iw = iw0 + pw;
    bool ok = idx < X_ELEMS && ih >= 0 && iw >= 0;
    if (DIL == 2) {
      ok = ok && !((ih | iw) & 1);
      ih >>= 1;
      iw >>= 1;
    }
    ok = ok && ih < a.H && iw < a.W;
    x_cc[i] = ok ? cc : -1;
    x_goff[i] = (cc * a.H + ih) * a.W + iw;
    x_loff[i] = cc * PLANE + ph * PWP + pw;
  }
  const float* xin = a.x + (size_t)n * a.C * a.H * a.W;
  const bool has_mask = a.mask != nullptr;
  const float* min_ = (has_mask ? a.mask : a.x) + (size_t)n * a.C * a.H * a.W;
  const int chunk_elems = CB * a.H * a.W;

#define FCD_LOAD_W(CCHUNK, RR)                                                                        \
  {                                                                                                   \
    const float* wsrc = a.wp + ((size_t)(CCHUNK) * CB * (R * S) + (RR) * (RCH * S)) * a.Kpad;         \
    _Pragma("unroll") for (int i = 0; i < W_PER_T; ++i) {                                             \
      if (W_F4 % 256 == 0 || tid + i * 256 < W_F4) wr[i] = *reinterpret_cast<const float4*>(wsrc + w_goff[i]); \
    }                                                                                                 \
  }
#define FCD_STORE_W()                                                                                 \
  {                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < W_PER_T; ++i) {                                             \
      if (W_F4 % 256 == 0 || tid + i * 256 < W_F4) *reinterpret_cast<float4*>(Ws + (tid + i * 256) * 4) = wr[i]; \
    }                                                                                                 \
  }
#define FCD_LOAD_X(CCHUNK)                                                                            \
  {                                                                                                   \
    const float* xsrc = xin + (size_t)(CCHUNK) * chunk_elems;                                      \
    const float* msrc = min_ + (size_t)(CCHUNK) * chunk_elems;                                        \
    const int cleft = a.C - (CCHUNK) * CB;                                                            \
    _Pragma("unroll") for (int i = 0; i < X_PER_T; ++i) {                                             \
      /* branch-free: out-of-range lanes read element 0 of the chunk and are zeroed by select */     \
      const bool ok = (unsigned)x_cc[i] < (unsigned)cleft;                                            \
      const unsigned off = ok ? (unsigned)x_goff[i] : 0u;                                             \
      const float v = xsrc[off];                                                                      \
      xr[i] = ok ? v : 0.f;                                                                           \
      if (has_mask) mr[i] = msrc[off];                                                                \
    }                                                                                                 \
  }
#define FCD_STORE_X()                                                                                 \
  {                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < X_PER_T; ++i) {                                             \
      if (X_ELEMS % 256 == 0 || tid + i * 256 < X_ELEMS) Xs[x_loff[i]] = (!has_mask || mr[i] > 0.f) ? xr[i] : 0.f; \
    }                                                                                                 \
  }

  // Sub-pixel data gradient of the stride-2 layers (fcd_conv2d_bwd_data_s2): output row block cls * shuf_C .. holds phase
  // (pi, pj) = (cls >> 1, cls & 1) of dx, whose 2 x 2 pseudo-filter has a tap (u, v) only if (pi || !u) && (pj || !v) -- 9 real
  // taps out of 16 over the four phases.  A wave skips the taps that are zero for every phase its 32 MI rows touch [r3].
  constexpr bool TAP_SKIP = (R == 2 && S == 2 && STRIDE == 1 && DIL == 1);
  unsigned tapmask = 0xFu;                   // bit u * 2 + v
  if (TAP_SKIP && a.shuf_C) {
    const int lo = ko0 + wm * (32 * MI), hi = min(lo + 32 * MI, a.K) - 1;
    unsigned m = 0;
    if (lo <= hi)
      for (int cls = lo / a.shuf_C; cls <= hi / a.shuf_C; ++cls)
        m |= 1u | ((cls & 1) ? 2u : 0u) | ((cls & 2) ? 4u : 0u) | ((cls & 3) == 3 ? 8u : 0u);
    tapmask = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
  }

  const int nsteps = a.nchunks * NR;
  FCD_LOAD_X(0)
  FCD_LOAD_W(0, 0)
  FCD_STORE_X()
  FCD_STORE_W()
  __syncthreads();
  IG_T(i_loop)
  IG_TACC(0, i_loop, i_begin)

  for (int step = 0; step < nsteps; ++step) {
    IG_T(is0)
    const int nxt = step + 1;
    const int rr = (NR == 1) ? 0 : step % NR;
    const bool have_next = nxt < nsteps;
    const bool next_patch = have_next && (NR == 1 || nxt % NR == 0);
    if (have_next) {
      if (next_patch) FCD_LOAD_X(nxt / NR)
      FCD_LOAD_W(nxt / NR, (NR == 1) ? 0 : nxt % NR)
    }
    IG_T(is1)

    const float* wl = Ws + woff;
    const float* xl = Xs + rr * RCH * PWP;
#pragma unroll
    for (int cc2 = 0; cc2 < CB / 2; ++cc2) {
#pragma unroll
      for (int rl = 0; rl < RCH; ++rl) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
          if (TAP_SKIP && !((tapmask >> (rl * S + s)) & 1u)) continue;
          float av[MI], bv[NI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) av[mi] = wl[((cc2 * 2) * (RCH * S) + rl * S + s) * BM + mi * 32];
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) bv[ni] = xl[xoff[ni] + (cc2 * 2) * PLANE + rl * PWP + s];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
        }
      }
    }
    IG_T(is2)
    __syncthreads();
    IG_T(is3)
    if (have_next) {
      if (next_patch) FCD_STORE_X()
      FCD_STORE_W()
    }
    IG_T(is4)
    __syncthreads();
    IG_T(is5)
    IG_TACC(1, is1, is0) IG_TACC(2, is2, is1) IG_TACC(3, is3, is2) IG_TACC(4, is4, is3) IG_TACC(5, is5, is4)
  }
  IG_T(i_epi)
#undef FCD_LOAD_W
#undef FCD_STORE_W
#undef FCD_LOAD_X
#undef FCD_STORE_X

  // epilogue: D layout col = lane&31 (pixel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (channel)
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int pidx = wn * (32 * NI) + ni * 32 + l31;
    const int p = p0 + pidx / TW, q = q0 + pidx % TW;
    if (p >= a.P || q >= a.Q) continue;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ko = ko0 + wm * (32 * MI) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (ko < a.K) {
          float v = acc[mi][ni][r];
          if (a.bias) v += a.bias[ko];
          if (a.relu) v = v > 0.f ? v : 0.f;
          if (a.act_slope) v = v > 0.f ? v : v * (a.slope_ptr ? a.slope_ptr[0] : a.slope_imm);
          if (a.shuf_C) {
            const int cls = ko / a.shuf_C, c = ko - cls * a.shuf_C;
            const int ph2 = 2 * p + (cls >> 1), qw2 = 2 * q + (cls & 1);
            if (ph2 < a.shuf_H && qw2 < a.shuf_W)
              a.y[(((size_t)n * a.shuf_C + c) * a.shuf_H + ph2) * a.shuf_W + qw2] = v;
            continue;
          }
          const size_t yo = (((size_t)n * a.K + ko) * a.P + p) * a.Q + q;
          if (a.residual) v += a.residual[yo];
          a.y[yo] = v;
        }
      }
    }
  }
#if IG_TIME
  {
    const unsigned long long i_end = __builtin_readcyclecounter();
    itacc[6] = i_end - i_epi; itacc[7] = i_end - i_begin;
    if (a.tbuf && lane == 0) {
      const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
#pragma unroll
      for (int i = 0; i < 8; ++i) a.tbuf[(wg * 4 + wave) * 8 + i] = itacc[i];
    }
  }
#endif
}

// ---------------------------------------------------------------------------
// <= 16 output channels (the Generator's last layer, 64 -> C bands through 9x9 taps: Module.py:165): the 32-row MFMA tile
// above does 13 useful rows out of 32.  Same staging, v_mfma_f32_16x16x4_f32 instead: a wave multiplies 16 rows x 64
// pixels (four 16-pixel column blocks) and FOUR input channels per instruction (lane group l >> 4 = channel), i.e. half
// the matrix-pipe cycles per (channel, pixel) of the 32x32x2 form.  Workgroup: 4 waves side by side on a TH x TW = 256
// pixel tile, CB = 8 channels per chunk, one filter-row chunk (RCH rows of taps) per step.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int R, int S, int RCH, int CB, int TH, int TW>
__global__ __launch_bounds__(256, 3) void conv_igemm_rows16_kernel(ConvArgs a) {
  constexpr int BM = 16, BN = 256;
  static_assert(TH * TW == BN && R % RCH == 0 && CB % 4 == 0, "tile");
  constexpr int PH = TH - 1 + R, PW = TW - 1 + S, PWP = PW | 1, PLANE = PH * PWP;
  constexpr int KC = CB * RCH * S, NR = R / RCH;
  constexpr int W_F4 = KC * BM / 4, W_PER_T = (W_F4 + 255) / 256;
  constexpr int X_ELEMS = CB * PH * PW, X_PER_T = (X_ELEMS + 255) / 256;
  __shared__ __attribute__((aligned(16))) float smem[KC * BM + CB * PLANE];
  float* Ws = smem;
  float* Xs = smem + KC * BM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kc = lane >> 4;
  int bx = blockIdx.x;
  const int tq = bx % a.tiles_q;
  bx /= a.tiles_q;
  const int tp = bx % a.tiles_p;
  const int n = bx / a.tiles_p;
  const int p0 = tp * TH, q0 = tq * TW;
  int xoff[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int pidx = wave * 64 + ni * 16 + l15;
    xoff[ni] = kc * PLANE + (pidx / TW) * PWP + (pidx % TW);
  }
  const int woff = kc * (RCH * S) * BM + l15;
  f32x4v acc[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) acc[ni] = f32x4v{0.f, 0.f, 0.f, 0.f};

  float4 wr[W_PER_T];
  float xr[X_PER_T];
  int w_goff[W_PER_T], x_goff[X_PER_T], x_loff[X_PER_T], x_cc[X_PER_T];
  const int ih0 = p0 - a.pad, iw0 = q0 - a.pad;
#pragma unroll
  for (int i = 0; i < W_PER_T; ++i) {
    const int idx = tid + i * 256;
    const int row = idx / (BM / 4), col4 = idx % (BM / 4);
    const int cc = row / (RCH * S), rem = row % (RCH * S);
    w_goff[i] = (cc * (R * S) + rem) * a.Kpad + col4 * 4;
  }
#pragma unroll
  for (int i = 0; i < X_PER_T; ++i) {
    const int idx = tid + i * 256;
    const int cc = idx / (PH * PW), rem = idx % (PH * PW);
    const int ph = rem / PW, pw = rem % PW;
    const int ih = ih0 + ph, iw = iw0 + pw;
    const bool ok = idx < X_ELEMS && ih >= 0 && iw >= 0 && ih < a.H && iw < a.W;
    x_cc[i] = ok ? cc : -1;
    x_goff[i] = (cc * a.H + ih) * a.W + iw;
    x_loff[i] = cc * PLANE + ph * PWP + pw;
  }
  const float* xin = a.x + (size_t)n * a.C * a.H * a.W;
  const int chunk_elems = CB * a.H * a.W;
#define R16_LOAD_W(CCHUNK, RR)                                                                        \
  {                                                                                                   \
    const float* wsrc = a.wp + ((size_t)(CCHUNK) * CB * (R * S) + (RR) * (RCH * S)) * a.Kpad;         \
    _Pragma("unroll") for (int i = 0; i < W_PER_T; ++i)                                               \
      if (W_F4 % 256 == 0 || tid + i * 256 < W_F4) wr[i] = *reinterpret_cast<const float4*>(wsrc + w_goff[i]); \
  }
#define R16_STORE_W()                                                                                 \
  {                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < W_PER_T; ++i)                                               \
      if (W_F4 % 256 == 0 || tid + i * 256 < W_F4) *reinterpret_cast<float4*>(Ws + (tid + i * 256) * 4) = wr[i]; \
  }
#define R16_LOAD_X(CCHUNK)                                                                            \
  {                                                                                                   \
    const float* xsrc = xin + (size_t)(CCHUNK) * chunk_elems;                                         \
    const int cleft = a.C - (CCHUNK) * CB;                                                            \
    _Pragma("unroll") for (int i = 0; i < X_PER_T; ++i) {                                             \
      const bool ok = (unsigned)x_cc[i] < (unsigned)cleft;                                            \
      const float v = xsrc[ok ? (unsigned)x_goff[i] : 0u];                                            \
      xr[i] = ok ? v : 0.f;                                                                           \
    }                                                                                                 \
  }
#define R16_STORE_X()                                                                                 \
  {                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < X_PER_T; ++i)                                               \
      if (X_ELEMS % 256 == 0 || tid + i * 256 < X_ELEMS) Xs[x_loff[i]] = xr[i];                       \
  }
  const int nsteps = a.nchunks * NR;
  R16_LOAD_X(0)
  R16_LOAD_W(0, 0)
  R16_STORE_X()
  R16_STORE_W()
  __syncthreads();
  for (int step = 0; step < nsteps; ++step) {
    const int nxt = step + 1;
    const int rr = (NR == 1) ? 0 : step % NR;
    const bool have_next = nxt < nsteps;
    const bool next_patch = have_next && (NR == 1 || nxt % NR == 0);
    if (have_next) {
      if (next_patch) R16_LOAD_X(nxt / NR)
      R16_LOAD_W(nxt / NR, (NR == 1) ? 0 : nxt % NR)
    }
    const float* wl = Ws + woff;
    const float* xl = Xs + rr * RCH * PWP;
#pragma unroll
    for (int cg = 0; cg < CB / 4; ++cg)
#pragma unroll
      for (int rl = 0; rl < RCH; ++rl)
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const float av = wl[((cg * 4) * (RCH * S) + rl * S + s) * BM];
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            acc[ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xl[xoff[ni] + (cg * 4) * PLANE + rl * PWP + s], acc[ni], 0, 0, 0);
        }
    __syncthreads();
    if (have_next) {
      if (next_patch) R16_STORE_X()
      R16_STORE_W()
    }
    __syncthreads();
  }
#undef R16_LOAD_W
#undef R16_STORE_W
#undef R16_LOAD_X
#undef R16_STORE_X
  // D layout: col = lane & 15 (pixel), row = 4 * (lane >> 4) + reg (channel)
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int pidx = wave * 64 + ni * 16 + l15;
    const int p = p0 + pidx / TW, q = q0 + pidx % TW;
    if (p >= a.P || q >= a.Q) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ko = 4 * kc + r;
      if (ko >= a.K) continue;
      float v = acc[ni][r];
      if (a.bias) v += a.bias[ko];
      if (a.relu) v = v > 0.f ? v : 0.f;
      if (a.act_slope) v = v > 0.f ? v : v * (a.slope_ptr ? a.slope_ptr[0] : a.slope_imm);
      const size_t yo = (((size_t)n * a.K + ko) * a.P + p) * a.Q + q;
      if (a.residual) v += a.residual[yo];
      a.y[yo] = v;
    }
  }
}

// ---------------------------------------------------------------------------
// v2: same math, different staging.  The packed filter slab of a step is a set of
// KC rows of BM contiguous floats, i.e. its LDS image [KC][BM] is lane-linear, so it is
// DMA'd straight into LDS with global_load_lds (16 B per lane, no VGPR staging, no
// ds_write pass).  Both LDS buffers (filter slab + input patch) are double-buffered:
// one s_barrier per step instead of two, and the registers freed by the filter staging
// buy a third resident workgroup per CU.
// channels per K-step of the v2 kernel (A/B on MI355X: 2 -> -2 %, 6 -> -5 % vs 4)
constexpr int FCD_CB2 = 4;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

// Filter-slab layout of the global_load_lds kernel ("T layout", written by pack_weights_t_kernel):
//   wp[q][h][m][FCD_KROW]   q = chunk of 4 input channels, h = lane half of the MFMA (channels 2h, 2h+1
//   of the chunk), m = GEMM row (output channel, padded to 128), and per row the 18 k-values
//   j = (c & 1) * 9 + tap this half consumes in one chunk, padded to 20 floats (16-B aligned rows;
//   stride 80 B => the b128 LDS reads of 8 consecutive lanes touch all 32 banks exactly once).
// One lane thus fetches its A operands for a whole chunk with 4 ds_read_b128 + 1 ds_read_b64 per
// m-tile instead of 18 ds_read_b32 with per-k address arithmetic.
#define FCD_KROW 20
#define FCD_KH 18
// SRC: how the loader treats the source tensor -- 0 plain, 1 gated by a ReLU mask (relu_out > 0),
// 2 pooled gradient routed by the argmax code byte.  Compile-time so the loop carries no mode branches.
template <int R, int S, int RCH, int STRIDE, int DIL, int CB, int MI, int NI, int WM, int WN, int TH,
          int TW, int SRC>
__global__ __launch_bounds__(256, (MI * NI > 4) ? 2 : 3) void conv_igemm_glds_kernel(ConvArgs a) {
  constexpr int BM = 32 * MI * WM;
  constexpr int BN = 32 * NI * WN;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(TH * TW == BN, "pixel tile");
  static_assert(R == 3 && S == 3 && RCH == 3 && CB == 4, "3x3 taps, 4-channel chunks");
  constexpr int PH = (TH - 1) * STRIDE + R;
  constexpr int PW = (TW - 1) * STRIDE + S;
  constexpr int PWP = PW | 1;
  constexpr int PLANE = PH * PWP;
  constexpr int WS_SZ = 2 * BM * FCD_KROW;         // floats per filter slab
  static_assert(WS_SZ % 256 == 0, "slab must be a whole number of wave loads");
  constexpr int W_INSTR = WS_SZ / 256;             // 1 KiB wave-instructions per slab
  constexpr int W_PER_WAVE = (W_INSTR + 3) / 4;
  constexpr int X_ELEMS = CB * PH * PW;
  constexpr int X_PER_T = (X_ELEMS + 255) / 256;
  constexpr int XS_SZ = CB * PLANE;

  // Distinct LDS objects for the filter slabs (written by global_load_lds = LDS DMA, counted by
  // vmcnt) and the input patches (ds_write), and a step loop unrolled by two so that the slab
  // being read and the slab being filled are different objects at compile time.
  __shared__ __attribute__((aligned(16))) float smem_w0[WS_SZ];
  __shared__ __attribute__((aligned(16))) float smem_w1[WS_SZ];
  __shared__ __attribute__((aligned(16))) float smem_x[2 * XS_SZ];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware tile order.  The dispatcher hands consecutive workgroup ids to the 8 XCDs round
  // robin (id % 8); each XCD has its own L2.  Remap so that every XCD walks a CONTIGUOUS range
  // of "virtual" ids, and let the k-tiles of one pixel tile be adjacent virtual ids: the blocks
  // that read the same input patch (and neighbouring patches sharing halo rows) then run on the
  // same XCD at about the same time and the patch is fetched from HBM once instead of once per
  // k-tile.  Placement only affects speed / traffic, never results.
  int ktile, bx;
  if (a.xcd_remap) {
    const unsigned total = gridDim.x, b = blockIdx.x;
    const unsigned q8 = total >> 3, r8 = total & 7u, xcd = b & 7u;
    const unsigned v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
    ktile = (int)(v % (unsigned)a.k_tiles);
    bx = (int)(v / (unsigned)a.k_tiles);
  } else {
    ktile = blockIdx.y;
    bx = blockIdx.x;
  }
  const int tq = bx % a.tiles_q;
  bx /= a.tiles_q;
  const int tp = bx % a.tiles_p;
  const int n = bx / a.tiles_p;
  const int ko0 = ktile * BM;
  const int p0 = tp * TH, q0 = tq * TW;

  // operand addresses inside the LDS buffers (floats)
  int xoff[NI], aoff[MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int pidx = wn * (32 * NI) + ni * 32 + l31;
    xoff[ni] = half * 2 * PLANE + (pidx / TW) * STRIDE * PWP + (pidx % TW) * STRIDE;
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) aoff[mi] = (half * BM + wm * (32 * MI) + mi * 32 + l31) * FCD_KROW;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // loop-invariant addresses.  Filter DMA: 16-B unit u of the slab image [h][m][KROW] comes from
  // row (h, ko0 + m) of the chunk's block in global memory.
  int w_goff[W_PER_WAVE];
#pragma unroll
  for (int j = 0; j < W_PER_WAVE; ++j) {
    const int u = (wave + 4 * j) * 64 + lane;
    const int h = u / (BM * (FCD_KROW / 4)), m = (u / (FCD_KROW / 4)) % BM, part = u % (FCD_KROW / 4);
    w_goff[j] = (h * a.Kpad + ko0 + m) * FCD_KROW + part * 4;
  }
  const size_t w_chunk_stride = (size_t)2 * a.Kpad * FCD_KROW;

  float xr[X_PER_T], mr[X_PER_T];
  unsigned mcode[X_PER_T], x_want[X_PER_T];   // raw code byte kept as an integer: no conversion => no early wait
  unsigned x_boff[X_PER_T];                    // byte offset inside a chunk of the source (0 for lanes that load nothing)
  int x_loff[X_PER_T], x_cc[X_PER_T];
  const int ih0 = p0 * STRIDE - a.pad, iw0 = q0 * STRIDE - a.pad;
#pragma unroll
  for (int i = 0; i < X_PER_T; ++i) {
    const int idx = tid + i * 256;
    const int cc = idx / (PH * PW), rem = idx % (PH * PW);
    const int ph = rem / PW, pw = rem % PW;
    int ih = ih0 + ph, iw = iw0 + pw;
    bool ok = idx < X_ELEMS && ih >= 0 && iw >= 0;
    if (DIL == 2) {
      ok = ok && !((ih | iw) & 1);
      ih >>= 1;
      iw >>= 1;
    }
    ok = ok && ih < a.H && iw < a.W;
    int goff = (cc * a.H + ih) * a.W + iw;
    x_loff[i] = cc * PLANE + ph * PWP + pw;
    if (SRC == 2) {   // gradient arrives pooled: index the (Hp, Wp) tensors, remember this element's slot
      const int hp = ih >> 1, wq = iw >> 1;
      if (hp >= a.Hp || wq >= a.Wp) ok = false;              // trailing odd row / column: never pooled
      goff = (cc * a.Hp + hp) * a.Wp + wq;
      x_want[i] = (unsigned)((((ih & 1) << 1) | (iw & 1)) | 4);
    }
    x_cc[i] = ok ? cc : -1;
    x_boff[i] = ok ? (unsigned)goff * 4u : 0u;
  }
  const int in_plane = (SRC == 2) ? a.Hp * a.Wp : a.H * a.W;
  const float* xin = a.x + (size_t)n * a.C * in_plane;
  const float* min_ = ((SRC == 1) ? a.mask : a.x) + (size_t)n * a.C * in_plane;
  const unsigned char* cin_ = (SRC == 2) ? a.pool_code_in + (size_t)n * a.C * in_plane : nullptr;
  const int chunk_elems = CB * in_plane;

#define FCD_GLDS_W(CCHUNK, WDST)                                                                      \
  {                                                                                                   \
    const float* wsrc = a.wp + (size_t)(CCHUNK) * w_chunk_stride;                                  \
    _Pragma("unroll") for (int j = 0; j < W_PER_WAVE; ++j) {                                          \
      if (W_INSTR % 4 == 0 || wave + 4 * j < W_INSTR)                                                 \
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(wsrc + w_goff[j]),                             \
                                         (lds_void_t*)((WDST) + (wave + 4 * j) * 256), 16, 0, 0);     \
    }                                                                                                 \
  }
  /* Loads only: any USE of a loaded value here would put a vmcnt(0) wait in front of the MFMA       \
     block.  Lanes that load nothing (halo outside the image, channel tail) read byte 0 of the      \
     chunk (of the tensor for the tail); they are zeroed at STORE time. */
#define FCD_LOAD_X2(CCHUNK)                                                                           \
  {                                                                                                   \
    const int cleft = a.C - (CCHUNK) * CB;                                                            \
    const bool tail = cleft < CB;                                                                     \
    const char* xsrc = (const char*)(xin + (size_t)(CCHUNK) * chunk_elems);                           \
    const char* msrc = (const char*)(min_ + (size_t)(CCHUNK) * chunk_elems);                          \
    const unsigned char* csrc = cin_ + (size_t)(CCHUNK) * chunk_elems;                                \
    _Pragma("unroll") for (int i = 0; i < X_PER_T; ++i) {                                             \
      unsigned off = x_boff[i];                                                                       \
      if (tail) off = (x_cc[i] < cleft) ? off : 0u;                                                   \
      xr[i] = *(const float*)(xsrc + off);                                                            \
      if (SRC == 1) mr[i] = *(const float*)(msrc + off);                                              \
      if (SRC == 2) mcode[i] = csrc[off >> 2];                                                        \
    }                                                                                                 \
  }
  /* bounds / ReLU-mask / pool-slot selects are applied HERE (after the MFMA block) */
#define FCD_STORE_X2(BUF, CCHUNK)                                                                     \
  {                                                                                                   \
    const int cleft = a.C - (CCHUNK) * CB;                                                            \
    _Pragma("unroll") for (int i = 0; i < X_PER_T; ++i) {                                             \
      if (X_ELEMS % 256 == 0 || tid + i * 256 < X_ELEMS) {                                            \
        bool keep = (unsigned)x_cc[i] < (unsigned)cleft;                                              \
        if (SRC == 2) keep = keep && mcode[i] == x_want[i];                                           \
        if (SRC == 1) keep = keep && mr[i] > 0.f;                                                     \
        smem_x[(BUF) * XS_SZ + x_loff[i]] = keep ? xr[i] : 0.f;                                       \
      }                                                                                               \
    }                                                                                                 \
  }

  const int nsteps = a.nchunks;
  FCD_GLDS_W(0, smem_w0)
  FCD_LOAD_X2(0)
  FCD_STORE_X2(0, 0)
  __syncthreads();

#define FCD_STEP(STEP, WCUR, WNXT)                                                                    \
  {                                                                                                   \
    const int chunk = (STEP);                                                                         \
    const bool have_next = chunk + 1 < nsteps;                                                        \
    const int xb = chunk & 1;                                                                         \
    if (have_next) {                                                                                  \
      FCD_GLDS_W(chunk + 1, WNXT)                                                                     \
      FCD_LOAD_X2(chunk + 1)                                                                          \
    }                                                                                                 \
    const float* xl = smem_x + xb * XS_SZ;                                                            \
    float av[MI][FCD_KROW];                                                                           \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) {                                               \
      const float* ap = (WCUR) + aoff[mi];                                                            \
      _Pragma("unroll") for (int v4 = 0; v4 < 4; ++v4) {                                              \
        const f32x4 t4 = *(const f32x4*)(ap + 4 * v4);                                                \
        av[mi][4 * v4] = t4[0]; av[mi][4 * v4 + 1] = t4[1];                                           \
        av[mi][4 * v4 + 2] = t4[2]; av[mi][4 * v4 + 3] = t4[3];                                       \
      }                                                                                               \
      const f32x2 t2 = *(const f32x2*)(ap + 16);                                                      \
      av[mi][16] = t2[0]; av[mi][17] = t2[1];                                                         \
    }                                                                                                 \
    _Pragma("unroll") for (int j = 0; j < FCD_KH; ++j) {                                              \
      float bv[NI];                                                                                   \
      _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                               \
        bv[ni] = xl[xoff[ni] + (j / 9) * PLANE + ((j % 9) / 3) * PWP + (j % 3)];                      \
      _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                               \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                             \
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][j], bv[ni], acc[mi][ni], 0, 0, 0); \
    }                                                                                                 \
    if (have_next) FCD_STORE_X2(xb ^ 1, chunk + 1)                                                    \
    __syncthreads();                                                                                  \
  }

  for (int step = 0; step < nsteps; step += 2) {
    FCD_STEP(step, smem_w0, smem_w1)
    if (step + 1 < nsteps) FCD_STEP(step + 1, smem_w1, smem_w0)
  }
#undef FCD_STEP
#undef FCD_GLDS_W
#undef FCD_LOAD_X2
#undef FCD_STORE_X2

  if (a.pool_y != nullptr) {
    // ---- fused bias + ReLU + MaxPool2d(2) epilogue.  D layout: col = l31 (pixel), NI = 2.
    //  TW == 32: the wave's two n-tiles are image rows 2wn, 2wn+1 -> vertical partner = other
    //            n-tile (same lane), horizontal partner = lane ^ 1
    //  TW == 16: one n-tile holds rows (2t, 2t+1) in lanes (0-15, 16-31) -> vertical partner =
    //            lane ^ 16, horizontal partner = lane ^ 1
    const int Pp = a.P >> 1, Qp = a.Q >> 1;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ko = ko0 + wm * (32 * MI) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const float bv = (a.bias && ko < a.K) ? a.bias[ko] : 0.f;
        float v[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          v[ni] = acc[mi][ni][r] + bv;
          v[ni] = v[ni] > 0.f ? v[ni] : 0.f;
        }
        if (TW >= 32) {
          // the wave's n-tiles are consecutive image rows: pool rows (2t, 2t+1) = n-tiles (2t, 2t+1)
#pragma unroll
          for (int np = 0; np + 1 < NI; np += 2) {
            const float v01 = __shfl_xor(v[np], 1, 64), v11 = __shfl_xor(v[np + 1], 1, 64);
            float m = v[np];
            int arg = 0;
            if (v01 > m) { m = v01; arg = 1; }
            if (v[np + 1] > m) { m = v[np + 1]; arg = 2; }
            if (v11 > m) { m = v11; arg = 3; }
            const int pidx = wn * (32 * NI) + np * 32 + l31;
            const int pp = (p0 + (pidx / TW)) >> 1, qq = (q0 + (pidx % TW)) >> 1;
            if (!(l31 & 1) && ko < a.K && pp < Pp && qq < Qp) {
              const size_t o = (((size_t)n * a.K + ko) * Pp + pp) * Qp + qq;
              a.pool_y[o] = m;
              a.pool_code_out[o] = (unsigned char)(arg | (m > 0.f ? 4 : 0));
            }
          }
        } else {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const float v01 = __shfl_xor(v[ni], 1, 64), v10 = __shfl_xor(v[ni], 16, 64),
                        v11 = __shfl_xor(v[ni], 17, 64);
            float m = v[ni];
            int arg = 0;
            if (v01 > m) { m = v01; arg = 1; }
            if (v10 > m) { m = v10; arg = 2; }
            if (v11 > m) { m = v11; arg = 3; }
            const int pidx = wn * (32 * NI) + ni * 32 + l31;
            const int pp = (p0 + (pidx / TW)) >> 1, qq = (q0 + (pidx % TW)) >> 1;
            if (!(l31 & 17) && ko < a.K && pp < Pp && qq < Qp) {
              const size_t o = (((size_t)n * a.K + ko) * Pp + pp) * Qp + qq;
              a.pool_y[o] = m;
              a.pool_code_out[o] = (unsigned char)(arg | (m > 0.f ? 4 : 0));
            }
          }
        }
      }
    }
    return;
  }

#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int pidx = wn * (32 * NI) + ni * 32 + l31;
    const int p = p0 + pidx / TW, q = q0 + pidx % TW;
    if (p >= a.P || q >= a.Q) continue;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ko = ko0 + wm * (32 * MI) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (ko < a.K) {
          float v = acc[mi][ni][r];
          if (a.bias) v += a.bias[ko];
          if (a.relu) v = v > 0.f ? v : 0.f;
          if (a.act_slope) v = v > 0.f ? v : v * (a.slope_ptr ? a.slope_ptr[0] : a.slope_imm);
          const size_t yo = (((size_t)n * a.K + ko) * a.P + p) * a.Q + q;
          if (a.residual) v += a.residual[yo];
          a.y[yo] = v;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------
// [r5] Sub-pixel data gradient of the stride-2 layers (see pack_weights_s2_kernel below for the algebra) in the global_load_lds
// form: the register-staged kernel above spent 45 - 55 % of a wave's life staging (profiles/r04_igemm_segments.md) -- a step of 8
// channels is only 16 ... 64 MFMAs per wave behind two barriers and a filter slab that goes global -> VGPR -> LDS.  Here the
// slab [h][m][20] (per lane half h the 16 k-values (4 channels x 2 x 2 pseudo-taps) of GEMM row m, 80-B rows as in the 3x3
// kernel) arrives by LDS-DMA into one of two buffers, the (TH + 1) x (TW + 1) patch of 8 dy channels is double-buffered too:
// one barrier per step, three workgroups per CU.  Rows stay class-major (row = phase * C + c), so a wave whose 64 rows lie
// in one phase skips that phase's zero taps as before (9 real taps of 16 over the four phases).
#define FCD_S2_KROW 20
#define FCD_S2_KH 16
template <int MI, int NI, int WM, int WN, int TH, int TW>
__global__ __launch_bounds__(256, 3) void conv_s2sub_glds_kernel(ConvArgs a) {
  constexpr int CB = 8;
  constexpr int BM = 32 * MI * WM;
  constexpr int BN = 32 * NI * WN;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(TH * TW == BN, "pixel tile");
  constexpr int PH = TH + 1, PW = TW + 1;
  constexpr int PWP = PW | 1;
  constexpr int PLANE = PH * PWP;
  constexpr int WS_SZ = 2 * BM * FCD_S2_KROW;
  static_assert(WS_SZ % 256 == 0, "slab must be a whole number of wave loads");
  constexpr int W_INSTR = WS_SZ / 256;
  constexpr int W_PER_WAVE = (W_INSTR + 3) / 4;
  constexpr int X_ELEMS = CB * PH * PW;
  constexpr int X_PER_T = (X_ELEMS + 255) / 256;
  constexpr int XS_SZ = CB * PLANE;
  __shared__ __attribute__((aligned(16))) float smem_w0[WS_SZ];
  __shared__ __attribute__((aligned(16))) float smem_w1[WS_SZ];
  __shared__ __attribute__((aligned(16))) float smem_x[2 * XS_SZ];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave / WN, wn = wave % WN;
  int ktile, bx;
  if (a.xcd_remap) {
    const unsigned total = gridDim.x, b = blockIdx.x;
    const unsigned q8 = total >> 3, r8 = total & 7u, xcd = b & 7u;
    const unsigned v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
    ktile = (int)(v % (unsigned)a.k_tiles);
    bx = (int)(v / (unsigned)a.k_tiles);
  } else {
    ktile = blockIdx.y;
    bx = blockIdx.x;
  }
  const int tq = bx % a.tiles_q;
  bx /= a.tiles_q;
  const int tp = bx % a.tiles_p;
  const int n = bx / a.tiles_p;
  const int ko0 = ktile * BM;
  const int p0 = tp * TH, q0 = tq * TW;

  int xoff[NI], aoff[MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int pidx = wn * (32 * NI) + ni * 32 + l31;
    xoff[ni] = half * (CB / 2) * PLANE + (pidx / TW) * PWP + (pidx % TW);
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) aoff[mi] = (half * BM + wm * (32 * MI) + mi * 32 + l31) * FCD_S2_KROW;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  int w_goff[W_PER_WAVE];
#pragma unroll
  for (int j = 0; j < W_PER_WAVE; ++j) {
    const int u = (wave + 4 * j) * 64 + lane;
    const int h = u / (BM * (FCD_S2_KROW / 4)), m = (u / (FCD_S2_KROW / 4)) % BM, part = u % (FCD_S2_KROW / 4);
    w_goff[j] = (h * a.Kpad + ko0 + m) * FCD_S2_KROW + part * 4;
  }
  const size_t w_chunk_stride = (size_t)2 * a.Kpad * FCD_S2_KROW;

  float xr[X_PER_T];
  unsigned x_boff[X_PER_T];
  int x_loff[X_PER_T], x_cc[X_PER_T];
#pragma unroll
  for (int i = 0; i < X_PER_T; ++i) {
    const int idx = tid + i * 256;
    const int cc = idx / (PH * PW), rem = idx % (PH * PW);
    const int ph = rem / PW, pw = rem % PW;
    const int ih = p0 + ph, iw = q0 + pw;
    const bool ok = idx < X_ELEMS && ih < a.H && iw < a.W;
    x_loff[i] = cc * PLANE + ph * PWP + pw;
    x_cc[i] = ok ? cc : -1;
    x_boff[i] = ok ? (unsigned)((cc * a.H + ih) * a.W + iw) * 4u : 0u;
  }
  const float* xin = a.x + (size_t)n * a.C * a.H * a.W;
  const int chunk_elems = CB * a.H * a.W;

  // taps (u, v) that are non-zero for some phase among this wave's rows (bit u * 2 + v)
  unsigned tapmask;
  {
    const int lo = ko0 + wm * (32 * MI), hi = min(lo + 32 * MI, a.K) - 1;
    unsigned m = 0;
    if (lo <= hi)
      for (int cls = lo / a.shuf_C; cls <= hi / a.shuf_C; ++cls)
        m |= 1u | ((cls & 1) ? 2u : 0u) | ((cls & 2) ? 4u : 0u) | ((cls & 3) == 3 ? 8u : 0u);
    tapmask = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
  }

#define S2_GLDS_W(CCHUNK, WDST)                                                                       \
  {                                                                                                   \
    const float* wsrc = a.wp + (size_t)(CCHUNK) * w_chunk_stride;                                     \
    _Pragma("unroll") for (int j = 0; j < W_PER_WAVE; ++j) {                                          \
      if (W_INSTR % 4 == 0 || wave + 4 * j < W_INSTR)                                                 \
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(wsrc + w_goff[j]),                             \
                                         (lds_void_t*)((WDST) + (wave + 4 * j) * 256), 16, 0, 0);     \
    }                                                                                                 \
  }
#define S2_LOAD_X(CCHUNK)                                                                             \
  {                                                                                                   \
    const int cleft = a.C - (CCHUNK) * CB;                                                            \
    const bool tail = cleft < CB;                                                                     \
    const char* xsrc = (const char*)(xin + (size_t)(CCHUNK) * chunk_elems);                           \
    _Pragma("unroll") for (int i = 0; i < X_PER_T; ++i) {                                             \
      unsigned off = x_boff[i];                                                                       \
      if (tail) off = (x_cc[i] < cleft) ? off : 0u;                                                   \
      xr[i] = *(const float*)(xsrc + off);                                                            \
    }                                                                                                 \
  }
#define S2_STORE_X(BUF, CCHUNK)                                                                       \
  {                                                                                                   \
    const int cleft = a.C - (CCHUNK) * CB;                                                            \
    _Pragma("unroll") for (int i = 0; i < X_PER_T; ++i) {                                             \
      if (X_ELEMS % 256 == 0 || tid + i * 256 < X_ELEMS)                                              \
        smem_x[(BUF) * XS_SZ + x_loff[i]] = ((unsigned)x_cc[i] < (unsigned)cleft) ? xr[i] : 0.f;      \
    }                                                                                                 \
  }

  const int nsteps = a.nchunks;
  S2_GLDS_W(0, smem_w0)
  S2_LOAD_X(0)
  S2_STORE_X(0, 0)
  __syncthreads();

#define S2_STEP(STEP, WCUR, WNXT)                                                                     \
  {                                                                                                   \
    const int chunk = (STEP);                                                                         \
    const bool have_next = chunk + 1 < nsteps;                                                        \
    const int xb = chunk & 1;                                                                         \
    if (have_next) {                                                                                  \
      S2_GLDS_W(chunk + 1, WNXT)                                                                      \
      S2_LOAD_X(chunk + 1)                                                                            \
    }                                                                                                 \
    const float* xl = smem_x + xb * XS_SZ;                                                            \
    float av[MI][FCD_S2_KH];                                                                          \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) {                                               \
      const float* ap = (WCUR) + aoff[mi];                                                            \
      _Pragma("unroll") for (int v4 = 0; v4 < 4; ++v4) {                                              \
        const f32x4 t4 = *(const f32x4*)(ap + 4 * v4);                                                \
        av[mi][4 * v4] = t4[0]; av[mi][4 * v4 + 1] = t4[1];                                           \
        av[mi][4 * v4 + 2] = t4[2]; av[mi][4 * v4 + 3] = t4[3];                                       \
      }                                                                                               \
    }                                                                                                 \
    _Pragma("unroll") for (int tap = 0; tap < 4; ++tap) {                                             \
      if (!((tapmask >> tap) & 1u)) continue;                                                         \
      _Pragma("unroll") for (int cl = 0; cl < CB / 2; ++cl) {                                         \
        float bv[NI];                                                                                 \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                             \
          bv[ni] = xl[xoff[ni] + cl * PLANE + (tap >> 1) * PWP + (tap & 1)];                          \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                             \
          _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                           \
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][cl * 4 + tap], bv[ni], acc[mi][ni], 0, 0, 0); \
      }                                                                                               \
    }                                                                                                 \
    if (have_next) S2_STORE_X(xb ^ 1, chunk + 1)                                                      \
    __syncthreads();                                                                                  \
  }

  for (int step = 0; step < nsteps; step += 2) {
    S2_STEP(step, smem_w0, smem_w1)
    if (step + 1 < nsteps) S2_STEP(step + 1, smem_w1, smem_w0)
  }
#undef S2_STEP
#undef S2_GLDS_W
#undef S2_LOAD_X
#undef S2_STORE_X

  // scatter epilogue: row ko = phase * shuf_C + c  ->  dx[n][c][2 p + (phase >> 1)][2 q + (phase & 1)]
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int pidx = wn * (32 * NI) + ni * 32 + l31;
    const int p = p0 + pidx / TW, q = q0 + pidx % TW;
    if (p >= a.P || q >= a.Q) continue;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ko = ko0 + wm * (32 * MI) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (ko < a.K) {
          const int cls = ko / a.shuf_C, c = ko - cls * a.shuf_C;
          const int ph2 = 2 * p + (cls >> 1), qw2 = 2 * q + (cls & 1);
          if (ph2 < a.shuf_H && qw2 < a.shuf_W)
            a.y[(((size_t)n * a.shuf_C + c) * a.shuf_H + ph2) * a.shuf_W + qw2] = acc[mi][ni][r];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// weight packing
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int K, int C,
                                    int R, int S, int rows, int cols, int mode) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % cols);
    const int row = (int)(i / cols);
    const int tap = row % (R * S), ch = row / (R * S);
    const int r = tap / S, s = tap % S;
    float v = 0.f;
    if (mode == 0) {  // row = (c, r, s), col = k
      if (ch < C && col < K) v = w[(((int64_t)col * C + ch) * R + r) * S + s];
    } else {  // row = (k, r', s'), col = c ; flipped taps
      if (ch < K && col < C) v = w[(((int64_t)ch * C + col) * R + (R - 1 - r)) * S + (S - 1 - s)];
    }
    wp[i] = v;
  }
}

static inline int cb_for(int R, int S) { return (R * S == 1) ? 32 : 8; }

// Which layout a filter is packed in is a pure function of its shape: 3x3 filters whose GEMM has
// more than 32 rows (output channels forward, input channels for the data gradient) run on the
// global_load_lds kernel and use the T layout; everything else the register-staged kernel's
// row layout wp[(c, r, s)][Mpad].
static inline bool t_layout(int M, int R, int S) { return R == 3 && S == 3 && M > 32; }

static void packed_dims(int K, int C, int R, int S, int mode, int* rows, int* cols) {
  const int cb = cb_for(R, S);
  const int M = mode == 0 ? K : C, Cg = mode == 0 ? C : K;      // GEMM rows / reduction channels
  if (t_layout(M, R, S)) {
    *rows = (round_up(Cg, cb) / 4) * 2 * round_up(M, 128);       // (chunk, half, m)
    *cols = FCD_KROW;
    return;
  }
  *rows = round_up(Cg, cb) * R * S;
  *cols = round_up(M, 128);
}

// T layout (see conv_igemm_glds_kernel): wp[q][h][m][KROW], j = (c & 1) * 9 + tap, zero padded.
__global__ void pack_weights_t_kernel(const float* __restrict__ w, float* __restrict__ wp, int K, int C,
                                      int64_t total, int Mpad, int mode) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % FCD_KROW);
    const int64_t row = i / FCD_KROW;
    const int m = (int)(row % Mpad);
    const int h = (int)((row / Mpad) & 1);
    const int q = (int)(row / (2 * (int64_t)Mpad));
    float v = 0.f;
    if (j < FCD_KH) {
      const int ch = q * 4 + h * 2 + j / 9, tap = j % 9;
      const int r = tap / 3, sx = tap % 3;
      if (mode == 0) {   // GEMM row = output channel k, reduction over (c, r, s)
        if (ch < C && m < K) v = w[(((int64_t)m * C + ch) * 3 + r) * 3 + sx];
      } else {           // GEMM row = input channel c, reduction over (k, r', s') with flipped taps
        if (ch < K && m < C) v = w[(((int64_t)ch * C + m) * 3 + (2 - r)) * 3 + (2 - sx)];
      }
    }
    wp[i] = v;
  }
}

extern "C" int64_t fcd_conv_packed_elems(int K, int C, int R, int S, int mode) {
  int rows, cols;
  packed_dims(K, C, R, S, mode, &rows, &cols);
  return (int64_t)rows * cols;
}

extern "C" int fcd_conv_pack_weights(const float* w, float* wp, int K, int C, int R, int S, int mode,
                                     void* stream) {
  FCD_CHECK_ARG(w && wp && K > 0 && C > 0 && R > 0 && S > 0 && (mode == 0 || mode == 1),
                "fcd_conv_pack_weights: bad arguments");
  int rows, cols;
  packed_dims(K, C, R, S, mode, &rows, &cols);
  const int64_t total = (int64_t)rows * cols;
  const int grid = (int)std::min<int64_t>(cdiv64(total, 256), 4096);
  FcdProfScope prof(FCD_K_PACK, (hipStream_t)stream, 0.0, 8.0 * total);
  if (t_layout(mode == 0 ? K : C, R, S))
    hipLaunchKernelGGL(pack_weights_t_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, wp, K, C, total,
                       round_up(mode == 0 ? K : C, 128), mode);
  else
    hipLaunchKernelGGL(pack_weights_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, wp, K, C, R,
                       S, rows, cols, mode);
  FCD_LAUNCH_CHECK("pack_weights");
  return FCD_OK;
}

// ---------------------------------------------------------------------------
// dispatch
static int xcd_remap_on() { return fcd_sw(FCD_SW_CONV_XCD); }

template <int R, int S, int RCH, int STRIDE, int DIL, int CB, int MI, int NI, int WM, int WN, int TH,
          int TW>
static int launch_cfg(const ConvArgs& a0, hipStream_t st) {
  ConvArgs a = a0;
  constexpr int BM = 32 * MI * WM;
  a.tiles_p = cdiv(a.P, TH);
  a.tiles_q = cdiv(a.Q, TW);
  dim3 grid((unsigned)(a.N * a.tiles_p * a.tiles_q), (unsigned)cdiv(a.K, BM));
  if constexpr (R == 3 && S == 3 && BM > 32) {
    // global_load_lds kernel (T-layout filters: must agree with t_layout() used by the packer)
    constexpr int CB2 = FCD_CB2;
    a.nchunks = cdiv(a.C, CB2);
    a.k_tiles = (int)grid.y;
    a.xcd_remap = xcd_remap_on();
    const dim3 grid1 = a.xcd_remap ? dim3(grid.x * grid.y) : grid;
    const int src = a.pool_code_in ? 2 : (a.mask ? 1 : 0);
    if (src == 0) {
      hipLaunchKernelGGL((conv_igemm_glds_kernel<R, S, RCH, STRIDE, DIL, CB2, MI, NI, WM, WN, TH, TW, 0>), grid1,
                         dim3(256), 0, st, a);
      return 0;
    }
    if constexpr (STRIDE == 1) {          // gated sources only occur in data gradients (unit stride)
      if (src == 1) {
        hipLaunchKernelGGL((conv_igemm_glds_kernel<R, S, RCH, STRIDE, DIL, CB2, MI, NI, WM, WN, TH, TW, 1>), grid1,
                           dim3(256), 0, st, a);
        return 0;
      }
      if constexpr (DIL == 1) {
        hipLaunchKernelGGL((conv_igemm_glds_kernel<R, S, RCH, STRIDE, DIL, CB2, MI, NI, WM, WN, TH, TW, 2>), grid1,
                           dim3(256), 0, st, a);
        return 0;
      }
    }
    return -1;
  } else {
    a.nchunks = cdiv(a.C, CB);
#if IG_TIME
    a.tbuf = g_ig_tbuf;
#endif
    hipLaunchKernelGGL((conv_igemm_kernel<R, S, RCH, STRIDE, DIL, CB, MI, NI, WM, WN, TH, TW>), grid,
                       dim3(256), 0, st, a);
    return 0;
  }
}

static int thin_fwd_on() { return fcd_sw(FCD_SW_CONV_THINFWD); }

static int big_tiles_on() { return fcd_sw(FCD_SW_CONV_BIG); }

template <int R, int S, int RCH, int STRIDE, int DIL, int CB>
static int launch_family(const ConvArgs& a, hipStream_t st) {
  const bool wide = a.Q > 16;  // 4x32 pixel tiles for wide maps, 8x16 otherwise
  if (a.K > 64) {
    // 128 x 256 tiles (8x32 pixels, each wave 64 x 128): one filter slab feeds twice the MFMAs, at two
    // workgroups per CU
    if constexpr (R == 3 && S == 3) {
      // ... when the launch still fills the 512 resident slots several times over
      if (wide && a.P >= 8 && big_tiles_on() &&
          (int64_t)a.N * cdiv(a.P, 8) * cdiv(a.Q, 32) * cdiv(a.K, 128) >= 2048)
        return launch_cfg<R, S, RCH, STRIDE, DIL, CB, 2, 4, 2, 2, 8, 32>(a, st);
    }
    return wide ? launch_cfg<R, S, RCH, STRIDE, DIL, CB, 2, 2, 2, 2, 4, 32>(a, st)
                : launch_cfg<R, S, RCH, STRIDE, DIL, CB, 2, 2, 2, 2, 8, 16>(a, st);
  } else if (a.K > 32) {
    // 64 output channels: 64 x 256-pixel tiles (8x32 / 16x16 pixels), each wave 64x64
    if ((R == 3 || R == 9) && a.P >= 8)      // (9x9: a 16 x 40 patch per 8 x 32 pixels instead of 12 x 40 per 4 x 32)
      return wide ? launch_cfg<R, S, RCH, STRIDE, DIL, CB, 2, 2, 1, 4, 8, 32>(a, st)
                  : launch_cfg<R, S, RCH, STRIDE, DIL, CB, 2, 2, 1, 4, 16, 16>(a, st);
    return wide ? launch_cfg<R, S, RCH, STRIDE, DIL, CB, 2, 1, 1, 4, 4, 32>(a, st)
                : launch_cfg<R, S, RCH, STRIDE, DIL, CB, 2, 1, 1, 4, 8, 16>(a, st);
  }
  return wide ? launch_cfg<R, S, RCH, STRIDE, DIL, CB, 1, 1, 1, 4, 4, 32>(a, st)
              : launch_cfg<R, S, RCH, STRIDE, DIL, CB, 1, 1, 1, 4, 8, 16>(a, st);
}

// generic entry: input tensor (N,C,H,W) -> output (N,K,P,Q) with filter RxS.
static int conv_dispatch(const ConvArgs& a, int R, int S, int stride, int dil, hipStream_t st) {
  if (R == 3 && S == 3 && stride == 1 && dil == 1) return launch_family<3, 3, 3, 1, 1, 8>(a, st);
  if (R == 3 && S == 3 && stride == 2 && dil == 1) return launch_family<3, 3, 3, 2, 1, 8>(a, st);
  if (R == 3 && S == 3 && stride == 1 && dil == 2) return launch_family<3, 3, 3, 1, 2, 8>(a, st);
  if (R == 9 && S == 9 && stride == 1 && dil == 1) {
    if (fcd_sw(FCD_SW_CONV_ROWS16) && a.K <= 16 && !a.mask && !a.pool_code_in && !a.shuf_C) {       // 16-row MFMA tiles
      ConvArgs b = a;
      const bool wide = a.Q > 16;
      b.tiles_p = cdiv(a.P, wide ? 8 : 16);
      b.tiles_q = cdiv(a.Q, wide ? 32 : 16);
      b.nchunks = cdiv(a.C, 8);
      const dim3 grid((unsigned)(a.N * b.tiles_p * b.tiles_q));
      if (wide) hipLaunchKernelGGL((conv_igemm_rows16_kernel<9, 9, 1, 8, 8, 32>), grid, dim3(256), 0, st, b);
      else hipLaunchKernelGGL((conv_igemm_rows16_kernel<9, 9, 1, 8, 16, 16>), grid, dim3(256), 0, st, b);
      return 0;
    }
    return launch_family<9, 9, 1, 1, 1, 8>(a, st);
  }
  if (R == 1 && S == 1 && stride == 1 && dil == 1) return launch_family<1, 1, 1, 1, 1, 32>(a, st);
  if (R == 2 && S == 2 && stride == 2 && dil == 1) return launch_family<2, 2, 2, 2, 1, 8>(a, st);
  if (R == 2 && S == 2 && stride == 1 && dil == 2) return launch_family<2, 2, 2, 1, 2, 8>(a, st);
  if (R == 2 && S == 2 && stride == 1 && dil == 1) return launch_family<2, 2, 2, 1, 1, 8>(a, st);
  return -1;
}

// ---------------------------------------------------------------------------
// 1x1 convolution on 1x1 maps with a handful of samples (the Discriminator's classifier after global average pooling:
// Module.py:205-213, 8..32 samples x 512 -> 1024 -> 1): a 2-MB filter read and ~10 MFLOP.  On the implicit-GEMM tiles this
// is ONE workgroup column walking the whole filter (87-161 us per call); here 64 outputs x 4 reduction slices per block,
// every lane streams its filter column ([reduction][Kpad] packing => coalesced), the samples' inputs are wave-uniform.
struct SmallFcArgs {
  const float* x;       // (N, C)
  const float* mask;    // optional gate on x (data gradient: ReLU output of the layer), same shape
  const float* wp;      // [C][Kpad]
  const float* bias;
  float* y;             // (N, K)
  int N, C, K, Kpad, relu;
};

// K >= 64: lanes over outputs.  The samples' inputs are staged through LDS in 128-element reduction chunks (coalesced
// load, gate applied once), every lane streams its filter column, LDS reads are broadcasts.
template <int NB>
__global__ __launch_bounds__(256) void small_fc_kernel(SmallFcArgs a) {
  constexpr int CH = 128;
  __shared__ float xs[NB][CH];
  __shared__ float red[3][64][NB + 1];
  const int l64 = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int o = blockIdx.x * 64 + l64;
  const bool live = o < a.K;
  float acc[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) acc[n] = 0.f;
  for (int c0 = 0; c0 < a.C; c0 += CH) {
    for (int idx = threadIdx.x; idx < NB * CH; idx += 256) {
      const int n = idx / CH, rr = idx % CH;
      float v = 0.f;
      if (n < a.N && c0 + rr < a.C) {
        v = a.x[(size_t)n * a.C + c0 + rr];
        if (a.mask && !(a.mask[(size_t)n * a.C + c0 + rr] > 0.f)) v = 0.f;
      }
      xs[n][rr] = v;
    }
    __syncthreads();
    // the 32 filter values of this slice first (independent loads in flight together), then the multiply-adds
    float wv[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int r = c0 + sl * 32 + i;
      wv[i] = (live && r < a.C) ? a.wp[(size_t)r * a.Kpad + o] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 32; ++i)
#pragma unroll
      for (int n = 0; n < NB; ++n) acc[n] += wv[i] * xs[n][sl * 32 + i];
    __syncthreads();
  }
  if (sl > 0) {
#pragma unroll
    for (int n = 0; n < NB; ++n) red[sl - 1][l64][n] = acc[n];
  }
  __syncthreads();
  if (sl == 0 && live) {
    const float b = a.bias ? a.bias[o] : 0.f;
#pragma unroll
    for (int n = 0; n < NB; ++n)
      if (n < a.N) {
        float v = ((acc[n] + red[0][l64][n]) + red[1][l64][n]) + red[2][l64][n] + b;
        if (a.relu) v = v > 0.f ? v : 0.f;
        a.y[(size_t)n * a.K + o] = v;
      }
  }
}

// K < 64 (the final 1024 -> 1 layer): block = one output, threads over the reduction, fixed-order block sums
template <int NB>
__global__ __launch_bounds__(256) void small_fc_narrow_kernel(SmallFcArgs a) {
  __shared__ float red[256][NB + 1];
  const int o = blockIdx.x;
  float acc[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) acc[n] = 0.f;
  for (int r = threadIdx.x; r < a.C; r += 256) {
    const float w = a.wp[(size_t)r * a.Kpad + o];
#pragma unroll
    for (int n = 0; n < NB; ++n)
      if (n < a.N) {
        float v = a.x[(size_t)n * a.C + r];
        if (a.mask && !(a.mask[(size_t)n * a.C + r] > 0.f)) v = 0.f;
        acc[n] += w * v;
      }
  }
#pragma unroll
  for (int n = 0; n < NB; ++n) red[threadIdx.x][n] = acc[n];
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
#pragma unroll
      for (int n = 0; n < NB; ++n) red[threadIdx.x][n] += red[threadIdx.x + st][n];
    }
    __syncthreads();
  }
  if ((int)threadIdx.x < a.N && (int)threadIdx.x < NB) {
    float v = red[0][threadIdx.x] + (a.bias ? a.bias[o] : 0.f);
    if (a.relu) v = v > 0.f ? v : 0.f;
    a.y[(size_t)threadIdx.x * a.K + o] = v;
  }
}

// 0 = launched
static int try_small_fc(int N, int C, int K, int H, int W, int R, int S, int stride, int pad, const float* x, const float* mask,
                        const float* wp, const float* bias, float* y, int relu, hipStream_t st) {
  if (!(R == 1 && S == 1 && stride == 1 && pad == 0 && H == 1 && W == 1 && N <= 32)) return 1;
  SmallFcArgs a;
  a.x = x; a.mask = mask; a.wp = wp; a.bias = bias; a.y = y;
  a.N = N; a.C = C; a.K = K; a.Kpad = round_up(K, 128); a.relu = relu;
  if (K >= 64) {
    const dim3 grid((unsigned)cdiv(K, 64));
    if (N <= 8) hipLaunchKernelGGL(small_fc_kernel<8>, grid, dim3(256), 0, st, a);
    else if (N <= 16) hipLaunchKernelGGL(small_fc_kernel<16>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(small_fc_kernel<32>, grid, dim3(256), 0, st, a);
  } else {
    const dim3 grid((unsigned)K);
    if (N <= 8) hipLaunchKernelGGL(small_fc_narrow_kernel<8>, grid, dim3(256), 0, st, a);
    else if (N <= 16) hipLaunchKernelGGL(small_fc_narrow_kernel<16>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(small_fc_narrow_kernel<32>, grid, dim3(256), 0, st, a);
  }
  return 0;
}

static int check_desc(const fcd_conv_desc* d, const char* who) {
  FCD_CHECK_ARG(d, "%s: null desc", who);
  FCD_CHECK_ARG(d->N > 0 && d->C > 0 && d->H > 0 && d->W > 0 && d->K > 0, "%s: non-positive dims", who);
  FCD_CHECK_ARG(d->stride == 1 || d->stride == 2, "%s: stride %d unsupported", who, d->stride);
  const int P = (d->H + 2 * d->pad - d->R) / d->stride + 1;
  const int Q = (d->W + 2 * d->pad - d->S) / d->stride + 1;
  FCD_CHECK_ARG(P == d->P && Q == d->Q && P > 0 && Q > 0, "%s: output size (%d,%d) != expected (%d,%d)",
                who, d->P, d->Q, P, Q);
  return FCD_OK;
}

extern "C" int fcd_conv2d_fwd_ex(const fcd_conv_desc* d, const float* x, const float* wp, const float* bias,
                                 float* y, int act, const float* slope_ptr, float slope_imm, const float* residual,
                                 void* stream);

extern "C" int fcd_conv2d_fwd(const fcd_conv_desc* d, const float* x, const float* wp, const float* bias,
                              float* y, int fuse_relu, void* stream) {
  return fcd_conv2d_fwd_ex(d, x, wp, bias, y, fuse_relu ? FCD_ACT_RELU : FCD_ACT_NONE, nullptr, 0.f, nullptr, stream);
}

extern "C" int fcd_conv2d_fwd_ex(const fcd_conv_desc* d, const float* x, const float* wp, const float* bias,
                                 float* y, int act, const float* slope_ptr, float slope_imm, const float* residual,
                                 void* stream) {
  int rc = check_desc(d, "fcd_conv2d_fwd");
  if (rc) return rc;
  FCD_CHECK_ARG(x && wp && y, "fcd_conv2d_fwd: null pointer");
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  FCD_CHECK_ARG(act >= FCD_ACT_NONE && act <= FCD_ACT_PRELU, "fcd_conv2d_fwd: bad activation code %d", act);
  a.x = x; a.wp = wp; a.bias = bias; a.y = y; a.relu = act == FCD_ACT_RELU ? 1 : 0;
  a.act_slope = (act == FCD_ACT_LEAKY || act == FCD_ACT_PRELU) ? 1 : 0;
  a.slope_ptr = slope_ptr; a.slope_imm = slope_imm; a.residual = residual;
  a.N = d->N; a.C = d->C; a.H = d->H; a.W = d->W;
  a.K = d->K; a.Kpad = round_up(d->K, 128);
  a.P = d->P; a.Q = d->Q; a.pad = d->pad;
  const double flops = 2.0 * d->N * d->K * (double)d->P * d->Q * d->C * d->R * d->S;
  const double bytes = 4.0 * ((double)d->N * d->C * d->H * d->W + (double)d->N * d->K * d->P * d->Q +
                              (double)d->K * d->C * d->R * d->S);
  FcdProfScope prof(FCD_K_CONV_FWD, (hipStream_t)stream, flops, bytes, fcd_prof_tag_desc("fwd", d));
  if (!a.act_slope && !residual && thin_fwd_on() &&
      fcd_try_fwd_thin(d, x, wp, bias, y, a.relu, (hipStream_t)stream, nullptr) == 0) {   // <= 4 input channels: VALU kernel
    FCD_LAUNCH_CHECK("conv2d_fwd(thin)");
    return FCD_OK;
  }
  if (!a.act_slope && !residual &&
      try_small_fc(d->N, d->C, d->K, d->H, d->W, d->R, d->S, d->stride, d->pad, x, nullptr, wp, bias, y, a.relu,
                   (hipStream_t)stream) == 0) {
    FCD_LAUNCH_CHECK("conv2d_fwd(small fc)");
    return FCD_OK;
  }
  rc = conv_dispatch(a, d->R, d->S, d->stride, 1, (hipStream_t)stream);
  FCD_CHECK_ARG(rc == 0, "fcd_conv2d_fwd: unsupported filter %dx%d stride %d", d->R, d->S, d->stride);
  FCD_LAUNCH_CHECK("conv2d_fwd");
  return FCD_OK;
}

extern "C" int fcd_conv2d_bwd_data(const fcd_conv_desc* d, const float* dy, const float* relu_out,
                                   const float* wp_bwd, float* dx, void* stream) {
  int rc = check_desc(d, "fcd_conv2d_bwd_data");
  if (rc) return rc;
  FCD_CHECK_ARG(dy && wp_bwd && dx, "fcd_conv2d_bwd_data: null pointer");
  FCD_CHECK_ARG(d->R - 1 - d->pad >= 0, "fcd_conv2d_bwd_data: pad > R-1 unsupported");
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.x = dy; a.wp = wp_bwd; a.bias = nullptr; a.y = dx; a.mask = relu_out;
  a.N = d->N; a.C = d->K; a.H = d->P; a.W = d->Q;     // "input" of the transposed conv = dy
  a.K = d->C; a.Kpad = round_up(d->C, 128);
  a.P = d->H; a.Q = d->W; a.pad = d->R - 1 - d->pad;
  const double flops = 2.0 * d->N * d->K * (double)d->P * d->Q * d->C * d->R * d->S;
  const double bytes = 4.0 * ((double)d->N * d->C * d->H * d->W + (double)d->N * d->K * d->P * d->Q +
                              (double)d->K * d->C * d->R * d->S);
  FcdProfScope prof(FCD_K_CONV_DGRAD, (hipStream_t)stream, flops, bytes, fcd_prof_tag_desc("dgrad", d));
  if (fcd_try_dgrad_thin(d, dy, relu_out, wp_bwd, dx, (hipStream_t)stream, 0) == 0) {   // <= 4 input channels: VALU kernel
    FCD_LAUNCH_CHECK("conv2d_bwd_data(thin)");
    return FCD_OK;
  }
  if (try_small_fc(d->N, d->K, d->C, d->P, d->Q, d->R, d->S, d->stride, d->pad, dy, relu_out, wp_bwd, nullptr, dx, 0,
                   (hipStream_t)stream) == 0) {      // transposed: reduction over K, outputs = C
    FCD_LAUNCH_CHECK("conv2d_bwd_data(small fc)");
    return FCD_OK;
  }
  rc = conv_dispatch(a, d->R, d->S, 1, d->stride, (hipStream_t)stream);
  FCD_CHECK_ARG(rc == 0, "fcd_conv2d_bwd_data: unsupported filter %dx%d stride %d", d->R, d->S, d->stride);
  FCD_LAUNCH_CHECK("conv2d_bwd_data");
  return FCD_OK;
}

// ---------------------------------------------------------------------------
// Data gradient of the 3x3 / stride-2 / pad-1 convolutions (the Discriminator's four layers) as ONE stride-1 pseudo-
// convolution with 2x2 taps over dy whose 4 C output rows are the four sub-pixel phases of dx:
//   dx[2a + pi][2b + pj] = sum_{u, v in {0,1}} f_(pi,pj)[u][v] * dy[a + u][b + v]
//   rows: pi = 0 -> only r = 1 (u = 0);  pi = 1 -> r = 2 (u = 0), r = 0 (u = 1);  columns alike
// 16 multiplies per input-pixel quad instead of the 36 of the zero-dilated read (DIL = 2) it replaces -- that form
// spends 3 of 4 MFMAs on zeros (measured 0.20 of peak on the algorithmic count).  The epilogue scatters the rows to
// their sub-pixel positions (ConvArgs::shuf_*).
__global__ void pack_weights_s2_kernel(const float* __restrict__ w, float* __restrict__ wp, int K, int C, int Mpad,
                                       int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i % Mpad);
    const int row = (int)(i / Mpad);          // (k, u, v)
    const int k = row >> 2, u = (row >> 1) & 1, v = row & 1;
    float val = 0.f;
    if (k < K && m < 4 * C) {
      const int cls = m / C, c = m - cls * C, pi = cls >> 1, pj = cls & 1;
      const int r = pi == 0 ? (u == 0 ? 1 : -1) : (u == 0 ? 2 : 0);
      const int sx = pj == 0 ? (v == 0 ? 1 : -1) : (v == 0 ? 2 : 0);
      if (r >= 0 && sx >= 0) val = w[(((int64_t)k * C + c) * 3 + r) * 3 + sx];
    }
    wp[i] = val;
  }
}

// the same pseudo-filters in the slab layout of conv_s2sub_glds_kernel: wt[q][h][m][20], k-value j = cl * 4 + u * 2 + v of
// forward filter k = 8 q + 4 h + cl (j >= 16: padding)
__global__ void pack_weights_s2t_kernel(const float* __restrict__ w, float* __restrict__ wt, int K, int C, int Mpad,
                                        int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % FCD_S2_KROW);
    int64_t t = i / FCD_S2_KROW;
    const int m = (int)(t % Mpad); t /= Mpad;
    const int h = (int)(t & 1), q = (int)(t >> 1);
    float val = 0.f;
    if (j < FCD_S2_KH) {
      const int k = q * 8 + h * 4 + (j >> 2), u = (j >> 1) & 1, v = j & 1;
      if (k < K && m < 4 * C) {
        const int cls = m / C, c = m - cls * C, pi = cls >> 1, pj = cls & 1;
        const int r = pi == 0 ? (u == 0 ? 1 : -1) : (u == 0 ? 2 : 0);
        const int sx = pj == 0 ? (v == 0 ? 1 : -1) : (v == 0 ? 2 : 0);
        if (r >= 0 && sx >= 0) val = w[(((int64_t)k * C + c) * 3 + r) * 3 + sx];
      }
    }
    wt[i] = val;
  }
}

extern "C" int fcd_conv_s2_dgrad_plan(const fcd_conv_desc* d) {
  return (fcd_sw(FCD_SW_S2_SUBPIXEL) && d && d->R == 3 && d->S == 3 && d->stride == 2 && d->pad == 1 && d->C >= 8) ? 1 : 0;
}

static int64_t s2_rows_elems(int K, int C) { return (int64_t)round_up(K, 8) * 4 * round_up(4 * C, 128); }
static int64_t s2_slab_elems(int K, int C) { return (int64_t)(round_up(K, 8) / 8) * 2 * round_up(4 * C, 128) * FCD_S2_KROW; }
// [row layout of the register-staged kernel | slab layout of the LDS-DMA kernel]
extern "C" int64_t fcd_conv_s2_dgrad_packed_elems(int K, int C) { return s2_rows_elems(K, C) + s2_slab_elems(K, C); }

extern "C" int fcd_conv_s2_dgrad_pack(const float* w, float* wp, int K, int C, void* stream) {
  FCD_CHECK_ARG(w && wp && K > 0 && C > 0, "fcd_conv_s2_dgrad_pack: bad arguments");
  const int64_t total = s2_rows_elems(K, C), total_t = s2_slab_elems(K, C);
  const int grid = (int)std::min<int64_t>(cdiv64(total, 256), 4096);
  FcdProfScope prof(FCD_K_PACK, (hipStream_t)stream, 0.0, 4.0 * (total + total_t));
  hipLaunchKernelGGL(pack_weights_s2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, wp, K, C,
                     round_up(4 * C, 128), total);
  hipLaunchKernelGGL(pack_weights_s2t_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total_t, 256), 4096)), dim3(256), 0,
                     (hipStream_t)stream, w, wp + total, K, C, round_up(4 * C, 128), total_t);
  FCD_LAUNCH_CHECK("pack_weights_s2");
  return FCD_OK;
}

extern "C" int fcd_conv2d_bwd_data_s2(const fcd_conv_desc* d, const float* dy, const float* relu_out, const float* wp_s2,
                                      float* dx, void* stream) {
  int rc = check_desc(d, "fcd_conv2d_bwd_data_s2");
  if (rc) return rc;
  FCD_CHECK_ARG(dy && wp_s2 && dx, "fcd_conv2d_bwd_data_s2: null pointer");
  FCD_CHECK_ARG(fcd_conv_s2_dgrad_plan(d), "fcd_conv2d_bwd_data_s2: only 3x3 / stride 2 / pad 1 layers");
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.x = dy; a.wp = wp_s2; a.y = dx; a.mask = relu_out;
  a.N = d->N; a.C = d->K; a.H = d->P; a.W = d->Q;
  a.K = 4 * d->C; a.Kpad = round_up(4 * d->C, 128);
  a.P = d->P; a.Q = d->Q; a.pad = 0;
  a.shuf_C = d->C; a.shuf_H = d->H; a.shuf_W = d->W;
  const double flops = 2.0 * d->N * d->K * (double)d->P * d->Q * d->C * 9;
  const double bytes = 4.0 * ((double)d->N * d->C * d->H * d->W + (double)d->N * d->K * d->P * d->Q + (double)d->K * d->C * 9);
  FcdProfScope prof(FCD_K_CONV_DGRAD, (hipStream_t)stream, flops, bytes, fcd_prof_tag_desc("dgrad_s2sub", d));
  if (fcd_sw(FCD_SW_S2_GLDS) &&      // S2_GLDS=0: the register-staged kernel for every layer
      !relu_out && a.K > 64) {
    a.wp = wp_s2 + s2_rows_elems(d->K, d->C);
    a.nchunks = cdiv(a.C, 8);
    const bool wide = a.Q > 16;
    // (256-pixel tiles, each wave 64 x 128 at two workgroups per CU -- half the filter-slab DMA per MFMA: no change, 297 vs 297 us)
    a.tiles_p = cdiv(a.P, wide ? 4 : 8);
    a.tiles_q = cdiv(a.Q, wide ? 32 : 16);
    a.k_tiles = cdiv(a.K, 128);
    a.xcd_remap = xcd_remap_on();
    const unsigned pix = (unsigned)(a.N * a.tiles_p * a.tiles_q);
    const dim3 grid = a.xcd_remap ? dim3(pix * (unsigned)a.k_tiles) : dim3(pix, (unsigned)a.k_tiles);
    if (wide) hipLaunchKernelGGL((conv_s2sub_glds_kernel<2, 2, 2, 2, 4, 32>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((conv_s2sub_glds_kernel<2, 2, 2, 2, 8, 16>), grid, dim3(256), 0, (hipStream_t)stream, a);
    FCD_LAUNCH_CHECK("conv2d_bwd_data_s2");
    return FCD_OK;
  }
  rc = conv_dispatch(a, 2, 2, 1, 1, (hipStream_t)stream);
  FCD_CHECK_ARG(rc == 0, "fcd_conv2d_bwd_data_s2: dispatch failed");
  FCD_LAUNCH_CHECK("conv2d_bwd_data_s2");
  return FCD_OK;
}

// ---------------------------------------------------------------------------
// Thin-channel layers (<= 4 input channels, e.g. VGG conv1_1 on single bands): fused-ReLU forward that also
// writes the ReLU mask as 4 bits per 1 x 4 pixel strip, and the data gradient that consumes it -- the fp32
// activation then need not be kept (or re-read) for the backward pass.
extern "C" size_t fcd_conv2d_relu_bits_bytes(const fcd_conv_desc* d) {
  if (!d || !(d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1 && d->C >= 1 && d->C <= 4 && d->K > 32 &&
              (d->W & 3) == 0 && (d->K % 8) == 0))
    return 0;
  return (size_t)d->N * d->K * d->H * (d->W >> 2);
}

extern "C" int fcd_conv2d_fwd_relu_bits(const fcd_conv_desc* d, const float* x, const float* wp, const float* bias,
                                        float* y, unsigned char* bits, void* stream) {
  int rc = check_desc(d, "fcd_conv2d_fwd_relu_bits");
  if (rc) return rc;
  FCD_CHECK_ARG(x && wp && y && bits, "fcd_conv2d_fwd_relu_bits: null pointer");
  FCD_CHECK_ARG(fcd_conv2d_relu_bits_bytes(d) > 0, "fcd_conv2d_fwd_relu_bits: layer has no bit-mask path");
  const double flops = 2.0 * d->N * d->K * (double)d->P * d->Q * d->C * 9;
  FcdProfScope prof(FCD_K_CONV_FWD, (hipStream_t)stream, flops,
                    4.0 * ((double)d->N * d->C * d->H * d->W + 1.0625 * d->N * d->K * (double)d->P * d->Q),
                    fcd_prof_tag_desc("fwd_bits", d));
  FCD_CHECK_ARG(fcd_try_fwd_thin(d, x, wp, bias, y, 1, (hipStream_t)stream, bits) == 0,
                "fcd_conv2d_fwd_relu_bits: unsupported shape");
  FCD_LAUNCH_CHECK("conv2d_fwd_relu_bits");
  return FCD_OK;
}

extern "C" int fcd_conv2d_bwd_data_bits(const fcd_conv_desc* d, const float* dy, const unsigned char* bits,
                                        const float* wp_bwd, float* dx, void* stream) {
  int rc = check_desc(d, "fcd_conv2d_bwd_data_bits");
  if (rc) return rc;
  FCD_CHECK_ARG(dy && bits && wp_bwd && dx, "fcd_conv2d_bwd_data_bits: null pointer");
  FCD_CHECK_ARG(fcd_conv2d_relu_bits_bytes(d) > 0, "fcd_conv2d_bwd_data_bits: layer has no bit-mask path");
  const double flops = 2.0 * d->N * d->K * (double)d->P * d->Q * d->C * 9;
  FcdProfScope prof(FCD_K_CONV_DGRAD, (hipStream_t)stream, flops,
                    4.0 * ((double)d->N * d->C * d->H * d->W + 1.0625 * d->N * d->K * (double)d->P * d->Q),
                    fcd_prof_tag_desc("dgrad_bits", d));
  FCD_CHECK_ARG(fcd_try_dgrad_thin(d, dy, (const float*)bits, wp_bwd, dx, (hipStream_t)stream, 1) == 0,
                "fcd_conv2d_bwd_data_bits: unsupported shape");
  FCD_LAUNCH_CHECK("conv2d_bwd_data_bits");
  return FCD_OK;
}

// ---------------------------------------------------------------------------
// conv3x3 + bias + ReLU + MaxPool2d(2) in one kernel, and its data gradient
static int pool_supported(const fcd_conv_desc* d, int out_channels, const char* who) {
  FCD_CHECK_ARG(d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1, "%s: only 3x3 / stride 1 / pad 1", who);
  FCD_CHECK_ARG(out_channels > 32, "%s: needs > 32 output channels of the launched GEMM (got %d)", who, out_channels);
  FCD_CHECK_ARG(d->P >= 2 && d->Q >= 2, "%s: map too small to pool", who);
  return FCD_OK;
}

extern "C" int fcd_conv2d_fwd_relu_pool(const fcd_conv_desc* d, const float* x, const float* wp, const float* bias,
                                        float* y_pool, unsigned char* code, void* stream) {
  int rc = check_desc(d, "fcd_conv2d_fwd_relu_pool");
  if (rc) return rc;
  FCD_CHECK_ARG(x && wp && y_pool && code, "fcd_conv2d_fwd_relu_pool: null pointer");
  rc = pool_supported(d, d->K, "fcd_conv2d_fwd_relu_pool");
  if (rc) return rc;
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.wp = wp; a.bias = bias; a.y = nullptr; a.relu = 1;
  a.pool_y = y_pool; a.pool_code_out = code;
  a.N = d->N; a.C = d->C; a.H = d->H; a.W = d->W;
  a.K = d->K; a.Kpad = round_up(d->K, 128);
  a.P = d->P; a.Q = d->Q; a.pad = d->pad;
  const double flops = 2.0 * d->N * d->K * (double)d->P * d->Q * d->C * 9;
  const double bytes = 4.0 * ((double)d->N * d->C * d->H * d->W + 0.3125 * d->N * d->K * d->P * d->Q + (double)d->K * d->C * 9);
  FcdProfScope prof(FCD_K_CONV_FWD, (hipStream_t)stream, flops, bytes, fcd_prof_tag_desc("fwd", d));
  rc = conv_dispatch(a, 3, 3, 1, 1, (hipStream_t)stream);
  FCD_CHECK_ARG(rc == 0, "fcd_conv2d_fwd_relu_pool: dispatch failed");
  FCD_LAUNCH_CHECK("conv2d_fwd_relu_pool");
  return FCD_OK;
}

extern "C" int fcd_conv2d_bwd_data_pooled(const fcd_conv_desc* d, const float* dy_pool, const unsigned char* code,
                                          const float* wp_bwd, float* dx, void* stream) {
  int rc = check_desc(d, "fcd_conv2d_bwd_data_pooled");
  if (rc) return rc;
  FCD_CHECK_ARG(dy_pool && code && wp_bwd && dx, "fcd_conv2d_bwd_data_pooled: null pointer");
  rc = pool_supported(d, d->C, "fcd_conv2d_bwd_data_pooled");
  if (rc) return rc;
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.x = dy_pool; a.wp = wp_bwd; a.y = dx;
  a.pool_code_in = code; a.Hp = d->P / 2; a.Wp = d->Q / 2;
  a.N = d->N; a.C = d->K; a.H = d->P; a.W = d->Q;
  a.K = d->C; a.Kpad = round_up(d->C, 128);
  a.P = d->H; a.Q = d->W; a.pad = 1;
  const double flops = 2.0 * d->N * d->K * (double)d->P * d->Q * d->C * 9;
  const double bytes = 4.0 * ((double)d->N * d->C * d->H * d->W + 0.3125 * d->N * d->K * d->P * d->Q + (double)d->K * d->C * 9);
  FcdProfScope prof(FCD_K_CONV_DGRAD, (hipStream_t)stream, flops, bytes, fcd_prof_tag_desc("dgrad", d));
  rc = conv_dispatch(a, 3, 3, 1, 1, (hipStream_t)stream);
  FCD_CHECK_ARG(rc == 0, "fcd_conv2d_bwd_data_pooled: dispatch failed");
  FCD_LAUNCH_CHECK("conv2d_bwd_data_pooled");
  return FCD_OK;
}
