// Weight gradient of conv2d on the fp32 matrix cores.
//
//   dW[k][c][r][s] = sum_{n,p,q} dY[n,k,p,q] * X[n,c,p*stride+r-pad,q*stride+s-pad]
//
// GEMM view: M = K, N = C x taps, reduction = output pixels.  The MFMA wants the
// NON-reduced index (k resp. c) across lanes, while NCHW keeps the reduced index
// (pixels) contiguous -- so the operands are first re-laid out channel-minor
// ("NHWC", channels zero-padded to a multiple of 64) by a small transpose kernel
// (one extra read+write of X and dY, ~1-2 % of the GEMM time).  In that layout
//   * the dY slab of a pixel-row tile is [TW pixels][64 k]   (256 B per pixel)
//   * the input patch is              [RB rows][PW cols][64 c]
// both lane-linear, so they are DMA'd straight into LDS with global_load_lds (no
// VGPR staging, zeros for halo/out-of-range lanes come from a zero page), every
// MFMA operand read is a conflict-free ds_read_b32 (lanes = consecutive channels),
// and every tap's MFMA reuses the SAME staged patch (shifted reads).
// Workgroup: 64(k) x 64(c) x (RB*S taps) slab of dW, 4 waves as 2x2, each wave
// 32x32xtaps accumulators (v_mfma_f32_32x32x2_f32, two pixels per instruction);
// LDS double-buffered, one barrier per pixel tile.  Split-K over pixel tiles across
// workgroups; partial slabs are summed by a second kernel (deterministic, no atomics).
#include "common.h"

#include <optional>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

// ---- NCHW -> N,HW,Cp (channels minor, zero padded) ---------------------------------
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src,
                                                           const float* __restrict__ mask, float* __restrict__ dst,
                                                           int C, int HW, int Cp) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const float* s = src + (size_t)n * C * HW;
  float* d = dst + (size_t)n * HW * Cp;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + ty + j * 8, p = p0 + tx;
    float v = 0.f;
    if (c < C && p < HW) {
      v = s[(size_t)c * HW + p];
      if (mask && !(mask[(size_t)n * C * HW + (size_t)c * HW + p] > 0.f)) v = 0.f;
    }
    tile[ty + j * 8][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = p0 + ty + j * 8, c = c0 + tx;
    if (p < HW && c < Cp) d[(size_t)p * Cp + c] = tile[tx][ty + j * 8];
  }
}

// 64x64 tile, float4 on both sides (HW % 4 == 0): 256-B rows in, 256-B channel rows out
// psum (optional): per-block channel sums of the (masked) tile, [n * gridDim.x + blockIdx.x][Cp] floats --
// the bias gradient falls out of the pass that re-lays dy out for the weight gradient.
__global__ __launch_bounds__(256) void nchw_to_nhwc_v4_kernel(const float* __restrict__ src,
                                                              const float* __restrict__ mask,
                                                              float* __restrict__ dst, int C, int HW, int Cp,
                                                              float* __restrict__ psum) {
  __shared__ float tile[64][65];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int t = threadIdx.x;
  const float* s = src + (size_t)n * C * HW;
  const float* m = mask ? mask + (size_t)n * C * HW : nullptr;
  float* d = dst + (size_t)n * HW * Cp;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = t + j * 256;          // 64 channel rows x 16 float4
    const int cr = idx >> 4, p4 = (idx & 15) * 4;
    const int c = c0 + cr, p = p0 + p4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C && p < HW) {                // HW % 4 == 0 => the whole float4 is in range
      v = *reinterpret_cast<const float4*>(s + (size_t)c * HW + p);
      if (m) {
        const float4 k = *reinterpret_cast<const float4*>(m + (size_t)c * HW + p);
        if (!(k.x > 0.f)) v.x = 0.f;
        if (!(k.y > 0.f)) v.y = 0.f;
        if (!(k.z > 0.f)) v.z = 0.f;
        if (!(k.w > 0.f)) v.w = 0.f;
      }
    }
    tile[cr][p4 + 0] = v.x; tile[cr][p4 + 1] = v.y; tile[cr][p4 + 2] = v.z; tile[cr][p4 + 3] = v.w;
  }
  __syncthreads();
  if (psum != nullptr) {                   // 4 threads per channel row, 16 pixels each, fixed order
    const int cr = t >> 2, q = t & 3;
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) a += tile[cr][q * 16 + i];
    a += __shfl_xor(a, 1, 64);
    a += __shfl_xor(a, 2, 64);
    if (q == 0) psum[((size_t)n * gridDim.x + blockIdx.x) * Cp + c0 + cr] = a;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = t + j * 256;          // 64 pixel rows x 16 float4 of channels
    const int pr = idx >> 4, c4 = (idx & 15) * 4;
    const int p = p0 + pr, c = c0 + c4;
    if (p < HW && c < Cp) {               // Cp % 64 == 0
      const float4 v = make_float4(tile[c4 + 0][pr], tile[c4 + 1][pr], tile[c4 + 2][pr], tile[c4 + 3][pr]);
      *reinterpret_cast<float4*>(d + (size_t)p * Cp + c) = v;
    }
  }
}

// db[c] = sum of the per-block partials (fp64 accumulation, fixed order)
__global__ __launch_bounds__(256) void channel_psum_fin_kernel(const float* __restrict__ psum, float* __restrict__ out,
                                                               int C, int Cp, int nblk) {
  __shared__ double red[16];
  const int c = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) s += (double)psum[(size_t)b * Cp + c];
  s = block_sum_d(s, red);
  if (threadIdx.x == 0 && c < C) out[c] = (float)s;
}

static void launch_transpose(const float* src, const float* mask, float* dst, int N, int C, int HW, int Cp,
                             hipStream_t st, float* psum = nullptr) {
  if ((HW & 3) == 0)
    hipLaunchKernelGGL(nchw_to_nhwc_v4_kernel, dim3(cdiv(HW, 64), Cp / 64, N), dim3(256), 0, st, src, mask, dst, C, HW,
                       Cp, psum);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(cdiv(HW, 32), Cp / 32, N), dim3(256), 0, st, src, mask, dst, C, HW,
                       Cp);
}

struct WgradArgs {
  const float* xt;    // [N][H][W][Cp]
  const float* dyt;   // [N][P][Q][Kp]
  const float* zeros; // >= 256 B of zeros
  float* out;
  int N, C, H, W, K, P, Q, pad;
  int Cp, Kp;
  int tiles_q, total_tiles, tiles_per_split;
  int c_tiles;
  long long split_stride;
  const float* x;     // [r5] NCHW-direct rolling kernel: the operands as they are (no channel-minor copy)
  const float* dy;
  float* db_part;     //      [split][Kp] partial bias gradients (null: none wanted)
  int tkc;                      // [r4] split partials as [tap][K][C] (lanes along C: 128-B runs) instead of dw's [K][C][tap], where the 64 lanes
                                //      of a store hit 64 different 36-B filters; the split reduction writes dw's layout
  unsigned long long* tbuf;     // WG_TIME builds
};

// WG_TIME: attribution build of the generic weight-gradient kernel (tools/wgrad_segments.py): per-wave cycle sums {prologue, LDS-DMA issue
// of the next tile, MFMA loop, barrier (+ wait for that DMA), epilogue, -, total} in a debug buffer.  Never defined in the product build.
#ifndef WG_TIME
#define WG_TIME 0
#endif
#if WG_TIME
#define WG_T(var) __builtin_amdgcn_sched_barrier(0); const unsigned long long var = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#define WG_TACC(slot, t1, t0) wtacc[slot] += (t1) - (t0);
unsigned long long* g_wg_tbuf = nullptr;
extern "C" void fcd_wgrad_time_buf(void* p) { g_wg_tbuf = (unsigned long long*)p; }
#else
#define WG_T(var)
#define WG_TACC(slot, t1, t0)
#endif

template <int R, int S, int RB, int STRIDE, int TW>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int T = RB * S;
  constexpr int PW = (TW - 1) * STRIDE + S;
  constexpr int NPOS = RB * PW;                       // patch positions (each 64 channels)
  constexpr int X_INSTR = (NPOS + 3) / 4;             // 4 positions (1 KiB) per wave instruction
  constexpr int DY_INSTR = TW / 4;
  constexpr int XS_SZ = X_INSTR * 4 * 64, DYS_SZ = TW * 64;
  constexpr int X_PER_WAVE = (X_INSTR + 3) / 4, DY_PER_WAVE = (DY_INSTR + 3) / 4;
  static_assert(TW % 4 == 0, "tile width");
  __shared__ __attribute__((aligned(16))) float smem[2 * (XS_SZ + DYS_SZ)];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wc = wave & 1;
  const int ko0 = (blockIdx.x / a.c_tiles) * 64;
  const int c0 = (blockIdx.x % a.c_tiles) * 64;
  const int r0 = blockIdx.z * RB;
  const int split = blockIdx.y;
  const int sub = lane >> 4, col4 = (lane & 15) * 4;   // position within an instruction, channel quad

#if WG_TIME
  unsigned long long wtacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  WG_T(w_begin)
  f32x16 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int tile_beg = split * a.tiles_per_split;
  const int tile_end = min(tile_beg + a.tiles_per_split, a.total_tiles);

#define FCD_WG_STAGE(TILE, BUF)                                                                       \
  {                                                                                                   \
    int tt = (TILE);                                                                                  \
    const int tq = tt % a.tiles_q;                                                                    \
    tt /= a.tiles_q;                                                                                  \
    const int p = tt % a.P, n = tt / a.P;                                                             \
    const int q0 = tq * TW;                                                                           \
    float* xs = smem + (BUF) * (XS_SZ + DYS_SZ);                                                      \
    float* dys = xs + XS_SZ;                                                                          \
    _Pragma("unroll") for (int j = 0; j < DY_PER_WAVE; ++j) {                                         \
      const int ins = wave + 4 * j;                                                                   \
      if (DY_INSTR % 4 == 0 || ins < DY_INSTR) {                                                      \
        const int q = q0 + ins * 4 + sub;                                                             \
        const float* src = (q < a.Q) ? a.dyt + (((size_t)n * a.P + p) * a.Q + q) * a.Kp + ko0 + col4  \
                                     : a.zeros + col4;                                                \
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(dys + ins * 256), 16, 0, 0); \
      }                                                                                               \
    }                                                                                                 \
    const int ih0 = p * STRIDE - a.pad + r0, iw0 = q0 * STRIDE - a.pad;                               \
    _Pragma("unroll") for (int j = 0; j < X_PER_WAVE; ++j) {                                          \
      const int ins = wave + 4 * j;                                                                   \
      if (X_INSTR % 4 == 0 || ins < X_INSTR) {                                                        \
        const int pos = ins * 4 + sub;                                                                \
        const int ph = pos / PW, pw = pos % PW;                                                       \
        const int ih = ih0 + ph, iw = iw0 + pw;                                                       \
        const bool ok = pos < NPOS && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;                     \
        const float* src = ok ? a.xt + (((size_t)n * a.H + ih) * a.W + iw) * a.Cp + c0 + col4         \
                              : a.zeros + col4;                                                       \
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(xs + ins * 256), 16, 0, 0);  \
      }                                                                                               \
    }                                                                                                 \
  }

  if (tile_beg < tile_end) {
    FCD_WG_STAGE(tile_beg, 0)
  }
  __syncthreads();
  WG_T(w_loop)
  WG_TACC(0, w_loop, w_begin)
  int buf = 0;
  for (int tile = tile_beg; tile < tile_end; ++tile) {
    WG_T(ws0)
    if (tile + 1 < tile_end) FCD_WG_STAGE(tile + 1, buf ^ 1)
    WG_T(ws1)
    const float* xs = smem + buf * (XS_SZ + DYS_SZ);
    const float* a_base = xs + XS_SZ + half * 64 + wm * 32 + l31;
    const float* b_base = xs + half * STRIDE * 64 + wc * 32 + l31;
#pragma unroll
    for (int t2 = 0; t2 < TW / 2; ++t2) {
      const float av = a_base[(2 * t2) * 64];
#pragma unroll
      for (int rl = 0; rl < RB; ++rl)
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const float bv = b_base[(rl * PW + (2 * t2) * STRIDE + s) * 64];
          acc[rl * S + s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[rl * S + s], 0, 0, 0);
        }
    }
    WG_T(ws2)
    __syncthreads();
    WG_T(ws3)
    WG_TACC(1, ws1, ws0) WG_TACC(2, ws2, ws1) WG_TACC(3, ws3, ws2)
    buf ^= 1;
  }
#undef FCD_WG_STAGE
  WG_T(w_epi)

  float* out = a.out + (size_t)split * a.split_stride;
  const int c = c0 + wc * 32 + l31;
  if (c < a.C) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int r = r0 + t / S, s = t % S;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int ko = ko0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
        if (ko < a.K) out[a.tkc ? ((size_t)(r * S + s) * a.K + ko) * a.C + c : (((size_t)ko * a.C + c) * R + r) * S + s] = acc[t][reg];
      }
    }
  }
#if WG_TIME
  {
    const unsigned long long w_end = __builtin_readcyclecounter();
    wtacc[4] = w_end - w_epi; wtacc[6] = w_end - w_begin;
    if (a.tbuf && lane == 0) {
      const size_t wg = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
#pragma unroll
      for (int i = 0; i < 8; ++i) a.tbuf[(wg * 4 + wave) * 8 + i] = wtacc[i];
    }
  }
#endif
}

// ---------------------------------------------------------------------------
// 3x3 / stride-1 / pad-1 specialisation with a ROLLING input patch: a workgroup walks its
// pixel tiles DOWN the image rows of one column strip (p fastest), so consecutive tiles share
// two of their three patch rows.  The patch lives in a 4-slot ring of row buffers
// [PW][64 c]; each tile only DMA-loads the one new row (and the next dY slab) while the
// current tile is being multiplied -- L2->LDS traffic per tile drops from 34 KB to 17 KB,
// which is what bounds the generic kernel (~7.4 B/clk/CU, the LDS-DMA rate of a CU).
template <int TW>
__global__ __launch_bounds__(256, 2) void conv_wgrad_roll_kernel(WgradArgs a) {
  constexpr int PW = TW + 2;
  constexpr int ROW_INSTR = (PW + 3) / 4;          // wave-instructions (4 positions each) per patch row
  constexpr int SLOT = ROW_INSTR * 4 * 64;         // floats per ring slot (with slack)
  constexpr int DY_INSTR = TW / 4;
  constexpr int DYS = TW * 64;
  __shared__ __attribute__((aligned(16))) float smem[4 * SLOT + 2 * DYS];
  float* ring = smem;
  float* dybuf = smem + 4 * SLOT;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wc = wave & 1;
  const int ko0 = (blockIdx.x / a.c_tiles) * 64;
  const int c0 = (blockIdx.x % a.c_tiles) * 64;
  const int split = blockIdx.y;
  const int sub = lane >> 4, col4 = (lane & 15) * 4;

#if WG_TIME
  unsigned long long wtacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  WG_T(w_begin)
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int tile_beg = split * a.tiles_per_split;
  const int tile_end = min(tile_beg + a.tiles_per_split, a.total_tiles);

  // tile t -> (n, tq, p) with p fastest
#define FCD_ROLL_DECODE(T, N_, TQ_, P_) \
  const int P_ = (T) % a.P;             \
  const int TQ_ = ((T) / a.P) % a.tiles_q; \
  const int N_ = (T) / (a.P * a.tiles_q);
  // DMA one input row ih (may be out of the image: zeros) of strip (n, tq) into its ring slot
#define FCD_ROLL_LOAD_ROW(N_, TQ_, IH)                                                                \
  {                                                                                                   \
    const int ih_ = (IH);                                                                             \
    float* dst = ring + ((ih_ + 1) & 3) * SLOT;                                                       \
    const int iw0 = (TQ_) * TW - 1;                                                                   \
    _Pragma("unroll") for (int j = 0; j < (ROW_INSTR + 3) / 4; ++j) {                                 \
      const int ins = wave + 4 * j;                                                                   \
      if (ins < ROW_INSTR) {                                                                          \
        const int pw = ins * 4 + sub;                                                                 \
        const int iw = iw0 + pw;                                                                      \
        const bool ok = pw < PW && ih_ >= 0 && ih_ < a.H && iw >= 0 && iw < a.W;                      \
        const float* src = ok ? a.xt + (((size_t)(N_) * a.H + ih_) * a.W + iw) * a.Cp + c0 + col4     \
                              : a.zeros + col4;                                                       \
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(dst + ins * 256), 16, 0, 0); \
      }                                                                                               \
    }                                                                                                 \
  }
#define FCD_ROLL_LOAD_DY(N_, TQ_, P_, BUF)                                                            \
  {                                                                                                   \
    float* dst = dybuf + (BUF) * DYS;                                                                 \
    _Pragma("unroll") for (int j = 0; j < (DY_INSTR + 3) / 4; ++j) {                                  \
      const int ins = wave + 4 * j;                                                                   \
      if (ins < DY_INSTR) {                                                                           \
        const int q = (TQ_) * TW + ins * 4 + sub;                                                     \
        const float* src = (q < a.Q) ? a.dyt + (((size_t)(N_) * a.P + (P_)) * a.Q + q) * a.Kp + ko0 + col4 \
                                     : a.zeros + col4;                                                \
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(dst + ins * 256), 16, 0, 0); \
      }                                                                                               \
    }                                                                                                 \
  }

  if (tile_beg < tile_end) {
    FCD_ROLL_DECODE(tile_beg, n0, tq0, p0)
    FCD_ROLL_LOAD_ROW(n0, tq0, p0 - 1)
    FCD_ROLL_LOAD_ROW(n0, tq0, p0)
    FCD_ROLL_LOAD_ROW(n0, tq0, p0 + 1)
    FCD_ROLL_LOAD_DY(n0, tq0, p0, 0)
  }
  __syncthreads();
  WG_T(w_loop)
  WG_TACC(0, w_loop, w_begin)
  int buf = 0;
  for (int tile = tile_beg; tile < tile_end; ++tile) {
    WG_T(ws0)
    FCD_ROLL_DECODE(tile, n, tq, p)
    const bool have_next = tile + 1 < tile_end;
    const bool same_strip = have_next && (p + 1 < a.P);
    if (have_next) {
      FCD_ROLL_DECODE(tile + 1, nn, tqn, pn)
      FCD_ROLL_LOAD_DY(nn, tqn, pn, buf ^ 1)
      if (same_strip) FCD_ROLL_LOAD_ROW(n, tq, p + 2)       // the one new row of the next tile
    }
    WG_T(ws1)
    const float* a_base = dybuf + buf * DYS + half * 64 + wm * 32 + l31;
    const float* b0 = ring + ((p + 0) & 3) * SLOT + half * 64 + wc * 32 + l31;   // input row p-1
    const float* b1 = ring + ((p + 1) & 3) * SLOT + half * 64 + wc * 32 + l31;   // row p
    const float* b2 = ring + ((p + 2) & 3) * SLOT + half * 64 + wc * 32 + l31;   // row p+1
#pragma unroll
    for (int t2 = 0; t2 < TW / 2; ++t2) {
      const float av = a_base[(2 * t2) * 64];
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        acc[0 + s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0[(2 * t2 + s) * 64], acc[0 + s], 0, 0, 0);
        acc[3 + s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1[(2 * t2 + s) * 64], acc[3 + s], 0, 0, 0);
        acc[6 + s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b2[(2 * t2 + s) * 64], acc[6 + s], 0, 0, 0);
      }
    }
    WG_T(ws2)
    __syncthreads();
    WG_T(ws3)
    WG_TACC(1, ws1, ws0) WG_TACC(2, ws2, ws1) WG_TACC(3, ws3, ws2)
    if (have_next && !same_strip) {      // strip change: (re)load the three rows of the new strip
      FCD_ROLL_DECODE(tile + 1, nn, tqn, pn)
      FCD_ROLL_LOAD_ROW(nn, tqn, pn - 1)
      FCD_ROLL_LOAD_ROW(nn, tqn, pn)
      FCD_ROLL_LOAD_ROW(nn, tqn, pn + 1)
      __syncthreads();
    }
    buf ^= 1;
  }
#undef FCD_ROLL_DECODE
#undef FCD_ROLL_LOAD_ROW
#undef FCD_ROLL_LOAD_DY
  WG_T(w_epi)

  float* out = a.out + (size_t)split * a.split_stride;
  const int c = c0 + wc * 32 + l31;
  if (c < a.C) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int ko = ko0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
        if (ko < a.K) out[a.tkc ? ((size_t)t * a.K + ko) * a.C + c : (((size_t)ko * a.C + c) * 3 + t / 3) * 3 + t % 3] = acc[t][reg];
      }
    }
  }
#if WG_TIME
  {
    const unsigned long long w_end = __builtin_readcyclecounter();
    wtacc[4] = w_end - w_epi; wtacc[6] = w_end - w_begin;
    if (a.tbuf && lane == 0) {
      const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
#pragma unroll
      for (int i = 0; i < 8; ++i) a.tbuf[(wg * 4 + wave) * 8 + i] = wtacc[i];
    }
  }
#endif
}

// ---------------------------------------------------------------------------
// [r5] The rolling 3x3 kernel on the NCHW operands themselves (W % 32 == 0, no ReLU mask): the channel-minor copies of x and dy in
// front of the kernel above are a second read and a write of both tensors -- a third of the 64 -> 64 call (profiles/r04_wgrad_segments.md),
// 3x the algorithmic HBM bytes of the family (profiles/r05_hbm_traffic.json).  Here the ring slots hold [64 c][44 pixels] (image columns
// q0 - 4 .. q0 + 39: whole 16-B groups, so a row of a channel is one 176-B run of the LDS-DMA) and the dY slab [64 k][36]; the 11- / 9-unit
// channel pitch is odd in 16-B units, so a lane reads ITS channel's pixels with conflict-free ds_read_b128.  The two reduction
// indices of a 32x32x2 MFMA are pixels t and t + 16 of the tile (lane half = which half of the tile row), so every lane holds a
// contiguous, 16-B aligned window of its row -- 16 + 2 halo pixels of x, 16 of dY -- and the operand of tap s at step t is
// register t + s + 3 of that window: 18 + 4 LDS instructions per 144 MFMAs instead of 160.  The bias gradient is the sum of the
// dY windows (workgroups of channel tile 0), per split, finished by channel_psum_fin_kernel.
// Measured (MI355X, whole call incl. split reduction): 16 x 64 x 256^2 -> 64: 0.86 -> 0.73 ms, 16 x 64 x 128^2 -> 128: 0.40 -> 0.34, 8 x 64 x 256^2: 0.42 -> 0.33
// (116 TF = 0.74 of the fp32 MFMA peak).  Splits ordered strip-fastest (concurrent workgroups covering whole image rows of a channel between
// them) changed nothing: the 128-B row pieces are not what bounds it.
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned wg_u32x4 __attribute__((ext_vector_type(4)));
// (x0, x1) -> three packed bf16 pairs h, m, l with x = h + m + l exactly (both residuals are exact in fp32, the last one fits 8 bits): the
// split of conv_wino.hip's GEMM (11 VALU per pair)
__device__ __forceinline__ void wg_split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  const wg_bf16x2 hp = {(__bf16)x0, (__bf16)x1};
  h = __builtin_bit_cast(unsigned, hp);
  const float r0 = x0 - __uint_as_float(__builtin_amdgcn_perm(h, 0u, 0x05040c0cu)), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  const wg_bf16x2 mp = {(__bf16)r0, (__bf16)r1};
  m = __builtin_bit_cast(unsigned, mp);
  const float q0 = r0 - __uint_as_float(__builtin_amdgcn_perm(m, 0u, 0x05040c0cu)), q1 = r1 - __uint_as_float(m & 0xffff0000u);
  const wg_bf16x2 lp = {(__bf16)q0, (__bf16)q1};
  l = __builtin_bit_cast(unsigned, lp);
}

// SPLIT: the same kernel on the bf16 matrix pipe -- every fp32 operand split exactly into three bf16 parts, six partial products per multiply
// accumulated in fp32 (fp32-equivalent results, as in the Winograd GEMM): 108 v_mfma_f32_32x32x16_bf16 of 8 passes instead of 144 32x32x2 of 16 per
// tile = 0.375x the matrix-pipe time, for ~480 VALU per tile and wave.  A 16-pixel reduction step j pairs pixel u with pixel u + 8 in one packed
// register (lane half h, register m: pixels 16 j + 4 h + m and + 8), so the operand of tap s is four CONSECUTIVE packed registers P[s .. s + 3] of the
// six pairs P[i] = (x[16 j + 4 h + i - 1], x[.. + 8]) -- the three taps of a row share one split, nothing is re-packed.
// STRIDE 2 (the Discriminator's 3x3 / stride-2 layers, SPLIT only): a tile is 16 output pixels of one output row = ONE reduction step; output pixel t
// under tap s reads input column 2 t + s - 1, so with the slot starting at column 2 q0 - 4 the pair of register m is slot columns 8 h + 2 m + s + 3 and
// + 16: nine pairs P[0 .. 8] per row, tap s takes P[s], P[s + 2], P[s + 4], P[s + 6].  Output row p needs input rows 2 p - 1 .. 2 p + 1: two new rows
// per tile, five ring slots (three in use, two in flight).
template <bool SPLIT, int STRIDE>
__global__ __launch_bounds__(256, 2) void conv_wgrad_roll_nchw_kernel(WgradArgs a) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  static_assert(STRIDE == 1 || (STRIDE == 2 && SPLIT), "stride 2 exists on the bf16 pipe only");
  constexpr int TW = STRIDE == 1 ? 32 : 16;
  constexpr int XG = STRIDE == 1 ? 11 : 9, XP = 4 * XG, DG = STRIDE == 1 ? 9 : 5, DP = 4 * DG;
  constexpr int NSLOT = STRIDE == 1 ? 4 : 5;
  constexpr int SLOT = 64 * XP, DYS = 64 * DP;
  __shared__ __attribute__((aligned(16))) float smem[NSLOT * SLOT + 2 * DYS];
  float* ring = smem;
  float* dybuf = smem + NSLOT * SLOT;
  auto slot_of = [](int ih) { return STRIDE == 1 ? ((ih + 1) & 3) : ((ih + 1) % 5); };      // ih >= -1

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wc = wave & 1;
  const int ko0 = (blockIdx.x / a.c_tiles) * 64;
  const int c0 = (blockIdx.x % a.c_tiles) * 64;
  const int split = blockIdx.y;
  const float* zsrc = a.zeros + (lane & 15) * 4;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float dbacc = 0.f;
  const bool want_db = a.db_part != nullptr && c0 == 0 && wc == 0;

  // loop-invariant part of the DMA addresses: unit u = instruction * 64 + lane -> (channel, 16-B group) of the slot image
  int x_off[3], x_iw[3], dy_off[3], dy_q[3];
  bool x_cok[3], dy_ok[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int u = (wave + 4 * j) * 64 + lane;
    const int xc = u / XG, xg = u % XG;
    x_off[j] = (c0 + xc) * a.H * a.W + 4 * xg - 4;
    x_iw[j] = 4 * xg - 4;
    x_cok[j] = (wave + 4 * j) < XG && c0 + xc < a.C;
    const int dk = u / DG, dg = u % DG;
    dy_off[j] = (ko0 + dk) * a.P * a.Q + 4 * dg;
    dy_q[j] = 4 * dg;
    dy_ok[j] = (wave + 4 * j) < DG && ko0 + dk < a.K && dg < DG - 1;       // the last group pads the pitch, never read
  }

  const int tile_beg = split * a.tiles_per_split;
  const int tile_end = min(tile_beg + a.tiles_per_split, a.total_tiles);

#define FCD_RN_DECODE(T, N_, TQ_, P_) \
  const int P_ = (T) % a.P;             \
  const int TQ_ = ((T) / a.P) % a.tiles_q; \
  const int N_ = (T) / (a.P * a.tiles_q);
#define FCD_RN_LOAD_ROW(N_, TQ_, IH)                                                                  \
  {                                                                                                   \
    const int ih_ = (IH);                                                                             \
    float* dst = ring + slot_of(ih_) * SLOT;                                                          \
    const bool rowok = ih_ >= 0 && ih_ < a.H;                                                         \
    const float* base = a.x + ((size_t)(N_) * a.C * a.H + (rowok ? ih_ : 0)) * a.W + (TQ_) * (TW * STRIDE); \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                   \
      const int ins = wave + 4 * j;                                                                   \
      if (ins < XG) {                                                                                 \
        const int iw = (TQ_) * (TW * STRIDE) + x_iw[j];                                               \
        const bool ok = rowok && x_cok[j] && iw >= 0 && iw < a.W;                                     \
        const float* src = ok ? base + x_off[j] : zsrc;                                               \
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(dst + ins * 256), 16, 0, 0); \
      }                                                                                               \
    }                                                                                                 \
  }
#define FCD_RN_LOAD_DY(N_, TQ_, P_, BUF)                                                              \
  {                                                                                                   \
    float* dst = dybuf + (BUF) * DYS;                                                                 \
    const float* base = a.dy + ((size_t)(N_) * a.K * a.P + (P_)) * a.Q + (TQ_) * TW;                 \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                   \
      const int ins = wave + 4 * j;                                                                   \
      if (ins < DG) {                                                                                 \
        const float* src = (dy_ok[j] && (TQ_) * TW + dy_q[j] < a.Q) ? base + dy_off[j] : zsrc;        \
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(dst + ins * 256), 16, 0, 0); \
      }                                                                                               \
    }                                                                                                 \
  }

  if (tile_beg < tile_end) {
    FCD_RN_DECODE(tile_beg, n0, tq0, p0)
    FCD_RN_LOAD_ROW(n0, tq0, p0 * STRIDE - 1)
    FCD_RN_LOAD_ROW(n0, tq0, p0 * STRIDE)
    FCD_RN_LOAD_ROW(n0, tq0, p0 * STRIDE + 1)
    FCD_RN_LOAD_DY(n0, tq0, p0, 0)
  }
  __syncthreads();
  int buf = 0;
  const int a_lane = (wm * 32 + l31) * DP + (SPLIT ? 4 : 16) * half;
  const int b_lane = (wc * 32 + l31) * XP + (SPLIT ? 4 * STRIDE : 16) * half;
  for (int tile = tile_beg; tile < tile_end; ++tile) {
    FCD_RN_DECODE(tile, n, tq, p)
    const bool have_next = tile + 1 < tile_end;
    const bool same_strip = have_next && (p + 1 < a.P);
    if (have_next) {
      FCD_RN_DECODE(tile + 1, nn, tqn, pn)
      FCD_RN_LOAD_DY(nn, tqn, pn, buf ^ 1)
      if (same_strip) {                                     // the new row(s) of the next tile
        FCD_RN_LOAD_ROW(n, tq, (p + 1) * STRIDE + 1)
        if (STRIDE == 2) FCD_RN_LOAD_ROW(n, tq, (p + 1) * STRIDE)
      }
    }
    if (SPLIT) {
      constexpr int NJ = STRIDE == 1 ? 2 : 1;               // 16-pixel reduction steps per tile
      // dY: step j pairs pixels 16 j + 4 h + m and + 8
      wg_u32x4 ah[NJ], am[NJ], al[NJ];
      {
        const float* ap = dybuf + buf * DYS + a_lane;
        f32x4 gq[2 * NJ];
#pragma unroll
        for (int g = 0; g < 2 * NJ; ++g) gq[g] = *(const f32x4*)(ap + 8 * g);
        if (want_db) {
#pragma unroll
          for (int g = 0; g < 2 * NJ; ++g) dbacc += (gq[g][0] + gq[g][1]) + (gq[g][2] + gq[g][3]);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            unsigned h_, m_, l_;
            wg_split_pair(gq[2 * j][m], gq[2 * j + 1][m], h_, m_, l_);
            ah[j][m] = h_; am[j][m] = m_; al[j][m] = l_;
          }
      }
#pragma unroll
      for (int rl = 0; rl < 3; ++rl) {
        const float* bp = ring + slot_of(p * STRIDE - 1 + rl) * SLOT + b_lane;      // input row p * STRIDE - 1 + rl; w[t] = slot column 4 STRIDE h + t
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          constexpr int NWF = STRIDE == 1 ? 20 : 28, NP = STRIDE == 1 ? 6 : 9, HI = STRIDE == 1 ? 8 : 16;
          float w[NWF];
#pragma unroll
          for (int v = 0; v < NWF / 4; ++v) {
            const f32x4 q = *(const f32x4*)(bp + 16 * j + 4 * v);
            w[4 * v] = q[0]; w[4 * v + 1] = q[1]; w[4 * v + 2] = q[2]; w[4 * v + 3] = q[3];
          }
          unsigned ph[NP], pm[NP], pl[NP];
#pragma unroll
          for (int i = 0; i < NP; ++i) wg_split_pair(w[i + 3], w[i + 3 + HI], ph[i], pm[i], pl[i]);
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            const wg_u32x4 bh = {ph[s], ph[s + STRIDE], ph[s + 2 * STRIDE], ph[s + 3 * STRIDE]},
                           bm = {pm[s], pm[s + STRIDE], pm[s + 2 * STRIDE], pm[s + 3 * STRIDE]},
                           bl = {pl[s], pl[s + STRIDE], pl[s + 2 * STRIDE], pl[s + 3 * STRIDE]};
#define FCD_WG_MF(AV, BV) acc[rl * 3 + s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wg_bf16x8, AV), __builtin_bit_cast(wg_bf16x8, BV), acc[rl * 3 + s], 0, 0, 0);
            FCD_WG_MF(al[j], bh) FCD_WG_MF(ah[j], bl) FCD_WG_MF(am[j], bm) FCD_WG_MF(am[j], bh) FCD_WG_MF(ah[j], bm) FCD_WG_MF(ah[j], bh)
#undef FCD_WG_MF
          }
        }
      }
    } else {
      float av[16];
      {
        const float* ap = dybuf + buf * DYS + a_lane;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const f32x4 q = *(const f32x4*)(ap + 4 * v);
          av[4 * v] = q[0]; av[4 * v + 1] = q[1]; av[4 * v + 2] = q[2]; av[4 * v + 3] = q[3];
        }
      }
      if (want_db) {
#pragma unroll
        for (int t = 0; t < 16; ++t) dbacc += av[t];
      }
#pragma unroll
      for (int rl = 0; rl < 3; ++rl) {
        const float* bp = ring + slot_of(p - 1 + rl) * SLOT + b_lane;      // input row p - 1 + rl
        float w[24];
#pragma unroll
        for (int v = 0; v < 6; ++v) {
          const f32x4 q = *(const f32x4*)(bp + 4 * v);
          w[4 * v] = q[0]; w[4 * v + 1] = q[1]; w[4 * v + 2] = q[2]; w[4 * v + 3] = q[3];
        }
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
          for (int s = 0; s < 3; ++s)
            acc[rl * 3 + s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], w[t + s + 3], acc[rl * 3 + s], 0, 0, 0);
      }
    }
    __syncthreads();
    if (have_next && !same_strip) {      // strip change: (re)load the three rows of the new strip
      FCD_RN_DECODE(tile + 1, nn, tqn, pn)
      FCD_RN_LOAD_ROW(nn, tqn, pn * STRIDE - 1)
      FCD_RN_LOAD_ROW(nn, tqn, pn * STRIDE)
      FCD_RN_LOAD_ROW(nn, tqn, pn * STRIDE + 1)
      __syncthreads();
    }
    buf ^= 1;
  }
#undef FCD_RN_DECODE
#undef FCD_RN_LOAD_ROW
#undef FCD_RN_LOAD_DY

  if (want_db) {
    const float v = dbacc + __shfl_xor(dbacc, 32, 64);
    const int ko = ko0 + wm * 32 + l31;
    if (half == 0) a.db_part[(size_t)split * a.Kp + ko] = ko < a.K ? v : 0.f;
  }
  float* out = a.out + (size_t)split * a.split_stride;
  const int c = c0 + wc * 32 + l31;
  if (c < a.C) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int ko = ko0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
        if (ko < a.K) out[a.tkc ? ((size_t)t * a.K + ko) * a.C + c : (((size_t)ko * a.C + c) * 3 + t / 3) * 3 + t % 3] = acc[t][reg];
      }
    }
  }
}

// dw = sum over split-K partials (fixed order => deterministic).  float4 streams, four independent
// loads in flight per partial; n4 = n / 4 vectors, the (n % 4) tail is handled by the last threads.
// (kc > 0: the partials are laid out [tap][K][C], kc = K * C, rs = taps: element e goes to dw[(e % kc) * rs + e / kc])
__device__ __forceinline__ long long wgrad_dst(long long e, long long kc, int rs) { return kc > 0 ? (e % kc) * rs + e / kc : e; }

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                           long long n, int splits, long long kc = 0, int rs = 1) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  // every partial starts 16-B aligned when n % 4 == 0 and the bases are
  const bool aligned = (n & 3) == 0 && (((size_t)part | (size_t)dw) & 15) == 0;
  if (aligned) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
      v4 s = *(const v4*)(part + 4 * i);
      for (int k = 1; k < splits; ++k) s += *(const v4*)(part + (long long)k * n + 4 * i);
      if (kc > 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dw[wgrad_dst(4 * i + e, kc, rs)] = s[e];
      } else {
        *(v4*)(dw + 4 * i) = s;
      }
    }
    return;
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += part[(long long)k * n + i];
    dw[wgrad_dst(i, kc, rs)] = s;
  }
}

// Many partials, few elements (small filters are split up to 1024 ways): 64 float4 columns per
// block, the partials dealt round-robin to 4 groups of 64 threads (4 loads in flight each), group
// sums combined in fixed order through LDS => deterministic.
__global__ __launch_bounds__(256) void wgrad_reduce_wide_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                long long n, int splits, long long kc = 0, int rs = 1) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  __shared__ v4 red[4][64];
  const int tx = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long long i = (long long)blockIdx.x * 64 + tx;
  const long long n4 = n >> 2;
  v4 s = {0.f, 0.f, 0.f, 0.f};
  if (i < n4) {
    const float* p = part + 4 * i;
    int k = g;
    for (; k + 12 < splits; k += 16) {
      const v4 a0 = *(const v4*)(p + (long long)k * n), a1 = *(const v4*)(p + (long long)(k + 4) * n),
               a2 = *(const v4*)(p + (long long)(k + 8) * n), a3 = *(const v4*)(p + (long long)(k + 12) * n);
      s += a0; s += a1; s += a2; s += a3;
    }
    for (; k < splits; k += 4) s += *(const v4*)(p + (long long)k * n);
  }
  red[g][tx] = s;
  __syncthreads();
  if (g == 0 && i < n4) {
    v4 t = red[0][tx];
    t += red[1][tx]; t += red[2][tx]; t += red[3][tx];
    if (kc > 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) dw[wgrad_dst(4 * i + e, kc, rs)] = t[e];
    } else {
      *(v4*)(dw + 4 * i) = t;
    }
  }
}

// out[c] = sum_{n,i} x[n,c,i] (* [mask > 0]); grid (C, N-splits): per-block partial via fp64 atomics-free
// two-stage reduce (partials in ws, finalized by the last kernel)
__global__ __launch_bounds__(256) void channel_sum_part_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ mask,
                                                               double* __restrict__ part, int N, int C, int HW,
                                                               int nsplit) {
  __shared__ double red[16];
  const int c = blockIdx.x, sp = blockIdx.y;
  const long long total = (long long)N * HW;
  const long long chunk = (total + nsplit - 1) / nsplit;
  const long long beg = sp * chunk, end = min(beg + chunk, total);
  double s = 0.0;
  for (long long e = beg + threadIdx.x; e < end; e += 256) {
    const int n = (int)(e / HW), i = (int)(e % HW);
    const size_t off = ((size_t)n * C + c) * HW + i;
    float v = x[off];
    if (mask && !(mask[off] > 0.f)) v = 0.f;
    s += (double)v;
  }
  s = block_sum_d(s, red);
  if (threadIdx.x == 0) part[(size_t)c * nsplit + sp] = s;
}

__global__ void channel_sum_fin_kernel(const double* __restrict__ part, float* __restrict__ out, int C, int nsplit) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int i = 0; i < nsplit; ++i) s += part[(size_t)c * nsplit + i];
  out[c] = (float)s;
}

#define CS_MAX_SPLIT 64
extern "C" size_t fcd_channel_sum_ws_bytes(int C) { return (size_t)C * CS_MAX_SPLIT * sizeof(double); }

extern "C" int fcd_channel_sum(const float* x, const float* relu_out, float* out, int N, int C, int HW, void* ws,
                               size_t ws_bytes, void* stream) {
  FCD_CHECK_ARG(x && out && N > 0 && C > 0 && HW > 0, "fcd_channel_sum: bad arguments");
  if (!ws || ws_bytes < fcd_channel_sum_ws_bytes(C)) {
    fcd_set_error("fcd_channel_sum: workspace too small");
    return FCD_ERR_WORKSPACE;
  }
  int nsplit = cdiv(1024, C);
  const long long maxs = std::max<long long>(1, (long long)N * HW / 1024);
  if (nsplit > maxs) nsplit = (int)maxs;
  if (nsplit > CS_MAX_SPLIT) nsplit = CS_MAX_SPLIT;
  hipStream_t st = (hipStream_t)stream;
  FcdProfScope prof(FCD_K_MISC, st, 0.0, 4.0 * N * C * (double)HW);
  hipLaunchKernelGGL(channel_sum_part_kernel, dim3(C, nsplit), dim3(256), 0, st, x, relu_out, (double*)ws, N, C, HW,
                     nsplit);
  hipLaunchKernelGGL(channel_sum_fin_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, (const double*)ws, out, C, nsplit);
  FCD_LAUNCH_CHECK("channel_sum");
  return FCD_OK;
}

// ---------------------------------------------------------------------------
struct WgradPlan {
  int TW, RB, tiles_q, total_tiles, tiles_per_split, splits, k_tiles, c_tiles, r_groups, Cp, Kp;
  size_t xt_bytes, dyt_bytes, part_bytes, zero_bytes, psum_bytes;
};

// Workgroups the split count aims at (FCD_WGRAD_WGS for A/B).  [r4] 512 = ONE round of the 512 resident slots (two 59-KB workgroups per CU):
// every workgroup pays a prologue and a 147-KB partial store, and every split adds a pass to the reduction -- 1024 (rounds 1-3) measured
// 0.30 / 0.25 / 0.23 ms on the Discriminator's stride-2 layers against 0.27 / 0.23 / 0.21 with 512, 2048: 0.36 / 0.33 / 0.29.
static int wgrad_target_wgs() { return fcd_sw(FCD_SW_WGRAD_WGS); }

static bool wgrad_plan(const fcd_conv_desc* d, WgradPlan* pl) {
  const int R = d->R, S = d->S, st = d->stride;
  if (R == 3 && S == 3 && st == 1) { pl->TW = 32; pl->RB = 3; }
  else if (R == 3 && S == 3 && st == 2) { pl->TW = 16; pl->RB = 3; }
  else if (R == 9 && S == 9 && st == 1) { pl->TW = 32; pl->RB = 1; }
  else if (R == 1 && S == 1 && st == 1) { pl->TW = 32; pl->RB = 1; }
  else if (R == 2 && S == 2 && st == 2) { pl->TW = 16; pl->RB = 2; }
  else return false;
  if (d->Q <= 16) pl->TW = 16;
  pl->tiles_q = cdiv(d->Q, pl->TW);
  pl->total_tiles = d->N * d->P * pl->tiles_q;
  pl->k_tiles = cdiv(d->K, 64);
  pl->c_tiles = cdiv(d->C, 64);
  pl->r_groups = R / pl->RB;
  pl->Cp = pl->c_tiles * 64;
  pl->Kp = pl->k_tiles * 64;
  const int base = pl->k_tiles * pl->c_tiles * pl->r_groups;
  int splits = cdiv(wgrad_target_wgs(), base);
  const long long dw_bytes = 4LL * d->K * d->C * R * S;
  const long long cap = std::max<long long>(1, (512LL << 20) / dw_bytes);
  if (splits > cap) splits = (int)cap;
  if (splits > pl->total_tiles) splits = pl->total_tiles;
  if (splits < 1) splits = 1;
  pl->tiles_per_split = cdiv(pl->total_tiles, splits);
  pl->splits = cdiv(pl->total_tiles, pl->tiles_per_split);
  pl->xt_bytes = (size_t)d->N * d->H * d->W * pl->Cp * sizeof(float);
  pl->dyt_bytes = (size_t)d->N * d->P * d->Q * pl->Kp * sizeof(float);
  pl->part_bytes = pl->splits > 1 ? (size_t)pl->splits * dw_bytes : 0;
  pl->zero_bytes = 1024;
  // bias gradient: per-block channel partials of the dy re-layout (float4 path), else the
  // two-stage channel_sum workspace
  pl->psum_bytes = std::max(std::max((size_t)d->N * cdiv(d->P * d->Q, 64) * pl->Kp * sizeof(float),
                                     (size_t)d->K * 64 * sizeof(double)),
                            (size_t)pl->splits * pl->Kp * sizeof(float));      // [r5] NCHW-direct kernel: [split][Kp]
  return true;
}

// conv_wino.hip: Winograd F(4x4, 3x3) weight gradient for the wide 3x3 / stride-1 layers
size_t fcd_wino_wgrad_ws_bytes(const fcd_conv_desc* d);
int fcd_wino_wgrad_run(const fcd_conv_desc* d, const float* x, const float* dy, const float* relu_out, float* dw,
                       float* db, void* ws, hipStream_t st);

// conv_wgrad_thin.hip: reduction-contiguous kernel for the first layers (C x 9 < 128 columns, <= 64 filters)
int fcd_wgrad_thin_plan(const fcd_conv_desc* d);
size_t fcd_wgrad_thin_ws_bytes(const fcd_conv_desc* d);
int fcd_wgrad_thin_run(const fcd_conv_desc* d, const float* x, const float* dy, const float* relu_out, float* dw, float* db,
                       void* ws, hipStream_t st);

// conv_wgrad_thin9.hip [r4]: the 9x9 layers with <= 4 channels on one side (the Generator's first / last convolution on 3- / 4-band data)
int fcd_wgrad_thin9_plan(const fcd_conv_desc* d);
size_t fcd_wgrad_thin9_ws_bytes(const fcd_conv_desc* d);
int fcd_wgrad_thin9_run(const fcd_conv_desc* d, const float* x, const float* dy, const float* relu_out, float* dw, float* db,
                        void* ws, hipStream_t st);

extern "C" size_t fcd_conv2d_bwd_weight_ws_bytes(const fcd_conv_desc* d) {
  WgradPlan pl;
  if (!d || !wgrad_plan(d, &pl)) return 0;
  return std::max(std::max(std::max(pl.zero_bytes + pl.xt_bytes + pl.dyt_bytes + pl.part_bytes + pl.psum_bytes,
                                    fcd_wino_wgrad_ws_bytes(d)), fcd_wgrad_thin_ws_bytes(d)), fcd_wgrad_thin9_ws_bytes(d));
}

static int wgrad_tkc() { return fcd_sw(FCD_SW_WGRAD_TKC); }      // 0: split partials in dw's own layout (round-3 behaviour)

static int wgrad_roll() { return fcd_sw(FCD_SW_WGRAD_ROLL); }

extern "C" int fcd_conv_wgrad_split_set(int on) {
  const int old = fcd_sw(FCD_SW_WGRAD_SPLIT);
  if (on >= 0) g_fcd_switch[FCD_SW_WGRAD_SPLIT] = on ? 1 : 0;
  return old;
}

// WGRAD_NCHW=0: channel-minor copies + conv_wgrad_roll_kernel for every 3x3 / stride-1 layer (rounds 1-4)
static bool wgrad_nchw_ok(const fcd_conv_desc* d, const WgradPlan& pl, const float* x, const float* dy, const float* relu_out) {
  if (!fcd_sw(FCD_SW_WGRAD_NCHW)) return false;
  const bool split = fcd_sw(FCD_SW_WGRAD_SPLIT) != 0;
  const bool geom = d->stride == 1 ? (pl.TW == 32 && (d->W % 32) == 0)
                                   : (d->stride == 2 && split && pl.TW == 16 && (d->W % 8) == 0 && d->Q == d->W / 2);     // [r5] stride 2: bf16 pipe only
  return d->R == 3 && d->S == 3 && d->pad == 1 && geom && relu_out == nullptr &&
         ((((size_t)x) | ((size_t)dy)) & 15) == 0 && (long long)pl.Cp * d->H * d->W < (1LL << 31) &&
         (long long)pl.Kp * d->P * d->Q < (1LL << 31);
}

template <int R, int S, int RB, int STRIDE, int TW>
static void launch_wgrad(const WgradArgs& a, const WgradPlan& pl, hipStream_t st) {
  dim3 grid((unsigned)(pl.k_tiles * pl.c_tiles), (unsigned)pl.splits, (unsigned)pl.r_groups);
#if WG_TIME
  WgradArgs b = a;
  b.tbuf = g_wg_tbuf;
  hipLaunchKernelGGL((conv_wgrad_kernel<R, S, RB, STRIDE, TW>), grid, dim3(256), 0, st, b);
#else
  hipLaunchKernelGGL((conv_wgrad_kernel<R, S, RB, STRIDE, TW>), grid, dim3(256), 0, st, a);
#endif
}

extern "C" int fcd_conv2d_bwd_weight_bias(const fcd_conv_desc* d, const float* x, const float* dy,
                                          const float* relu_out, float* dw, float* db, void* ws, size_t ws_bytes,
                                          void* stream);

extern "C" int fcd_conv2d_bwd_weight(const fcd_conv_desc* d, const float* x, const float* dy, const float* relu_out,
                                     float* dw, void* ws, size_t ws_bytes, void* stream) {
  return fcd_conv2d_bwd_weight_bias(d, x, dy, relu_out, dw, nullptr, ws, ws_bytes, stream);
}

extern "C" int fcd_conv2d_bwd_weight_bias(const fcd_conv_desc* d, const float* x, const float* dy,
                                          const float* relu_out, float* dw, float* db, void* ws, size_t ws_bytes,
                                          void* stream) {
  FCD_CHECK_ARG(d && x && dy && dw, "fcd_conv2d_bwd_weight: null pointer");
  WgradPlan pl;
  FCD_CHECK_ARG(wgrad_plan(d, &pl), "fcd_conv2d_bwd_weight: unsupported filter %dx%d stride %d", d->R, d->S,
                d->stride);
  const size_t need = fcd_conv2d_bwd_weight_ws_bytes(d);
  if (ws == nullptr || ws_bytes < need) {
    fcd_set_error("fcd_conv2d_bwd_weight: workspace %zu < %zu bytes", ws_bytes, need);
    return FCD_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  char* wsp = (char*)ws;
  float* zeros = (float*)wsp; wsp += pl.zero_bytes;
  float* xt = (float*)wsp;    wsp += pl.xt_bytes;
  float* dyt = (float*)wsp;   wsp += pl.dyt_bytes;
  float* part = (float*)wsp;  wsp += pl.part_bytes;
  float* psum = (float*)wsp;
  const double flops = 2.0 * d->N * d->K * (double)d->P * d->Q * d->C * d->R * d->S;
  const double bytes = 4.0 * ((double)d->N * d->C * d->H * d->W + (double)d->N * d->K * d->P * d->Q +
                              (double)d->K * d->C * d->R * d->S);
  FcdProfScope prof(FCD_K_CONV_WGRAD, st, flops, bytes, fcd_prof_tag_desc("wgrad", d));
  if (fcd_wino_wgrad_ws_bytes(d) > 0 && fcd_wino_wgrad_run(d, x, dy, relu_out, dw, db, ws, st) == 0) {
    FCD_LAUNCH_CHECK("conv2d_bwd_weight(winograd)");
    return FCD_OK;
  }
  if (fcd_wgrad_thin9_plan(d)) {
    if (fcd_wgrad_thin9_run(d, x, dy, relu_out, dw, db, ws, st) != 0) {
      fcd_set_error("fcd_conv2d_bwd_weight: 9x9 thin-channel kernel launch failed");
      return FCD_ERR_LAUNCH;
    }
    return FCD_OK;
  }
  if (fcd_wgrad_thin_plan(d)) {
    if (fcd_wgrad_thin_run(d, x, dy, relu_out, dw, db, ws, st) != 0) {
      fcd_set_error("fcd_conv2d_bwd_weight: thin-channel kernel launch failed");
      return FCD_ERR_LAUNCH;
    }
    return FCD_OK;
  }
  if (hipMemsetAsync(zeros, 0, pl.zero_bytes, st) != hipSuccess) {
    fcd_set_error("fcd_conv2d_bwd_weight: memset failed");
    return FCD_ERR_LAUNCH;
  }
  const bool nchw = wgrad_roll() && wgrad_nchw_ok(d, pl, x, dy, relu_out);
  if (!nchw) {
    const int HW = d->H * d->W;
    launch_transpose(x, nullptr, xt, d->N, d->C, HW, pl.Cp, st);
    const int PQ = d->P * d->Q;
    const bool fused_db = db != nullptr && (PQ & 3) == 0;
    launch_transpose(dy, relu_out, dyt, d->N, d->K, PQ, pl.Kp, st, fused_db ? psum : nullptr);
    if (fused_db) {
      hipLaunchKernelGGL(channel_psum_fin_kernel, dim3(d->K), dim3(256), 0, st, (const float*)psum, db, d->K, pl.Kp,
                         d->N * cdiv(PQ, 64));
    } else if (db != nullptr) {
      int nsplit = std::min(std::min<long long>(cdiv(1024, d->K), std::max<long long>(1, (long long)d->N * PQ / 1024)),
                            (long long)64);
      hipLaunchKernelGGL(channel_sum_part_kernel, dim3(d->K, nsplit), dim3(256), 0, st, dy, relu_out, (double*)psum,
                         d->N, d->K, PQ, nsplit);
      hipLaunchKernelGGL(channel_sum_fin_kernel, dim3(cdiv(d->K, 128)), dim3(128), 0, st, (const double*)psum, db,
                         d->K, nsplit);
    }
  }
  WgradArgs a;
  memset(&a, 0, sizeof(a));
  a.xt = xt; a.dyt = dyt; a.zeros = zeros;
  a.out = pl.splits > 1 ? part : dw;
  a.N = d->N; a.C = d->C; a.H = d->H; a.W = d->W; a.K = d->K; a.P = d->P; a.Q = d->Q; a.pad = d->pad;
  a.Cp = pl.Cp; a.Kp = pl.Kp;
  a.tiles_q = pl.tiles_q; a.total_tiles = pl.total_tiles;
  a.tiles_per_split = pl.tiles_per_split; a.c_tiles = pl.c_tiles;
  a.split_stride = (long long)d->K * d->C * d->R * d->S;
  a.tkc = (pl.splits > 1 && wgrad_tkc()) ? 1 : 0;
  const int R = d->R, S = d->S, sd = d->stride;
  const bool narrow = pl.TW == 16;
  if (nchw) {
    a.x = x; a.dy = dy; a.db_part = db ? psum : nullptr;
    dim3 grid((unsigned)(pl.k_tiles * pl.c_tiles), (unsigned)pl.splits, 1);
    const bool bf16_pipe = d->stride == 2 || fcd_sw(FCD_SW_WGRAD_SPLIT);      // WGRAD_SPLIT=0: the fp32 matrix pipe (A/B, tests, bench.py fp32_mfma_only)
    {
      // nested in the call's FCD_K_CONV_WGRAD scope: the launch on the bf16 pipe, so that a report can price it against that peak
      std::optional<FcdProfScope> ps;
      if (bf16_pipe) ps.emplace(FCD_K_WGRAD_SPLIT, st, flops, bytes);
      if (d->stride == 2) hipLaunchKernelGGL((conv_wgrad_roll_nchw_kernel<true, 2>), grid, dim3(256), 0, st, a);
      else if (!bf16_pipe) hipLaunchKernelGGL((conv_wgrad_roll_nchw_kernel<false, 1>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((conv_wgrad_roll_nchw_kernel<true, 1>), grid, dim3(256), 0, st, a);
    }
    if (db) hipLaunchKernelGGL(channel_psum_fin_kernel, dim3(d->K), dim3(256), 0, st, (const float*)psum, db, d->K, pl.Kp, pl.splits);
  } else if (R == 3 && S == 3 && sd == 1 && d->pad == 1 && wgrad_roll()) {
    // p-fastest tile order + rolling 4-row ring (see conv_wgrad_roll_kernel)
    dim3 grid((unsigned)(pl.k_tiles * pl.c_tiles), (unsigned)pl.splits, 1);
#if WG_TIME
    a.tbuf = g_wg_tbuf;
#endif
    if (narrow) hipLaunchKernelGGL(conv_wgrad_roll_kernel<16>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(conv_wgrad_roll_kernel<32>, grid, dim3(256), 0, st, a);
  } else if (R == 3 && S == 3 && sd == 1) {
    if (narrow) launch_wgrad<3, 3, 3, 1, 16>(a, pl, st); else launch_wgrad<3, 3, 3, 1, 32>(a, pl, st);
  } else if (R == 3 && S == 3 && sd == 2) {
    launch_wgrad<3, 3, 3, 2, 16>(a, pl, st);
  } else if (R == 9 && S == 9) {
    if (narrow) launch_wgrad<9, 9, 1, 1, 16>(a, pl, st); else launch_wgrad<9, 9, 1, 1, 32>(a, pl, st);
  } else if (R == 1 && S == 1) {
    if (narrow) launch_wgrad<1, 1, 1, 1, 16>(a, pl, st); else launch_wgrad<1, 1, 1, 1, 32>(a, pl, st);
  } else {
    launch_wgrad<2, 2, 2, 2, 16>(a, pl, st);
  }
  FCD_LAUNCH_CHECK("conv2d_bwd_weight");
  if (pl.splits > 1) {
    const long long n = a.split_stride;
    if (pl.splits >= 8 && (n & 3) == 0 && (((size_t)part | (size_t)dw) & 15) == 0) {
      hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3((unsigned)cdiv64(n >> 2, 64)), dim3(256), 0, st,
                         (const float*)part, dw, n, pl.splits, a.tkc ? (long long)d->K * d->C : 0LL, d->R * d->S);
    } else {
      const int grid = (int)std::min<long long>(cdiv64(cdiv64(n, 4), 256), 4096);
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid), dim3(256), 0, st, (const float*)part, dw, n, pl.splits,
                         a.tkc ? (long long)d->K * d->C : 0LL, d->R * d->S);
    }
    FCD_LAUNCH_CHECK("wgrad_reduce");
  }
  return FCD_OK;
}
