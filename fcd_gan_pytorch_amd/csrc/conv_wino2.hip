// Fused Winograd F(2x2, 3x3) convolution for the 64-row layers (3x3 / stride 1 / pad 1 with <= 64 GEMM rows: VGG
// conv1_2 forward + data gradient, the data gradient of conv2_1, the Generator's eleven 64 -> 64 layers, the second
// conv of the Segmentor's first stage).  For these the three-kernel F(4x4) form loses -- with only 64 rows the
// transformed activations V / products M (2.25x the tensors each) cost more HBM time than the 4x smaller GEMM saves
// (measured: 10.8 ms vs 8.3 ms direct on conv1_2) -- and the direct kernel sits at 0.79 of the fp32 MFMA peak with
// nothing left but its FLOP count.  Here ONE kernel does input transform, the 16 batched GEMMs and the output
// transform, so V and M never exist in memory: the FLOP count drops 2.25x (16 multiplies per 2x2 output tile and
// channel pair instead of 36) at the direct kernel's HBM traffic (x read once + halo, y written once).
//
//   y_tile = A^T [ sum_c (G g_kc G^T) .* (B^T d_c B) ] A        4x4 patch d -> 2x2 outputs, 16 positions xi
//
// Why this shape works where a fused F(4x4) does not: with v_mfma_f32_16x16x4_f32 a lane's B operand is ONE float
// = V_xi[channel kc][tile n] and its A operand ONE float = U_xi[row][channel kc].  A lane therefore needs, per
// 4-channel chunk, all 16 xi of ITS (tile, channel): exactly the B^T d B of the 4x4 patch it reads from LDS with eight
// ds_read_b64 -- 32 VALU adds for 16 MFMA operands (2 per MFMA, 0.5 LDS instructions per MFMA), no cross-lane
// traffic.  The 16 accumulators of a (row, tile) end up in ONE lane (C/D layout: col = tile, 4 rows per lane), so the
// output transform is 24 in-register adds, and a 2x2 tile is exactly one MaxPool2d(2) window: the pooled epilogue needs
// no shuffles.
//
// Workgroup = 512 threads = 8 waves on an 8 x 32 pixel block (4 tile rows x 16 tiles), all 64 rows:
//   wave w: rows 16 (w & 3) .. +15, tile rows 2 (w >> 2), 2 (w >> 2) + 1  -> 16 xi x 2 accumulators of 4 VGPRs = 128.
// K loop over chunks of 4 reduction channels: transformed filters U2[chunk][row group][lane][16 xi (+4 pad)] arrive by
// global_load_lds (20 KiB per chunk, lane-linear, 80-B lane pitch => conflict-free ds_read_b128), the 10 x 34 input
// patch by dword loads issued one chunk ahead and stored to LDS after the MFMA block (source gating -- ReLU mask or
// pooled-gradient routing -- applied at store time, as in conv_igemm.hip).
#include "common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

int fcd_wino_mode_now();   // conv_wino.hip: 0 = direct kernels only (tests' A/B switch), 2 / 4 otherwise

// [r5] The wrong-result diagnostic switches (W2_EXP: no memory instruction in the loop, no barrier, operands from registers, one
// patch buffer ...) and the W2_LATE / W2_DEEP A/B switches live in csrc/lab/conv_wino2.hip (`make lab`), not in this source.
// W2_TIME: attribution build (tools/w2_segments.py): every wave stamps s_memtime at the segment borders of its
// stage loop and writes the per-segment cycle sums {prologue, issue (filter DMA + patch loads), operands + transforms + MFMA issue,
// LDS commit of the next patch (waits for its global loads), barrier, epilogue, total} to a debug buffer.  Results stay correct;
// the stamps cost ~10 % (each waits for the wave's outstanding LDS reads).  Never defined in the product build.
#ifndef W2_TIME
#define W2_TIME 0
#endif
#if W2_TIME
// (scheduling barriers on both sides: the stamp is a scalar instruction with no dependences, and without them the compiler
//  floats it 25 MFMAs up into the block it is meant to close)
#define W2_T(var) __builtin_amdgcn_sched_barrier(0); const unsigned long long var = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#define W2_TACC(slot, t1, t0) tacc[slot] += (t1) - (t0);
#else
#define W2_T(var)
#define W2_TACC(slot, t1, t0)
#endif
namespace {
constexpr int W2_ROWS = 64;                 // GEMM rows per workgroup (all of them)
constexpr int W2_LP = 12;                   // floats per lane row in the filter slab: the ROW-transformed filter G g (4 x 3),
                                            // 48-B pitch (conflict-free ds_read_b128); the column transform (.) G^T runs in the kernel
constexpr int W2_SLAB = 4 * 64 * W2_LP;     // floats per chunk: [row group][lane][20]
constexpr int W2_TW = 32;                   // output columns per workgroup (rows: 4 per 4 waves)
constexpr int W2_PW = W2_TW + 2;
// LDS image of the input patch: PAIR rows.  Element (r, c) of pair row r holds (x[r][c], x[r + 2][c]) in two adjacent
// floats: a wave's two tile rows (patch rows 4h + i and 4h + 2 + i, i = 0..3) then arrive as ready-made (tile row 0,
// tile row 1) register pairs, four columns per two ds_read_b128, and every transform add is one v_pk_add_f32 serving
// two MFMA operands.  (Each patch element is stored twice: as .x of pair row r and as .y of pair row r - 2.)
constexpr int W2_RP = 72;                   // pair-row pitch (floats): 34 columns x 2, padded to a multiple of 4
// Workgroup shapes (template NW, KS; FCD_WINO2_WAVES / FCD_WINO2_KS select them for A/B runs):
//   8 waves, 8 x 32 pixels, 8 channels per pipeline stage (KS = 2)  <- default: 6.2 ms on VGG conv1_2 (N = 208) vs 8.2 direct
//   8 waves, 4 channels per stage (KS = 1): 6.8 ms (twice the barriers)
//   4 waves, 4 x 32 pixels (two or three independent workgroups per CU): 9.1 - 10.2 ms -- half the pixels per filter
//   fetch and 1.5x halo rows; kept for the record
//   4 waves x 4 tile rows each, 8 x 32 pixels (FCD_WINO2_WAVES=1: one wave per SIMD, 256 VGPRs + 256 AGPRs): 8.9 ms as
//   the compiler schedules it -- a single wave per SIMD needs a hand-pipelined operand prefetch to hide its LDS reads
// Where the 6.2 ms go (diagnostic builds, lab/conv_wino2.hip): pure MFMA floor 2.85 ms; MFMA + transforms + per-workgroup prologue /
// epilogue with NO memory instruction in the loop 4.0 ms (one workgroup per CU: nothing hides the prologue's memory
// round trip and the epilogue's stores); + LDS operand reads 0.9, + filter DMA / patch loads / LDS stores 0.9, + barrier
// 0.3.  PMC: 47 % MFMA-busy, 32 % of wave cycles parked at s_waitcnt / s_barrier (profiles/r02_pmc_wino2.md).

struct Wino2Args {
  const float* x;       // source (N, C, H, W) -- or the pooled gradient (N, C, Hp, Wp) when SRC == 2
  const float* U;       // packed transformed filters
  const float* bias;
  const float* mask;    // SRC == 1: same shape as x, source is read as x * (mask > 0)
  const unsigned char* code_in;   // SRC == 2
  float* y;             // (N, K, P, Q)   (EPI == 0)
  float* pool_y;        // (N, K, P/2, Q/2) (EPI == 1)
  unsigned char* code_out;
  const float* residual;
  const float* slope_ptr;
  float slope_imm;
  int relu, act_slope;
  int N, C, H, W, K, Hp, Wp, nchunks, tiles_p, tiles_q, xcd_remap;
  unsigned long long* tbuf;   // W2_TIME builds: [workgroup][wave][8] cycle sums
};

// Row-transformed filters G g (4 x 3 per (row, channel)), G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]: 12 floats instead of the 16
// of G g G^T -- the kernel is bound by the filter traffic from L2 (LDS-DMA ~9 B/cycle/CU), so 40 % fewer bytes beat 0.5 extra
// VALU operations per MFMA
__global__ void wino2_pack_kernel(const float* __restrict__ w, float* __restrict__ U, int K, int C, int mode,
                                  long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int e = (int)(i % W2_LP);
    long long t = i / W2_LP;
    const int l = (int)(t % 64); t /= 64;
    const int g = (int)(t % 4);
    const int q = (int)(t / 4);
    const int row = 16 * g + (l & 15), red = 4 * q + (l >> 4);
    const int rows = mode == 0 ? K : C, reds = mode == 0 ? C : K;
    float v = 0.f;
    if (row < rows && red < reds) {
      // element e = 3 a + s of the half-transformed filter: T[a][s] = sum_r G[a][r] g[r][s]; the middle column is stored
      // halved (the kernel forms u1 = m + h, u2 = m - h with m = (T[a][0] + T[a][2]) / 2, h = T[a][1] / 2)
      const int a = e / 3, sc = e % 3;
      float g0, g1, g2;
      if (mode == 0) {
        g0 = w[(((size_t)row * C + red) * 3 + 0) * 3 + sc];
        g1 = w[(((size_t)row * C + red) * 3 + 1) * 3 + sc];
        g2 = w[(((size_t)row * C + red) * 3 + 2) * 3 + sc];
      } else {
        g0 = w[(((size_t)red * C + row) * 3 + 2) * 3 + (2 - sc)];
        g1 = w[(((size_t)red * C + row) * 3 + 1) * 3 + (2 - sc)];
        g2 = w[(((size_t)red * C + row) * 3 + 0) * 3 + (2 - sc)];
      }
      v = a == 0 ? g0 : (a == 1 ? 0.5f * (g0 + g1 + g2) : (a == 2 ? 0.5f * (g0 - g1 + g2) : g2));
      if (sc == 1) v *= 0.5f;
    }
    U[i] = v;
  }
}

template <int SRC, int EPI, int NW, int W2_KS, int PG>     // PG: pairs of tile rows per wave (2 PG accumulators per xi)
__global__ __launch_bounds__(64 * NW) void conv_wino2_kernel(Wino2Args a) {
  constexpr int NT = 64 * NW;                         // threads
  constexpr int W2_CB = 4 * W2_KS;                    // channels per pipeline stage
  constexpr int W2_TH = NW * PG;                      // output rows per workgroup: 2 PG tile rows per 4 waves
  constexpr int W2_PH = W2_TH + 2;
  // pair rows per channel plane, padded to a multiple of 64 floats: the four channel groups of a wave (lanes 16 kc ..)
  // must start on the same bank, or the 16-lane groups in which ds_read_b128 is served (they mix lanes of two channel
  // groups) collide two ways on every operand read
  constexpr int W2_PL = (W2_TH * W2_RP + 63) / 64 * 64;
  constexpr int X_ELEMS = W2_CB * W2_PH * W2_PW;
  constexpr int X_PER_T = (X_ELEMS + NT - 1) / NT;
  constexpr int XS_SZ = W2_CB * W2_PL;
  constexpr int U_STAGE = W2_KS * W2_SLAB;            // floats per stage of the filter pipeline
  constexpr int U_INSTR = U_STAGE / 256;              // wave-instructions of 1 KiB per stage
  constexpr int U_PER_W = (U_INSTR + NW - 1) / NW;
  __shared__ __attribute__((aligned(16))) float su0[U_STAGE];
  __shared__ __attribute__((aligned(16))) float su1[U_STAGE];
  __shared__ __attribute__((aligned(16))) float sx[2 * XS_SZ];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = wave & 3, hrow = wave >> 2;
  const int ln = lane & 15, kc = lane >> 4;
#if W2_TIME
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  W2_T(t_begin)

  unsigned v;
  {
    const unsigned total = gridDim.x, b = blockIdx.x;
    if (a.xcd_remap) {
      const unsigned q8 = total >> 3, r8 = total & 7u, xcd = b & 7u;
      v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
    } else {
      v = b;
    }
  }
  int bx = (int)v;
  const int tq = bx % a.tiles_q;
  bx /= a.tiles_q;
  const int tp = bx % a.tiles_p;
  const int n = bx / a.tiles_p;
  const int p0 = tp * W2_TH, q0 = tq * W2_TW;

  f32x4 acc[16][2 * PG];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi)
#pragma unroll
    for (int gr = 0; gr < 2 * PG; ++gr) acc[xi][gr] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- loop-invariant addresses
  const int aoff = (g * 64 + lane) * W2_LP;                         // A operands: 16 consecutive floats
  int boff[PG];
#pragma unroll
  for (int pg = 0; pg < PG; ++pg) boff[pg] = kc * W2_PL + (4 * (hrow * PG + pg)) * W2_RP + 4 * ln;   // pair rows of group pg

  // DEEP [r3]: plain sources keep TWO chunks of patch loads in flight (the chunk two stages ahead is requested while the next
  // one is still on its way): a stage lasts ~3.6 us, an HBM round trip under load is not much shorter.  Costs X_PER_T registers.
  // [r4] The pooled-gradient source (SRC == 2) stages UNIQUE pooled elements: the 10 x 34 patch of a channel is the 2 x 2
  // expansion of 6 x 18 pooled gradients, and the element-per-patch-position staging loaded each of them (and its argmax code)
  // four times -- 12 loads + 30 staging registers per thread and stage, requested in the stage that consumes them (the s_memtime
  // attribution, profiles/r04_w2_segments.md: 18 % of a wave's life issuing them, 17 % waiting for them at the LDS commit).  Now a
  // thread loads L_PER_T = 2 pooled gradients + codes and routes each into its window's four patch positions at commit time;
  // that fits two chunks in flight (DEEP) like the plain source.
  constexpr int PPH = W2_PH / 2 + 1, PPW = W2_PW / 2 + 1;           // pooled rows / columns under the patch (6 x 18)
  constexpr int P_ELEMS = W2_CB * PPH * PPW;
  constexpr int L_PER_T = (SRC == 2) ? (P_ELEMS + NT - 1) / NT : X_PER_T;
  constexpr int L_ELEMS = (SRC == 2) ? P_ELEMS : X_ELEMS;
  constexpr bool DEEP = (SRC == 0 || SRC == 2);
  float xr[L_PER_T], xr2[DEEP ? L_PER_T : 1], mr[SRC == 1 ? L_PER_T : 1];
  unsigned xr_c[SRC == 2 ? L_PER_T : 1], xr2_c[SRC == 2 ? L_PER_T : 1], x_boff[L_PER_T];
  // per staged element: byte offset in the source chunk + ONE packed word.  Patch-position staging: {LDS offset of the .x copy
  // : 16, channel : 4, .x copy exists : 1, .y copy exists : 1}; pooled staging: {LDS offset of patch position (2a, 2b) : 16,
  // channel : 4, pooled row a : 3, pooled column b : 5} -- staging registers are what pushes this kernel against the 256-VGPR limit
  unsigned x_pk[L_PER_T];
  const int ih0 = p0 - 1, iw0 = q0 - 1;
#pragma unroll
  for (int i = 0; i < L_PER_T; ++i) {
    const int idx = tid + i * NT;
    if (SRC == 2) {
      const int cc = idx / (PPH * PPW), rem = idx % (PPH * PPW);
      const int pa = rem / PPW, pb = rem % PPW;
      const int hp = (ih0 >> 1) + pa, wq = (iw0 >> 1) + pb;         // ih0, iw0 are odd (or -1): patch rows 2 pa - 1, 2 pa belong to hp
      const bool ok = idx < P_ELEMS && hp >= 0 && wq >= 0 && hp < a.Hp && wq < a.Wp;
      const unsigned lo = (unsigned)((cc < W2_CB ? cc : 0) * W2_PL + (2 * pa) * W2_RP + 2 * (2 * pb));
      x_pk[i] = lo | ((ok ? (unsigned)cc : 15u) << 16) | ((unsigned)pa << 20) | ((unsigned)pb << 23);
      x_boff[i] = ok ? (unsigned)((cc * a.Hp + hp) * a.Wp + wq) * 4u : 0u;
    } else {
      const int cc = idx / (W2_PH * W2_PW), rem = idx % (W2_PH * W2_PW);
      const int ph = rem / W2_PW, pw = rem % W2_PW;
      const int ih = ih0 + ph, iw = iw0 + pw;
      const bool ok = idx < X_ELEMS && ih >= 0 && iw >= 0 && ih < a.H && iw < a.W;
      const unsigned lo = (unsigned)(cc * W2_PL + ph * W2_RP + 2 * pw);       // .y copy lives at lo - 2 * W2_RP + 1
      x_pk[i] = lo | ((ok ? (unsigned)cc : 15u) << 16) | ((ph < W2_TH ? 1u : 0u) << 20) | ((ph >= 2 ? 1u : 0u) << 21);
      x_boff[i] = ok ? (unsigned)((cc * a.H + ih) * a.W + iw) * 4u : 0u;
    }
  }
  const int in_plane = (SRC == 2) ? a.Hp * a.Wp : a.H * a.W;
  const float* xin = a.x + (size_t)n * a.C * in_plane;
  const float* min_ = ((SRC == 1) ? a.mask : a.x) + (size_t)n * a.C * in_plane;
  const unsigned char* cin_ = (SRC == 2) ? a.code_in + (size_t)n * a.C * in_plane : nullptr;
  const int chunk_elems = W2_CB * in_plane;

#define W2_DMA(CH, DST)                                                                              \
  {                                                                                                  \
    const float* usrc = a.U + (size_t)(CH) * U_STAGE + lane * 4;                                     \
    _Pragma("unroll") for (int j = 0; j < U_PER_W; ++j) {                                            \
      const int ins = wave + NW * j;                                                                 \
      if (U_INSTR % NW == 0 || ins < U_INSTR)                                                        \
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(usrc + ins * 256), (lds_void_t*)((DST) + ins * 256), 16, 0, 0); \
    }                                                                                                \
  }
#define W2_LOAD_X(CH, XR)                                                                            \
  {                                                                                                  \
    const int cleft = a.C - (CH) * W2_CB;                                                            \
    const bool tail = cleft < W2_CB;                                                                 \
    const char* xsrc = (const char*)(xin + (size_t)(CH) * chunk_elems);                              \
    const char* msrc = (const char*)(min_ + (size_t)(CH) * chunk_elems);                             \
    const unsigned char* csrc = cin_ + (size_t)(CH) * chunk_elems;                                   \
    _Pragma("unroll") for (int i = 0; i < L_PER_T; ++i) {                                            \
      unsigned off = x_boff[i];                                                                      \
      if (tail) off = ((int)((x_pk[i] >> 16) & 15u) < cleft) ? off : 0u;                             \
      XR[i] = *(const float*)(xsrc + off);                                                           \
      if (SRC == 1) mr[i] = *(const float*)(msrc + off);                                             \
      if (SRC == 2) XR##_c[i] = csrc[off >> 2];                                                      \
    }                                                                                                \
  }
#define W2_STORE_X(BUF, CH, XR)                                                                      \
  {                                                                                                  \
    const int cleft = a.C - (CH) * W2_CB;                                                            \
    _Pragma("unroll") for (int i = 0; i < L_PER_T; ++i) {                                            \
      if (tid + i * NT < L_ELEMS) {                                                                  \
        bool keep = ((x_pk[i] >> 16) & 15u) < (unsigned)min(cleft, 15);                              \
        if (SRC == 1) keep = keep && mr[i] > 0.f;                                                    \
        const float xv = keep ? XR[i] : 0.f;                                                         \
        const int lo_ = (int)(x_pk[i] & 0xFFFFu);                                                    \
        if (SRC == 2) {                                                                              \
          /* pooled gradient xv of window (pa, pb): patch rows 2 pa - 1 + dy, columns 2 pb - 1 + dx; the argmax code names */ \
          /* the one position that receives it, the other three get zeros (every patch position is written: no stale data) */ \
          const int pa_ = (int)((x_pk[i] >> 20) & 7u), pb_ = (int)((x_pk[i] >> 23) & 31u);           \
          const unsigned code_ = XR##_c[i];                                                          \
          _Pragma("unroll") for (int dy = 0; dy < 2; ++dy)                                           \
            _Pragma("unroll") for (int dx = 0; dx < 2; ++dx) {                                       \
              const int ph_ = 2 * pa_ - 1 + dy, pw_ = 2 * pb_ - 1 + dx;                              \
              if (ph_ >= 0 && ph_ < W2_PH && pw_ >= 0 && pw_ < W2_PW) {                              \
                const float v_ = (code_ == (unsigned)((dy << 1) | dx | 4)) ? xv : 0.f;               \
                const int at_ = lo_ + (dy - 1) * W2_RP + 2 * (dx - 1);                               \
                if (ph_ < W2_TH) sx[(BUF) * XS_SZ + at_] = v_;                                       \
                if (ph_ >= 2) sx[(BUF) * XS_SZ + at_ - 2 * W2_RP + 1] = v_;                          \
              }                                                                                      \
            }                                                                                        \
        } else {                                                                                     \
          if (x_pk[i] & (1u << 20)) sx[(BUF) * XS_SZ + lo_] = xv;                                    \
          if (x_pk[i] & (1u << 21)) sx[(BUF) * XS_SZ + lo_ - 2 * W2_RP + 1] = xv;                    \
        }                                                                                            \
      }                                                                                              \
    }                                                                                                \
  }
#define W2_STEP(CH, UCUR, UNXT, XNEXT, XFAR)      /* XNEXT: registers of chunk CH + 1; XFAR (DEEP): loaded now with CH + 2 */ \
  {                                                                                                  \
    const int cch = (CH);                                                                             \
    const bool have_next = cch + 1 < a.nchunks;                                                       \
    const int xb = cch & 1;                                                                           \
    W2_T(ts0)                                                                                         \
    if (have_next) {                                                                                 \
      W2_DMA(cch + 1, UNXT)                                                                           \
      if (!DEEP) W2_LOAD_X(cch + 1, XNEXT)                                                            \
    }                                                                                                \
    if (DEEP && cch + 2 < a.nchunks) W2_LOAD_X(cch + 2, XFAR)                                         \
    W2_T(ts1)                                                                                         \
    const float* xl = sx + xb * XS_SZ;                                                               \
    _Pragma("unroll") for (int ks = 0; ks < W2_KS; ++ks) {                                           \
    if (W2_KS == 3 && ks > 0 && cch * W2_CB + ks * 4 >= a.C) continue;   /* 12-channel stages: the last one may hold 4 or 8 */ \
    float av[16];                                                                                    \
    {                                                                                                \
      const float* up = (UCUR) + ks * W2_SLAB + aoff;                                                \
      const f32x4 r0 = *(const f32x4*)up;    \
      const f32x4 r1 = *(const f32x4*)(up + 4);     \
      const f32x4 r2 = *(const f32x4*)(up + 8);     \
      const float tt[12] = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3], r2[0], r2[1], r2[2], r2[3]}; \
      _Pragma("unroll") for (int ar = 0; ar < 4; ++ar) {   /* (G g) G^T: u0 = x0, u1 = m + h, u2 = m - h, u3 = x2 */ \
        const float x0 = tt[3 * ar], hh = tt[3 * ar + 1], x2 = tt[3 * ar + 2];                       \
        const float mm = 0.5f * (x0 + x2);                                                           \
        av[4 * ar] = x0; av[4 * ar + 1] = mm + hh; av[4 * ar + 2] = mm - hh; av[4 * ar + 3] = x2;     \
      }                                                                                              \
    }                                                                                                \
    /* both tile rows of the wave at once: every quantity is a (tile row 0, tile row 1) pair in two adjacent VGPRs */ \
    /* (ds_read2_b32 fills such a pair from two addresses), so each transform add is ONE v_pk_add_f32 for two MFMA */ \
    /* operands -- 1 VALU op per MFMA instead of 2.8 */                                                \
    _Pragma("unroll") for (int pg = 0; pg < PG; ++pg) {                                               \
      const float* xp = xl + ks * 4 * W2_PL + boff[pg];                                               \
      f32x2 d[4][4];                                                                                 \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                \
        const f32x4 q0 = *(const f32x4*)(xp + i * W2_RP);      \
        const f32x4 q1 = *(const f32x4*)(xp + i * W2_RP + 4);  \
        d[i][0] = __builtin_shufflevector(q0, q0, 0, 1); d[i][1] = __builtin_shufflevector(q0, q0, 2, 3); \
        d[i][2] = __builtin_shufflevector(q1, q1, 0, 1); d[i][3] = __builtin_shufflevector(q1, q1, 2, 3); \
      }                                                                                              \
      f32x2 t[4][4];   /* B^T d */                                                                   \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)   /* t0 = d0 - d2, t1 = d1 + d2, t2 = d2 - d1, t3 = d1 - d3 */ \
        asm("v_pk_add_f32 %0, %4, %6 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %1, %5, %6\n\t"         \
            "v_pk_add_f32 %2, %6, %5 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %3, %5, %7 neg_lo:[0,1] neg_hi:[0,1]" \
            : "=&v"(t[0][j]), "=&v"(t[1][j]), "=&v"(t[2][j]), "=&v"(t[3][j])                         \
            : "v"(d[0][j]), "v"(d[1][j]), "v"(d[2][j]), "v"(d[3][j]));                               \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {   /* (B^T d) B, then the 8 MFMAs of row i */  \
        /* v0 = t0 - t2, v1 = t1 + t2, v2 = t2 - t1, v3 = t1 - t3 on (tile row 0, tile row 1) pairs.  Written as */ \
        /* asm because LLVM scalarises a <2 x float> op whose lanes are only extracted (here: MFMA operands) into */ \
        /* two v_add_f32; the trailing s_nop 1 = the 2 wait states a VALU result needs before an MFMA reads it */ \
        f32x2 v0, v1, v2, v3;                                                                        \
        asm("v_pk_add_f32 %0, %4, %6 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %1, %5, %6\n\t"         \
            "v_pk_add_f32 %2, %6, %5 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %3, %5, %7 neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 1" \
            : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(t[i][0]), "v"(t[i][1]), "v"(t[i][2]), "v"(t[i][3])); \
        acc[4 * i + 0][2 * pg] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * i + 0], v0[0], acc[4 * i + 0][2 * pg], 0, 0, 0); \
        acc[4 * i + 1][2 * pg] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * i + 1], v1[0], acc[4 * i + 1][2 * pg], 0, 0, 0); \
        acc[4 * i + 2][2 * pg] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * i + 2], v2[0], acc[4 * i + 2][2 * pg], 0, 0, 0); \
        acc[4 * i + 3][2 * pg] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * i + 3], v3[0], acc[4 * i + 3][2 * pg], 0, 0, 0); \
        acc[4 * i + 0][2 * pg + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * i + 0], v0[1], acc[4 * i + 0][2 * pg + 1], 0, 0, 0); \
        acc[4 * i + 1][2 * pg + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * i + 1], v1[1], acc[4 * i + 1][2 * pg + 1], 0, 0, 0); \
        acc[4 * i + 2][2 * pg + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * i + 2], v2[1], acc[4 * i + 2][2 * pg + 1], 0, 0, 0); \
        acc[4 * i + 3][2 * pg + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * i + 3], v3[1], acc[4 * i + 3][2 * pg + 1], 0, 0, 0); \
      }                                                                                              \
    }                                                                                                \
    }                                                                                                \
    W2_T(ts2)                                                                                         \
    if (have_next) W2_STORE_X(xb ^ 1, cch + 1, XNEXT)                                                 \
    W2_T(ts3)                                                                                         \
    __syncthreads();     /* (a hand-written `s_waitcnt vmcnt(X_PER_T)` + s_barrier that keeps the DEEP loads in flight makes LLVM's waitcnt pass put a vmcnt(0) right behind the next stage's loads: measured on the ISA, not kept) */ \
    W2_T(ts4)                                                                                         \
    W2_TACC(1, ts1, ts0) W2_TACC(2, ts2, ts1) W2_TACC(3, ts3, ts2) W2_TACC(4, ts4, ts3)               \
  }

  W2_DMA(0, su0)
  W2_LOAD_X(0, xr)
  if (DEEP && 1 < a.nchunks) W2_LOAD_X(1, xr2)
  W2_STORE_X(0, 0, xr)
  __syncthreads();
  W2_T(t_loop)
  W2_TACC(0, t_loop, t_begin)
  // (loop peeled rather than "if (ch + 1 < n) STEP" inside the body: that form was MIScompiled by this toolchain --
  // the second step's contributions vanished -- tools/debug/dbg_wino2*.py)
  {
    int ch = 0;
#pragma unroll 1
    for (; ch + 1 < a.nchunks; ch += 2) {        // DEEP: chunk ch + 1 sits in xr2 (ch even), chunk ch + 2 goes to xr
      if (DEEP) {
        W2_STEP(ch, su0, su1, xr2, xr)
        W2_STEP(ch + 1, su1, su0, xr, xr2)
      } else {
        W2_STEP(ch, su0, su1, xr, xr)
        W2_STEP(ch + 1, su1, su0, xr, xr)
      }
    }
    if (ch < a.nchunks) {
      if (DEEP) W2_STEP(ch, su0, su1, xr2, xr) else W2_STEP(ch, su0, su1, xr, xr)
    }
  }
#undef W2_STEP
#undef W2_STORE_X
#undef W2_LOAD_X
#undef W2_DMA

  // ---- output transform + epilogue.  Lane: tile column ln, rows 16 g + 4 kc + reg; A^T = [[1,1,1,0],[0,1,-1,-1]]
  W2_T(t_epi)
  const int P = a.H, Q = a.W;          // stride 1 / pad 1: output extent == input extent
#pragma unroll
  for (int gr = 0; gr < 2 * PG; ++gr) {
    const int trow = 2 * hrow * PG + gr;
    const int p = p0 + 2 * trow, q = q0 + 2 * ln;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int k = 16 * g + 4 * kc + reg;
      float s[2][4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float m0 = acc[b][gr][reg], m1 = acc[4 + b][gr][reg], m2 = acc[8 + b][gr][reg], m3 = acc[12 + b][gr][reg];
        s[0][b] = m0 + m1 + m2;
        s[1][b] = m1 - m2 - m3;
      }
      float o[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        o[i][0] = s[i][0] + s[i][1] + s[i][2];
        o[i][1] = s[i][1] - s[i][2] - s[i][3];
      }
      if (k >= a.K) continue;
      const float bv = a.bias ? a.bias[k] : 0.f;
      if (EPI == 1) {
        float m = fmaxf(o[0][0] + bv, 0.f);
        int arg = 0;
        const float v01 = fmaxf(o[0][1] + bv, 0.f), v10 = fmaxf(o[1][0] + bv, 0.f), v11 = fmaxf(o[1][1] + bv, 0.f);
        if (v01 > m) { m = v01; arg = 1; }
        if (v10 > m) { m = v10; arg = 2; }
        if (v11 > m) { m = v11; arg = 3; }
        const int pp = p >> 1, qq = q >> 1, Pp = P >> 1, Qp = Q >> 1;
        if (pp < Pp && qq < Qp) {
          const size_t oo = (((size_t)n * a.K + k) * Pp + pp) * Qp + qq;
          a.pool_y[oo] = m;
          a.code_out[oo] = (unsigned char)(arg | (m > 0.f ? 4 : 0));
        }
      } else {
        const float slope = a.act_slope ? (a.slope_ptr ? a.slope_ptr[0] : a.slope_imm) : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (p + i >= P) continue;
          const size_t yo = (((size_t)n * a.K + k) * P + (p + i)) * Q + q;
          float v0 = o[i][0] + bv, v1 = o[i][1] + bv;
          if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
          if (a.act_slope) { v0 = v0 > 0.f ? v0 : v0 * slope; v1 = v1 > 0.f ? v1 : v1 * slope; }
          if (q + 1 < Q && !(Q & 1)) {           // 8-B aligned pair
            if (a.residual) {
              const f32x2 rr = *(const f32x2*)(a.residual + yo);
              v0 += rr[0]; v1 += rr[1];
            }
            *(f32x2*)(a.y + yo) = f32x2{v0, v1};
          } else {
            if (q < Q) a.y[yo] = v0 + (a.residual ? a.residual[yo] : 0.f);
            if (q + 1 < Q) a.y[yo + 1] = v1 + (a.residual ? a.residual[yo + 1] : 0.f);
          }
        }
      }
    }
  }
#if W2_TIME
  {
    const unsigned long long t_end = __builtin_readcyclecounter();
    tacc[5] = t_end - t_epi; tacc[6] = t_end - t_begin; tacc[7] = t_begin;
    if (a.tbuf && lane == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a.tbuf[((size_t)blockIdx.x * NW + wave) * 8 + i] = tacc[i];
    }
  }
#endif
}

#if W2_TIME
unsigned long long* g_w2_tbuf = nullptr;
#endif

int w2_env() { return fcd_sw(FCD_SW_WINO2); }
}  // namespace

// mode 0 forward / 1 data gradient: 1 when the layer runs on the fused F(2x2,3x3) kernel
extern "C" int fcd_conv_wino2_plan(const fcd_conv_desc* d, int mode) {
  if (!d || !w2_env() || fcd_wino_mode_now() == 0) return 0;
  if (!(d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1)) return 0;
  const int rows = mode == 0 ? d->K : d->C, red = mode == 0 ? d->C : d->K;
  const int min_red = fcd_sw(FCD_SW_WINO2_MINC);
  if (rows <= 32 || rows > W2_ROWS || red < min_red) return 0;
  if (d->H < 4 || d->W < 4) return 0;
  return 1;
}

extern "C" int64_t fcd_conv_wino2_filter_elems(int K, int C, int mode) {
  const int red = mode == 0 ? C : K;
  return (int64_t)round_up(cdiv(red, 4), 6) * W2_SLAB;     // whole 8- or 12-channel stages: the tail chunks are zero-filled by the packer
}

extern "C" int fcd_conv_wino2_pack(const float* w, float* U, int K, int C, int mode, void* stream) {
  FCD_CHECK_ARG(w && U && K > 0 && C > 0 && (mode == 0 || mode == 1), "fcd_conv_wino2_pack: bad arguments");
  FCD_CHECK_ARG((mode == 0 ? K : C) <= W2_ROWS, "fcd_conv_wino2_pack: more than %d GEMM rows", W2_ROWS);
  const long long total = fcd_conv_wino2_filter_elems(K, C, mode);
  const int grid = (int)std::min<long long>(cdiv64(total, 256), 4096);
  FcdProfScope prof(FCD_K_PACK, (hipStream_t)stream, 0.0, 4.0 * total);
  hipLaunchKernelGGL(wino2_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, U, K, C, mode, total);
  FCD_LAUNCH_CHECK("wino2_pack");
  return FCD_OK;
}

static int w2_xcd() { return fcd_sw(FCD_SW_CONV_XCD); }

static int w2_waves() { return fcd_sw(FCD_SW_WINO2_WAVES); }     // 8 (one 8 x 32 workgroup per CU) | 4 (two 4 x 32 workgroups per CU) | 1

#if W2_TIME
extern "C" void fcd_wino2_time_buf(void* p) { g_w2_tbuf = (unsigned long long*)p; }
#endif

template <int SRC, int EPI>
static void w2_launch(Wino2Args& a, int red, hipStream_t st) {
#if W2_TIME
  a.tbuf = g_w2_tbuf;
#endif
  a.tiles_q = cdiv(a.W, W2_TW);
  a.xcd_remap = w2_xcd();
  const int ks = fcd_sw(FCD_SW_WINO2_KS);
  if (w2_waves() == 1) {          // one wave per SIMD: 4 waves x 4 tile rows, 512-register budget (accumulators in AGPRs)
    a.tiles_p = cdiv(a.H, 8);
    a.nchunks = cdiv(red, 8);
    hipLaunchKernelGGL((conv_wino2_kernel<SRC, EPI, 4, 2, 2>), dim3((unsigned)(a.N * a.tiles_p * a.tiles_q)), dim3(256), 0, st, a);
  } else if (w2_waves() == 8 && ks == 3) {      // 12-channel stages: 145 KB of LDS, 6 instead of 8 barriers per 64 channels
    a.tiles_p = cdiv(a.H, 8);
    a.nchunks = cdiv(red, 12);
    hipLaunchKernelGGL((conv_wino2_kernel<SRC, EPI, 8, 3, 1>), dim3((unsigned)(a.N * a.tiles_p * a.tiles_q)), dim3(512), 0, st, a);
  } else if (w2_waves() == 8 && ks == 2) {
    a.tiles_p = cdiv(a.H, 8);
    a.nchunks = cdiv(red, 8);
    hipLaunchKernelGGL((conv_wino2_kernel<SRC, EPI, 8, 2, 1>), dim3((unsigned)(a.N * a.tiles_p * a.tiles_q)), dim3(512), 0, st, a);
  } else if (w2_waves() == 8) {
    a.tiles_p = cdiv(a.H, 8);
    a.nchunks = cdiv(red, 4);
    hipLaunchKernelGGL((conv_wino2_kernel<SRC, EPI, 8, 1, 1>), dim3((unsigned)(a.N * a.tiles_p * a.tiles_q)), dim3(512), 0, st, a);
  } else if (ks == 2) {
    a.tiles_p = cdiv(a.H, 4);
    a.nchunks = cdiv(red, 8);
    hipLaunchKernelGGL((conv_wino2_kernel<SRC, EPI, 4, 2, 1>), dim3((unsigned)(a.N * a.tiles_p * a.tiles_q)), dim3(256), 0, st, a);
  } else {
    a.tiles_p = cdiv(a.H, 4);
    a.nchunks = cdiv(red, 4);
    hipLaunchKernelGGL((conv_wino2_kernel<SRC, EPI, 4, 1, 1>), dim3((unsigned)(a.N * a.tiles_p * a.tiles_q)), dim3(256), 0, st, a);
  }
}

// y = act(conv(x, w) + bias) + residual, or (pool_y, code) = maxpool2(relu(conv + bias)) when pool_y != NULL
extern "C" int fcd_conv2d_fwd_wino2(const fcd_conv_desc* d, const float* x, const float* U, const float* bias, float* y,
                                    int act, const float* slope_ptr, float slope_imm, const float* residual, float* pool_y,
                                    unsigned char* code, void* stream) {
  FCD_CHECK_ARG(d && x && U && (y || (pool_y && code)), "fcd_conv2d_fwd_wino2: null pointer");
  FCD_CHECK_ARG(fcd_conv_wino2_plan(d, 0), "fcd_conv2d_fwd_wino2: layer is not planned for the fused F(2x2,3x3) kernel");
  FCD_CHECK_ARG(act >= FCD_ACT_NONE && act <= FCD_ACT_PRELU, "fcd_conv2d_fwd_wino2: bad activation code %d", act);
  Wino2Args a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.U = U; a.bias = bias; a.y = y; a.pool_y = pool_y; a.code_out = code; a.residual = residual;
  a.relu = act == FCD_ACT_RELU; a.act_slope = (act == FCD_ACT_LEAKY || act == FCD_ACT_PRELU) ? 1 : 0;
  a.slope_ptr = slope_ptr; a.slope_imm = slope_imm;
  a.N = d->N; a.C = d->C; a.H = d->H; a.W = d->W; a.K = d->K;
  const double flops = 2.0 * d->N * d->K * (double)d->P * d->Q * d->C * 9;
  const double bytes = 4.0 * ((double)d->N * d->C * d->H * d->W + (pool_y ? 0.3125 : 1.0) * d->N * d->K * d->P * d->Q +
                              (double)d->K * d->C * 16);
  FcdProfScope prof(FCD_K_WINO2_FWD, (hipStream_t)stream, flops, bytes, fcd_prof_tag_desc(pool_y ? "w2_fwd_pool" : "w2_fwd", d));
  if (pool_y) w2_launch<0, 1>(a, d->C, (hipStream_t)stream); else w2_launch<0, 0>(a, d->C, (hipStream_t)stream);
  FCD_LAUNCH_CHECK("conv2d_fwd_wino2");
  return FCD_OK;
}

// dx = conv_transpose(dy') with dy' = dy, dy * [relu_out > 0], or the pooled gradient routed by pool_code
extern "C" int fcd_conv2d_bwd_data_wino2(const fcd_conv_desc* d, const float* dy, const float* relu_out,
                                         const unsigned char* pool_code, const float* U, float* dx, void* stream) {
  FCD_CHECK_ARG(d && dy && U && dx, "fcd_conv2d_bwd_data_wino2: null pointer");
  FCD_CHECK_ARG(fcd_conv_wino2_plan(d, 1), "fcd_conv2d_bwd_data_wino2: layer is not planned for the fused F(2x2,3x3) kernel");
  Wino2Args a;
  memset(&a, 0, sizeof(a));
  a.x = dy; a.U = U; a.y = dx; a.mask = pool_code ? nullptr : relu_out; a.code_in = pool_code;
  a.N = d->N; a.C = d->K; a.H = d->P; a.W = d->Q; a.K = d->C; a.Hp = d->P / 2; a.Wp = d->Q / 2;
  const double flops = 2.0 * d->N * d->K * (double)d->P * d->Q * d->C * 9;
  const double bytes = 4.0 * ((double)d->N * d->C * d->H * d->W + (pool_code ? 0.3125 : (relu_out ? 2.0 : 1.0)) * d->N * d->K * d->P * d->Q +
                              (double)d->K * d->C * 16);
  FcdProfScope prof(FCD_K_WINO2_DGRAD, (hipStream_t)stream, flops, bytes, fcd_prof_tag_desc("w2_dgrad", d));
  if (pool_code) w2_launch<2, 0>(a, d->K, (hipStream_t)stream);
  else if (relu_out) w2_launch<1, 0>(a, d->K, (hipStream_t)stream);
  else w2_launch<0, 0>(a, d->K, (hipStream_t)stream);
  FCD_LAUNCH_CHECK("conv2d_bwd_data_wino2");
  return FCD_OK;
}
