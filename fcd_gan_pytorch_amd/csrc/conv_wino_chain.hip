// Output transform of layer l -> (bias, ReLU | ReLU gate) -> input transform of layer l + 1 in ONE kernel, for runs of
// F(4x4,3x3) layers with nothing but a ReLU between them: the frozen VGG16 stack of the perception term (reference
// Loss.py:25-36 -- conv2_1 .. conv5_3, requires_grad = False, only relu5_3 tapped by default) forward, and its data gradient
// backward.  The activation between two such layers is never a tensor in HBM: the stand-alone pair of transform kernels
// moves M (2.25 units) -> y (1 write + 1 read) -> V (2.25), this kernel M -> V: 4.5 instead of 6.5 tensor units per layer
// boundary, and the ReLU sign bits the backward pass needs (16 per tile) fall out of it as before.
//
// One workgroup = (image n, CHB channels, band of BW tile columns), walking DOWN the image.  Per step: the 36 M values of
// the next TRS tile rows arrive as 36 float4 per thread (thread = tile x 4 channels, the split GEMM's MFMA-native blocks:
// conv_wino.hip, wino_output_blk_kernel), are output-transformed (+bias, ReLU, sign bits | gate by the sign bits of the
// tensor being produced) and written into an LDS ring of 4 TRS + 5 image rows; then every thread runs the input
// transform of (tile, channel) items one tile row BEHIND -- the 6 x 6 patch of a tile needs the row above and below its own
// tile row -- with the channel fastest across lanes, so that a V row [tile][32 ch] leaves as one 128-B segment.  The M
// loads of step s + 1 are issued before the input transforms of step s (registers: 36 float4), so that one workgroup per CU
// keeps the HBM stream going; the ring is LDS-resident for the whole image, no halo row is ever re-read.  A band narrower
// than the image (W = 128: two bands of 16 tiles) recomputes the ONE neighbouring tile column on each inner side (+6 % M reads).
// Same tile functions as the stand-alone kernels (conv_wino.h) => V is bit-identical to output transform -> y -> input transform.
#include "conv_wino.h"

// Ring: image row y lives in ring row (y + 4) % RR.  A step writes 4 TRS rows while the rows from one above the step's first input
// tile row are still to be read: 4 TRS + 5 rows are live at most, RR = that (9 rows of 66 columns x 32 channels = 76 KB: two
// workgroups per CU, which is what overlaps one workgroup's load / output phase with the other's input / store phase).
// VW = channels per thread in the output phase: 4 (one 16-B word of the GEMM's blocks per position; BW + 2 tile columns incl. the
// halo candidates) or 2 (8-B halves: lanes (tile, pair) still cover 256 contiguous bytes, every thread of the workgroup has an
// item and holds 72 instead of 144 prefetch registers; bands that span the image only -- no halo column).
template <int BW, int TRS, int VW, int NT, int MINB>
__global__ __launch_bounds__(NT, MINB) void wino_oi_kernel(WinoOiArgs a) {
  constexpr int A = 6, CHB = 32;           // one 32-channel chunk of V per workgroup: whole 128-B V rows (16-channel workgroups, 64-B rows: 3.7 vs 4.5 TB/s)
  constexpr int RR = 4 * TRS + 5;          // ring rows
  constexpr int CW = 4 * BW + 2;           // ring columns: left halo, band, right halo
  constexpr int PL = (RR * CW) | 1;        // odd channel pitch: the input phase reads with lanes = channels
  constexpr int JT = VW == 4 ? BW + 2 : BW;   // tile columns of the output phase (VW == 4: incl. the two halo candidates)
  constexpr int GP = 4 / VW;               // channel groups per 16-B word
  constexpr int QD = CHB / 4;              // channel quads
  static_assert(TRS * JT * QD * GP <= NT, "output phase: one item per thread");
  typedef float mvec __attribute__((ext_vector_type(VW)));
  __shared__ float ring[CHB * PL];
  const int tid = threadIdx.x;
  const int tx0 = blockIdx.x * BW;
  const int kb = blockIdx.y * CHB;         // first channel of the block
  const int n = blockIdx.z;

  // zero what is read but never written: the halo columns at the image border, the row above tile row 0
  for (int i = tid; i < CHB * RR; i += NT) {
    const int c = i / RR, r = i % RR;
    ring[c * PL + r * CW] = 0.f;
    ring[c * PL + r * CW + CW - 1] = 0.f;
  }
  for (int i = tid; i < CHB * CW; i += NT) ring[(i / CW) * PL + 3 * CW + i % CW] = 0.f;      // y = -1

  // output phase: thread = (channel group og of its 16-B word, tile column oj of the band (-1 / BW: halo), tile row otr of the step,
  // channel quad ocq), the group fastest across lanes
  const int og = tid % GP, oj = (tid / GP) % JT - (VW == 4 ? 1 : 0), orest = tid / (GP * JT);
  const int otr = orest % TRS, ocq = orest / TRS;
  const int otx = tx0 + oj;
  const bool o_on = ocq < QD && otx >= 0 && otx < a.TW;
  const int k0 = kb + 4 * ocq + VW * og;   // first of the thread's VW channels
  const int cl0 = 4 * ocq + VW * og;       // ... inside the block
  mvec m4[A * A];
  unsigned gw[VW];
  auto issue = [&](int s) {
    const int ty = s * TRS + otr;
    if (!o_on || ty >= a.TH) return;
    const long long t = ((long long)n * a.TH + ty) * a.TW + otx;
    const float* mp = a.Mb + ((size_t)(k0 >> 5) * a.tblk + (size_t)(t >> 5)) * 1024 +
                      ((((k0 >> 2) & 1) * 4 + ((k0 >> 3) & 3)) * 32 + (int)(t & 31)) * 4 + (k0 & 3);
#pragma unroll
    for (int q = 0; q < A * A; ++q) m4[q] = __builtin_nontemporal_load((const mvec*)(mp + (size_t)q * a.xs_blk));
    if (a.gate) {
#pragma unroll
      for (int e = 0; e < VW; ++e) gw[e] = a.gate[(((size_t)n * a.K + k0 + e) * a.TH + ty) * a.TW + otx];
    }
  };
  auto out_phase = [&](int s) {
    const int ty = s * TRS + otr;
    if (ty == a.TH && ocq < QD) {        // the row below the last tile row (a banded ring holds an old halo value there)
#pragma unroll
      for (int e = 0; e < VW; ++e) {
        float* rp = ring + (cl0 + e) * PL + ((4 * ty + 4) % RR) * CW;
        if (oj < 0) rp[0] = 0.f;
        else if (oj >= BW) rp[CW - 1] = 0.f;
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) rp[1 + 4 * oj + j] = 0.f;
        }
      }
    }
    if (!o_on || ty >= a.TH) return;
    int rrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rrow[i] = ((4 * ty + 4 + i) % RR) * CW;
#pragma unroll
    for (int e = 0; e < VW; ++e) {
      float mv[A][A];
#pragma unroll
      for (int q = 0; q < A * A; ++q) mv[q / A][q % A] = m4[q][e];
      float o[4][4];
      wino_out_tile<4>(mv, a.bias ? a.bias[k0 + e] : 0.f, a.relu, o);
      if (a.gate) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) o[i][j] = ((gw[e] >> (4 * i + j)) & 1u) ? o[i][j] : 0.f;
      }
      float* rp = ring + (cl0 + e) * PL;
      if (oj >= 0 && oj < BW) {
        if (a.bits_out) {
          unsigned w = 0;
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) w |= (o[i][j] > 0.f ? 1u : 0u) << (4 * i + j);
          a.bits_out[(((size_t)n * a.K + k0 + e) * a.TH + ty) * a.TW + otx] = (unsigned short)w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) rp[rrow[i] + 1 + 4 * oj + j] = o[i][j];
      } else if (oj < 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) rp[rrow[i]] = o[i][3];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) rp[rrow[i] + CW - 1] = o[i][0];
      }
    }
  };

  const size_t xi_stride = (size_t)a.Q * a.T * 32;
  const int steps = a.TH / TRS + 1;
  issue(0);
#pragma unroll 1
  for (int s = 0; s < steps; ++s) {
    out_phase(s);
    if (s + 1 < steps) issue(s + 1);
    __syncthreads();
    // input phase: tile rows s TRS - 1 .. s TRS + TRS - 2
#pragma unroll 1
    for (int it = tid; it < TRS * BW * CHB; it += NT) {
      const int c = it % CHB, tl = it / CHB;
      const int tc = tl % BW, r = s * TRS - 1 + tl / BW;
      if (r < 0 || r >= a.TH || tx0 + tc >= a.TW) continue;
      const float* rc = ring + c * PL + 4 * tc;
      float d[A][A];
#pragma unroll
      for (int i = 0; i < A; ++i) {
        const int ro = ((4 * r + 3 + i) % RR) * CW;       // image row 4 r - 1 + i
#pragma unroll
        for (int j = 0; j < A; ++j) d[i][j] = rc[ro + j];
      }
      float t1[A][A];
      wino_in_rows<4>(d, t1);
      const size_t t = ((size_t)n * a.TH + r) * a.TW + tx0 + tc;
      const int kg = kb + c;
      float* vout = a.V + ((size_t)(kg >> 5) * a.T + t) * 32 + (kg & 31);
#pragma unroll
      for (int i = 0; i < A; ++i)
#pragma unroll
        for (int j = 0; j < A; ++j) vout[(size_t)(i * A + j) * xi_stride] = wino_in_col<4>(t1, i, j);
    }
    __syncthreads();
  }
}

// 0: no kernel for this geometry (the caller runs output transform -> tensor -> input transform instead)
int wino_oi_ok(int K, int H, int W) {
  if (!fcd_sw(FCD_SW_WINO_CHAIN) || (K & 31) || (H & 3) || (W & 3)) return 0;
  const int TW = W / 4, TH = H / 4;
  if (TW >= 16) return (TW % 16) == 0;
  if (TW == 8) return (TH % 2) == 0;
  if (TW == 4) return (TH % 4) == 0;
  return 0;
}

void wino_oi_launch(const WinoOiArgs& a, hipStream_t st) {
  // measured on MI355X (208 band images; bytes = M read + V written): 16 / 8 / 4 tiles wide, channel pairs per thread, two workgroups per
  // CU: 4.8 - 5.1 / 5.2 - 5.5 / 5.2 - 5.5 TB/s (quads: 4.5 / 4.9 / 4.2; one workgroup per CU: 4.1 / 4.3 / 3.3); 32 tiles wide as ONE
  // 8-wave workgroup per CU over the whole row: 4.5 - 4.7 TB/s (two 16-tile bands: 3.6 -- the one halo tile column of a band costs a
  // whole 128-B line per position and row); unrolling the input phase by two: no change
  const dim3 grid((unsigned)cdiv(a.TW, a.TW >= 16 ? 16 : a.TW), (unsigned)(a.K / 32), (unsigned)a.N);
  if (a.TW == 32) hipLaunchKernelGGL((wino_oi_kernel<32, 1, 2, 512, 1>), dim3(1, (unsigned)(a.K / 32), (unsigned)a.N), dim3(512), 0, st, a);
  else if (a.TW > 16) hipLaunchKernelGGL((wino_oi_kernel<16, 1, 4, 256, 2>), grid, dim3(256), 0, st, a);      // bands, neighbouring tile columns recomputed
  else if (a.TW == 16) hipLaunchKernelGGL((wino_oi_kernel<16, 1, 2, 256, 2>), grid, dim3(256), 0, st, a);
  else if (a.TW == 8) hipLaunchKernelGGL((wino_oi_kernel<8, 2, 2, 256, 2>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((wino_oi_kernel<4, 4, 2, 256, 2>), grid, dim3(256), 0, st, a);
}
