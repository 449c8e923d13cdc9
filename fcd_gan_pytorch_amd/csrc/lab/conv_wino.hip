// Winograd F(m x m, 3 x 3) path for the 3x3 / stride-1 / pad-1 layers whose GEMM has >= 128 rows and
// >= 64 reduction channels (VGG conv2_1 .. conv5_3, all but the first U-Net stage) -- Module.py:25-31,
// Loss.py:25.
//
//   y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A            (Lavin & Gray 2016, m = 2 or 4)
//
// per (m+2)^2 transform position xi the channel sum is a plain GEMM
//   M_xi[k][t] = sum_c U_xi[k][c] * V_xi[t][c]             t = image tile index
// with 2.25x (m = 2) / 4x (m = 4) fewer multiplies than the direct convolution.  Three kernels per layer call:
//   wino_input_roll_kernel   x (or the gated / pooled gradient; or a LIST of tensors standing for their channel
//   (wino_input_kernel        concatenation)  -> V   [xi][chunk of 32 ch][t][32]                      HBM-bound
//    on maps one strip high)
//   wino_gemm_split256_kernel / wino_gemm_split_kernel   batched TN GEMM on v_mfma_f32_32x32x16_bf16 with every
//                            fp32 operand split EXACTLY into three bf16 parts (six partial products, fp32
//                            accumulation: fp32-equivalent)                                            MFMA-bound
//   (wino_gemm_kernel         the same GEMM on v_mfma_f32_32x32x2_f32: 64-row GEMMs, FCD_WINO_SPLIT=0)
//   wino_output_kernel       M [xi][k][t] -> y (+bias, ReLU, 2x2 max-pool with argmax code; or into a list
//                            of tensors)                                                               HBM-bound
// The weight gradient of the K, C >= 128 layers takes the same form (reduction over the tiles): wino_wg_* below.
// The transformed filters U [xi][rows][channels padded to 32] -- fp32, or their three bf16 planes -- are packed once per
// weight version (fcd_conv_wino_pack): mode 0 forward, mode 1 data gradient (flipped taps, channels swapped).
// The same arithmetic order is used for every launch => bit-reproducible; fp32 rounding of the
// m = 4 transforms is ~1e-5 relative (tests/test_gpu_ops.py), m = 2 ~3e-7.
#include "conv_wino.h"

#ifndef FCD_NT_EXP
#define FCD_NT_EXP 0  // experiments with non-temporal hints: 2 input-transform x loads, 4 B stream of the filter-resident GEMM, 8 blocked C stores
#endif

// --------------------------------------------------------------------------------------------
// filter transform: U[xi][row][kc] = (G g G^T)[xi]
//   mode 0: row = k (output channel), kc = c, g = w[k][c]
//   mode 1: row = c (input channel),  kc = k, g = w[k][c] with both taps flipped
// planes != NULL: additionally the exact three-way bf16 split u = h + m + l (see wino_gemm_split_kernel), plane p at
// planes + p * 36 * rows * Kc, same [xi][row][kc] order
__device__ __forceinline__ unsigned short bf16_rn_bits(float x) {
  const __bf16 b = (__bf16)x;
  return __builtin_bit_cast(unsigned short, b);
}

template <int MM>
__global__ void wino_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int K, int C, int rows, int Kc,
                                   int mode, unsigned short* __restrict__ planes) {
  // one thread = TWO adjacent reduction channels (Kc is a multiple of 32): the bf16 planes are written as 4-byte pairs
  // and the fp32 U as float2 -- the pack is store-bound (40 B written per filter tap read), 2-byte stores halve its rate
  constexpr int A = WinoMat<MM>::A;
  const long long total = (long long)rows * Kc;
  const long long pairs = total >> 1;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < pairs;
       i += (long long)gridDim.x * blockDim.x) {
    const int kc0 = (int)((2 * i) % Kc), row = (int)((2 * i) / Kc);
    float g[2][3][3];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int kc = kc0 + e;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          float v = 0.f;
          if (mode == 0) {
            if (kc < C) v = w[(((long long)row * C + kc) * 3 + r) * 3 + s];
          } else {
            if (kc < K) v = w[(((long long)kc * C + row) * 3 + (2 - r)) * 3 + (2 - s)];
          }
          g[e][r][s] = v;
        }
    }
    float t[2][A][3];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int a = 0; a < A; ++a)
#pragma unroll
        for (int s = 0; s < 3; ++s)
          t[e][a][s] = WinoMat<MM>::G(a, 0) * g[e][0][s] + WinoMat<MM>::G(a, 1) * g[e][1][s] + WinoMat<MM>::G(a, 2) * g[e][2][s];
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
      for (int b = 0; b < A; ++b) {
        float u[2];
#pragma unroll
        for (int e = 0; e < 2; ++e)
          u[e] = t[e][a][0] * WinoMat<MM>::G(b, 0) + t[e][a][1] * WinoMat<MM>::G(b, 1) + t[e][a][2] * WinoMat<MM>::G(b, 2);
        const long long o = ((long long)(a * A + b) * rows + row) * Kc + kc0;       // even
        if (U) *(float2*)(U + o) = make_float2(u[0], u[1]);
        if (planes) {
          const long long ps = (long long)A * A * total;
          unsigned h[2], m[2], l[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            h[e] = bf16_rn_bits(u[e]);
            const float r = u[e] - __uint_as_float(h[e] << 16);
            m[e] = bf16_rn_bits(r);
            const float q = r - __uint_as_float(m[e] << 16);
            l[e] = bf16_rn_bits(q);
          }
          *(unsigned*)(planes + o) = h[0] | (h[1] << 16);
          *(unsigned*)(planes + ps + o) = m[0] | (m[1] << 16);
          *(unsigned*)(planes + 2 * ps + o) = l[0] | (l[1] << 16);
        }
      }
  }
}

// --------------------------------------------------------------------------------------------
// input transform.  One block: image n, tile row ty, TWB consecutive tiles, one 32-channel chunk.
// The raw (m+2)-row strip is staged in LDS with coalesced row reads (source gating applied here),
// then every thread transforms (tile, channel) items with the channel fastest across lanes, so
// that each V row [t][32 ch] is written as one 128-B segment.
// ReLU mask of source element (plane index pc = n * C + c, row ih, column iw .. iw + 3, iw % 4 == 0) as 0 / 1 floats: from
// the fp32 activation (off = its element offset) or from the 16-bit tile words
__device__ __forceinline__ f32x4 wino_mask4(const WinoInArgs& a, size_t off, size_t pc, int ih, int iw) {
  if (a.mbits) {
    const unsigned w = a.mbits[(pc * a.TH + (ih >> 2)) * a.TW + (iw >> 2)] >> (4 * (ih & 3));
    return f32x4{(float)(w & 1u), (float)((w >> 1) & 1u), (float)((w >> 2) & 1u), (float)((w >> 3) & 1u)};
  }
  return *(const f32x4*)(a.mask + off);
}
__device__ __forceinline__ float wino_mask1(const WinoInArgs& a, size_t off, size_t pc, int ih, int iw) {
  if (a.mbits) return (float)((a.mbits[(pc * a.TH + (ih >> 2)) * a.TW + (iw >> 2)] >> (4 * (ih & 3) + (iw & 3))) & 1u);
  return a.mask[off];
}

// TRB x TWB tiles per block (16 tiles for m = 4, 32 for m = 2): 1 x 16 strips for wide maps, 2 x 8 / 4 x 4
// patches for the 32- and 16-pixel maps deep in the nets, so that no thread idles on tiles outside
// the image.  VEC: W % 4 == 0 -- the interior of every strip row is fetched as float4 (the strip
// starts one pixel left of a 16-B boundary), the two halo columns as scalars.
template <int MM, int SRC, int TRB, int TWB, bool VEC>
__global__ __launch_bounds__(256) void wino_input_kernel(WinoInArgs a) {
  constexpr int A = WinoMat<MM>::A;
  constexpr int RH = TRB * MM + 2;        // input rows of the block's patch
  constexpr int CW = TWB * MM + 2;        // input columns
  constexpr int PL = (RH * CW) | 1;       // odd plane pitch: conflict-free across the 32 channel lanes
  __shared__ float tile[32 * PL];
  const int tid = threadIdx.x;
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (a.xcd) {
    const unsigned lin = xcd_contiguous_id(bx + gridDim.x * (by + gridDim.y * bz), gridDim.x * gridDim.y * gridDim.z);
    bx = lin % gridDim.x;
    by = (lin / gridDim.x) % gridDim.y;
    bz = lin / (gridDim.x * gridDim.y);
  }
  const int tx0 = bx * TWB, ty0 = by * TRB;
  const int n = bz / a.Q, q = bz % a.Q;
  const int ih0 = ty0 * MM - 1, iw0 = tx0 * MM - 1;
  const int plane = (SRC == 2) ? a.Hp * a.Wp : a.H * a.W;
  const size_t img = ((size_t)n * a.C + (size_t)q * 32) * plane;

  // one element of the source at (channel c of the chunk, ih, iw), gated by the source mode
  auto fetch = [&](int c, int ih, int iw) -> float {
    float v = 0.f;
    if (q * 32 + c < a.C && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) {
      if (SRC == 2) {
        const int hp = ih >> 1, wq = iw >> 1;
        if (hp < a.Hp && wq < a.Wp) {
          const size_t off = img + (size_t)c * plane + (size_t)hp * a.Wp + wq;
          const unsigned want = (unsigned)((((ih & 1) << 1) | (iw & 1)) | 4);
          if ((unsigned)a.code[off] == want) v = a.x[off];
        }
      } else {
        const size_t off = img + (size_t)c * plane + (size_t)ih * a.W + iw;
        v = a.x[off];
        if (SRC == 1 && !(wino_mask1(a, off, (size_t)n * a.C + q * 32 + c, ih, iw) > 0.f)) v = 0.f;
      }
    }
    return v;
  };

  if (a.exp & 4) {
  } else if (VEC && SRC == 2) {
    // pooled gradient: 4 strip columns = 2 pooled elements (one 8-B load + 2 code bytes), each routed to
    // the slot its argmax code names
    constexpr int V4 = (CW - 2) / 4;
    for (int idx = tid; idx < 32 * RH * V4; idx += 256) {
      const int c = idx / (RH * V4), rem = idx % (RH * V4);
      const int r = rem / V4, v4 = rem % V4;
      const int ih = ih0 + r, iw = iw0 + 1 + v4 * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const int hp = ih >> 1, wq = iw >> 1;
      if (q * 32 + c < a.C && ih >= 0 && ih < a.H && iw < a.W && hp < a.Hp && wq + 1 < a.Wp + 1) {
        const size_t off = img + (size_t)c * plane + (size_t)hp * a.Wp + wq;
        const unsigned rowbit = (unsigned)((ih & 1) << 1) | 4u;
        if (wq + 1 < a.Wp) {
          const f32x2 g2 = *(const f32x2*)(a.x + off);
          const unsigned c0 = a.code[off], c1 = a.code[off + 1];
          v[0] = c0 == rowbit ? g2[0] : 0.f;
          v[1] = c0 == (rowbit | 1u) ? g2[0] : 0.f;
          v[2] = c1 == rowbit ? g2[1] : 0.f;
          v[3] = c1 == (rowbit | 1u) ? g2[1] : 0.f;
        } else if (wq < a.Wp) {
          const float g0 = a.x[off];
          const unsigned c0 = a.code[off];
          v[0] = c0 == rowbit ? g0 : 0.f;
          v[1] = c0 == (rowbit | 1u) ? g0 : 0.f;
        }
      }
      float* t = tile + c * PL + r * CW + 1 + v4 * 4;
      t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
    }
    for (int idx = tid; idx < 32 * RH * 2; idx += 256) {
      const int c = idx / (RH * 2), rem = idx % (RH * 2);
      const int r = rem >> 1, side = rem & 1;
      const int col = side ? CW - 1 : 0;
      tile[c * PL + r * CW + col] = fetch(c, ih0 + r, iw0 + col);
    }
  } else if (VEC) {
    constexpr int V4 = (CW - 2) / 4;                    // float4 per strip row
    for (int idx = tid; idx < 32 * RH * V4; idx += 256) {
      const int c = idx / (RH * V4), rem = idx % (RH * V4);
      const int r = rem / V4, v4 = rem % V4;
      const int ih = ih0 + r, iw = iw0 + 1 + v4 * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (q * 32 + c < a.C && ih >= 0 && ih < a.H && iw < a.W) {     // W % 4 == 0: whole float4 in range
        const size_t off = img + (size_t)c * plane + (size_t)ih * a.W + iw;
        v = *(const f32x4*)(a.x + off);
        if (SRC == 1) {
          const f32x4 k = wino_mask4(a, off, (size_t)n * a.C + q * 32 + c, ih, iw);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (!(k[e] > 0.f)) v[e] = 0.f;
        }
      }
      float* t = tile + c * PL + r * CW + 1 + v4 * 4;
      t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
    }
    for (int idx = tid; idx < 32 * RH * 2; idx += 256) {
      const int c = idx / (RH * 2), rem = idx % (RH * 2);
      const int r = rem >> 1, side = rem & 1;
      const int col = side ? CW - 1 : 0;
      tile[c * PL + r * CW + col] = fetch(c, ih0 + r, iw0 + col);
    }
  } else {
    for (int idx = tid; idx < 32 * RH * CW; idx += 256) {
      const int c = idx / (RH * CW), rem = idx % (RH * CW);
      const int r = rem / CW, col = rem % CW;
      tile[c * PL + r * CW + col] = fetch(c, ih0 + r, iw0 + col);
    }
  }
  __syncthreads();
  const size_t xi_stride = (size_t)a.Q * a.T * 32;
#pragma unroll 1
  for (int it = tid; it < TRB * TWB * 32; it += 256) {
    const int c = it & 31, tl = it >> 5;
    const int tr = tl / TWB, tc = tl % TWB;
    const int tx = tx0 + tc, ty = ty0 + tr;
    if (tx >= a.TW || ty >= a.TH) continue;
    float d[A][A];
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
      for (int j = 0; j < A; ++j) d[i][j] = tile[c * PL + (tr * MM + i) * CW + tc * MM + j];
    float t1[A][A];   // B^T d
    wino_in_rows<MM>(d, t1);
    const size_t t = ((size_t)n * a.TH + ty) * a.TW + tx;
    float* vout = a.V + ((size_t)q * a.T + t) * 32 + c;
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
      for (int j = 0; j < A; ++j) {   // (B^T d) B : column j of B = row j of B^T
        const float s = wino_in_col<MM>(t1, i, j);
        if (!(a.exp & 2) || s == 1.2345e-30f) vout[(size_t)(i * A + j) * xi_stride] = s;
      }
  }
}

// The same transform for the plain source on wide maps (1 x TWB strips, W % 4 == 0), one block walking ROLL
// consecutive tile rows of its 32-channel chunk: while the strip in LDS is transformed and its 36 V planes are stored,
// the next strip's rows are already in flight into registers (12 float4 + 2 halo scalars per thread) -- the load,
// transform and store phases of the one-strip kernel run back to back (measured: load + compute 0.86 ms, compute +
// store 0.65 ms of a 1.42 ms launch), here they overlap inside the block as well as across blocks.
template <int MM, int SRC, int TRB, int TWB>      // SRC 0 plain, 1 gated by the ReLU output `mask` (kept raw in registers too),
                                                  // 2 pooled gradient + argmax codes (routed when the strip is written to LDS)
__global__ __launch_bounds__(256) void wino_input_roll_kernel(WinoInArgs a, int roll) {
  constexpr int A = WinoMat<MM>::A;
  constexpr int RH = TRB * MM + 2, CW = TWB * MM + 2;
  constexpr int PL = (RH * CW) | 1;
  constexpr int V4 = (CW - 2) / 4;
  constexpr int NV = (32 * RH * V4 + 255) / 256, NH = (32 * RH * 2 + 255) / 256;
  __shared__ float tile[32 * PL];
  const int tid = threadIdx.x;
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (a.xcd) {
    const unsigned lin = xcd_contiguous_id(bx + gridDim.x * (by + gridDim.y * bz), gridDim.x * gridDim.y * gridDim.z);
    bx = lin % gridDim.x;
    by = (lin / gridDim.x) % gridDim.y;
    bz = lin / (gridDim.x * gridDim.y);
  }
  const int tx0 = bx * TWB;
  const int n = bz / a.Q, q = bz % a.Q;
  const int iw0 = tx0 * MM - 1;
  const int plane = (SRC == 2) ? a.Hp * a.Wp : a.H * a.W;
  const float* xsrc = a.x;
  size_t img = ((size_t)n * a.C + (size_t)q * 32) * plane;
  if (SRC == 0 && a.cat.n > 0) {          // chunk q lies in ONE of the concatenated tensors (32 | their channel counts)
    int ch = q * 32, chans;
    xsrc = wino_cat_pick(a.cat, ch, chans);
    img = ((size_t)n * chans + ch) * plane;
  }
  const int ty_beg = by * roll * TRB, ty_end = min(a.TH, ty_beg + roll * TRB);      // tile rows [ty_beg, ty_end), TRB per strip

  f32x4 pre[NV], msk[SRC == 1 ? NV : 1];
  float preh[NH], mskh[SRC != 0 ? NH : 1];
  auto issue = [&](int ty) {
    const int ih0 = ty * MM - 1;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int idx = tid + 256 * k;
      const int c = idx / (RH * V4), rem = idx % (RH * V4);
      const int r = rem / V4, v4 = rem % V4;
      const int ih = ih0 + r, iw = iw0 + 1 + v4 * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f}, m = {1.f, 1.f, 1.f, 1.f};
      if (SRC == 2) {
        // 4 strip columns = 2 pooled elements: {g0, g1, code0 | code1 << 8 (as bits), -}
        const int hp = ih >> 1, wq = iw >> 1;
        unsigned cc = 0;
        if (idx < 32 * RH * V4 && q * 32 + c < a.C && ih >= 0 && ih < a.H && iw < a.W && hp < a.Hp && wq < a.Wp) {
          const size_t off = img + (size_t)c * plane + (size_t)hp * a.Wp + wq;
          if (wq + 1 < a.Wp) {
            const f32x2 g2 = *(const f32x2*)(a.x + off);
            v[0] = g2[0]; v[1] = g2[1];
            cc = (unsigned)a.code[off] | ((unsigned)a.code[off + 1] << 8);
          } else {
            v[0] = a.x[off];
            cc = (unsigned)a.code[off];
          }
        }
        v[2] = __uint_as_float(cc);
      } else if (idx < 32 * RH * V4 && q * 32 + c < a.C && ih >= 0 && ih < a.H && iw < a.W) {
        const size_t off = img + (size_t)c * plane + (size_t)ih * a.W + iw;
        v = (FCD_NT_EXP & 2) ? __builtin_nontemporal_load((const f32x4*)(xsrc + off)) : *(const f32x4*)(xsrc + off);
        if (SRC == 1) {
          // bit mask: keep the RAW tile word in the register (decoded in commit): arithmetic on it here would wait for the
          // load and serialise the strip's prefetch (measured: +2 ms per step with the decode at issue time)
          if (a.mbits) m[0] = __uint_as_float((unsigned)a.mbits[(((size_t)n * a.C + q * 32 + c) * a.TH + (ih >> 2)) * a.TW + (iw >> 2)]);
          else m = *(const f32x4*)(a.mask + off);
        }
      }
      pre[k] = v;
      if (SRC == 1) msk[k] = m;
    }
#pragma unroll
    for (int k = 0; k < NH; ++k) {
      const int idx = tid + 256 * k;
      const int c = idx / (RH * 2), rem = idx % (RH * 2);
      const int r = rem >> 1, col = (rem & 1) ? CW - 1 : 0;
      const int ih = ih0 + r, iw = iw0 + col;
      float v = 0.f, m = 1.f;
      if (idx < 32 * RH * 2 && q * 32 + c < a.C && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) {
        if (SRC == 2) {
          const int hp = ih >> 1, wq = iw >> 1;
          if (hp < a.Hp && wq < a.Wp) {
            const size_t off = img + (size_t)c * plane + (size_t)hp * a.Wp + wq;
            v = a.x[off];
            m = ((unsigned)a.code[off] == (unsigned)((((ih & 1) << 1) | (iw & 1)) | 4)) ? 1.f : 0.f;
          }
        } else {
          const size_t off = img + (size_t)c * plane + (size_t)ih * a.W + iw;
          v = xsrc[off];
          if (SRC == 1) {
            if (a.mbits) m = __uint_as_float((unsigned)a.mbits[(((size_t)n * a.C + q * 32 + c) * a.TH + (ih >> 2)) * a.TW + (iw >> 2)]);
            else m = a.mask[off];
          }
        }
      }
      preh[k] = v;
      if (SRC != 0) mskh[k] = m;
    }
  };
  auto commit = [&](int ty) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int idx = tid + 256 * k;
      if (idx >= 32 * RH * V4) continue;
      const int c = idx / (RH * V4), rem = idx % (RH * V4);
      const int r = rem / V4, v4 = rem % V4;
      float* t = tile + c * PL + r * CW + 1 + v4 * 4;
      if (SRC == 2) {
        const unsigned cc = __float_as_uint(pre[k][2]), c0 = cc & 0xffu, c1 = cc >> 8;
        const unsigned rowbit = (unsigned)(((ty * MM - 1 + r) & 1) << 1) | 4u;
        t[0] = c0 == rowbit ? pre[k][0] : 0.f;
        t[1] = c0 == (rowbit | 1u) ? pre[k][0] : 0.f;
        t[2] = c1 == rowbit ? pre[k][1] : 0.f;
        t[3] = c1 == (rowbit | 1u) ? pre[k][1] : 0.f;
      } else {
        if (SRC == 1 && a.mbits) {
          const unsigned wv = __float_as_uint(msk[k][0]) >> (4 * ((ty * MM - 1 + r) & 3));
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] = ((wv >> e) & 1u) ? pre[k][e] : 0.f;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] = (SRC == 1 && !(msk[k][e] > 0.f)) ? 0.f : pre[k][e];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NH; ++k) {
      const int idx = tid + 256 * k;
      if (idx >= 32 * RH * 2) continue;
      const int c = idx / (RH * 2), rem = idx % (RH * 2);
      const int r = rem >> 1, col = (rem & 1) ? CW - 1 : 0;
      if (SRC == 1 && a.mbits) {
        const unsigned bit = (__float_as_uint(mskh[k]) >> (4 * ((ty * MM - 1 + r) & 3) + ((iw0 + col) & 3))) & 1u;
        tile[c * PL + r * CW + col] = bit ? preh[k] : 0.f;
      } else {
        tile[c * PL + r * CW + col] = (SRC != 0 && !(mskh[k] > 0.f)) ? 0.f : preh[k];
      }
    }
  };

  const size_t xi_stride = (size_t)a.Q * a.T * 32;
  issue(ty_beg);
  commit(ty_beg);
  __syncthreads();
#pragma unroll 1
  for (int ty0 = ty_beg; ty0 < ty_end; ty0 += TRB) {
    if (ty0 + TRB < ty_end) issue(ty0 + TRB);
#pragma unroll 1
    for (int it = tid; it < TRB * TWB * 32; it += 256) {
      const int c = it & 31, tl = it >> 5;
      const int tr = tl / TWB, tc = tl % TWB;
      const int tx = tx0 + tc, ty = ty0 + tr;
      if (tx >= a.TW || ty >= a.TH) continue;
      float d[A][A];
#pragma unroll
      for (int i = 0; i < A; ++i)
#pragma unroll
        for (int j = 0; j < A; ++j) d[i][j] = tile[c * PL + (tr * MM + i) * CW + tc * MM + j];
      float t1[A][A];
      wino_in_rows<MM>(d, t1);
      const size_t t = ((size_t)n * a.TH + ty) * a.TW + tx;
      float* vout = a.V + ((size_t)q * a.T + t) * 32 + c;
#pragma unroll
      for (int i = 0; i < A; ++i)
#pragma unroll
        for (int j = 0; j < A; ++j) vout[(size_t)(i * A + j) * xi_stride] = wino_in_col<MM>(t1, i, j);
    }
    if (ty0 + TRB < ty_end) {
      __syncthreads();       // every thread is done reading this strip
      commit(ty0 + TRB);
      __syncthreads();
    }
  }
}

// --------------------------------------------------------------------------------------------
// batched TN GEMM:  C[b][m][n] = sum_k A[b][m][k] * B[b][k / 32][n][k % 32]
// 128 x 128 block tile, 4 waves of 64 x 64 (2 x 2 MFMA 32x32x2 tiles), 32-channel chunks.  Both
// operand slabs arrive by global_load_lds (16 B / lane).  LDS image per operand and chunk:
// [lane half h][row][4 units of 16 B], unit j of a row stored at j ^ ((row >> 3) & 3) -- the swizzle
// is applied on the SOURCE address of the lane-linear DMA -- so the four ds_read_b128 with which a
// lane fetches its 16 operands of the chunk are bank-conflict free without padding.  Two LDS stages
// as distinct objects, chunk loop unrolled by two (see conv_igemm.hip for why).
#ifndef FCD_GEXP
#define FCD_GEXP 0   // diagnostic builds only (wrong results): 1 no DMA in the loop, 2 operands from registers, 4 no barrier
#endif
template <int WM, int WN>      // waves along M / N, each 64 x 64: block tile (64 WM) x (64 WN)
__global__ __launch_bounds__(64 * WM * WN, (WM * WN >= 8) ? 1 : 2) void wino_gemm_kernel(WinoGemmArgs a) {
  constexpr int KC = 32;                   // reduction elements per pipeline stage
  constexpr int NW = WM * WN;
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int UH = KC / 8;               // 16-B units per lane half and row
  constexpr int PPW_A = BM * UH * 2 / 64 / NW;   // 1 KiB wave-instructions per wave, operand and stage
  constexpr int PPW_B = BN * UH * 2 / 64 / NW;
  static_assert(PPW_A * NW * 64 == BM * UH * 2 && PPW_B * NW * 64 == BN * UH * 2, "DMA split");
  // ds_read_b128 is served in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md,
  // LDS): with a 64-B row pitch the 16 rows of a group hit 16 distinct 16-B slots of the 256-B bank row iff the
  // unit permutation differs between the row blocks {0,3,5,6} resp. {1,2,4,7} of 4 rows: swizzle = (row >> 3) & 3
  constexpr int SWS = 3;
  __shared__ __attribute__((aligned(16))) float sa0[BM * KC];
  __shared__ __attribute__((aligned(16))) float sa1[BM * KC];
  __shared__ __attribute__((aligned(16))) float sb0[BN * KC];
  __shared__ __attribute__((aligned(16))) float sb1[BN * KC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave / WN, wn = wave % WN;
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
  const int mt = (int)(v % (unsigned)a.m_tiles), nt = (int)(v / (unsigned)a.m_tiles);
  const int m0 = mt * BM, n0 = nt * BN;
  const int q_all = a.Kc / KC;
  const int q_beg = (int)blockIdx.z * a.stages_per_split;
  const int Q = min(a.stages_per_split, q_all - q_beg);      // pipeline stages per batch of this block
  const int b_first = (int)blockIdx.y * a.xb;
  const int nb = min(a.xb, a.batches - b_first);             // batches of this block
  const int S = nb * Q;                                      // stages of the whole pipeline
  // 64-bit part of the addresses in the (scalar) bases, the per-lane offsets stay 32-bit
  const float* Ab = a.A + (size_t)b_first * a.a_batch + (size_t)q_beg * KC + (size_t)m0 * a.a_ld;
  const float* Bb = a.B + (size_t)b_first * a.b_batch + (size_t)q_beg * a.b_adv + (size_t)n0 * a.b_ld;

  int a_goff[PPW_A], b_goff[PPW_B];
#pragma unroll
  for (int j = 0; j < PPW_A; ++j) {
    const int u = (wave + NW * j) * 64 + lane;
    const int h = u / (BM * UH), row = (u / UH) % BM, pj = u % UH;
    const int jl = pj ^ ((row >> SWS) & (UH - 1));
    a_goff[j] = (int)(min(row, a.M - 1 - m0) * a.a_ld) + h * (KC / 2) + jl * 4;
  }
#pragma unroll
  for (int j = 0; j < PPW_B; ++j) {
    const int u = (wave + NW * j) * 64 + lane;
    const int h = u / (BN * UH), row = (u / UH) % BN, pj = u % UH;
    const int jl = pj ^ ((row >> SWS) & (UH - 1));
    b_goff[j] = (int)(min(row, a.N - 1 - n0) * a.b_ld) + h * (KC / 2) + jl * 4;
  }

  int aoff[2], boff[2], uoff[UH];
  const int sw = (l31 >> SWS) & (UH - 1);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    aoff[i] = (half * BM + wm * 64 + i * 32 + l31) * (UH * 4);
    boff[i] = (half * BN + wn * 64 + i * 32 + l31) * (UH * 4);
  }
#pragma unroll
  for (int j = 0; j < UH; ++j) uoff[j] = (j ^ sw) * 4;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fq / fb: chunk and batch of the NEXT slab pair to fetch.  The prefetch runs one stage ahead of the MFMAs and does
  // not stop at a batch boundary: while the last chunk of batch b is multiplied and its C tile stored, chunk 0 of batch
  // b + 1 is already in flight.  (Launcher guarantees an even Q whenever nb > 1, so every batch starts on stage 0.)
  int fq = 0, fb = 0;
#define WG_DMA(SA, SB)                                                                           \
  {                                                                                              \
    const float* as_ = Ab + (size_t)fb * a.a_batch + (size_t)fq * KC;                            \
    const float* bs_ = Bb + (size_t)fb * a.b_batch + (size_t)fq * a.b_adv;                       \
    _Pragma("unroll") for (int j = 0; j < PPW_A; ++j)                                            \
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(as_ + a_goff[j]),                           \
                                       (lds_void_t*)((SA) + (wave + NW * j) * 256), 16, 0, 0);   \
    _Pragma("unroll") for (int j = 0; j < PPW_B; ++j)                                            \
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(bs_ + b_goff[j]),                           \
                                       (lds_void_t*)((SB) + (wave + NW * j) * 256), 16, 0, 0);   \
    if (++fq == Q) { fq = 0; ++fb; }                                                             \
  }
#define WG_STEP(SA, SB, SAN, SBN)                                                                \
  {                                                                                              \
    if (!(FCD_GEXP & 1)) if (fb < nb) WG_DMA(SAN, SBN)                                           \
    _Pragma("unroll") for (int j4 = 0; j4 < UH; ++j4) {                                          \
      f32x4 av[2], bv[2];                                                                        \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                            \
        av[i] = (FCD_GEXP & 2) ? f32x4{(float)lane, 1.f, 2.f, (float)j4} : *(const f32x4*)((SA) + aoff[i] + uoff[j4]); \
        bv[i] = (FCD_GEXP & 2) ? f32x4{(float)i, 1.f, (float)lane, 3.f} : *(const f32x4*)((SB) + boff[i] + uoff[j4]); \
      }                                                                                          \
      _Pragma("unroll") for (int e = 0; e < 4; ++e)                                              \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                            \
          _Pragma("unroll") for (int j = 0; j < 2; ++j)                                          \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv[j][e], acc[i][j], 0, 0, 0); \
    }                                                                                            \
    if (!(FCD_GEXP & 4)) __syncthreads();                                                        \
  }

  WG_DMA(sa0, sb0)
  __syncthreads();
#pragma unroll 1
  for (int cb = 0; cb < nb; ++cb) {
    for (int qc = 0; qc < Q; qc += 2) {
      WG_STEP(sa0, sb0, sa1, sb1)
      if (qc + 1 < Q) WG_STEP(sa1, sb1, sa0, sb0)
    }
    // (the row pitch goes through an opaque copy: otherwise the 64 store offsets are hoisted out of the batch loop
    // and held in registers across the MFMA loop -- +50 VGPRs and 60 spilled SGPRs)
    int ldc = a.N;
    asm volatile("" : "+s"(ldc));
    float* Cb = a.C + ((size_t)blockIdx.z * a.batches + (b_first + cb)) * a.M * ldc;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = n0 + wn * 64 + j * 32 + l31;
          if (m < a.M && n < ldc) Cb[(size_t)m * ldc + n] = acc[i][j][r];
          acc[i][j][r] = 0.f;
        }
      }
  }
#undef WG_STEP
#undef WG_DMA
}

// --------------------------------------------------------------------------------------------
// The same batched GEMM on the bf16 matrix pipe (16x the fp32 MFMA rate) WITHOUT giving up fp32 arithmetic: every fp32
// operand is split exactly into three bf16 parts x = h + m + l (round-to-nearest at each step; the residuals x - h and
// x - h - m are exact in fp32 and the last one fits the 8-bit significand), a bf16 x bf16 product is exact in the fp32
// accumulator, and the six partial products of weight >= 2^-16 are accumulated in fp32:
//     a b  =  ah bh + (ah bm + am bh) + (am bm + ah bl + al bh)  +  [am bl + al bm + al bl  -- dropped, <= 2^-24 |a b|]
// i.e. each product carries a relative error of ~2^-24, the size of ONE fp32 rounding, before the same fp32 accumulation
// the v_mfma_f32_32x32x2_f32 chain performs.  A (the transformed filters) is split once per weight version by the pack
// kernel (three bf16 planes); B (the transformed activations V, fp32 in HBM and in LDS) is split in registers right
// before the MFMAs.  Lane (row, half) owns reduction elements half*16 .. half*16+15 of the 32-chunk, 8 per MFMA step.
// Where in a stage the LDS-DMA of the NEXT stage is issued (FCD_YEXP, results identical): [r4] after the SECOND of the four MFMA groups
// (bit 4, the default) instead of at the top of the stage -- measured on conv3_x / conv4_x / conv3_1: 1.58 -> 1.485, 1.297 -> 1.257,
// 0.727 -> 0.699 ms (the requests no longer sit in front of the stage's own LDS operand reads, and half a stage is still enough for
// them to land); after the first group (bit 2) 1.53 / 1.285 / 0.706, after the third (bit 8) 1.51 / 1.264 / 0.709.  Bit 1 = s_setprio 2
// over the MFMA part of a stage: no effect.  Bit 16 = the same move in the 128-tile kernel: no gain (9.73 - 9.86 vs 9.69 - 9.72 ms over the
// 12 layer shapes of tools/bench_wino_gemm.py).
// YG_TIME: attribution build of the 256-tile split GEMM (tools/gemm_segments.py): every wave stamps s_memtime at the segment borders
// of its stage loop -- top of the stage, in front of the first MFMA group (after the stage's first LDS reads and the split of step 0),
// behind the last MFMA group, behind the stage barrier -- and around the C store of a batch, and writes the per-segment cycle sums
// {prologue, MFMA groups, DMA wait + barrier, C store, total} to a debug buffer.  Results stay correct; the stamps are scheduling
// fences and wait for the wave's outstanding LDS reads.  Never defined in the product build.
#ifndef YG_TIME
#define YG_TIME 0
#endif
#ifndef YG_PLAINC
#define YG_PLAINC 0     // experiment: the 256-tile kernel's C tile with plain instead of non-temporal stores
#endif
#if YG_TIME
#define YG_T(var) __builtin_amdgcn_sched_barrier(0); const unsigned long long var = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#define YG_TACC(slot, t1, t0) ytacc[slot] += (t1) - (t0);
unsigned long long* g_yg_tbuf = nullptr;
extern "C" void fcd_wino_gemm_time_buf(void* p) { g_yg_tbuf = (unsigned long long*)p; }
#else
#define YG_T(var)
#define YG_TACC(slot, t1, t0)
#endif
#ifndef FCD_YEXP
#define FCD_YEXP 4
#endif
#ifndef FCD_SEXP
#define FCD_SEXP 0   // diagnostic builds only (wrong results): 1 no operand split, 2 one MFMA of the six, 4 no barrier, 8 no DMA,
                     // 64 no A-operand DMA (128-tile kernel), 128 no C stores (128-tile kernel)
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  // low half << 16 through v_perm_b32: written as a shift, LLVM narrows the packed conversion to its low element and emits
  // a second v_cvt_pk_bf16_f32 for it (13 instead of 11 VALU per pair)
  const bf16x2 hp = {(__bf16)x0, (__bf16)x1};
  h = __builtin_bit_cast(unsigned, hp);
  const float r0 = x0 - __uint_as_float(__builtin_amdgcn_perm(h, 0u, 0x05040c0cu)), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  const bf16x2 mp = {(__bf16)r0, (__bf16)r1};
  m = __builtin_bit_cast(unsigned, mp);
  const float q0 = r0 - __uint_as_float(__builtin_amdgcn_perm(m, 0u, 0x05040c0cu)), q1 = r1 - __uint_as_float(m & 0xffff0000u);
  const bf16x2 lp = {(__bf16)q0, (__bf16)q1};
  l = __builtin_bit_cast(unsigned, lp);
}

// BT: the B operand is read TRANSPOSED from the forward pass's transformed activations V [xi][channel chunk][tile][32 ch]
// (weight gradient: GEMM row n = channel, reduction = tiles).  A stage's slab is four contiguous 4-KiB blocks [32 tiles][32
// ch] (one per 32-channel chunk of the 128 columns), DMA'd as they lie; a lane then picks its 16 reduction elements with
// ds_read_b32 (lanes = consecutive channels: conflict-free) instead of four ds_read_b128.  This is what lets the weight
// gradient skip its own input transform (wino_wg_input_kernel: x read again, 2.25x written, 1.6 ms per step).
template <int WM, int WN, bool BT = false>      // waves along M / N, each 64 x 64
__global__ __launch_bounds__(64 * WM * WN, (WM * WN >= 8) ? 1 : 2) void wino_gemm_split_kernel(WinoGemmArgs a) {
  constexpr int KC = 32;
  constexpr int NW = WM * WN;
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int UH = KC / 8;                       // B: 16-B units (4 fp32) per lane half and row
  constexpr int A_UNITS = 3 * 2 * BM * 2;          // A: [plane][half][row][2 units of 8 bf16]
  constexpr int PPW_A = A_UNITS / 64 / NW;
  constexpr int PPW_B = BN * UH * 2 / 64 / NW;
  static_assert(PPW_A * NW * 64 == A_UNITS && PPW_B * NW * 64 == BN * UH * 2, "DMA split");
  constexpr int SWS = 3;
  __shared__ __attribute__((aligned(16))) unsigned short sa0[A_UNITS * 8];
  __shared__ __attribute__((aligned(16))) unsigned short sa1[A_UNITS * 8];
  __shared__ __attribute__((aligned(16))) float sb0[BN * KC];
  __shared__ __attribute__((aligned(16))) float sb1[BN * KC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave / WN, wn = wave % WN;
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
  const int mt = (int)(v % (unsigned)a.m_tiles), nt = (int)(v / (unsigned)a.m_tiles);
  const int m0 = mt * BM, n0 = nt * BN;
  const int q_all = a.Kc / KC;
  const int q_beg = (int)blockIdx.z * a.stages_per_split;      // split of the reduction (weight gradient)
  const int Q = min(a.stages_per_split, q_all - q_beg);
  const int b_first = (int)blockIdx.y * a.xb;
  const int nb = min(a.xb, a.batches - b_first);
  const unsigned short* Ab = a.As + (size_t)b_first * a.a_batch + (size_t)q_beg * KC + (size_t)m0 * a.a_ld;
  const float* Bb = BT ? a.B + (size_t)b_first * a.b_batch
                       : a.B + (size_t)b_first * a.b_batch + (size_t)q_beg * a.b_adv + (size_t)n0 * a.b_ld;

  // A slab in LDS: unit u = ((plane * 2 + half) * BM + row) * 2 + pj, 32-B row pitch.  A 16-lane ds_read_b128 group
  // ({0-3,12-15,20-27}, {4-11,16-19,28-31}) covers 16 distinct 16-B slots iff rows 16..31 take the other unit parity.
  long long a_goff[PPW_A];
  int b_goff[PPW_B];
#pragma unroll
  for (int j = 0; j < PPW_A; ++j) {
    const int u = (wave + NW * j) * 64 + lane;
    const int plane = u / (2 * BM * 2), rem = u % (2 * BM * 2);
    const int h = rem / (BM * 2), row = (rem >> 1) % BM, pj = rem & 1;
    const int jl = pj ^ ((row >> 4) & 1);
    a_goff[j] = plane * a.as_plane + (long long)min(row, a.M - 1 - m0) * a.a_ld + h * (KC / 2) + jl * 8;
  }
  int bt_t[PPW_B];                 // BT: tile (reduction element) of the stage this unit holds
#pragma unroll
  for (int j = 0; j < PPW_B; ++j) {
    const int u = (wave + NW * j) * 64 + lane;
    if (BT) {                      // unit u of the slab [chunk][32 tiles][8 units of 4 channels]
      const int chunk = min(n0 / 32 + (u >> 8), a.N / 32 - 1);
      bt_t[j] = (u >> 3) & 31;
      b_goff[j] = (int)((long long)chunk * a.bt_T * 32) + (u & 7) * 4;      // + tile * 32, per stage
    } else {
      const int h = u / (BN * UH), row = (u / UH) % BN, pj = u % UH;
      const int jl = pj ^ ((row >> SWS) & (UH - 1));
      b_goff[j] = (int)(min(row, a.N - 1 - n0) * a.b_ld) + h * (KC / 2) + jl * 4;
      bt_t[j] = 0;
    }
  }

  int aoff[2], boff[2];
  const int swb = (l31 >> SWS) & (UH - 1), swa = (l31 >> 4) & 1;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    aoff[i] = ((half * BM + wm * 64 + i * 32 + l31) * 2) * 8;      // in bf16 elements, plane 0, unit 0
    boff[i] = BT ? ((wn * 2 + i) * 32 + half * 16) * 32 + l31      // [chunk][tile = half * 16 + e][channel l31]
                 : (half * BN + wn * 64 + i * 32 + l31) * (UH * 4);
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int fq = 0, fb = 0;
#define WS_DMA(SA, SB)                                                                           \
  {                                                                                              \
    const unsigned short* as_ = Ab + (size_t)fb * a.a_batch + (size_t)fq * KC;                   \
    const float* bs_ = BT ? Bb + (size_t)fb * a.b_batch : Bb + (size_t)fb * a.b_batch + (size_t)fq * a.b_adv; \
    if (!(FCD_SEXP & 64)) { _Pragma("unroll") for (int j = 0; j < PPW_A; ++j)                    \
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(as_ + a_goff[j]),                           \
                                       (lds_void_t*)((SA) + (wave + NW * j) * 512), 16, 0, 0); } \
    _Pragma("unroll") for (int j = 0; j < PPW_B; ++j) {                                          \
      /* BT: tile (q_beg + fq) * 32 + t of the forward V, clamped to the last real tile (A is zero there) */ \
      const long long bo_ = BT ? (long long)b_goff[j] + min((long long)(q_beg + fq) * KC + bt_t[j], a.bt_T - 1) * 32 \
                               : (long long)b_goff[j];                                           \
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(bs_ + bo_),                                 \
                                       (lds_void_t*)((SB) + (wave + NW * j) * 256), 16, 0, 0);   \
    }                                                                                            \
    if (++fq == Q) { fq = 0; ++fb; }                                                             \
  }
#define WS_MFMA(AV, BV)                                                                          \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                  \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, AV[i]),     \
                                                          __builtin_bit_cast(bf16x8, BV[j]), acc[i][j], 0, 0, 0);
#define WS_SPLIT(X0, X1, H, M, L)                                                                \
  {                                                                                              \
    unsigned th[4], tm[4], tl[4];                                                                \
    if (FCD_SEXP & 1) {                                                                          \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                            \
        th[e] = __float_as_uint(X0[e]); tm[e] = __float_as_uint(X1[e]); tl[e] = th[e] ^ tm[e]; } \
    } else {                                                                                     \
      split_pair(X0[0], X0[1], th[0], tm[0], tl[0]);                                             \
      split_pair(X0[2], X0[3], th[1], tm[1], tl[1]);                                             \
      split_pair(X1[0], X1[1], th[2], tm[2], tl[2]);                                             \
      split_pair(X1[2], X1[3], th[3], tm[3], tl[3]);                                             \
    }                                                                                            \
    H = u32x4{th[0], th[1], th[2], th[3]};                                                       \
    M = u32x4{tm[0], tm[1], tm[2], tm[3]};                                                       \
    L = u32x4{tl[0], tl[1], tl[2], tl[3]};                                                       \
  }
#define WS_SIX(AH, AM, AL, BH, BM_, BL)                                                          \
  if (!(FCD_SEXP & 2)) { WS_MFMA(AL, BH) WS_MFMA(AH, BL) WS_MFMA(AM, BM_) WS_MFMA(AM, BH) WS_MFMA(AH, BM_) } \
  WS_MFMA(AH, BH)
// One 32-element stage = two MFMA steps.  Program order = the software pipeline the scheduler is then pinned to with
// sched_group_barrier: all LDS reads of the stage, the split of step 0, then the 24 MFMAs of step 0 with the split of
// step 1 in their shadows (4 VALU per MFMA), then the 24 MFMAs of step 1.
#define WS_STEP(SA, SB, SAN, SBN)                                                                \
  {                                                                                              \
    if (!(FCD_SEXP & 8) && !(FCD_YEXP & 16)) if (fb < nb) WS_DMA(SAN, SBN)                       \
    f32x4 xr[2][2][2];                                                                           \
    u32x4 ah[2][2], am[2][2], al[2][2], bh[2][2], bm[2][2], bl[2][2];                            \
    _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                            \
        if (BT) {                                                                                \
          const float* bp_ = (SB) + boff[i] + s * 8 * 32;                                        \
          xr[s][i][0] = f32x4{bp_[0], bp_[32], bp_[64], bp_[96]};                                \
          xr[s][i][1] = f32x4{bp_[128], bp_[160], bp_[192], bp_[224]};                           \
        } else {                                                                                 \
        xr[s][i][0] = (FCD_SEXP & 32) ? f32x4{(float)lane, 1.f, (float)s, 2.f} : *(const f32x4*)((SB) + boff[i] + (((2 * s) ^ swb) * 4));     \
        xr[s][i][1] = (FCD_SEXP & 32) ? f32x4{(float)i, 3.f, (float)lane, 2.f} : *(const f32x4*)((SB) + boff[i] + (((2 * s + 1) ^ swb) * 4)); \
        }                                                                                        \
      }                                                                                          \
    _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                            \
        const unsigned short* ap = (SA) + aoff[i] + ((s ^ swa) * 8);                             \
        ah[s][i] = (FCD_SEXP & 32) ? u32x4{(unsigned)lane, 1u, (unsigned)s, 7u} : *(const u32x4*)(ap);                        \
        am[s][i] = (FCD_SEXP & 32) ? u32x4{(unsigned)i, 1u, (unsigned)lane, 7u} : *(const u32x4*)(ap + 2 * BM * 2 * 8);       \
        al[s][i] = (FCD_SEXP & 32) ? u32x4{(unsigned)lane, 3u, (unsigned)s, 9u} : *(const u32x4*)(ap + 2 * (2 * BM * 2 * 8)); \
      }                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) WS_SPLIT(xr[0][i][0], xr[0][i][1], bh[0][i], bm[0][i], bl[0][i]) \
    WS_SIX(ah[0], am[0], al[0], bh[0], bm[0], bl[0])                                             \
    if ((FCD_YEXP & 16) && !(FCD_SEXP & 8)) if (fb < nb) WS_DMA(SAN, SBN)                        \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) WS_SPLIT(xr[1][i][0], xr[1][i][1], bh[1][i], bm[1][i], bl[1][i]) \
    WS_SIX(ah[1], am[1], al[1], bh[1], bm[1], bl[1])                                             \
    if (!(FCD_SEXP & 16)) {                                                                      \
      __builtin_amdgcn_sched_group_barrier(0x100, BT ? 44 : 20, 0);                              \
      __builtin_amdgcn_sched_group_barrier(0x002, 88, 0);                                        \
      _Pragma("unroll") for (int k = 0; k < 22; ++k) {                                           \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                       \
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                                       \
      }                                                                                          \
      __builtin_amdgcn_sched_group_barrier(0x008, 26, 0);                                        \
    }                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    if (!(FCD_SEXP & 4)) __syncthreads();                                                        \
  }

  WS_DMA(sa0, sb0)
  __syncthreads();
#pragma unroll 1
  for (int cb = 0; cb < nb; ++cb) {
    for (int qc = 0; qc < Q; qc += 2) {
      WS_STEP(sa0, sb0, sa1, sb1)
      if (qc + 1 < Q) WS_STEP(sa1, sb1, sa0, sb0)
    }
    if (a.c_blk) {             // MFMA-native blocks: registers 4g .. 4g + 3 of a lane are rows (8g + 4 half) + 0..3 of ITS column
      float* Cb = a.C + (size_t)(b_first + cb) * a.c_batch;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int mb = (m0 + wm * 64 + i * 32) >> 5, tb = (n0 + wn * 64 + j * 32) >> 5;
          if (mb < a.c_mblk && tb < a.c_tblk) {
            float* blk = Cb + ((size_t)mb * a.c_tblk + tb) * 1024 + (half * 4 * 32 + l31) * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g)
              { const f32x4 cv_ = f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                if (FCD_NT_EXP & 8) __builtin_nontemporal_store(cv_, (f32x4*)(blk + g * 128)); else *(f32x4*)(blk + g * 128) = cv_; }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
      continue;
    }
    int ldc = a.N;
    asm volatile("" : "+s"(ldc));
    float* Cb = a.C + ((size_t)blockIdx.z * a.batches + (b_first + cb)) * a.M * ldc;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = n0 + wn * 64 + j * 32 + l31;
          if (m < a.M && n < ldc && (!(FCD_SEXP & 128) || acc[i][j][r] == 12345.f)) Cb[(size_t)m * ldc + n] = acc[i][j][r];
          acc[i][j][r] = 0.f;
        }
      }
  }
#undef WS_STEP
#undef WS_MFMA
#undef WS_DMA
}

// [r3] Filter-resident split GEMM for the ONE-row-tile launches with a short reduction (<= 128 rows, Kc <= 128: VGG conv2_x,
// the Segmentor's 128-channel stages).  In the 128 x 128 kernel every column-tile workgroup streams the same 24-KB filter
// planes per stage again -- 60 % of its L2 -> LDS bytes; with the A stream cut out (diagnostic build) conv2_2 ran 1.60
// instead of 1.94 ms.  Here the planes of ALL Kc / 32 stages of a transform position stay in LDS (4 x 24 KB), the
// workgroup (8 waves, 128 x 256 tile, each wave 64 x 64 as above) walks a contiguous range of column tiles of that
// position and only V moves: two 32-KB B stages, 96 + 64 KB = the whole LDS, one workgroup per CU.
__global__ __launch_bounds__(512, 1) void wino_gemm_split_res_kernel(WinoGemmArgs a) {
  constexpr int KC = 32, WM = 2, WN = 4, NW = 8, BM = 128, BN = 256, QMAX = 4;
  constexpr int UH = KC / 8;
  constexpr int A_UNITS = 3 * 2 * BM * 2;          // per stage: [plane][half][row][2 units of 8 bf16]
  constexpr int PPW_A = A_UNITS / 64 / NW;         // 3
  constexpr int PPW_B = BN * UH * 2 / 64 / NW;     // 4
  static_assert(PPW_A * NW * 64 == A_UNITS && PPW_B * NW * 64 == BN * UH * 2, "DMA split");
  constexpr int SWS = 3;
  __shared__ __attribute__((aligned(16))) unsigned short sa[QMAX * A_UNITS * 8];
  __shared__ __attribute__((aligned(16))) float sb0[BN * KC];          // two arrays, not one ring: the waitcnt pass only lets a
  __shared__ __attribute__((aligned(16))) float sb1[BN * KC];          // ds_read pass an LDS-DMA in flight to a DIFFERENT object
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave / WN, wn = wave % WN;
  const int Q = a.Kc / KC;                          // <= QMAX
  const int xi = blockIdx.y;
  const int chunks = gridDim.x;
  const int nt_beg = (int)((long long)a.n_tiles * blockIdx.x / chunks), nt_end = (int)((long long)a.n_tiles * (blockIdx.x + 1) / chunks);
  if (nt_beg >= nt_end) return;
  const unsigned short* Ab = a.As + (size_t)xi * a.a_batch;
  const float* Bb = a.B + (size_t)xi * a.b_batch;

  long long a_goff[PPW_A];
  int b_row[PPW_B], b_in[PPW_B];
#pragma unroll
  for (int j = 0; j < PPW_A; ++j) {
    const int u = (wave + NW * j) * 64 + lane;
    const int plane = u / (2 * BM * 2), rem = u % (2 * BM * 2);
    const int h = rem / (BM * 2), row = (rem >> 1) % BM, pj = rem & 1;
    const int jl = pj ^ ((row >> 4) & 1);
    a_goff[j] = plane * a.as_plane + (long long)min(row, a.M - 1) * a.a_ld + h * (KC / 2) + jl * 8;
  }
#pragma unroll
  for (int j = 0; j < PPW_B; ++j) {
    const int u = (wave + NW * j) * 64 + lane;
    const int h = u / (BN * UH), row = (u / UH) % BN, pj = u % UH;
    const int jl = pj ^ ((row >> SWS) & (UH - 1));
    b_row[j] = row;
    b_in[j] = h * (KC / 2) + jl * 4;
  }
  int aoff[2], boff[2];
  const int swb = (l31 >> SWS) & (UH - 1), swa = (l31 >> 4) & 1;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    aoff[i] = ((half * BM + wm * 64 + i * 32 + l31) * 2) * 8;
    boff[i] = (half * BN + wn * 64 + i * 32 + l31) * (UH * 4);
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // all filter stages of this transform position, once
  for (int q = 0; q < Q; ++q) {
#pragma unroll
    for (int j = 0; j < PPW_A; ++j)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(Ab + (size_t)q * KC + a_goff[j]),
                                       (lds_void_t*)(sa + (size_t)q * (A_UNITS * 8) + (wave + NW * j) * 512), 16, 0, 0);
  }
  int fq = 0, fnt = nt_beg;             // next B stage to fetch
#define R_DMA(SB)                                                                                \
  {                                                                                              \
    const float* bs_ = Bb + (size_t)fq * a.b_adv;                                                \
    const int n0f = fnt * BN;                                                                    \
    _Pragma("unroll") for (int j = 0; j < PPW_B; ++j)                                            \
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(bs_ + (size_t)min(n0f + b_row[j], a.N - 1) * a.b_ld + b_in[j]), \
                                       (lds_void_t*)((SB) + (wave + NW * j) * 256), 16, 0, (FCD_NT_EXP & 4) ? 2 : 0);   \
    if (++fq == Q) { fq = 0; ++fnt; }                                                            \
  }
#define R_MFMA(AV, BV)                                                                           \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                  \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, AV[i]),     \
                                                          __builtin_bit_cast(bf16x8, BV[j]), acc[i][j], 0, 0, 0);
#define R_SIX(AH, AM, AL, BH, BM_, BL)                                                           \
  R_MFMA(AL, BH) R_MFMA(AH, BL) R_MFMA(AM, BM_) R_MFMA(AM, BH) R_MFMA(AH, BM_) R_MFMA(AH, BH)
#define R_STEP(SA, SB, SBN)                                                                      \
  {                                                                                              \
    if (fnt < nt_end) R_DMA(SBN)                                                                 \
    f32x4 xr[2][2][2];                                                                           \
    u32x4 ah[2][2], am[2][2], al[2][2], bh[2][2], bm[2][2], bl[2][2];                            \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                             \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                            \
        xr[s_][i][0] = *(const f32x4*)((SB) + boff[i] + (((2 * s_) ^ swb) * 4));                 \
        xr[s_][i][1] = *(const f32x4*)((SB) + boff[i] + (((2 * s_ + 1) ^ swb) * 4));             \
      }                                                                                          \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                             \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                            \
        const unsigned short* ap = (SA) + aoff[i] + ((s_ ^ swa) * 8);                            \
        ah[s_][i] = *(const u32x4*)(ap);                                                         \
        am[s_][i] = *(const u32x4*)(ap + 2 * BM * 2 * 8);                                        \
        al[s_][i] = *(const u32x4*)(ap + 2 * (2 * BM * 2 * 8));                                  \
      }                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) WS_SPLIT(xr[0][i][0], xr[0][i][1], bh[0][i], bm[0][i], bl[0][i]) \
    R_SIX(ah[0], am[0], al[0], bh[0], bm[0], bl[0])                                              \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) WS_SPLIT(xr[1][i][0], xr[1][i][1], bh[1][i], bm[1][i], bl[1][i]) \
    R_SIX(ah[1], am[1], al[1], bh[1], bm[1], bl[1])                                              \
    __builtin_amdgcn_sched_group_barrier(0x100, 20, 0);                                          \
    __builtin_amdgcn_sched_group_barrier(0x002, 88, 0);                                          \
    _Pragma("unroll") for (int k = 0; k < 22; ++k) {                                             \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                                         \
    }                                                                                            \
    __builtin_amdgcn_sched_group_barrier(0x008, 26, 0);                                          \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    __syncthreads();                                                                             \
  }

  // Stage stream of the whole column range, two stages per trip (B buffers alternate; a tile's Q may be odd, so the tile end
  // is tested after every stage).  An `if (parity) STEP(sb0, sb1) else STEP(sb1, sb0)` diamond inside the q loop made LLVM
  // give the two copies different accumulator registers and move all 64 across every stage (7.5 VALU per MFMA, PMC).
#define R_TAIL                                                                                   \
  if (++q == Q) {                                                                                \
    q = 0;                                                                                       \
    /* C tile of column tile nt, in 32 x 32 MFMA-native blocks (this kernel only runs with c_blk) */ \
    float* Cb = a.C + (size_t)xi * a.c_batch;                                                    \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                            \
        const int mb = (wm * 64 + i * 32) >> 5, tb = (nt * BN + wn * 64 + j * 32) >> 5;          \
        if (mb < a.c_mblk && tb < a.c_tblk) {                                                    \
          float* blk = Cb + ((size_t)mb * a.c_tblk + tb) * 1024 + (half * 4 * 32 + l31) * 4;     \
          _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                        \
            const f32x4 cv_ = f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]}; \
            if (FCD_NT_EXP & 8) __builtin_nontemporal_store(cv_, (f32x4*)(blk + g * 128)); else *(f32x4*)(blk + g * 128) = cv_; \
          }                                                                                      \
        }                                                                                        \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;                       \
      }                                                                                          \
    ++nt;                                                                                        \
  }
  R_DMA(sb0)
  __syncthreads();
  const int S = (nt_end - nt_beg) * Q;
  int q = 0, nt = nt_beg;
#pragma unroll 1
  for (int st = 0; st < S; st += 2) {
    R_STEP(sa + (size_t)q * (A_UNITS * 8), sb0, sb1)
    R_TAIL
    if (st + 1 < S) {
      R_STEP(sa + (size_t)q * (A_UNITS * 8), sb1, sb0)
      R_TAIL
    }
  }
#undef R_TAIL
#undef R_STEP
#undef R_SIX
#undef R_MFMA
#undef R_DMA
}

// The split GEMM on 256 x 256 workgroup tiles.  With the matrix pipe 2.7x faster per fp32-equivalent FLOP the 128 x 128
// kernel above is bound by the operand stream L2 -> LDS (40 KB and 40 LDS-DMA wave-instructions per 1 MFLOP stage); a
// 256 x 256 tile halves both per FLOP.  8 waves (2 x 4), each 128 x 64 (4 x 2 MFMA blocks, 128 accumulator registers),
// two 80-KB LDS stages of 32 reduction elements = the whole 160-KB LDS, one workgroup per CU.  A stage is 96 MFMAs per
// wave in four groups of 24 (step 0 / 1 x upper / lower 64 rows of the wave tile); the A fragments of the next group
// and the split of step 1's B fragments are issued in the shadow of the running group.
template <int DUMMY>
__global__ __launch_bounds__(512, 1) void wino_gemm_split256_kernel(WinoGemmArgs a) {
  constexpr int KC = 32, BM = 256, BN = 256, NW = 8;
  constexpr int UH = KC / 8;
  constexpr int A_UNITS = 3 * 2 * BM * 2;          // [plane][half][row][2 units of 8 bf16]
  constexpr int B_UNITS = 2 * BN * UH;             // [half][row][4 units of 4 fp32]
  constexpr int PPW_A = A_UNITS / 64 / NW, PPW_B = B_UNITS / 64 / NW;     // 6, 4
  static_assert(PPW_A * NW * 64 == A_UNITS && PPW_B * NW * 64 == B_UNITS, "DMA split");
  constexpr int SWS = 3;
  __shared__ __attribute__((aligned(16))) unsigned short sa0[A_UNITS * 8];
  __shared__ __attribute__((aligned(16))) unsigned short sa1[A_UNITS * 8];
  __shared__ __attribute__((aligned(16))) float sb0[BN * KC];
  __shared__ __attribute__((aligned(16))) float sb1[BN * KC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 2, wn = wave & 3;
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
  const int mt = (int)(v % (unsigned)a.m_tiles), nt = (int)(v / (unsigned)a.m_tiles);
  const int m0 = mt * BM, n0 = nt * BN;
  const int Q = a.Kc / KC;
  const int b_first = (int)blockIdx.y * a.xb;
  const int nb = min(a.xb, a.batches - b_first);
  const unsigned short* Ab = a.As + (size_t)b_first * a.a_batch + (size_t)m0 * a.a_ld;
  const float* Bb = a.B + (size_t)b_first * a.b_batch + (size_t)n0 * a.b_ld;

  // (a wave-instruction of the A stream lies in ONE plane: 1024 units per plane, 64 per instruction, plane = j / 2)
  int a_goff[2], b_goff[PPW_B];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rem = (wave + NW * j) * 64 + lane;
    const int h = rem / (BM * 2), row = (rem >> 1) % BM, pj = rem & 1;
    const int jl = pj ^ ((row >> 4) & 1);
    a_goff[j] = (int)(min(row, a.M - 1 - m0) * a.a_ld) + h * (KC / 2) + jl * 8;
  }
#pragma unroll
  for (int j = 0; j < PPW_B; ++j) {
    const int u = (wave + NW * j) * 64 + lane;
    const int h = u / (BN * UH), row = (u / UH) % BN, pj = u % UH;
    const int jl = pj ^ ((row >> SWS) & (UH - 1));
    b_goff[j] = (int)(min(row, a.N - 1 - n0) * a.b_ld) + h * (KC / 2) + jl * 4;
  }
  int aoff[4], boff[2];
  const int swb = (l31 >> SWS) & (UH - 1), swa = (l31 >> 4) & 1;
#pragma unroll
  for (int i = 0; i < 4; ++i) aoff[i] = ((half * BM + wm * 128 + i * 32 + l31) * 2) * 8;
#pragma unroll
  for (int j = 0; j < 2; ++j) boff[j] = (half * BN + wn * 64 + j * 32 + l31) * (UH * 4);

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#if YG_TIME
  unsigned long long ytacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  YG_T(y_begin)
  int fq = 0, fb = 0;
#define Y_DMA(SA, SB)                                                                            \
  {                                                                                              \
    const unsigned short* as_ = Ab + (size_t)fb * a.a_batch + (size_t)fq * KC;                   \
    const float* bs_ = Bb + (size_t)fb * a.b_batch + (size_t)fq * a.b_adv;                       \
    _Pragma("unroll") for (int j = 0; j < PPW_A; ++j)                                            \
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(as_ + (size_t)(j >> 1) * a.as_plane + a_goff[j & 1]), \
                                       (lds_void_t*)((SA) + (wave + NW * j) * 512), 16, 0, 0);   \
    _Pragma("unroll") for (int j = 0; j < PPW_B; ++j)                                            \
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(bs_ + b_goff[j]),                           \
                                       (lds_void_t*)((SB) + (wave + NW * j) * 256), 16, 0, 0);   \
    if (++fq == Q) { fq = 0; ++fb; }                                                             \
  }
#define Y_LOADA(SA, S, IH, AH, AM, AL)                                                           \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                \
    const unsigned short* ap = (SA) + aoff[2 * (IH) + i] + (((S) ^ swa) * 8);                    \
    AH[i] = (FCD_SEXP & 32) ? u32x4{(unsigned)lane, 1u, (unsigned)(S), 7u} : *(const u32x4*)(ap);                        \
    AM[i] = (FCD_SEXP & 32) ? u32x4{(unsigned)i, 1u, (unsigned)lane, 7u} : *(const u32x4*)(ap + 2 * BM * 2 * 8);         \
    AL[i] = (FCD_SEXP & 32) ? u32x4{(unsigned)lane, 3u, (unsigned)(IH), 9u} : *(const u32x4*)(ap + 2 * (2 * BM * 2 * 8)); \
  }
#define Y_MFMA(IH, AV, BV)                                                                       \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                  \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                \
      acc[2 * (IH) + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                            \
          __builtin_bit_cast(bf16x8, AV[i]), __builtin_bit_cast(bf16x8, BV[j]), acc[2 * (IH) + i][j], 0, 0, 0);
#define Y_SIX(IH, AH, AM, AL, BH, BM_, BL)                                                       \
  Y_MFMA(IH, AL, BH) Y_MFMA(IH, AH, BL) Y_MFMA(IH, AM, BM_) Y_MFMA(IH, AM, BH) Y_MFMA(IH, AH, BM_) Y_MFMA(IH, AH, BH)
#define Y_STEP(SA, SB, SAN, SBN)                                                                 \
  {                                                                                              \
    YG_T(ys0)                                                                                    \
    if (!(FCD_SEXP & 8) && !(FCD_YEXP & 14)) if (fb < nb) Y_DMA(SAN, SBN)                         \
    f32x4 xa[2][2];                                                                              \
    u32x4 pah[2], pam[2], pal[2], qah[2], qam[2], qal[2];                                        \
    u32x4 bh0[2], bm0[2], bl0[2], bh1[2], bm1[2], bl1[2];                                        \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                              \
      xa[j][0] = (FCD_SEXP & 32) ? f32x4{(float)lane, 1.f, 0.f, 2.f} : *(const f32x4*)((SB) + boff[j] + ((0 ^ swb) * 4)); \
      xa[j][1] = (FCD_SEXP & 32) ? f32x4{(float)j, 3.f, (float)lane, 2.f} : *(const f32x4*)((SB) + boff[j] + ((1 ^ swb) * 4)); \
    }                                                                                            \
    Y_LOADA(SA, 0, 0, pah, pam, pal)                                                             \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) WS_SPLIT(xa[j][0], xa[j][1], bh0[j], bm0[j], bl0[j]) \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                              \
      xa[j][0] = (FCD_SEXP & 32) ? f32x4{(float)lane, 1.f, 2.f, 2.f} : *(const f32x4*)((SB) + boff[j] + ((2 ^ swb) * 4)); \
      xa[j][1] = (FCD_SEXP & 32) ? f32x4{(float)j, 3.f, (float)lane, 2.f} : *(const f32x4*)((SB) + boff[j] + ((3 ^ swb) * 4)); \
    }                                                                                            \
    if (FCD_YEXP & 1) __builtin_amdgcn_s_setprio(2);                                             \
    YG_T(ys1)                                                                                    \
    Y_SIX(0, pah, pam, pal, bh0, bm0, bl0)                                                       \
    if ((FCD_YEXP & 2) && !(FCD_SEXP & 8)) if (fb < nb) Y_DMA(SAN, SBN)                          \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) WS_SPLIT(xa[j][0], xa[j][1], bh1[j], bm1[j], bl1[j]) \
    Y_LOADA(SA, 0, 1, qah, qam, qal)                                                             \
    Y_SIX(1, qah, qam, qal, bh0, bm0, bl0)                                                       \
    if ((FCD_YEXP & 4) && !(FCD_SEXP & 8)) if (fb < nb) Y_DMA(SAN, SBN)                          \
    Y_LOADA(SA, 1, 0, pah, pam, pal)                                                             \
    Y_SIX(0, pah, pam, pal, bh1, bm1, bl1)                                                       \
    if ((FCD_YEXP & 8) && !(FCD_SEXP & 8)) if (fb < nb) Y_DMA(SAN, SBN)                          \
    Y_LOADA(SA, 1, 1, qah, qam, qal)                                                             \
    Y_SIX(1, qah, qam, qal, bh1, bm1, bl1)                                                       \
    __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);     /* B raw step 0, A (0, 0) */         \
    __builtin_amdgcn_sched_group_barrier(0x002, 88, 0);     /* split step 0 */                   \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                           \
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);      /* B raw step 1 */                   \
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                           \
    _Pragma("unroll") for (int k = 0; k < 14; ++k) {        /* group (0, 0) over the split of step 1 */ \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
      __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);                                         \
    }                                                                                            \
    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                                           \
    _Pragma("unroll") for (int k = 0; k < 6; ++k) {         /* ... and the A loads of (0, 1) */  \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                         \
    }                                                                                            \
    _Pragma("unroll") for (int k = 0; k < 6; ++k) {         /* group (0, 1) over the A loads of (1, 0) */ \
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                         \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                         \
    }                                                                                            \
    __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);                                          \
    _Pragma("unroll") for (int k = 0; k < 6; ++k) {         /* group (1, 0) over the A loads of (1, 1) */ \
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                         \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                         \
    }                                                                                            \
    __builtin_amdgcn_sched_group_barrier(0x008, 36, 0);                                          \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    if (FCD_YEXP & 1) __builtin_amdgcn_s_setprio(0);                                             \
    YG_T(ys2)                                                                                    \
    if (YG_TIME) __builtin_amdgcn_s_waitcnt(0x0F70);        /* vmcnt(0): the wave's share of the next stage's LDS-DMA has landed */ \
    YG_T(ys2b)                                                                                   \
    __syncthreads();                                                                             \
    YG_T(ys3)                                                                                    \
    YG_TACC(0, ys1, ys0) YG_TACC(1, ys2, ys1) YG_TACC(2, ys3, ys2b) YG_TACC(4, ys2b, ys2)         \
  }

  Y_DMA(sa0, sb0)
  __syncthreads();
#pragma unroll 1
  for (int cb = 0; cb < nb; ++cb) {
    for (int qc = 0; qc < Q; qc += 2) {
      Y_STEP(sa0, sb0, sa1, sb1)
      if (qc + 1 < Q) Y_STEP(sa1, sb1, sa0, sb0)
    }
    YG_T(yc0)
    if (a.c_blk) {
      float* Cb = a.C + (size_t)(b_first + cb) * a.c_batch;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int mb = (m0 + wm * 128 + i * 32) >> 5, tb = (n0 + wn * 64 + j * 32) >> 5;
          if (mb < a.c_mblk && tb < a.c_tblk) {
            float* blk = Cb + ((size_t)mb * a.c_tblk + tb) * 1024 + (half * 4 * 32 + l31) * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g)
              { const f32x4 cv_ = f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                if (YG_PLAINC) *(f32x4*)(blk + g * 128) = cv_; else __builtin_nontemporal_store(cv_, (f32x4*)(blk + g * 128)); }      /* M is consumed once, by the output transform: measured -2 ... -4 % on the 256-row launches and their output transforms; the 128-row kernels lose with the same hint */
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
      YG_T(yc1)
      YG_TACC(3, yc1, yc0)
      continue;
    }
    int ldc = a.N;
    asm volatile("" : "+s"(ldc));
    float* Cb = a.C + (size_t)(b_first + cb) * a.M * ldc;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = n0 + wn * 64 + j * 32 + l31;
          if (m < a.M && n < ldc) Cb[(size_t)m * ldc + n] = acc[i][j][r];
          acc[i][j][r] = 0.f;
        }
      }
  }
#if YG_TIME
  {
    const unsigned long long y_end = __builtin_readcyclecounter();
    ytacc[6] = y_end - y_begin;
    if (a.tbuf && lane == 0) {
      const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
#pragma unroll
      for (int i = 0; i < 8; ++i) a.tbuf[(wg * NW + wave) * 8 + i] = ytacc[i];
    }
  }
#endif
#undef Y_STEP
#undef Y_SIX
#undef Y_MFMA
#undef Y_LOADA
#undef Y_DMA
}

// The same 256 x 256 tile as a two-group ping-pong (MI355X_MICROARCH.md, "Two waves per SIMD"): waves 0-3 (upper 128
// rows, group A) and their SIMD partners 4-7 (lower 128 rows, group B) run the same stage loop ONE barrier interval
// apart, so that in every interval each SIMD has one wave in its compute segment (48 MFMAs of a 16-element reduction
// step, the second half's A fragments read from LDS in their shadow) beside one in its load segment (B fragments and the
// first A fragments LDS -> registers, exact bf16 split of B, then the wave's 5 of the 40 LDS-DMA wave-instructions of
// the stage two ahead).  The matrix pipe never waits for a split or an LDS round trip of its own wave.  Ring of four
// 40-KB LDS stages = the whole LDS:
//   interval 2n  : A load(n), A's half of DMA(n+2)    | B compute(n-1)
//   interval 2n+1: A compute(n)                        | B load(n), B's half of DMA(n+2)
// Slot (n+2) % 4 held stage n-2, last read in B's compute(n-2) = interval 2n-2; stage n+2 is first read in interval
// 2n+4, after each group waited for its own half (vmcnt) before a barrier that precedes it.
template <int DUMMY>
__global__ __launch_bounds__(512, 1) void wino_gemm_split_pp_kernel(WinoGemmArgs a) {
  constexpr int BM = 256, BN = 256;
  constexpr int A_UNITS = 3 * 2 * BM;              // [plane][half][row] x 16 B (8 bf16)
  constexpr int B_UNITS = 2 * BN * 2;              // [half][row][2 units of 4 fp32]
    __shared__ __attribute__((aligned(16))) unsigned short sa0[A_UNITS * 8];
  __shared__ __attribute__((aligned(16))) unsigned short sa1[A_UNITS * 8];
  __shared__ __attribute__((aligned(16))) unsigned short sa2[A_UNITS * 8];
  __shared__ __attribute__((aligned(16))) unsigned short sa3[A_UNITS * 8];
  __shared__ __attribute__((aligned(16))) float sb0[B_UNITS * 4];
  __shared__ __attribute__((aligned(16))) float sb1[B_UNITS * 4];
  __shared__ __attribute__((aligned(16))) float sb2[B_UNITS * 4];
  __shared__ __attribute__((aligned(16))) float sb3[B_UNITS * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 2, wn = wave & 3;
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
  const int mt = (int)(v % (unsigned)a.m_tiles), nt = (int)(v / (unsigned)a.m_tiles);
  const int m0 = mt * BM, n0 = nt * BN;
  const int Q = a.Kc / 16;                         // stages per batch (even)
  const int b_first = (int)blockIdx.y * a.xb;
  const int nb = min(a.xb, a.batches - b_first);
  const int S = nb * Q;
  const unsigned short* Ab = a.As + (size_t)b_first * a.a_batch + (size_t)m0 * a.a_ld;
  const float* Bb = a.B + (size_t)b_first * a.b_batch + (size_t)n0 * a.b_ld;

  // DMA (group B only; wave-instruction p of wave wn covers units (wn + 4 p) * 64 + lane).  A: 512 units per plane,
  // so instruction p lies in plane p / 2.
  // every wave issues 5 of the 40 wave-instructions of a stage: its group's half of each A plane and of the B slab
  int a_goff[1], b_goff[2];
  {
    const int rem = (wn + 4 * wm) * 64 + lane;     // unit inside the plane: half * 256 + row
    const int h = rem / BM, row = rem % BM;
    a_goff[0] = (int)(min(row, a.M - 1 - m0) * a.a_ld) + h * 8;
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int u = (wn + 4 * (2 * p + wm)) * 64 + lane;
    const int h = u / (BN * 2), row = (u >> 1) % BN, pj = u & 1;
    const int jl = pj ^ ((row >> 4) & 1);          // 32-B row pitch: rows 16..31 of a block take the other unit parity
    b_goff[p] = (int)(min(row, a.N - 1 - n0) * a.b_ld) + h * 8 + jl * 4;
  }
  int aoff[4], boff[2];
  const int swb = (l31 >> 4) & 1;
#pragma unroll
  for (int i = 0; i < 4; ++i) aoff[i] = (half * BM + wm * 128 + i * 32 + l31) * 8;        // bf16 elements, plane 0
#pragma unroll
  for (int j = 0; j < 2; ++j) boff[j] = (half * BN + wn * 64 + j * 32 + l31) * 8;         // floats, unit 0

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int fq = 0, fb = 0;        // stage / batch of the next slab pair to fetch (group B)
  int cq = 0, cb = 0;        // stage / batch being multiplied
#define P_DMA(SA, SB)                                                                            \
  {                                                                                              \
    const unsigned short* as_ = Ab + (size_t)fb * a.a_batch + (size_t)fq * 16;                   \
    const float* bs_ = Bb + (size_t)fb * a.b_batch + (size_t)(fq >> 1) * a.b_adv + (fq & 1) * 16; \
    _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)       /* half `wm` of plane pl */             \
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(as_ + (size_t)pl * a.as_plane + a_goff[0]),  \
                                       (lds_void_t*)((SA) + (pl * 8 + wn + 4 * wm) * 512), 16, 0, 0); \
    _Pragma("unroll") for (int p = 0; p < 2; ++p)                                                \
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(bs_ + b_goff[p]),                           \
                                       (lds_void_t*)((SB) + (wn + 4 * (2 * p + wm)) * 256), 16, 0, 0); \
    /* past the last stage the last slabs are fetched again into a slot nobody reads any more: the number of DMA */ \
    /* instructions of this wave in flight behind the stage being waited for is 5 on every path                              */ \
    if (!(fb == nb - 1 && fq == Q - 1)) { if (++fq == Q) { fq = 0; ++fb; } }                     \
  }
#define P_LOADA(SA, IH, AH, AM, AL)                                                              \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                \
    const unsigned short* ap = (SA) + aoff[2 * (IH) + i];                                        \
    AH[i] = *(const u32x4*)(ap);                                                                 \
    AM[i] = *(const u32x4*)(ap + 2 * BM * 8);                                                    \
    AL[i] = *(const u32x4*)(ap + 2 * (2 * BM * 8));                                              \
  }
#define P_MFMA(IH, AV, BV)                                                                       \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                  \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                \
      acc[2 * (IH) + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                            \
          __builtin_bit_cast(bf16x8, AV[i]), __builtin_bit_cast(bf16x8, BV[j]), acc[2 * (IH) + i][j], 0, 0, 0);
#define P_SIX(IH, AH, AM, AL)                                                                    \
  P_MFMA(IH, AL, bh) P_MFMA(IH, AH, bl) P_MFMA(IH, AM, bm) P_MFMA(IH, AM, bh) P_MFMA(IH, AH, bm) P_MFMA(IH, AH, bh)
// one stage of one wave: load segment, barrier, compute segment, barrier
#define P_STAGE(SA, SB, FA, FB)                                                                  \
  {                                                                                              \
    u32x4 bh[2], bm[2], bl[2], pah[2], pam[2], pal[2], qah[2], qam[2], qal[2];                   \
    {                                                                                            \
      f32x4 xa[2][2];                                                                            \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                            \
        xa[j][0] = *(const f32x4*)((SB) + boff[j] + ((0 ^ swb) * 4));                            \
        xa[j][1] = *(const f32x4*)((SB) + boff[j] + ((1 ^ swb) * 4));                            \
      }                                                                                          \
      P_LOADA(SA, 0, pah, pam, pal)                                                              \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) WS_SPLIT(xa[j][0], xa[j][1], bh[j], bm[j], bl[j]) \
    }                                                                                            \
    if (!(FCD_SEXP & 8)) {                                                                       \
      P_DMA(FA, FB)                                                                              \
      __builtin_amdgcn_s_waitcnt(0x0F75);                                /* vmcnt(5) */          \
    }                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    __builtin_amdgcn_s_barrier();                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    P_LOADA(SA, 1, qah, qam, qal)                                                                \
    P_SIX(0, pah, pam, pal)                                                                      \
    P_SIX(1, qah, qam, qal)                                                                      \
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);                                           \
    __builtin_amdgcn_sched_group_barrier(0x008, 48, 0);                                          \
    if (++cq == Q) {                                                                             \
      cq = 0;                                                                                    \
      int ldc = a.N;                                                                             \
      asm volatile("" : "+s"(ldc));                                                              \
      float* Cb = a.C + (size_t)(b_first + cb) * a.M * ldc;                                      \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                              \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                         \
          const int m = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;              \
          _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                        \
            const int n = n0 + wn * 64 + j * 32 + l31;                                           \
            if (m < a.M && n < ldc) Cb[(size_t)m * ldc + n] = acc[i][j][r];                      \
            acc[i][j][r] = 0.f;                                                                  \
          }                                                                                      \
        }                                                                                        \
      ++cb;                                                                                      \
    }                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    __builtin_amdgcn_s_barrier();                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                           \
  }

  P_DMA(sa0, sb0)
  P_DMA(sa1, sb1)              // (S >= 2: reductions of >= 32 elements only)
  __builtin_amdgcn_s_waitcnt(0x0F75);            // vmcnt(5): this wave's part of stage 0 has landed
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();     // group B runs one interval behind
#pragma unroll 1
  for (int g = 0; g < S; g += 4) {
    P_STAGE(sa0, sb0, sa2, sb2)
    if (g + 1 < S) P_STAGE(sa1, sb1, sa3, sb3)
    if (g + 2 < S) P_STAGE(sa2, sb2, sa0, sb0)
    if (g + 3 < S) P_STAGE(sa3, sb3, sa1, sb1)
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();     // ... and group A waits for it at the end
  __builtin_amdgcn_s_waitcnt(0x0F70);            // the surplus prefetches must land before the LDS allocation is released
#undef P_STAGE
#undef P_SIX
#undef P_MFMA
#undef P_LOADA
#undef P_DMA
}

static int wino_gemm_cfg() { return fcd_sw(FCD_SW_WINO_TILE); }     // 0 = 128x128 (4 waves), 1 = 256x128 (8 waves), 2 = 256x256 (16 waves)

// Batches (transform positions) per workgroup.  The per-batch pipeline is only Kc / 32 stages long (2 ... 16 for the VGG
// layers), so its fill and the C-tile store are a large share of a workgroup's life; chaining xb batches into one
// pipeline pays them once.  Keep >= ~4 full rounds of the 512 resident workgroup slots (256 CUs x 2) so that the
// tail stays small.  FCD_WINO_XB forces a value (1 = one batch per workgroup, the round-1 behaviour).
static int wino_gemm_xb(long long tiles, int batches, int splits, int q_stages) {
  const int forced = fcd_sw(FCD_SW_WINO_XB);
  if (forced > 0) return std::min(forced, batches);
  if (q_stages & 1) return 1;          // the two LDS stages alternate: a batch must start on stage 0
  static const int cand[] = {36, 18, 12, 9, 6, 4, 3, 2};
  for (int xb : cand) {
    if (batches % xb) continue;
    if (tiles * (batches / xb) * splits >= 2048) return xb;
  }
  return 1;
}

// FCD_WINO_SPLIT / fcd_conv_wino_split_set: 1 (default) = conv / data-gradient GEMMs on the bf16 matrix pipe with exact
// three-way operand splitting (wino_gemm_split_kernel), 0 = v_mfma_f32_32x32x2_f32 (wino_gemm_kernel)
static int wino_split() { return fcd_sw(FCD_SW_WINO_SPLIT); }
static int wino_split_state() { return wino_split(); }
extern "C" int fcd_conv_wino_split_set(int on) {
  const int old = wino_split();
  if (on >= 0) g_fcd_switch[FCD_SW_WINO_SPLIT] = on > 2 ? 1 : on;      // 2: as 1, and the 256 x 256 kernel for every GEMM with >= 256 rows (tests)
  return old;
}

static void wino_gemm_launch(WinoGemmArgs ga, int batches, int splits, hipStream_t st) {
  int cfg = wino_gemm_cfg();
  if (ga.As && ga.M > 64 && wino_split()) {
    ga.batches = batches;
    if (splits > 1 || ga.bt) {          // weight gradient: split reduction, one transform position per workgroup
      ga.m_tiles = cdiv(ga.M, 128); ga.n_tiles = cdiv(ga.N, 128);
      ga.xb = 1;
      const dim3 grid((unsigned)(ga.m_tiles * ga.n_tiles), (unsigned)batches, (unsigned)splits);
      if (ga.bt) hipLaunchKernelGGL((wino_gemm_split_kernel<2, 2, true>), grid, dim3(256), 0, st, ga);
      else hipLaunchKernelGGL((wino_gemm_split_kernel<2, 2>), grid, dim3(256), 0, st, ga);
      return;
    }
    // [r3] one row tile, short reduction, M in blocks: the filter-resident kernel (FCD_WINO_RES=0: off)
    if (fcd_sw(FCD_SW_WINO_RES) && ga.c_blk && ga.M <= 128 && ga.Kc <= 128 && (long long)cdiv(ga.N, 256) * batches >= 512) {
      ga.m_tiles = 1; ga.n_tiles = cdiv(ga.N, 256);
      ga.xb = 1;
      const int wgs = fcd_sw(FCD_SW_WINO_RES_WGS);           // workgroups in flight: one per CU
      const int chunks = std::max(1, std::min(ga.n_tiles, wgs / batches));
      hipLaunchKernelGGL(wino_gemm_split_res_kernel, dim3((unsigned)chunks, (unsigned)batches), dim3(512), 0, st, ga);
      return;
    }
    // FCD_WINO_SPLIT_BIG: 0 = 128 x 128 tiles only; 1 (default) = 256 x 256 two-stage kernel for GEMMs with >= 256
    // rows and enough tiles to fill the chip; 2 = ... for every GEMM with >= 256 rows (tests); 4 / 5 = the same two
    // policies with the ping-pong kernel (measured slower: DESIGN.md)
    const int big = fcd_sw(FCD_SW_WINO_SPLIT_BIG);
    const bool force = big == 2 || big == 5 || wino_split() == 2;
    if (big && ga.M >= 256 && (force || (long long)cdiv(ga.M, 256) * cdiv(ga.N, 256) * batches >= 1024)) {
      ga.m_tiles = cdiv(ga.M, 256); ga.n_tiles = cdiv(ga.N, 256);
      ga.xb = wino_gemm_xb((long long)ga.m_tiles * ga.n_tiles * 2, batches, 1, ga.Kc / 32);
      const dim3 grid((unsigned)(ga.m_tiles * ga.n_tiles), (unsigned)cdiv(batches, ga.xb));
      if (big >= 4) hipLaunchKernelGGL((wino_gemm_split_pp_kernel<0>), grid, dim3(512), 0, st, ga);
      else {
#if YG_TIME
        ga.tbuf = g_yg_tbuf;
#endif
        hipLaunchKernelGGL((wino_gemm_split256_kernel<0>), grid, dim3(512), 0, st, ga);
      }
      return;
    }
    ga.m_tiles = cdiv(ga.M, 128); ga.n_tiles = cdiv(ga.N, 128);
    ga.xb = wino_gemm_xb((long long)ga.m_tiles * ga.n_tiles, batches, 1, ga.Kc / 32);
    hipLaunchKernelGGL((wino_gemm_split_kernel<2, 2>), dim3((unsigned)(ga.m_tiles * ga.n_tiles), (unsigned)cdiv(batches, ga.xb)),
                       dim3(256), 0, st, ga);
    return;
  }
  if (cfg == 2 && (ga.M < 256 || ga.N < 256)) cfg = 1;
  if (cfg == 1 && ga.M < 256) cfg = 0;
  ga.batches = batches;
#define WG_LAUNCH(WM_, WN_, BM_, BN_, THREADS_)                                                              \
  {                                                                                                          \
    ga.m_tiles = cdiv(ga.M, BM_); ga.n_tiles = cdiv(ga.N, BN_);                                              \
    ga.xb = splits > 1 ? 1 : wino_gemm_xb((long long)ga.m_tiles * ga.n_tiles, batches, splits, ga.Kc / 32);                               \
    hipLaunchKernelGGL((wino_gemm_kernel<WM_, WN_>),                                                         \
                       dim3((unsigned)(ga.m_tiles * ga.n_tiles), (unsigned)cdiv(batches, ga.xb), (unsigned)splits), \
                       dim3(THREADS_), 0, st, ga);                                                           \
  }
  if (ga.M <= 64) {        // 64-row GEMMs: 64 x 256 tiles, no wasted MFMA rows
    WG_LAUNCH(1, 4, 64, 256, 256)
  } else if (cfg == 2) {
    WG_LAUNCH(4, 4, 256, 256, 1024)
  } else if (cfg == 1) {
    WG_LAUNCH(4, 2, 256, 128, 512)
  } else {
    WG_LAUNCH(2, 2, 128, 128, 256)
  }
#undef WG_LAUNCH
}

// --------------------------------------------------------------------------------------------
// output transform + epilogue.  Thread = (output channel k, tile t), t fastest.
template <int MM>
__device__ __forceinline__ void wino_out_one(const WinoOutArgs& a, const float (&mv)[WinoMat<MM>::A][WinoMat<MM>::A], int k,
                                             long long t, double* s1 = nullptr, double* s2 = nullptr) {
  constexpr int A = WinoMat<MM>::A;
  const float b = a.bias ? a.bias[k] : 0.f;
  float o[MM][MM];
  wino_out_tile<MM>(mv, b, a.relu, o);
  const int tx = (int)(t % a.TW);
  const long long r2 = t / a.TW;
  const int ty = (int)(r2 % a.TH), n = (int)(r2 / a.TH);
  if (MM == 4 && a.gate != nullptr) {      // data gradient handed to a consumer that cannot gate it itself (chain, fused F(2x2) kernel)
    const unsigned gw = a.gate[(((size_t)n * a.K + k) * a.TH + ty) * a.TW + tx];
#pragma unroll
    for (int i = 0; i < MM; ++i)
#pragma unroll
      for (int j = 0; j < MM; ++j) o[i][j] = ((gw >> (4 * i + j)) & 1u) ? o[i][j] : 0.f;
  }
  if (s1 != nullptr) {           // statistics of the outputs that exist (ragged last tile row / column left out)
    double a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int i = 0; i < MM; ++i)
#pragma unroll
      for (int j = 0; j < MM; ++j)
        if (ty * MM + i < a.P && tx * MM + j < a.Q) { a1 += (double)o[i][j]; a2 += (double)o[i][j] * (double)o[i][j]; }
    *s1 = a1; *s2 = a2;
  }
  if (MM == 4 && a.bits != nullptr) {
    unsigned w = 0;
#pragma unroll
    for (int i = 0; i < MM; ++i)
#pragma unroll
      for (int j = 0; j < MM; ++j) w |= (o[i][j] > 0.f ? 1u : 0u) << (4 * i + j);
    a.bits[(((size_t)n * a.K + k) * a.TH + ty) * a.TW + tx] = (unsigned short)w;
  }
  const int p0 = ty * MM, q0 = tx * MM;
  if (a.pool_y != nullptr) {
    // The four outputs of a pooling window come out of the same transform-domain values through
    // different coefficient patterns: mathematically equal outputs (constant image regions, zero
    // padding) differ by transform rounding (<= ~2e-5 relative for m = 4).  Candidates that close to
    // the maximum count as tied and the FIRST one takes the gradient, as MaxPool2d does on exact ties
    // (Loss.py:25 -> torchvision vgg16.features[4,9,16,23]); the pooled VALUE is always the maximum.
    const int Pp = a.P >> 1, Qp = a.Q >> 1;
    // [r4] ... plus a floor from the transform domain: where the products M cancel (flat image regions: zero padding, nodata
    // areas -- every output of the region is the same small number) the rounding of A^T M A scales with max |M| of the tile, not
    // with the outputs, and the relative rule alone would break such exact ties at random instead of first-wins
    float mabs = 0.f;
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
      for (int j = 0; j < A; ++j) mabs = fmaxf(mabs, fabsf(mv[i][j]));
    const float tie_floor = (MM == 4 ? 4e-6f : 5e-7f) * mabs;
#pragma unroll
    for (int wi = 0; wi < MM / 2; ++wi)
#pragma unroll
      for (int wj = 0; wj < MM / 2; ++wj) {
        const int pp = (p0 >> 1) + wi, qq = (q0 >> 1) + wj;
        if (pp >= Pp || qq >= Qp) continue;
        const float c0 = o[2 * wi][2 * wj], c1 = o[2 * wi][2 * wj + 1], c2 = o[2 * wi + 1][2 * wj],
                    c3 = o[2 * wi + 1][2 * wj + 1];
        const float m = fmaxf(fmaxf(c0, c1), fmaxf(c2, c3));
        const float tie = (MM == 4 ? 3e-5f : 2e-6f) * fmaxf(fmaxf(fabsf(c0), fabsf(c1)), fmaxf(fabsf(c2), fabsf(c3))) + tie_floor;
        const float mx = m - tie;
        int arg = 3;
        if (c2 >= mx) arg = 2;
        if (c1 >= mx) arg = 1;
        if (c0 >= mx) arg = 0;
        const size_t oo = (((size_t)n * a.K + k) * Pp + pp) * Qp + qq;
        a.pool_y[oo] = m;
        a.code[oo] = (unsigned char)(arg | (m > 0.f ? 4 : 0));
      }
    return;
  }
  float* yo = a.y + (((size_t)n * a.K + k) * a.P + p0) * a.Q + q0;
  if (a.cat.n > 0) {
    int ch = k, chans;
    float* base = const_cast<float*>(wino_cat_pick(a.cat, ch, chans));
    yo = base + (((size_t)n * chans + ch) * a.P + p0) * a.Q + q0;
  }
  const bool vec = (MM == 4) ? ((a.Q & 3) == 0) : ((a.Q & 1) == 0);
#pragma unroll
  for (int i = 0; i < MM; ++i) {
    if (p0 + i >= a.P) continue;
    if (vec && q0 + MM <= a.Q) {
      if (MM == 4) {
        f32x4 v4 = {o[i][0], o[i][1], o[i][2], o[i][3]};
        *(f32x4*)(yo + (size_t)i * a.Q) = v4;
      } else {
        f32x2 v2 = {o[i][0], o[i][1]};
        *(f32x2*)(yo + (size_t)i * a.Q) = v2;
      }
    } else {
#pragma unroll
      for (int j = 0; j < MM; ++j)
        if (q0 + j < a.Q) yo[(size_t)i * a.Q + j] = o[i][j];
    }
  }
}

template <int MM>
__global__ __launch_bounds__(256) void wino_output_kernel(WinoOutArgs a) {
  constexpr int A = WinoMat<MM>::A;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int k = blockIdx.y;
  if (t >= a.T) return;
  const size_t xs = (size_t)a.K * a.T;
  const float* mp = a.Mb + (size_t)k * a.T + t;
  float mv[A][A];
#pragma unroll
  for (int i = 0; i < A; ++i)
#pragma unroll
    for (int j = 0; j < A; ++j) mv[i][j] = mp[(size_t)(i * A + j) * xs];
  wino_out_one<MM>(a, mv, k, t);
}

// [r3] M in the split GEMM's MFMA-native 32 x 32 blocks: a thread takes FOUR channels of its tile -- the four rows a GEMM
// lane held in adjacent registers lie in one 16-B word, so the 36 positions arrive as 36 float4 loads (lanes = consecutive
// tiles: 512 B contiguous per 32 lanes) and the GEMM's epilogue is a quarter of the store instructions it was.
__global__ __launch_bounds__(256) void wino_output_blk_kernel(WinoOutArgs a) {
  constexpr int A = 6;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int k0 = blockIdx.y * 4;
  if (t >= a.T) return;
  const float* mp = a.Mb + ((size_t)(k0 >> 5) * a.tblk + (size_t)(t >> 5)) * 1024 +
                    ((((k0 >> 2) & 1) * 4 + ((k0 >> 3) & 3)) * 32 + (int)(t & 31)) * 4;
  f32x4 m4[A * A];
#pragma unroll
  // M is read exactly once: non-temporal loads (measured 5.1 -> 5.6 ... 6.0 TB/s on the 64 x 64 ... 32 x 32 maps; the same hint on
  // the input transform's V STORES lost 10 - 25 %: dword stores want the L2's write combining)
  for (int q = 0; q < A * A; ++q) m4[q] = __builtin_nontemporal_load((const f32x4*)(mp + (size_t)q * a.xs_blk));
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (k0 + c >= a.K) break;
    float mv[A][A];
#pragma unroll
    for (int q = 0; q < A * A; ++q) mv[q / A][q % A] = m4[q][c];
    wino_out_one<4>(a, mv, k0 + c, t);
  }
}

// the same with the BatchNorm partial sums: every thread stays to the end (block reduction), fp64, fixed order
__global__ __launch_bounds__(256) void wino_output_blk_bn_kernel(WinoOutArgs a) {
  constexpr int A = 6;
  __shared__ double red[4][8];
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int k0 = blockIdx.y * 4;
  const bool live = t < a.T;
  double bs[4][2];
#pragma unroll
  for (int c = 0; c < 4; ++c) bs[c][0] = bs[c][1] = 0.0;
  if (live) {
    const float* mp = a.Mb + ((size_t)(k0 >> 5) * a.tblk + (size_t)(t >> 5)) * 1024 +
                      ((((k0 >> 2) & 1) * 4 + ((k0 >> 3) & 3)) * 32 + (int)(t & 31)) * 4;
    f32x4 m4[A * A];
#pragma unroll
    for (int q = 0; q < A * A; ++q) m4[q] = __builtin_nontemporal_load((const f32x4*)(mp + (size_t)q * a.xs_blk));
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (k0 + c >= a.K) break;
      float mv[A][A];
#pragma unroll
      for (int q = 0; q < A * A; ++q) mv[q / A][q % A] = m4[q][c];
      wino_out_one<4>(a, mv, k0 + c, t, &bs[c][0], &bs[c][1]);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const double v = wave_sum_d(bs[c][e]);
      if (lane == 0) red[wave][c * 2 + e] = v;
    }
  __syncthreads();
  if (threadIdx.x < 8) {
    const int c = threadIdx.x >> 1, e = threadIdx.x & 1;
    if (k0 + c < a.K) {
      const double v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
      const int g = (int)(blockIdx.x / a.bn_bpg), bi = (int)(blockIdx.x % a.bn_bpg);
      a.bn_part[(((size_t)g * a.K + (k0 + c)) * a.bn_bpg + bi) * 3 + e] = v;
    }
  }
}

// --------------------------------------------------------------------------------------------
// host side
static int wino_env() { return fcd_sw(FCD_SW_WINO); }

int fcd_wino_mode_now() { return wino_env(); }   // conv_wino2.hip: the fused F(2x2) kernel follows the same switch

// 0 = direct kernels only, 2 / 4 = Winograd tile size for the planned layers; returns the previous value.
// (Filters packed for another tile size stay valid: packs are keyed by m.)
extern "C" int fcd_conv_wino_set(int m) {
  const int prev = wino_env();
  if (m == 0 || m == 2 || m == 4) g_fcd_switch[FCD_SW_WINO] = m;
  return prev;
}

// mode 0 forward / 1 data gradient.  Returns the output tile size m (2 or 4), or 0 when the layer
// runs on the direct kernels.  The transform passes stream 22 (m = 4) / 36 (m = 2) bytes per input and
// per output element, which the 4x / 2.25x smaller GEMM only pays back from 128 GEMM rows and 64 reduction
// channels upwards -- measured per layer on MI355X with tools/bench_conv.py.  64-row layers (64 x 256 GEMM
// tiles, FCD_WINO_MINROWS=64) are a wash: 64 -> 64 at 208 x 256 x 256 runs 8.3 / 8.4 ms vs 8.2 / 7.8 direct, the
// ReLU-gated 128 -> 64 data gradient 4.4 vs 4.2.
extern "C" int fcd_conv_wino_plan(const fcd_conv_desc* d, int mode) {
  if (!d || wino_env() == 0) return 0;
  if (!(d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1)) return 0;
  const int red = mode == 0 ? d->C : d->K;       // reduction channels of the GEMM
  const int rows = mode == 0 ? d->K : d->C;      // GEMM rows
  const int min_red = fcd_sw(FCD_SW_WINO_MINC), min_rows = fcd_sw(FCD_SW_WINO_MINROWS);
  if (red % 32 != 0 || red < min_red || rows < min_rows) return 0;
  if (d->P < 4 || d->Q < 4) return 0;
  return wino_env();
}

static bool wino_plan(const fcd_conv_desc* d, int mode, WinoPlan* pl) {
  pl->m = fcd_conv_wino_plan(d, mode);
  if (!pl->m) return false;
  const int a = pl->m + 2;
  pl->A2 = a * a;
  pl->rows = mode == 0 ? d->K : d->C;
  pl->red = mode == 0 ? d->C : d->K;
  pl->Kc = round_up(pl->red, 32);
  pl->Q = pl->Kc / 32;
  // forward: tiles over the output (P, Q) = (H, W); data gradient: tiles over dx (H, W) = (P, Q)
  pl->TH = cdiv(d->H, pl->m);
  pl->TW = cdiv(d->W, pl->m);
  pl->T = (long long)d->N * pl->TH * pl->TW;
  pl->v_bytes = (size_t)pl->A2 * pl->Q * pl->T * 32 * sizeof(float);
  // (rows and tiles rounded up to 32: the split GEMM may write M in whole 32 x 32 blocks, wino_output_blk_kernel)
  pl->m_bytes = (size_t)pl->A2 * round_up(pl->rows, 32) * ((pl->T + 31) / 32 * 32) * sizeof(float);
  return true;
}

extern "C" size_t fcd_conv_wino_ws_bytes(const fcd_conv_desc* d, int mode) {
  WinoPlan pl;
  if (!d || !wino_plan(d, mode, &pl)) return 0;
  return pl.v_bytes + pl.m_bytes + 256;
}

extern "C" int64_t fcd_conv_wino_filter_elems(int K, int C, int mode, int m) {
  if (m != 2 && m != 4) return 0;
  const int rows = mode == 0 ? K : C, red = mode == 0 ? C : K;
  const int64_t elems = (int64_t)(m + 2) * (m + 2) * rows * round_up(red, 32);
  return elems + elems / 2 * 3;      // fp32 U, then three bf16 planes (the split GEMM's A operand)
}

static int wino_split_state();      // fcd_conv_wino_split_set(-1)
extern "C" int fcd_conv_wino_pack(const float* w, float* U, int K, int C, int mode, int m, void* stream) {
  FCD_CHECK_ARG(w && U && K > 0 && C > 0 && (mode == 0 || mode == 1) && (m == 2 || m == 4),
                "fcd_conv_wino_pack: bad arguments");
  const int rows = mode == 0 ? K : C, Kc = round_up(mode == 0 ? C : K, 32);
  const long long total = (long long)rows * Kc;
  const int grid = (int)std::min<long long>(cdiv64(total / 2, 256), 4096);
  FcdProfScope prof(FCD_K_PACK, (hipStream_t)stream, 0.0, 4.0 * total * (9 + 1.5 * (m + 2) * (m + 2)));
  // the GEMM reads EITHER the fp32 U (fp32 matrix pipe) OR its three bf16 planes behind it (split pipe, every GEMM of
  // >= 128 rows = every layer the plan sends here): only that one is written.  The host cache keys the packed buffer by
  // fcd_conv_wino_split_set(-1), so flipping the switch re-packs.
  unsigned short* planes = (unsigned short*)(U + (long long)(m + 2) * (m + 2) * total);
  const bool split = wino_split_state() != 0 && rows > 64;
  float* Uw = split ? nullptr : U;
  if (!split) planes = nullptr;
  if (m == 2)
    hipLaunchKernelGGL(wino_filter_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, Uw, K, C, rows, Kc, mode, planes);
  else
    hipLaunchKernelGGL(wino_filter_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, Uw, K, C, rows, Kc, mode, planes);
  FCD_LAUNCH_CHECK("wino_pack");
  return FCD_OK;
}

static int wino_xcd() { return fcd_sw(FCD_SW_CONV_XCD); }

template <int MM, int TRB, int TWB>
static void wino_launch_input_cfg(const WinoInArgs& ia, int src, hipStream_t st) {
  dim3 grid((unsigned)cdiv(ia.TW, TWB), (unsigned)cdiv(ia.TH, TRB), (unsigned)(ia.N * ia.Q));
  // float4 strip rows: W % 4 == 0; the pooled source additionally needs 8-B aligned pooled rows
  const bool vec = (ia.W & 3) == 0 && (src != 2 || (ia.Wp & 1) == 0);
#define WINO_IN(SRC_, VEC_) \
  hipLaunchKernelGGL((wino_input_kernel<MM, SRC_, TRB, TWB, VEC_>), grid, dim3(256), 0, st, ia)
  if (src == 0) { if (vec) WINO_IN(0, true); else WINO_IN(0, false); }
  else if (src == 1) { if (vec) WINO_IN(1, true); else WINO_IN(1, false); }
  else { if (vec) WINO_IN(2, true); else WINO_IN(2, false); }
#undef WINO_IN
}

template <int MM>
static void wino_launch_input(const WinoInArgs& ia, int src, hipStream_t st) {
  constexpr int NT = 64 / MM;             // tiles per block: 64 output pixels per patch row at 1 x NT
  const int roll = fcd_sw(FCD_SW_WINO_IN_ROLL);
  // maps >= 2 strips high: the rolling kernel (next strip prefetched into registers)
  if ((ia.W & 3) == 0 && (src != 2 || (ia.Wp & 1) == 0) && roll > 1 && ia.TW > NT / 4) {
    const int trb = ia.TW > NT / 2 ? 1 : 2, strips = cdiv(ia.TH, trb);
    if (strips >= 2) {
      const int r = std::min(roll, strips);
      dim3 grid((unsigned)cdiv(ia.TW, NT / trb), (unsigned)cdiv(strips, r), (unsigned)(ia.N * ia.Q));
#define WINO_ROLL(SRC_, TRB_) hipLaunchKernelGGL((wino_input_roll_kernel<MM, SRC_, TRB_, NT / TRB_>), grid, dim3(256), 0, st, ia, r)
      if (src == 0) { if (trb == 1) WINO_ROLL(0, 1); else WINO_ROLL(0, 2); }
      else if (src == 1) { if (trb == 1) WINO_ROLL(1, 1); else WINO_ROLL(1, 2); }
      else { if (trb == 1) WINO_ROLL(2, 1); else WINO_ROLL(2, 2); }
#undef WINO_ROLL
      return;
    }
  }
  if (ia.TW > NT / 2 && (ia.exp & 1)) wino_launch_input_cfg<MM, 1, NT / 2>(ia, src, st);
  else if (ia.TW > NT / 2) wino_launch_input_cfg<MM, 1, NT>(ia, src, st);
  else if (ia.TW > NT / 4) wino_launch_input_cfg<MM, 2, NT / 2>(ia, src, st);
  else wino_launch_input_cfg<MM, 4, NT / 4>(ia, src, st);
}

// a virtually concatenated input needs the rolling input kernel (the only one that takes source lists): m = 4, W % 4 == 0,
// map at least two strips high and more than 4 tiles wide
static bool wino_cat_input_ok(const WinoPlan& pl, int W) {
  if (pl.m != 4 || (W & 3) != 0 || pl.TW <= 4) return false;
  const int trb = pl.TW > 8 ? 1 : 2;
  return cdiv(pl.TH, trb) >= 2;
}

// shared by forward and data gradient: src tensor (in_ch channels, H x W logical extent) -> out tensor
// (rows channels, H x W)
// [r3] M in 32 x 32 MFMA-native blocks whenever the split kernels run the GEMM (FCD_WINO_CBLK=0: row-major M, A/B)
static bool wino_blk_path(const WinoPlan& pl) {
  static int cblk_on = -1;
  if (cblk_on < 0) {
    const char* e = getenv("FCD_WINO_CBLK");
    const char* e2 = getenv("FCD_WINO_SPLIT_BIG");
    cblk_on = ((e && e[0] == '0') || (e2 && atoi(e2) >= 4)) ? 0 : 1;      // (the ping-pong experiment kernel writes row-major)
  }
  return cblk_on && pl.m == 4 && pl.rows > 64 && wino_split() && (pl.rows & 3) == 0;
}

// The three stages of a layer call.  wino_run = input -> GEMM -> output; the chains further down put the fused output -> input
// kernel of conv_wino_chain.hip between the GEMMs of consecutive layers.
struct WinoInSrc {
  const float* src; const float* mask; const unsigned short* mask_bits; const unsigned char* code; int Hp, Wp;
  const WinoCat* cat;
};
static void wino_stage_input(const WinoPlan& pl, int N, int in_ch, int H, int W, const WinoInSrc& in, float* V, hipStream_t st) {
  WinoInArgs ia;
  memset(&ia, 0, sizeof(ia));
  if (in.cat) ia.cat = *in.cat;
  ia.x = in.src; ia.mask = in.mask; ia.mbits = in.mask_bits; ia.code = in.code; ia.V = V;
  ia.N = N; ia.C = in_ch; ia.H = H; ia.W = W; ia.Hp = in.Hp; ia.Wp = in.Wp;
  ia.TH = pl.TH; ia.TW = pl.TW; ia.Q = pl.Q; ia.T = pl.T;
  const int srcmode = in.code ? 2 : ((in.mask || in.mask_bits) ? 1 : 0);
  {
    const int exp = fcd_sw(FCD_SW_WINO_IN_EXP);
    ia.exp = exp;
    ia.xcd = (wino_xcd() && !(exp & 8)) ? 1 : 0;
  }
  const double in_elems = (double)N * in_ch * H * W;
  FcdProfScope p1(FCD_K_WINO_XFORM, st, 0.0, 4.0 * in_elems * (srcmode == 1 ? (in.mask_bits ? 1.03125 : 2.0) : 1.0) + (double)pl.v_bytes,
                  fcd_prof_tagf("in src=%d C=%d img=%dx%dx%d", srcmode, in_ch, N, H, W));
  if (pl.m == 2) wino_launch_input<2>(ia, srcmode, st); else wino_launch_input<4>(ia, srcmode, st);
}

// M = U V; returns the launch's arguments (the output stage needs the block geometry of M)
static WinoGemmArgs wino_stage_gemm(const WinoPlan& pl, const float* U, const float* V, float* Mb, int N, int H, int W, hipStream_t st) {
  WinoGemmArgs ga;
  memset(&ga, 0, sizeof(ga));
  ga.A = U; ga.B = V; ga.C = Mb;
  ga.M = pl.rows; ga.N = (int)pl.T; ga.Kc = pl.Kc;
  ga.m_tiles = cdiv(pl.rows, 128); ga.n_tiles = cdiv((int)pl.T, 128);
  ga.xcd_remap = wino_xcd();
  ga.a_ld = pl.Kc; ga.a_batch = (long long)pl.rows * pl.Kc;                 // U [xi][rows][Kc]
  ga.as_plane = (long long)pl.A2 * pl.rows * pl.Kc;                          // bf16 planes behind the fp32 U
  ga.As = (const unsigned short*)(U + ga.as_plane);
  ga.b_ld = 32; ga.b_adv = (long long)pl.T * 32; ga.b_batch = (long long)pl.Q * pl.T * 32;   // V [xi][Q][T][32]
  ga.stages_per_split = pl.Q;
  if (wino_blk_path(pl)) {
    ga.c_blk = 1; ga.c_mblk = cdiv(pl.rows, 32); ga.c_tblk = (int)((pl.T + 31) / 32);
    ga.c_batch = (long long)ga.c_mblk * ga.c_tblk * 1024;
  }
  const bool split = pl.rows > 64 && wino_split();
  FcdProfScope p2(split ? FCD_K_WINO_GEMM_SPLIT : FCD_K_WINO_GEMM, st, 2.0 * pl.A2 * pl.rows * (double)pl.Kc * (double)pl.T,
                  (double)pl.v_bytes + (double)pl.m_bytes + (split ? 6.0 : 4.0) * pl.A2 * pl.rows * pl.Kc,
                  fcd_prof_tagf("conv M=%d N=%lld Kc=%d batch=%d img=%dx%dx%d", pl.rows, pl.T, pl.Kc, pl.A2, N, H, W));
  wino_gemm_launch(ga, pl.A2, 1, st);
  return ga;
}

struct WinoOutDst {
  const float* bias; int relu; float* y; float* pool_y; unsigned char* code;
  const WinoCat* cat; unsigned short* bits; const unsigned short* gate; double* bn_part; int bn_bpg;
};
static void wino_stage_output(const WinoPlan& pl, const WinoGemmArgs& ga, const float* Mb, int N, int H, int W, const WinoOutDst& o,
                              hipStream_t st) {
  WinoOutArgs oa;
  memset(&oa, 0, sizeof(oa));
  if (o.cat) oa.cat = *o.cat;
  oa.Mb = Mb; oa.bias = o.bias; oa.y = o.y; oa.pool_y = o.pool_y; oa.code = o.code; oa.bits = o.bits; oa.gate = o.gate;
  oa.K = pl.rows; oa.P = H; oa.Q = W; oa.TH = pl.TH; oa.TW = pl.TW; oa.relu = o.relu; oa.T = pl.T;
  dim3 og((unsigned)cdiv64(pl.T, 256), (unsigned)pl.rows);
  const double mm = pl.m * pl.m;
  FcdProfScope p3(FCD_K_WINO_XFORM, st, 0.0,
                  (double)pl.m_bytes + (double)pl.m_bytes / pl.A2 * mm * (o.pool_y ? 0.3125 : 1.0),
                  fcd_prof_tagf("out pool=%d K=%d img=%dx%dx%d", o.pool_y ? 1 : 0, pl.rows, N, H, W));
  if (ga.c_blk) {
    oa.tblk = ga.c_tblk; oa.xs_blk = ga.c_batch;
    const dim3 ogb((unsigned)cdiv64(pl.T, 256), (unsigned)cdiv(pl.rows, 4));
    if (o.bn_part) {
      oa.bn_part = o.bn_part; oa.bn_bpg = o.bn_bpg;
      hipLaunchKernelGGL(wino_output_blk_bn_kernel, ogb, dim3(256), 0, st, oa);
    } else {
      hipLaunchKernelGGL(wino_output_blk_kernel, ogb, dim3(256), 0, st, oa);
    }
  } else if (pl.m == 2) hipLaunchKernelGGL(wino_output_kernel<2>, og, dim3(256), 0, st, oa);
  else hipLaunchKernelGGL(wino_output_kernel<4>, og, dim3(256), 0, st, oa);
}

static int wino_run(const WinoPlan& pl, int N, int in_ch, int H, int W, const float* src, const float* mask,
                    const unsigned char* code_in, int Hp, int Wp, const float* U, const float* bias, int relu, float* y,
                    float* pool_y, unsigned char* code_out, void* ws, hipStream_t st,
                    const WinoCat* in_cat = nullptr, const WinoCat* out_cat = nullptr, float* v_keep = nullptr,
                    const unsigned short* mask_bits = nullptr, unsigned short* relu_bits_out = nullptr,
                    double* bn_part = nullptr, int bn_bpg = 0) {
  float* V = v_keep ? v_keep : (float*)ws;       // v_keep: the caller keeps the transformed input for the weight gradient
  float* Mb = (float*)((char*)ws + ((pl.v_bytes + 255) & ~(size_t)255));
  const WinoInSrc in = {src, mask, mask_bits, code_in, Hp, Wp, in_cat};
  wino_stage_input(pl, N, in_ch, H, W, in, V, st);
  const WinoGemmArgs ga = wino_stage_gemm(pl, U, V, Mb, N, H, W, st);
  const WinoOutDst out = {bias, relu, y, pool_y, code_out, out_cat, relu_bits_out, nullptr, bn_part, bn_bpg};
  wino_stage_output(pl, ga, Mb, N, H, W, out, st);
  return 0;
}

static double conv_flops(const fcd_conv_desc* d) { return 2.0 * d->N * d->K * (double)d->P * d->Q * d->C * 9; }
// bytes the three passes stream: x in, V out + in, M out + in, y out, transformed filters in
static double wino_bytes(const WinoPlan& pl) {
  const double v = (double)pl.v_bytes, m = (double)pl.m_bytes;
  const int a2 = pl.A2, mm = pl.m * pl.m;
  return v / a2 * mm + 2.0 * v + 2.0 * m + m / a2 * mm + 4.0 * a2 * pl.rows * pl.Kc;
}

// y = [relu](conv(x, w) + bias)  or, when pool_y != NULL, pool_y / code = maxpool2(relu(conv + bias))
// Bytes of the forward pass's transformed input V the WEIGHT GRADIENT of this layer can consume instead of transforming x
// again (fcd_conv2d_fwd_wino_keepv -> fcd_conv2d_bwd_weight_bias_v); 0 when the layer's passes do not line up that way
extern "C" size_t fcd_conv_wino_keepv_bytes(const fcd_conv_desc* d);

extern "C" int fcd_conv2d_fwd_wino_x(const fcd_conv_desc* d, const float* x, const float* U, const float* bias, float* y,
                                     int fuse_relu, float* pool_y, unsigned char* code, void* ws, size_t ws_bytes,
                                     const fcd_wino_fwd_extras* ex, void* stream);
extern "C" int fcd_conv2d_fwd_wino_keepv(const fcd_conv_desc* d, const float* x, const float* U, const float* bias, float* y,
                                         int fuse_relu, float* pool_y, unsigned char* code, void* ws, size_t ws_bytes,
                                         float* v_keep, void* stream);
extern "C" int fcd_conv2d_fwd_wino(const fcd_conv_desc* d, const float* x, const float* U, const float* bias, float* y,
                                   int fuse_relu, float* pool_y, unsigned char* code, void* ws, size_t ws_bytes,
                                   void* stream) {
  return fcd_conv2d_fwd_wino_keepv(d, x, U, bias, y, fuse_relu, pool_y, code, ws, ws_bytes, nullptr, stream);
}
extern "C" int fcd_conv2d_fwd_wino_keepv(const fcd_conv_desc* d, const float* x, const float* U, const float* bias, float* y,
                                         int fuse_relu, float* pool_y, unsigned char* code, void* ws, size_t ws_bytes,
                                         float* v_keep, void* stream) {
  fcd_wino_fwd_extras ex;
  memset(&ex, 0, sizeof(ex));
  ex.v_keep = v_keep;
  return fcd_conv2d_fwd_wino_x(d, x, U, bias, y, fuse_relu, pool_y, code, ws, ws_bytes, &ex, stream);
}

// workgroups of the output transform per BatchNorm sample group, 0 when the statistics cannot come out of it (layer not
// on the blocked F(4x4) path, or a workgroup's 256 tiles would straddle two groups)
extern "C" int fcd_conv_wino_bn_split(const fcd_conv_desc* d, int groups) {
  WinoPlan pl;
  if (!d || !fcd_sw(FCD_SW_WINO_BNSTATS) || groups < 1 || !wino_plan(d, 0, &pl) || !wino_blk_path(pl) || d->N % groups) return 0;
  const long long per_group = pl.T / groups;
  if (per_group < 256 || per_group % 256) return 0;
  return (int)(per_group / 256);
}
extern "C" size_t fcd_conv_wino_bn_part_bytes(const fcd_conv_desc* d, int groups) {
  return (size_t)fcd_conv_wino_bn_split(d, groups) * groups * d->K * 3 * sizeof(double);
}

extern "C" int fcd_conv2d_fwd_wino_x(const fcd_conv_desc* d, const float* x, const float* U, const float* bias, float* y,
                                     int fuse_relu, float* pool_y, unsigned char* code, void* ws, size_t ws_bytes,
                                     const fcd_wino_fwd_extras* ex, void* stream) {
  float* v_keep = ex ? ex->v_keep : nullptr;
  double* bn_part = ex ? ex->bn_part : nullptr;
  const int bn_bpg = bn_part ? fcd_conv_wino_bn_split(d, ex->bn_groups) : 0;
  FCD_CHECK_ARG(d && x && U && (y || (pool_y && code)), "fcd_conv2d_fwd_wino: null pointer");
  FCD_CHECK_ARG(!bn_part || (bn_bpg > 0 && !pool_y && !fuse_relu), "fcd_conv2d_fwd_wino_x: BatchNorm partial sums not available for this call "
                "(fcd_conv_wino_bn_split(d, groups) == 0, or a fused ReLU / pool in front of the BatchNorm)");
  FCD_CHECK_ARG(!v_keep || fcd_conv_wino_keepv_bytes(d) > 0, "fcd_conv2d_fwd_wino_keepv: fcd_conv_wino_keepv_bytes(d) == 0 for this layer");
  WinoPlan pl;
  FCD_CHECK_ARG(wino_plan(d, 0, &pl), "fcd_conv2d_fwd_wino: layer is not planned for the Winograd path");
  if (!ws || ws_bytes < fcd_conv_wino_ws_bytes(d, 0)) {
    fcd_set_error("fcd_conv2d_fwd_wino: workspace %zu < %zu bytes", ws_bytes, fcd_conv_wino_ws_bytes(d, 0));
    return FCD_ERR_WORKSPACE;
  }
  FcdProfScope prof(FCD_K_WINO_FWD, (hipStream_t)stream, conv_flops(d), wino_bytes(pl), fcd_prof_tag_desc("wino_fwd", d));
  wino_run(pl, d->N, d->C, d->H, d->W, x, nullptr, nullptr, 0, 0, U, bias, (fuse_relu || pool_y) ? 1 : 0,
           pool_y ? nullptr : y, pool_y, code, ws, (hipStream_t)stream, nullptr, nullptr, v_keep, nullptr, nullptr, bn_part, bn_bpg);
  FCD_LAUNCH_CHECK("conv2d_fwd_wino");
  return FCD_OK;
}

// [r3] ReLU mask of a frozen F(4x4) layer as 16 bits per (n, k, 4 x 4 tile) instead of the fp32 activation: the forward
// output transform writes it next to y, the gated input transform of the data gradient reads it (4.25 -> 3.28 tensor
// passes on the gated transforms; y itself is free to go once the next layer has consumed it).  0 when the layer's
// forward and data gradient do not both run as F(4x4).
extern "C" size_t fcd_conv_wino_relu_bits_bytes(const fcd_conv_desc* d) {
  WinoPlan pf, pd;
  if (!d || !wino_plan(d, 0, &pf) || !wino_plan(d, 1, &pd) || pf.m != 4 || pd.m != 4) return 0;
  return (size_t)d->N * d->K * pf.TH * pf.TW * sizeof(unsigned short);
}

extern "C" int fcd_conv2d_fwd_wino_relu_bits(const fcd_conv_desc* d, const float* x, const float* U, const float* bias, float* y,
                                             unsigned short* relu_bits, void* ws, size_t ws_bytes, void* stream) {
  FCD_CHECK_ARG(d && x && U && y && relu_bits, "fcd_conv2d_fwd_wino_relu_bits: null pointer");
  FCD_CHECK_ARG(fcd_conv_wino_relu_bits_bytes(d) > 0, "fcd_conv2d_fwd_wino_relu_bits: fcd_conv_wino_relu_bits_bytes(d) == 0");
  WinoPlan pl;
  wino_plan(d, 0, &pl);
  if (!ws || ws_bytes < fcd_conv_wino_ws_bytes(d, 0)) {
    fcd_set_error("fcd_conv2d_fwd_wino_relu_bits: workspace %zu < %zu bytes", ws_bytes, fcd_conv_wino_ws_bytes(d, 0));
    return FCD_ERR_WORKSPACE;
  }
  FcdProfScope prof(FCD_K_WINO_FWD, (hipStream_t)stream, conv_flops(d), wino_bytes(pl), fcd_prof_tag_desc("wino_fwd", d));
  wino_run(pl, d->N, d->C, d->H, d->W, x, nullptr, nullptr, 0, 0, U, bias, 1, y, nullptr, nullptr, ws, (hipStream_t)stream,
           nullptr, nullptr, nullptr, nullptr, relu_bits);
  FCD_LAUNCH_CHECK("conv2d_fwd_wino_relu_bits");
  return FCD_OK;
}

extern "C" int fcd_conv2d_bwd_data_wino_bits(const fcd_conv_desc* d, const float* dy, const unsigned short* relu_bits,
                                             const float* U, float* dx, void* ws, size_t ws_bytes, void* stream) {
  FCD_CHECK_ARG(d && dy && relu_bits && U && dx, "fcd_conv2d_bwd_data_wino_bits: null pointer");
  FCD_CHECK_ARG(fcd_conv_wino_relu_bits_bytes(d) > 0, "fcd_conv2d_bwd_data_wino_bits: fcd_conv_wino_relu_bits_bytes(d) == 0");
  WinoPlan pl;
  wino_plan(d, 1, &pl);
  if (!ws || ws_bytes < fcd_conv_wino_ws_bytes(d, 1)) {
    fcd_set_error("fcd_conv2d_bwd_data_wino_bits: workspace %zu < %zu bytes", ws_bytes, fcd_conv_wino_ws_bytes(d, 1));
    return FCD_ERR_WORKSPACE;
  }
  FcdProfScope prof(FCD_K_WINO_DGRAD, (hipStream_t)stream, conv_flops(d), wino_bytes(pl), fcd_prof_tag_desc("wino_dgrad", d));
  wino_run(pl, d->N, d->K, d->P, d->Q, dy, nullptr, nullptr, 0, 0, U, nullptr, 0, dx, nullptr, nullptr, ws, (hipStream_t)stream,
           nullptr, nullptr, nullptr, relu_bits, nullptr);
  FCD_LAUNCH_CHECK("conv2d_bwd_data_wino_bits");
  return FCD_OK;
}

// dx = conv_transpose(dy') with dy' = dy, dy * [relu_out > 0], or the pooled gradient routed by pool_code
extern "C" int fcd_conv2d_bwd_data_wino(const fcd_conv_desc* d, const float* dy, const float* relu_out,
                                        const unsigned char* pool_code, const float* U, float* dx, void* ws,
                                        size_t ws_bytes, void* stream) {
  FCD_CHECK_ARG(d && dy && U && dx, "fcd_conv2d_bwd_data_wino: null pointer");
  WinoPlan pl;
  FCD_CHECK_ARG(wino_plan(d, 1, &pl), "fcd_conv2d_bwd_data_wino: layer is not planned for the Winograd path");
  if (!ws || ws_bytes < fcd_conv_wino_ws_bytes(d, 1)) {
    fcd_set_error("fcd_conv2d_bwd_data_wino: workspace %zu < %zu bytes", ws_bytes, fcd_conv_wino_ws_bytes(d, 1));
    return FCD_ERR_WORKSPACE;
  }
  FcdProfScope prof(FCD_K_WINO_DGRAD, (hipStream_t)stream, conv_flops(d), wino_bytes(pl), fcd_prof_tag_desc("wino_dgrad", d));
  wino_run(pl, d->N, d->K, d->P, d->Q, dy, pool_code ? nullptr : relu_out, pool_code, d->P / 2, d->Q / 2, U, nullptr, 0,
           dx, nullptr, nullptr, ws, (hipStream_t)stream);
  FCD_LAUNCH_CHECK("conv2d_bwd_data_wino");
  return FCD_OK;
}

// =============================================================================================
// [r5] Runs of frozen 3x3 layers with only a ReLU between them (the VGG16 stack of the perception term, reference Loss.py:25-36:
// conv + ReLU pairs between two max-pools, requires_grad = False): input transform -> GEMM -> [fused output -> input transform
// (conv_wino_chain.hip) -> GEMM] x (n - 1) -> output transform.  The activations between the layers of a run are never written
// as tensors; the backward run needs only their sign bits.  d[0 .. n) are the layers' descriptors in forward order.
static bool wino_chain_plans(const fcd_conv_desc* d, int n, int mode, WinoPlan* pls) {
  if (!d || n < 1 || n > 8) return false;
  for (int i = 0; i < n; ++i) {
    if (!wino_plan(&d[i], mode, &pls[i]) || pls[i].m != 4 || !wino_blk_path(pls[i])) return false;
    if (d[i].N != d[0].N || d[i].H != d[0].H || d[i].W != d[0].W || d[i].P != d[0].H || d[i].Q != d[0].W) return false;
    if (i > 0 && (d[i].C != d[i - 1].K || !wino_oi_ok(d[i].C, d[i].H, d[i].W))) return false;
  }
  return true;
}
// 1: the run d[0 .. n) can go through fcd_conv2d_fwd_wino_chain (mode 0) / fcd_conv2d_bwd_data_wino_chain (mode 1)
extern "C" int fcd_conv_wino_chain_ok(const fcd_conv_desc* d, int n, int mode) {
  WinoPlan pls[8];
  return (mode == 0 || mode == 1) && wino_chain_plans(d, n, mode, pls) ? 1 : 0;
}
// bytes of relu_bits[i] of a run: 16 sign bits per (n, k, 4 x 4 output tile) of layer d (0: the layer's forward is not F(4x4))
extern "C" size_t fcd_conv_wino_chain_bits_bytes(const fcd_conv_desc* d) {
  WinoPlan pf;
  if (!d || !wino_plan(d, 0, &pf) || pf.m != 4) return 0;
  return (size_t)d->N * d->K * pf.TH * pf.TW * sizeof(unsigned short);
}
extern "C" size_t fcd_conv_wino_chain_ws_bytes(const fcd_conv_desc* d, int n, int mode) {
  WinoPlan pls[8];
  if (!(mode == 0 || mode == 1) || !wino_chain_plans(d, n, mode, pls)) return 0;
  size_t v = 0, m = 0;
  for (int i = 0; i < n; ++i) { v = std::max(v, pls[i].v_bytes); m = std::max(m, pls[i].m_bytes); }
  return ((v + 255) & ~(size_t)255) + m + 256;
}

static void wino_stage_oi(const WinoPlan& prod, const WinoGemmArgs& ga, const float* Mb, const WinoPlan& cons, int N, int H, int W,
                          const float* bias, int relu, unsigned short* bits_out, const unsigned short* gate, float* V, hipStream_t st) {
  WinoOiArgs oa;
  memset(&oa, 0, sizeof(oa));
  oa.Mb = Mb; oa.xs_blk = ga.c_batch; oa.tblk = ga.c_tblk;
  oa.bias = bias; oa.relu = relu; oa.bits_out = bits_out; oa.gate = gate; oa.V = V;
  oa.N = N; oa.K = prod.rows; oa.H = H; oa.W = W; oa.TH = cons.TH; oa.TW = cons.TW; oa.Q = cons.Q; oa.T = cons.T;
  FcdProfScope p(FCD_K_WINO_XFORM, st, 0.0, (double)prod.m_bytes + (double)cons.v_bytes + 2.0 * prod.rows * (double)cons.T,
                 fcd_prof_tagf("oi gate=%d K=%d img=%dx%dx%d", gate ? 1 : 0, prod.rows, N, H, W));
  wino_oi_launch(oa, st);
}

// y = relu(conv_{n-1}( ... relu(conv_0(x)) ... )), or pool_y / code = its 2 x 2 max-pool.  relu_bits[i] (may be NULL, and the
// array itself may be NULL): sign bits of layer i's activation, fcd_conv_wino_chain_bits_bytes(&d[i]) bytes each; with pool_y the
// last entry is ignored (the pool's argmax code carries the sign).
extern "C" int fcd_conv2d_fwd_wino_chain(const fcd_conv_desc* d, int n, const float* x, const float* const* U,
                                         const float* const* bias, float* y, float* pool_y, unsigned char* code,
                                         unsigned short* const* relu_bits, void* ws, size_t ws_bytes, void* stream) {
  WinoPlan pls[8];
  FCD_CHECK_ARG(d && x && U && bias && (y || (pool_y && code)), "fcd_conv2d_fwd_wino_chain: null pointer");
  FCD_CHECK_ARG(wino_chain_plans(d, n, 0, pls), "fcd_conv2d_fwd_wino_chain: fcd_conv_wino_chain_ok(d, n, 0) == 0 for this run");
  const size_t need = fcd_conv_wino_chain_ws_bytes(d, n, 0);
  if (!ws || ws_bytes < need) {
    fcd_set_error("fcd_conv2d_fwd_wino_chain: workspace %zu < %zu bytes", ws_bytes, need);
    return FCD_ERR_WORKSPACE;
  }
  size_t vmax = 0;
  for (int i = 0; i < n; ++i) vmax = std::max(vmax, pls[i].v_bytes);
  float* V = (float*)ws;
  float* Mb = (float*)((char*)ws + ((vmax + 255) & ~(size_t)255));
  hipStream_t st = (hipStream_t)stream;
  const int N = d[0].N, H = d[0].H, W = d[0].W;
  WinoGemmArgs ga;
  for (int i = 0; i < n; ++i) {
    FCD_CHECK_ARG(U[i], "fcd_conv2d_fwd_wino_chain: null filter pack");
    const WinoPlan& pl = pls[i];
    const double v = (double)pl.v_bytes, m = (double)pl.m_bytes;
    // bytes of this layer's share of the run: its input side (x or the fused kernel: M of the previous layer -> V), the GEMM,
    // and for the last layer the output transform
    const double in_b = i == 0 ? 4.0 * N * d[0].C * H * W + v : (double)pls[i - 1].m_bytes + v;
    const double out_b = i == n - 1 ? m + m / pl.A2 * 16.0 * (pool_y ? 0.3125 : 1.0) : 0.0;
    FcdProfScope prof(FCD_K_WINO_FWD, st, conv_flops(&d[i]), in_b + v + m + out_b + 4.0 * pl.A2 * pl.rows * pl.Kc,
                      fcd_prof_tag_desc("wino_fwd_chain", &d[i]));
    if (i == 0) {
      const WinoInSrc in = {x, nullptr, nullptr, nullptr, 0, 0, nullptr};
      wino_stage_input(pl, N, d[0].C, H, W, in, V, st);
    } else {
      wino_stage_oi(pls[i - 1], ga, Mb, pl, N, H, W, bias[i - 1], 1, relu_bits ? relu_bits[i - 1] : nullptr, nullptr, V, st);
    }
    ga = wino_stage_gemm(pl, U[i], V, Mb, N, H, W, st);
    if (i == n - 1) {
      const WinoOutDst out = {bias[i], 1, pool_y ? nullptr : y, pool_y, code, nullptr,
                              (!pool_y && relu_bits) ? relu_bits[i] : nullptr, nullptr, nullptr, 0};
      wino_stage_output(pl, ga, Mb, N, H, W, out, st);
    }
  }
  FCD_LAUNCH_CHECK("conv2d_fwd_wino_chain");
  return FCD_OK;
}

// Data gradient of the run: dy is the gradient w.r.t. the run's output -- pooled (pool_code != NULL) or not, in which case it is
// gated with relu_bits[n - 1] when that entry is not NULL.  relu_bits[i], i < n - 1: the sign bits the forward run wrote (all
// needed).  gate_in (optional): sign bits of the run's INPUT tensor (the activation of the layer in front of the run): dx is
// zeroed where that activation was <= 0, for a preceding layer whose own data-gradient kernel cannot gate by bits.
// U1[i]: mode-1 filter packs.  dx: (N, C_0, H, W).
extern "C" int fcd_conv2d_bwd_data_wino_chain(const fcd_conv_desc* d, int n, const float* dy, const unsigned char* pool_code,
                                              const unsigned short* const* relu_bits, const unsigned short* gate_in,
                                              const float* const* U1, float* dx, void* ws, size_t ws_bytes, void* stream) {
  WinoPlan pls[8];
  FCD_CHECK_ARG(d && dy && U1 && dx && (n == 1 || relu_bits), "fcd_conv2d_bwd_data_wino_chain: null pointer");
  FCD_CHECK_ARG(wino_chain_plans(d, n, 1, pls), "fcd_conv2d_bwd_data_wino_chain: fcd_conv_wino_chain_ok(d, n, 1) == 0 for this run");
  const size_t need = fcd_conv_wino_chain_ws_bytes(d, n, 1);
  if (!ws || ws_bytes < need) {
    fcd_set_error("fcd_conv2d_bwd_data_wino_chain: workspace %zu < %zu bytes", ws_bytes, need);
    return FCD_ERR_WORKSPACE;
  }
  size_t vmax = 0;
  for (int i = 0; i < n; ++i) vmax = std::max(vmax, pls[i].v_bytes);
  float* V = (float*)ws;
  float* Mb = (float*)((char*)ws + ((vmax + 255) & ~(size_t)255));
  hipStream_t st = (hipStream_t)stream;
  const int N = d[0].N, H = d[0].H, W = d[0].W;
  WinoGemmArgs ga;
  for (int i = n - 1; i >= 0; --i) {
    FCD_CHECK_ARG(U1[i] && (i == 0 || relu_bits[i - 1]), "fcd_conv2d_bwd_data_wino_chain: null filter pack / sign bits");
    const WinoPlan& pl = pls[i];       // rows = C_i, reduction = K_i
    const double v = (double)pl.v_bytes, m = (double)pl.m_bytes;
    const double in_b = i == n - 1 ? 4.0 * N * d[i].K * H * W * (pool_code ? 0.3125 : 1.03125) + v : (double)pls[i + 1].m_bytes + v;
    const double out_b = i == 0 ? m + m / pl.A2 * 16.0 : 0.0;
    FcdProfScope prof(FCD_K_WINO_DGRAD, st, conv_flops(&d[i]), in_b + v + m + out_b + 4.0 * pl.A2 * pl.rows * pl.Kc,
                      fcd_prof_tag_desc("wino_dgrad_chain", &d[i]));
    if (i == n - 1) {
      const WinoInSrc in = {dy, nullptr, pool_code ? nullptr : (relu_bits ? relu_bits[i] : nullptr), pool_code, H / 2, W / 2, nullptr};
      wino_stage_input(pl, N, d[i].K, H, W, in, V, st);
    } else {
      // products of layer i + 1's data gradient = gradient w.r.t. layer i's activation: gate with its sign bits
      wino_stage_oi(pls[i + 1], ga, Mb, pl, N, H, W, nullptr, 0, nullptr, relu_bits[i], V, st);
    }
    ga = wino_stage_gemm(pl, U1[i], V, Mb, N, H, W, st);
    if (i == 0) {
      const WinoOutDst out = {nullptr, 0, dx, nullptr, nullptr, nullptr, nullptr, gate_in, nullptr, 0};
      wino_stage_output(pl, ga, Mb, N, H, W, out, st);
    }
  }
  FCD_LAUNCH_CHECK("conv2d_bwd_data_wino_chain");
  return FCD_OK;
}

// =============================================================================================
// Weight gradient through the same transforms (m = 4):
//   dU_xi[k][c] = sum_t W_xi[k][t] * V_xi[c][t],  W = A dY_t A^T (6x6 from the 4x4 gradient tile),
//   V = B^T d_t B as in the forward pass,  dw[k][c] = G^T dU G  (3x3 from 6x6)
// i.e. 36 GEMMs with the reduction over the image tiles t -- a quarter of the direct weight-gradient
// multiplies.  Both operands are written tile-contiguous ([xi][channel][Tpad], zero padded to a
// multiple of 32 tiles), so the transform kernels are pure streaming passes (thread = tile) and the
// GEMM is the same kernel with a row-major B and, for small filters, a split reduction whose partial
// dU blocks are summed in fixed order by the final transform.  The bias gradient falls out of the dY
// pass (per-block channel sums).
struct WinoWgArgs {
  const float* src;      // x (N, C, H, W) or dy (N, K, H, W)
  const float* mask;     // dy only: ReLU output of the layer (NULL: none)
  float* dst;            // [36][ch][Tpad]
  float* psum;           // dy only: [gridDim.x][ch] per-block channel sums (NULL: none)
  unsigned short* planes;  // dy only, split GEMM: the three bf16 parts [3][36][ch][Tpad] INSTEAD of dst (NULL: fp32 dst)
  WinoCat cat;             // x only: src = cat(cat.p[...]) (cat.n > 0)
  int N, CH, H, W, TH, TW;
  long long T, Tpad;
};

// V'[xi][c][t]: thread = tile, block = 256 consecutive tiles of one channel
__global__ __launch_bounds__(256) void wino_wg_input_kernel(WinoWgArgs a) {
  constexpr int A = 6;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (t >= a.Tpad) return;
  float* vout = a.dst + (size_t)c * a.Tpad + t;
  const size_t xs = (size_t)a.CH * a.Tpad;
  if (t >= a.T) {
#pragma unroll
    for (int i = 0; i < A * A; ++i) vout[(size_t)i * xs] = 0.f;
    return;
  }
  const int tx = (int)(t % a.TW);
  const long long r2 = t / a.TW;
  const int ty = (int)(r2 % a.TH), n = (int)(r2 / a.TH);
  const float* xp = a.src + ((size_t)n * a.CH + c) * a.H * a.W;
  if (a.cat.n > 0) {
    int ch = c, chans;
    const float* base = wino_cat_pick(a.cat, ch, chans);
    xp = base + ((size_t)n * chans + ch) * a.H * a.W;
  }
  const int ih0 = ty * 4 - 1, iw0 = tx * 4 - 1;
  float d[A][A];
#pragma unroll
  for (int i = 0; i < A; ++i) {
    const int ih = ih0 + i;
    const bool rok = ih >= 0 && ih < a.H;
#pragma unroll
    for (int j = 0; j < A; ++j) {
      const int iw = iw0 + j;
      d[i][j] = (rok && iw >= 0 && iw < a.W) ? xp[(size_t)ih * a.W + iw] : 0.f;
    }
  }
  float t1[A][A];
#pragma unroll
  for (int i = 0; i < A; ++i)
#pragma unroll
    for (int j = 0; j < A; ++j) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < A; ++k)
        if (WinoMat<4>::BT(i, k) != 0.f) s += WinoMat<4>::BT(i, k) * d[k][j];
      t1[i][j] = s;
    }
#pragma unroll
  for (int i = 0; i < A; ++i)
#pragma unroll
    for (int j = 0; j < A; ++j) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < A; ++k)
        if (WinoMat<4>::BT(j, k) != 0.f) s += t1[i][k] * WinoMat<4>::BT(j, k);
      vout[(size_t)(i * A + j) * xs] = s;
    }
}

// W[xi][k][t] = A dY A^T, plus per-block channel sums of dY (bias gradient)
__global__ __launch_bounds__(256) void wino_wg_dy_kernel(WinoWgArgs a) {
  // one thread = TWO consecutive tiles (Tpad is even): float4 row reads where the rows allow it, and the three bf16
  // planes leave as 4-byte pairs (2-byte stores run at half the rate: the pass is store-bound, 3.4x the bytes it reads)
  constexpr int A = 6;
  __shared__ double red[16];
  const long long t0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 2;
  const int k = blockIdx.y;
  float dy[2][4][4];
  float lsum = 0.f;
  const bool vec = (a.W & 3) == 0;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const long long t = t0 + e;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) dy[e][i][j] = 0.f;
    if (t < a.T) {
      const int tx = (int)(t % a.TW);
      const long long r2 = t / a.TW;
      const int ty = (int)(r2 % a.TH), n = (int)(r2 / a.TH);
      const size_t base = ((size_t)n * a.CH + k) * a.H * a.W;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ih = ty * 4 + i;
        if (ih >= a.H) continue;
        const size_t off = base + (size_t)ih * a.W + tx * 4;
        if (vec) {                          // W % 4 == 0: the whole 4-column group is inside the row
          f32x4 v = *(const f32x4*)(a.src + off);
          if (a.mask) {
            const f32x4 m = *(const f32x4*)(a.mask + off);
#pragma unroll
            for (int j = 0; j < 4; ++j) if (!(m[j] > 0.f)) v[j] = 0.f;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) dy[e][i][j] = v[j];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (tx * 4 + j < a.W) {
              float v = a.src[off + j];
              if (a.mask && !(a.mask[off + j] > 0.f)) v = 0.f;
              dy[e][i][j] = v;
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) lsum += dy[e][i][j];
    }
  }
  if (a.psum != nullptr) {       // uniform: every thread of the block takes part
    const double bs = block_sum_d((double)lsum, red);
    if (threadIdx.x == 0) a.psum[(size_t)blockIdx.x * a.CH + k] = (float)bs;
  }
  if (t0 >= a.Tpad) return;
  float* wout = a.dst + (size_t)k * a.Tpad + t0;
  const size_t xs = (size_t)a.CH * a.Tpad;
  unsigned short* pout = a.planes ? a.planes + (size_t)k * a.Tpad + t0 : nullptr;
  const size_t ps = 36 * xs;
  float t1[2][A][4];   // A dY : A[q][i] = AT(i, q)
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int q = 0; q < A; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (WinoMat<4>::AT(i, q) != 0.f) s += WinoMat<4>::AT(i, q) * dy[e][i][j];
        t1[e][q][j] = s;
      }
#pragma unroll
  for (int q = 0; q < A; ++q)
#pragma unroll
    for (int p2 = 0; p2 < A; ++p2) {
      float sv[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (WinoMat<4>::AT(j, p2) != 0.f) s += t1[e][q][j] * WinoMat<4>::AT(j, p2);
        sv[e] = s;                           // tiles past T: dy = 0 => exact zeros in every plane
      }
      const size_t o = (size_t)(q * A + p2) * xs;
      if (pout) {      // exact three-way split, as wino_filter_kernel does for U
        unsigned h[2], m[2], l[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          h[e] = bf16_rn_bits(sv[e]);
          const float r = sv[e] - __uint_as_float(h[e] << 16);
          m[e] = bf16_rn_bits(r);
          l[e] = bf16_rn_bits(r - __uint_as_float(m[e] << 16));
        }
        *(unsigned*)(pout + o) = h[0] | (h[1] << 16);
        *(unsigned*)(pout + ps + o) = m[0] | (m[1] << 16);
        *(unsigned*)(pout + 2 * ps + o) = l[0] | (l[1] << 16);
      } else {
        *(float2*)(wout + o) = make_float2(sv[0], sv[1]);
      }
    }
}

// dw[k][c][3][3] = G^T (sum over splits of dU[split][xi][k][c]) G.  Thread = one (k, c) filter; the nine
// taps of the block's 256 filters are staged in LDS so that dw is written as 2304 consecutive floats.
__global__ __launch_bounds__(256) void wino_wg_final_kernel(const float* __restrict__ dU, float* __restrict__ dw, int K,
                                                            int C, int splits) {
  constexpr int A = 6;
  __shared__ float st9[256 * 9];
  const long long total = (long long)K * C;
  const long long i0 = (long long)blockIdx.x * 256;
  const long long i = i0 + threadIdx.x;
  if (i < total) {
    const size_t xs = (size_t)K * C, ss = 36 * xs;
    float u[A][A];
#pragma unroll
    for (int q = 0; q < A * A; ++q) {
      float s = 0.f;
      for (int sp = 0; sp < splits; ++sp) s += dU[(size_t)sp * ss + (size_t)q * xs + i];
      u[q / A][q % A] = s;
    }
    float t1[3][A];   // G^T u : G^T[r][q] = G(q, r)
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int p2 = 0; p2 < A; ++p2) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < A; ++q)
          if (WinoMat<4>::G(q, r) != 0.f) s += WinoMat<4>::G(q, r) * u[q][p2];
        t1[r][p2] = s;
      }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) {
        float s = 0.f;
#pragma unroll
        for (int p2 = 0; p2 < A; ++p2)
          if (WinoMat<4>::G(p2, s3) != 0.f) s += t1[r][p2] * WinoMat<4>::G(p2, s3);
        st9[threadIdx.x * 9 + r * 3 + s3] = s;
      }
  }
  __syncthreads();
  const long long nout = (total - i0 < 256 ? total - i0 : 256) * 9;
  for (int j = threadIdx.x; j < nout; j += 256) dw[i0 * 9 + j] = st9[j];
}

// Split partials of dU summed in fixed (ascending split) order, in place into split 0: one float4 of the 36 K C block per
// thread => a grid of 9 K C / 256 workgroups streaming the partials coalesced.  (The final transform used to do this sum
// itself with K C threads -- 64 workgroups for a 128 x 128 filter walking 43 x 36 strided reads each: 0.5 ms for 100 MB.)
__global__ __launch_bounds__(256) void wino_wg_splitsum_kernel(float* __restrict__ dU, long long n4, long long ss4, int splits) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4* p = (float4*)dU;
  float4 s = p[i];
  for (int sp = 1; sp < splits; ++sp) {
    const float4 v = p[(size_t)sp * ss4 + i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  p[i] = s;
}

__global__ __launch_bounds__(256) void wino_psum_fin_kernel(const float* __restrict__ psum, float* __restrict__ out, int C,
                                                            int nblk) {
  __shared__ double red[16];
  const int c = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) s += (double)psum[(size_t)b * C + c];
  s = block_sum_d(s, red);
  if (threadIdx.x == 0) out[c] = (float)s;
}

struct WinoWgPlan {
  int TH, TW, splits, sps, stages;
  long long T, Tpad;
  size_t w_bytes, v_bytes, du_bytes, psum_bytes;
};

// 0 = direct weight-gradient kernel
int fcd_wino_wgrad_plan(const fcd_conv_desc* d, WinoWgPlan* pl) {
  if (wino_env() != 4) return 0;
  if (!(d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1)) return 0;
  const int min_k = fcd_sw(FCD_SW_WINO_WG_MINK), min_c = fcd_sw(FCD_SW_WINO_WG_MINC);
  if (d->K < min_k || d->C < min_c || d->P < 4 || d->Q < 4) return 0;
  pl->TH = cdiv(d->H, 4);
  pl->TW = cdiv(d->W, 4);
  pl->T = (long long)d->N * pl->TH * pl->TW;
  pl->Tpad = (pl->T + 31) / 32 * 32;
  pl->stages = (int)(pl->Tpad / 32);
  const int blocks = cdiv(d->K, 128) * cdiv(d->C, 128) * 36;
  static int wg_wgs = -1;             // FCD_WINO_WG_WGS: workgroups the reduction split aims at
  if (wg_wgs < 0) { const char* e = getenv("FCD_WINO_WG_WGS"); wg_wgs = e ? atoi(e) : 1024; if (wg_wgs < 1) wg_wgs = 1024; }   // [r4] 1024 (was 1536): -2 % over the Segmentor's 16 layers
  int splits = cdiv(wg_wgs, blocks);
  if (splits > pl->stages) splits = pl->stages;
  if (splits > 64) splits = 64;
  if (splits < 1) splits = 1;
  pl->sps = cdiv(pl->stages, splits);
  pl->splits = cdiv(pl->stages, pl->sps);
  pl->w_bytes = (size_t)36 * d->K * pl->Tpad * 6;      // fp32, or three bf16 planes for the split GEMM
  pl->v_bytes = (size_t)36 * d->C * pl->Tpad * sizeof(float);
  pl->du_bytes = (size_t)pl->splits * 36 * d->K * d->C * sizeof(float);
  pl->psum_bytes = (size_t)cdiv64(pl->Tpad, 256) * d->K * sizeof(float);
  return 4;
}

size_t fcd_wino_wgrad_ws_bytes(const fcd_conv_desc* d) {
  WinoWgPlan pl;
  if (!fcd_wino_wgrad_plan(d, &pl)) return 0;
  return pl.w_bytes + pl.v_bytes + pl.du_bytes + pl.psum_bytes + 1024;
}

// called by fcd_conv2d_bwd_weight_bias (conv_wgrad.hip) when the plan says so; ws holds fcd_wino_wgrad_ws_bytes
static int wino_wgrad_run_impl(const fcd_conv_desc* d, const float* x, const WinoCat* xcat, const float* dy,
                               const float* relu_out, float* dw, float* db, void* ws, hipStream_t st, const float* v_fwd = nullptr);
int fcd_wino_wgrad_run(const fcd_conv_desc* d, const float* x, const float* dy, const float* relu_out, float* dw,
                       float* db, void* ws, hipStream_t st) {
  return wino_wgrad_run_impl(d, x, nullptr, dy, relu_out, dw, db, ws, st);
}

extern "C" size_t fcd_conv_wino_keepv_bytes(const fcd_conv_desc* d) {
  WinoPlan pf;
  WinoWgPlan pw;
  if (!d || !fcd_sw(FCD_SW_WINO_KEEPV) || !wino_split() || (d->C & 31) || !wino_plan(d, 0, &pf) || pf.m != 4 || !fcd_wino_wgrad_plan(d, &pw)) return 0;
  if (pf.T != pw.T || pf.TW != pw.TW) return 0;
  // the forward dispatch (_ops._fwd_conv) tries the fused F(2x2) kernel first, which never writes V, and a <= 64-row GEMM takes the
  // non-split kernel, which ignores the transposed-B layout: only layers whose forward really runs the split F(4x4) GEMM keep V
  // (reachable with FCD_WINO_MINROWS <= 64 only)
  if (d->K <= 64 || fcd_conv_wino2_plan(d, 0)) return 0;
  return pf.v_bytes;
}

// weight (+ bias) gradient from the forward pass's transformed input (fcd_conv2d_fwd_wino[_cat]_keepv) instead of x
extern "C" int fcd_conv2d_bwd_weight_bias_v(const fcd_conv_desc* d, const float* v_fwd, const float* dy, const float* relu_out,
                                            float* dw, float* db, void* ws, size_t ws_bytes, void* stream) {
  FCD_CHECK_ARG(d && v_fwd && dy && dw, "fcd_conv2d_bwd_weight_bias_v: null pointer");
  FCD_CHECK_ARG(fcd_conv_wino_keepv_bytes(d) > 0, "fcd_conv2d_bwd_weight_bias_v: this layer's weight gradient does not take the forward V");
  const size_t need = fcd_wino_wgrad_ws_bytes(d);
  if (!ws || ws_bytes < need) {
    fcd_set_error("fcd_conv2d_bwd_weight_bias_v: workspace %zu < %zu bytes", ws_bytes, need);
    return FCD_ERR_WORKSPACE;
  }
  FcdProfScope prof(FCD_K_CONV_WGRAD, (hipStream_t)stream, conv_flops(d),
                    4.0 * (2.25 * d->N * d->C * (double)d->H * d->W + (double)d->N * d->K * d->P * d->Q + (double)d->K * d->C * 9),
                    fcd_prof_tag_desc("wgrad", d));
  wino_wgrad_run_impl(d, nullptr, nullptr, dy, relu_out, dw, db, ws, (hipStream_t)stream, v_fwd);
  FCD_LAUNCH_CHECK("conv2d_bwd_weight_bias_v");
  return FCD_OK;
}

static int wino_wgrad_run_impl(const fcd_conv_desc* d, const float* x, const WinoCat* xcat, const float* dy,
                               const float* relu_out, float* dw, float* db, void* ws, hipStream_t st, const float* v_fwd) {
  WinoWgPlan pl;
  if (!fcd_wino_wgrad_plan(d, &pl)) return 1;
  FcdProfScope pw(FCD_K_WGRAD_WINO, st, 2.0 * d->N * d->K * (double)d->P * d->Q * d->C * 9, 0.0,
                  fcd_prof_tag_desc("wgrad_wino", d));
  char* wsp = (char*)ws;
  float* Wb = (float*)wsp; wsp += (pl.w_bytes + 255) & ~(size_t)255;
  float* Vb = (float*)wsp; wsp += (pl.v_bytes + 255) & ~(size_t)255;
  float* dU = (float*)wsp; wsp += (pl.du_bytes + 255) & ~(size_t)255;
  float* psum = (float*)wsp;
  const unsigned tb = (unsigned)cdiv64(pl.Tpad, 256);
  const bool split = wino_split() != 0;      // dY~ written as three bf16 planes, GEMM on the bf16 matrix pipe
  {
    FcdProfScope p1(FCD_K_WINO_XFORM, st, 0.0,
                    4.0 * ((v_fwd ? 0.0 : (double)d->N * d->C * d->H * d->W) + (double)d->N * d->K * d->P * d->Q) +
                        (double)pl.w_bytes / 6 * (split ? 6 : 4) + (v_fwd ? 0.0 : (double)pl.v_bytes),
                    fcd_prof_tag_desc(v_fwd ? "wgrad_in_dy_only" : "wgrad_in", d));
    WinoWgArgs ia;
    memset(&ia, 0, sizeof(ia));
    if (xcat) ia.cat = *xcat;
    ia.src = x; ia.dst = Vb; ia.N = d->N; ia.CH = d->C; ia.H = d->H; ia.W = d->W;
    ia.TH = pl.TH; ia.TW = pl.TW; ia.T = pl.T; ia.Tpad = pl.Tpad;
    if (!v_fwd) hipLaunchKernelGGL(wino_wg_input_kernel, dim3(tb, (unsigned)d->C), dim3(256), 0, st, ia);
    WinoWgArgs ya = ia;
    memset(&ya.cat, 0, sizeof(ya.cat));
    ya.src = dy; ya.mask = relu_out; ya.dst = Wb; ya.CH = d->K; ya.psum = db ? psum : nullptr;
    ya.planes = split ? (unsigned short*)Wb : nullptr;
    const unsigned tb2 = (unsigned)cdiv64(pl.Tpad, 512);     // two tiles per thread
    hipLaunchKernelGGL(wino_wg_dy_kernel, dim3(tb2, (unsigned)d->K), dim3(256), 0, st, ya);
    if (db) hipLaunchKernelGGL(wino_psum_fin_kernel, dim3((unsigned)d->K), dim3(256), 0, st, (const float*)psum, db, d->K, (int)tb2);
  }
  WinoGemmArgs ga;
  memset(&ga, 0, sizeof(ga));
  ga.A = Wb; ga.B = Vb; ga.C = dU;
  if (split) { ga.As = (const unsigned short*)Wb; ga.as_plane = 36LL * d->K * pl.Tpad; }
  ga.M = d->K; ga.N = d->C; ga.Kc = (int)pl.Tpad;
  ga.m_tiles = cdiv(d->K, 128); ga.n_tiles = cdiv(d->C, 128);
  ga.xcd_remap = 0;
  ga.a_ld = pl.Tpad; ga.a_batch = (long long)d->K * pl.Tpad;
  ga.b_ld = pl.Tpad; ga.b_adv = 32; ga.b_batch = (long long)d->C * pl.Tpad;
  if (v_fwd) {                     // forward V [xi][C / 32][T][32], read transposed by the split kernel (BT)
    ga.B = v_fwd; ga.bt = 1; ga.bt_T = pl.T;
    ga.b_batch = (long long)(d->C / 32) * pl.T * 32;
  }
  ga.stages_per_split = pl.sps;
  {
    FcdProfScope p2(split ? FCD_K_WINO_GEMM_SPLIT : FCD_K_WINO_GEMM, st, 2.0 * 36 * d->K * (double)d->C * (double)pl.Tpad,
                    (double)pl.w_bytes / 6 * (split ? 6 : 4) + (double)pl.v_bytes + (double)pl.du_bytes,
                    fcd_prof_tagf("wgrad M=%d N=%d Kc=%lld batch=36 splits=%d img=%dx%dx%d", d->K, d->C, pl.Tpad, pl.splits,
                                  d->N, d->H, d->W));
    wino_gemm_launch(ga, 36, pl.splits, st);
  }
  {
    FcdProfScope p3(FCD_K_WINO_XFORM, st, 0.0, (double)pl.du_bytes + 4.0 * 9 * d->K * d->C, fcd_prof_tag_desc("wgrad_fin", d));
    int fin_splits = pl.splits;
    const long long n = 36LL * d->K * d->C;
    if (pl.splits > 1 && (n & 3) == 0) {
      hipLaunchKernelGGL(wino_wg_splitsum_kernel, dim3((unsigned)cdiv64(n >> 2, 256)), dim3(256), 0, st, dU, n >> 2, n >> 2,
                         pl.splits);
      fin_splits = 1;
    }
    hipLaunchKernelGGL(wino_wg_final_kernel, dim3((unsigned)cdiv64((long long)d->K * d->C, 256)), dim3(256), 0, st,
                       (const float*)dU, dw, d->K, d->C, fin_splits);
  }
  return 0;
}

// =============================================================================================
// Virtual channel concatenation: entry points
static bool wino_cat_fill(WinoCat* k, const float* const* ptrs, const int* chans, int n, int total) {
  memset(k, 0, sizeof(*k));
  if (!ptrs || !chans || n < 1 || n > 3) return false;
  int sum = 0;
  for (int i = 0; i < n; ++i) {
    if (!ptrs[i] || chans[i] <= 0 || (chans[i] & 31)) return false;
    k->p[i] = ptrs[i]; k->c[i] = chans[i]; sum += chans[i];
  }
  k->n = n;
  return sum == total;
}

// 1 = forward, data gradient and weight gradient of this layer all run on kernels that take tensor lists
extern "C" int fcd_conv_wino_cat_ok(const fcd_conv_desc* d) {
  WinoPlan pf, pd;
  WinoWgPlan pw;
  if (!d || !wino_plan(d, 0, &pf) || !wino_plan(d, 1, &pd) || !fcd_wino_wgrad_plan(d, &pw)) return 0;
  if (fcd_sw(FCD_SW_WINO_IN_ROLL) <= 1) return 0;
  return wino_cat_input_ok(pf, d->W) ? 1 : 0;
}

extern "C" int fcd_conv2d_fwd_wino_cat_keepv(const fcd_conv_desc* d, const float* const* src, const int* chans, int nsrc,
                                             const float* U, const float* bias, float* y, int fuse_relu, void* ws,
                                             size_t ws_bytes, float* v_keep, void* stream);
extern "C" int fcd_conv2d_fwd_wino_cat(const fcd_conv_desc* d, const float* const* src, const int* chans, int nsrc,
                                       const float* U, const float* bias, float* y, int fuse_relu, void* ws, size_t ws_bytes,
                                       void* stream) {
  return fcd_conv2d_fwd_wino_cat_keepv(d, src, chans, nsrc, U, bias, y, fuse_relu, ws, ws_bytes, nullptr, stream);
}
extern "C" int fcd_conv2d_fwd_wino_cat_x(const fcd_conv_desc* d, const float* const* src, const int* chans, int nsrc,
                                         const float* U, const float* bias, float* y, int fuse_relu, void* ws,
                                         size_t ws_bytes, const fcd_wino_fwd_extras* ex, void* stream);
extern "C" int fcd_conv2d_fwd_wino_cat_keepv(const fcd_conv_desc* d, const float* const* src, const int* chans, int nsrc,
                                             const float* U, const float* bias, float* y, int fuse_relu, void* ws,
                                             size_t ws_bytes, float* v_keep, void* stream) {
  fcd_wino_fwd_extras ex;
  memset(&ex, 0, sizeof(ex));
  ex.v_keep = v_keep;
  return fcd_conv2d_fwd_wino_cat_x(d, src, chans, nsrc, U, bias, y, fuse_relu, ws, ws_bytes, &ex, stream);
}
extern "C" int fcd_conv2d_fwd_wino_cat_x(const fcd_conv_desc* d, const float* const* src, const int* chans, int nsrc,
                                         const float* U, const float* bias, float* y, int fuse_relu, void* ws,
                                         size_t ws_bytes, const fcd_wino_fwd_extras* ex, void* stream) {
  float* v_keep = ex ? ex->v_keep : nullptr;
  double* bn_part = ex ? ex->bn_part : nullptr;
  const int bn_bpg = bn_part ? fcd_conv_wino_bn_split(d, ex->bn_groups) : 0;
  FCD_CHECK_ARG(d && U && y, "fcd_conv2d_fwd_wino_cat: null pointer");
  FCD_CHECK_ARG(!bn_part || (bn_bpg > 0 && !fuse_relu), "fcd_conv2d_fwd_wino_cat_x: BatchNorm partial sums not available for this call");
  FCD_CHECK_ARG(!v_keep || fcd_conv_wino_keepv_bytes(d) > 0, "fcd_conv2d_fwd_wino_cat_keepv: fcd_conv_wino_keepv_bytes(d) == 0 for this layer");
  WinoCat cat;
  FCD_CHECK_ARG(wino_cat_fill(&cat, src, chans, nsrc, d->C),
                "fcd_conv2d_fwd_wino_cat: 1..3 tensors, channel counts multiples of 32 that add up to C");
  FCD_CHECK_ARG(fcd_conv_wino_cat_ok(d), "fcd_conv2d_fwd_wino_cat: layer does not take tensor lists (fcd_conv_wino_cat_ok)");
  WinoPlan pl;
  wino_plan(d, 0, &pl);
  if (!ws || ws_bytes < fcd_conv_wino_ws_bytes(d, 0)) {
    fcd_set_error("fcd_conv2d_fwd_wino_cat: workspace %zu < %zu bytes", ws_bytes, fcd_conv_wino_ws_bytes(d, 0));
    return FCD_ERR_WORKSPACE;
  }
  FcdProfScope prof(FCD_K_WINO_FWD, (hipStream_t)stream, conv_flops(d), wino_bytes(pl), fcd_prof_tag_desc("wino_fwd", d));
  wino_run(pl, d->N, d->C, d->H, d->W, cat.p[0], nullptr, nullptr, 0, 0, U, bias, fuse_relu ? 1 : 0, y, nullptr, nullptr, ws,
           (hipStream_t)stream, &cat, nullptr, v_keep, nullptr, nullptr, bn_part, bn_bpg);
  FCD_LAUNCH_CHECK("conv2d_fwd_wino_cat");
  return FCD_OK;
}

extern "C" int fcd_conv2d_bwd_data_wino_cat(const fcd_conv_desc* d, const float* dy, const float* relu_out, const float* U,
                                            float* const* dsrc, const int* chans, int nsrc, void* ws, size_t ws_bytes,
                                            void* stream) {
  FCD_CHECK_ARG(d && dy && U, "fcd_conv2d_bwd_data_wino_cat: null pointer");
  WinoCat cat;
  FCD_CHECK_ARG(wino_cat_fill(&cat, (const float* const*)dsrc, chans, nsrc, d->C),
                "fcd_conv2d_bwd_data_wino_cat: 1..3 tensors, channel counts multiples of 32 that add up to C");
  WinoPlan pl;
  FCD_CHECK_ARG(wino_plan(d, 1, &pl), "fcd_conv2d_bwd_data_wino_cat: layer is not planned for the Winograd path");
  if (!ws || ws_bytes < fcd_conv_wino_ws_bytes(d, 1)) {
    fcd_set_error("fcd_conv2d_bwd_data_wino_cat: workspace %zu < %zu bytes", ws_bytes, fcd_conv_wino_ws_bytes(d, 1));
    return FCD_ERR_WORKSPACE;
  }
  FcdProfScope prof(FCD_K_WINO_DGRAD, (hipStream_t)stream, conv_flops(d), wino_bytes(pl), fcd_prof_tag_desc("wino_dgrad", d));
  wino_run(pl, d->N, d->K, d->P, d->Q, dy, relu_out, nullptr, 0, 0, U, nullptr, 0, const_cast<float*>(cat.p[0]), nullptr, nullptr,
           ws, (hipStream_t)stream, nullptr, &cat);
  FCD_LAUNCH_CHECK("conv2d_bwd_data_wino_cat");
  return FCD_OK;
}


// weight (+ bias) gradient of a layer whose input is the virtual concatenation of src[...]
extern "C" int fcd_conv2d_bwd_weight_bias_cat(const fcd_conv_desc* d, const float* const* src, const int* chans, int nsrc,
                                              const float* dy, const float* relu_out, float* dw, float* db, void* ws,
                                              size_t ws_bytes, void* stream) {
  FCD_CHECK_ARG(d && dy && dw, "fcd_conv2d_bwd_weight_bias_cat: null pointer");
  WinoCat cat;
  FCD_CHECK_ARG(wino_cat_fill(&cat, src, chans, nsrc, d->C),
                "fcd_conv2d_bwd_weight_bias_cat: 1..3 tensors, channel counts multiples of 32 that add up to C");
  const size_t need = fcd_wino_wgrad_ws_bytes(d);
  FCD_CHECK_ARG(need > 0, "fcd_conv2d_bwd_weight_bias_cat: layer does not take the Winograd weight-gradient form");
  if (!ws || ws_bytes < need) {
    fcd_set_error("fcd_conv2d_bwd_weight_bias_cat: workspace %zu < %zu bytes", ws_bytes, need);
    return FCD_ERR_WORKSPACE;
  }
  FcdProfScope prof(FCD_K_CONV_WGRAD, (hipStream_t)stream, conv_flops(d),
                    4.0 * ((double)d->N * d->C * d->H * d->W + (double)d->N * d->K * d->P * d->Q + (double)d->K * d->C * 9),
                    fcd_prof_tag_desc("wgrad", d));
  wino_wgrad_run_impl(d, cat.p[0], &cat, dy, relu_out, dw, db, ws, (hipStream_t)stream);
  FCD_LAUNCH_CHECK("conv2d_bwd_weight_bias_cat");
  return FCD_OK;
}
