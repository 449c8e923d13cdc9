/* Oracle (test infrastructure): first-principles C restatement of the two operators that
 * carry ~98 % of the hot path's arithmetic -- direct 2-D convolution (forward, data-gradient,
 * weight-gradient) and train-mode BatchNorm statistics -- with double accumulation.
 * It pins the torch-op oracle (oracle/nets.py) itself: tests/test_oracle_c.py checks that
 * F.conv2d / F.batch_norm as used there (reference Module.py:25-31) equal these loops.
 * Plain C99, no dependencies.   cc -O2 -shared -fPIC conv_ref.c -o libfcd_oracle_c.so
 */
#include <stddef.h>

#define X(n, c, h, w) x[(((size_t)(n) * C + (c)) * H + (h)) * W + (w)]
#define Y(n, k, p, q) y[(((size_t)(n) * K + (k)) * P + (p)) * Q + (q)]
#define WT(k, c, r, s) wt[(((size_t)(k) * C + (c)) * R + (r)) * S + (s)]

void fcd_ref_conv2d_fwd(const float* x, const float* wt, const float* bias, float* y, int N, int C, int H,
                        int W, int K, int R, int S, int stride, int pad) {
  const int P = (H + 2 * pad - R) / stride + 1, Q = (W + 2 * pad - S) / stride + 1;
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k)
      for (int p = 0; p < P; ++p)
        for (int q = 0; q < Q; ++q) {
          double acc = bias ? (double)bias[k] : 0.0;
          for (int c = 0; c < C; ++c)
            for (int r = 0; r < R; ++r) {
              const int h = p * stride + r - pad;
              if (h < 0 || h >= H) continue;
              for (int s = 0; s < S; ++s) {
                const int w = q * stride + s - pad;
                if (w < 0 || w >= W) continue;
                acc += (double)X(n, c, h, w) * (double)WT(k, c, r, s);
              }
            }
          Y(n, k, p, q) = (float)acc;
        }
}

/* dx = d/dx, dw = d/dw of sum(y * dy) */
void fcd_ref_conv2d_bwd(const float* x, const float* wt, const float* y /* = dy */, float* dx, float* dw, int N,
                        int C, int H, int W, int K, int R, int S, int stride, int pad) {
  const int P = (H + 2 * pad - R) / stride + 1, Q = (W + 2 * pad - S) / stride + 1;
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c)
      for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w) {
          double acc = 0.0;
          for (int k = 0; k < K; ++k)
            for (int r = 0; r < R; ++r) {
              const int ph = h + pad - r;
              if (ph < 0 || ph % stride) continue;
              const int p = ph / stride;
              if (p >= P) continue;
              for (int s = 0; s < S; ++s) {
                const int qw = w + pad - s;
                if (qw < 0 || qw % stride) continue;
                const int q = qw / stride;
                if (q >= Q) continue;
                acc += (double)Y(n, k, p, q) * (double)WT(k, c, r, s);
              }
            }
          dx[(((size_t)n * C + c) * H + h) * W + w] = (float)acc;
        }
  for (int k = 0; k < K; ++k)
    for (int c = 0; c < C; ++c)
      for (int r = 0; r < R; ++r)
        for (int s = 0; s < S; ++s) {
          double acc = 0.0;
          for (int n = 0; n < N; ++n)
            for (int p = 0; p < P; ++p) {
              const int h = p * stride + r - pad;
              if (h < 0 || h >= H) continue;
              for (int q = 0; q < Q; ++q) {
                const int w = q * stride + s - pad;
                if (w < 0 || w >= W) continue;
                acc += (double)Y(n, k, p, q) * (double)X(n, c, h, w);
              }
            }
          dw[(((size_t)k * C + c) * R + r) * S + s] = (float)acc;
        }
}

/* train-mode BatchNorm2d statistics: mean, biased var per channel (Module.py:26) */
void fcd_ref_bn_stats(const float* x, double* mean, double* var, int N, int C, int HW) {
  for (int c = 0; c < C; ++c) {
    double s = 0.0, s2 = 0.0;
    for (int n = 0; n < N; ++n)
      for (int i = 0; i < HW; ++i) s += (double)x[((size_t)n * C + c) * HW + i];
    const double m = s / ((double)N * HW);
    for (int n = 0; n < N; ++n)
      for (int i = 0; i < HW; ++i) {
        const double d = (double)x[((size_t)n * C + c) * HW + i] - m;
        s2 += d * d;
      }
    mean[c] = m;
    var[c] = s2 / ((double)N * HW);
  }
}
