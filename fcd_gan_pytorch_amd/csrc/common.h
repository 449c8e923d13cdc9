// Shared host/device helpers for libfcdgan_hip.so (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>

#include "../../include/fcdgan_hip.h"
#include "switches.h"

#define FCD_WAVE 64

// ---- error plumbing: never throw across the C ABI --------------------------
void fcd_set_error(const char* fmt, ...);

#define FCD_CHECK_ARG(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      fcd_set_error(__VA_ARGS__);                \
      return FCD_ERR_INVALID;                    \
    }                                            \
  } while (0)

#define FCD_LAUNCH_CHECK(name)                                              \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      fcd_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return FCD_ERR_LAUNCH;                                                \
    }                                                                       \
  } while (0)

// ---- per-kernel-family timing (HIP events on the launch stream) ------------
// Family ids index the table returned by fcd_prof_read().
enum {
  FCD_K_CONV_FWD = 0,   // implicit-GEMM MFMA conv used as forward
  FCD_K_CONV_DGRAD = 1, // same kernel used as data-gradient
  FCD_K_CONV_WGRAD = 2, // MFMA weight-gradient kernel (+ split reduce)
  FCD_K_PACK = 3,       // weight repacking
  FCD_K_NORM = 4,       // BN statistics / apply / backward, activations
  FCD_K_POOL = 5,       // maxpool / bilinear / avgpool
  FCD_K_LOSS = 6,       // masked recon, SSIM, reductions
  FCD_K_OPTIM = 7,      // Adam / RMSprop
  FCD_K_MISC = 8,
  FCD_K_WINO_FWD = 9,   // Winograd path (input transform + batched MFMA GEMM + output transform), forward
  FCD_K_WINO_DGRAD = 10,// ... used as data gradient
  FCD_K_WINO_GEMM = 11, // nested in 9/10: the batched MFMA GEMM alone (FLOPs = executed GEMM FLOPs)
  FCD_K_WINO_XFORM = 12,// nested in 9/10: input + output transform kernels (bytes streamed)
  FCD_K_WGRAD_WINO = 13,// nested in 2: weight-gradient calls that take the Winograd form (FLOPs = direct count)
  FCD_K_WINO2_FWD = 14, // fused Winograd F(2x2,3x3) kernel (64-row layers), forward; FLOPs = direct count (executed: x 16/36)
  FCD_K_WINO2_DGRAD = 15,
  FCD_K_WINO_GEMM_SPLIT = 16, // nested in 9/10: the batched GEMM on the bf16 matrix pipe (exact three-way operand split);
                              // FLOPs = fp32-equivalent GEMM FLOPs (the bf16 MFMAs execute 6x that)
  FCD_K_WGRAD_SPLIT = 17,     // nested in 2: the NCHW-direct 3x3 weight-gradient kernel on the bf16 matrix pipe (exact three-way split of
                              // x and dY); FLOPs = fp32-equivalent weight-gradient FLOPs (the bf16 MFMAs execute 6x that)
  FCD_K_COUNT = 18
};

struct FcdProfScope {
  int fam;
  hipStream_t st;
  hipEvent_t e0, e1;
  bool on;
  int detail_idx;
  // tag: optional per-launch label (layer geometry) kept when the detail log is on (fcd_prof_enable(2)); use
  // fcd_prof_tagf() to format one only when somebody is listening
  FcdProfScope(int family, hipStream_t stream, double flops, double bytes, const char* tag = nullptr);
  ~FcdProfScope();
};
// printf into a thread-local buffer when the per-launch detail log is on; returns NULL otherwise
const char* fcd_prof_tagf(const char* fmt, ...);
static inline const char* fcd_prof_tag_desc(const char* what, const fcd_conv_desc* d) {
  return fcd_prof_tagf("%s N=%d C=%d H=%d W=%d K=%d R=%d s=%d", what, d->N, d->C, d->H, d->W, d->K, d->R, d->stride);
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return cdiv(a, b) * b; }

// ---- device helpers ---------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum (blockDim.x multiple of 64, <= 1024).  Result valid in thread 0.
__device__ __forceinline__ double block_sum_d(double v, double* smem /* >= 16 doubles */) {
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) r += smem[i];
  }
  return r;
}

// activation codes shared by host + device
// act: 0 none, 1 ReLU, 2 LeakyReLU(slope scalar), 3 PReLU(slope from device pointer)
__device__ __forceinline__ float act_apply(float v, int act, float slope) {
  if (act == FCD_ACT_NONE) return v;
  if (act == FCD_ACT_RELU) return v > 0.f ? v : 0.f;
  return v > 0.f ? v : v * slope;
}
// derivative factor d act(v) / d v  (PyTorch convention: slope branch at v <= 0)
__device__ __forceinline__ float act_grad(float v, int act, float slope) {
  if (act == FCD_ACT_NONE) return 1.f;
  if (act == FCD_ACT_RELU) return v > 0.f ? 1.f : 0.f;
  return v > 0.f ? 1.f : slope;
}
