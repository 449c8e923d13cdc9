// torch.optim.Adam / torch.optim.RMSprop single-tensor update rules on flat fp32
// buffers (one launch per network).  HBM-bound: Adam 16 B read + 12 B write per
// parameter, RMSprop 12 B + 8 B; float4 access, grid-stride.
// Semantics follow torch 2.10 _single_tensor_adam / _single_tensor_rmsprop with
// the defaults the demos use (amsgrad=False, maximize=False, momentum=0,
// centered=False): Demo_USSS.py:121-122, Demo_RSSS.py:151-158, Demo_WSSS.py:116-122.
#include "common.h"

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float lr, float b1, float b2, float eps, float wd,
                            float bc1, float bc2_sqrt, float gs, const float* __restrict__ hyper) {
  if (hyper) { lr = hyper[0]; bc1 = hyper[1]; bc2_sqrt = hyper[2]; }      // hipGraph replays: per-step scalars live on the device
  const float step_size = lr / bc1;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    float gi = g[i] * gs;
    const float pi = p[i];
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    // exp_avg.lerp_(grad, 1-beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
    m[i] = mi;
    v[i] = vi;
  }
}

__global__ void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq,
                               long long n, float lr, float alpha, float eps, float wd, float gs,
                               const float* __restrict__ hyper) {
  if (hyper) lr = hyper[0];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    float gi = g[i] * gs;
    const float pi = p[i];
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    const float s = sq[i] * alpha + (1.f - alpha) * gi * gi;
    sq[i] = s;
    p[i] = pi - lr * (gi / (sqrtf(s) + eps));
  }
}

static inline int ew_grid(long long total) { return (int)std::min<long long>(cdiv64(total, 256), 4096); }

extern "C" int fcd_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream) {
  FCD_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "fcd_adam_step: bad arguments");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  FcdProfScope prof(FCD_K_OPTIM, (hipStream_t)stream, 0.0, 28.0 * n);
  hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n, lr,
                     beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale, (const float*)nullptr);
  FCD_LAUNCH_CHECK("adam_step");
  return FCD_OK;
}

// The same update with the per-step scalars read from DEVICE memory: hyper = {lr, 1 - beta1^t, sqrt(1 - beta2^t)} as floats (the
// host writes them before every step / graph replay -- the values fcd_adam_step derives from its arguments).  For train steps
// captured in a hipGraph: a replay re-issues the recorded launch with its recorded arguments, so a learning-rate schedule or Adam's
// bias correction can only reach the kernel through memory.
extern "C" int fcd_adam_step_h(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, float beta1,
                               float beta2, float eps, float weight_decay, float grad_scale, void* stream) {
  FCD_CHECK_ARG(p && g && m && v && hyper && n > 0, "fcd_adam_step_h: bad arguments");
  FcdProfScope prof(FCD_K_OPTIM, (hipStream_t)stream, 0.0, 28.0 * n);
  hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n, 0.f,
                     beta1, beta2, eps, weight_decay, 1.f, 1.f, grad_scale, hyper);
  FCD_LAUNCH_CHECK("adam_step_h");
  return FCD_OK;
}

extern "C" int fcd_rmsprop_step(float* p, const float* g, float* sq, int64_t n, float lr, float alpha, float eps,
                                float weight_decay, float grad_scale, void* stream) {
  FCD_CHECK_ARG(p && g && sq && n > 0, "fcd_rmsprop_step: bad arguments");
  FcdProfScope prof(FCD_K_OPTIM, (hipStream_t)stream, 0.0, 20.0 * n);
  hipLaunchKernelGGL(rmsprop_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, sq, (long long)n, lr,
                     alpha, eps, weight_decay, grad_scale, (const float*)nullptr);
  FCD_LAUNCH_CHECK("rmsprop_step");
  return FCD_OK;
}

// hyper = {lr} on the device: see fcd_adam_step_h
extern "C" int fcd_rmsprop_step_h(float* p, const float* g, float* sq, int64_t n, const float* hyper, float alpha, float eps,
                                  float weight_decay, float grad_scale, void* stream) {
  FCD_CHECK_ARG(p && g && sq && hyper && n > 0, "fcd_rmsprop_step_h: bad arguments");
  FcdProfScope prof(FCD_K_OPTIM, (hipStream_t)stream, 0.0, 20.0 * n);
  hipLaunchKernelGGL(rmsprop_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, sq, (long long)n, 0.f,
                     alpha, eps, weight_decay, grad_scale, hyper);
  FCD_LAUNCH_CHECK("rmsprop_step_h");
  return FCD_OK;
}
