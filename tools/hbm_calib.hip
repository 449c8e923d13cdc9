// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts, in the access patterns the
// library's kernels use (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming
// read ... other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access
// pattern").  Each kernel streams `bytes` (default 2 GiB, far beyond the 256 MiB Infinity Cache) exactly once in and
// once out:
//   calib_copy_b128     global_load_dwordx4  -> global_store_dwordx4     (transform kernels, float4 paths)
//   calib_copy_b32      global_load_dword    -> global_store_dword       (direct-conv patch loads, scalar epilogues)
//   calib_copy_lds_dma  global_load_lds 16 B/lane -> ds_read_b128 -> global_store_dwordx4   (GEMM / conv slab DMA)
//   calib_read_b128     global_load_dwordx4 only (sum reduced to one store per block)
// Run:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o c -- tools/hbm_calib.bin
//       rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out2 -o c -- tools/hbm_calib.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

__global__ __launch_bounds__(256) void calib_copy_b128(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void calib_copy_b32(const float* __restrict__ a, float* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void calib_read_b128(const float4* __restrict__ a, float* __restrict__ out, size_t n4) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = a[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 12345.678f) out[blockIdx.x] = s;      // keeps the loads alive, (practically) never stores
}
__global__ __launch_bounds__(256) void calib_copy_lds_dma(const float* __restrict__ a, float4* __restrict__ b, size_t n4) {
  __shared__ __attribute__((aligned(16))) float buf[256 * 4];
  const int wave = threadIdx.x >> 6;
  for (size_t base = (size_t)blockIdx.x * 256; base < n4; base += (size_t)gridDim.x * 256) {
    // one 1 KiB wave-instruction per wave: lane l of wave w fetches 16 B from a + (base + 64 w + l) * 4 floats
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(a + (base + threadIdx.x) * 4), (lds_void_t*)(buf + wave * 256), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) (and everything else)
    __syncthreads();
    if (base + threadIdx.x < n4) b[base + threadIdx.x] = *(const float4*)(buf + threadIdx.x * 4);
    __syncthreads();
  }
}

int main(int argc, char** argv) {
  const size_t bytes = argc > 1 ? (size_t)atoll(argv[1]) : ((size_t)2 << 30);
  const size_t n = bytes / 4, n4 = bytes / 16;
  float *a, *b;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  hipMemset(a, 0x3c, bytes);
  hipMemset(b, 0, bytes);
  hipDeviceSynchronize();
  const int grid = 256 * 16;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(calib_copy_b128, dim3(grid), dim3(256), 0, 0, (const float4*)a, (float4*)b, n4);
    hipLaunchKernelGGL(calib_copy_b32, dim3(grid), dim3(256), 0, 0, (const float*)a, b, n);
    hipLaunchKernelGGL(calib_read_b128, dim3(grid), dim3(256), 0, 0, (const float4*)a, b, n4);
    hipLaunchKernelGGL(calib_copy_lds_dma, dim3(grid), dim3(256), 0, 0, (const float*)a, (float4*)b, n4);
  }
  if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 2; }
  float h[4];
  hipMemcpy(h, b, sizeof(h), hipMemcpyDeviceToHost);
  printf("calib: streamed %zu bytes per kernel per direction, check %g\n", bytes, (double)h[0]);
  hipFree(a); hipFree(b);
  return 0;
}
