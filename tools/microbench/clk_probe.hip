// Clock probe (gfx950): the shader clock a kernel actually runs at, as a function of how much of the chip it occupies.
// Each wave runs a chain of dependent v_fma_f32; wall time from HIP events, s_memtime and s_memrealtime deltas from wave 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
__global__ void chain(float* out, unsigned long long* tk, int n, float w) {
  float a = threadIdx.x * 0.001f;
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 64; ++u) a = __builtin_fmaf(a, w, 1.0f);
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
  if (blockIdx.x == 0 && threadIdx.x == 0) tk[0] = t1 - t0, tk[1] = r1 - r0;
}
int main() {
  float* out; unsigned long long* tk;
  CK(hipMalloc(&out, 4096 * 256 * 4)); CK(hipMalloc(&tk, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int n = 4096;   // 262144 dependent FMAs per wave
  for (int rep = 0; rep < 2; ++rep)
  for (int wgs : {1, 16, 64, 256, 1024}) for (int threads : {64, 256}) {
    hipLaunchKernelGGL(chain, dim3(wgs), dim3(threads), 0, 0, out, tk, n, 0.999f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(chain, dim3(wgs), dim3(threads), 0, 0, out, tk, n, 0.999f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, tk, 16, hipMemcpyDeviceToHost));
    const double ninst = 64.0 * n;
    printf("wgs %4d x %3d threads: %8.1f us  memtime ticks/fma %.3f  ns/fma %.3f  memtime GHz %.3f  realtime MHz %.1f\n", wgs, threads,
           ms * 1e3, h[0] / ninst, ms * 1e6 / ninst, h[0] / (ms * 1e6), h[1] / (ms * 1e3));
  }
  return 0;
}
