// Which SIMD does each wave of a 320-thread (5-wave) / 384-thread (6-wave) workgroup land on?  (HW_ID.SIMD_ID)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned* out) {
  const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all bits
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = hw;
}
int main() {
  unsigned* d; hipMalloc(&d, 64 * 16 * 4);
  for (int threads : {320, 384, 512}) {
    hipMemset(d, 0, 64 * 16 * 4);
    hipLaunchKernelGGL(probe, dim3(8), dim3(threads), 0, 0, d);
    hipDeviceSynchronize();
    unsigned h[8 * 16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 3; ++b) {
      printf("threads %d block %d: simd of waves:", threads, b);
      for (int w = 0; w < threads / 64; ++w) printf(" %u", (h[b * 16 + w] >> 4) & 3);
      printf("   (cu %u)\n", (h[b * 16] >> 8) & 15);
    }
  }
  return 0;
}
