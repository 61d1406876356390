// Does global_load_lds accept the SGPR-base form (saddr + 32-bit VGPR offset) on gfx950, and does the data land where the
// VGPR-address form puts it?   hipcc --offload-arch=gfx950 lds_dma_saddr.hip -o /tmp/t && /tmp/t
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(1))) float gf32;
__global__ void k(const float* p, float* o, int mode) {
  extern __shared__ float sm[];
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)sm);
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) sm[i] = -1.f;
  __syncthreads();
  if (threadIdx.x < 64) {
    if (mode == 0) {
      const float* src = p + 256 + threadIdx.x * 4;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"((const gf32*)src), "s"(dst) : "memory");
    } else {
      const unsigned voff = threadIdx.x * 16;
      const float* base = p + 256;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) o[i] = sm[i];
}
int main() {
  float *p, *o, h[1024], hp[2048];
  for (int i = 0; i < 2048; ++i) hp[i] = (float)i;
  hipMalloc(&p, sizeof(hp)); hipMalloc(&o, sizeof(h));
  hipMemcpy(p, hp, sizeof(hp), hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, p, o, mode);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += h[i] != (float)(256 + i);
    for (int i = 256; i < 1024; ++i) bad += h[i] != -1.f;
    printf("mode %d (%s): %s, %d wrong, first floats %g %g %g %g\n", mode, mode ? "saddr" : "vaddr", hipGetErrorString(e), bad, h[0], h[1], h[2], h[255]);
  }
  return 0;
}
