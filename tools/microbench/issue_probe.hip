// Issue-rate probe (gfx950): cycles per VALU instruction for ONE wave per SIMD vs 2 / 4 waves per SIMD,
// independent vs dependent chains, v_fma_f32 vs v_pk_fma_f32 vs v_fmac_dpp.  One workgroup on one CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1;} } while (0)

template <int MODE, int NACC>
__global__ void probe(float* out, long long* cyc, float seed) {
  float acc[8];
  f32x2 pacc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { acc[i] = seed + i + threadIdx.x; pacc[i] = f32x2{acc[i], acc[i] + 1.f}; }
  const float w = seed * 0.5f;
  const f32x2 pw = f32x2{w, w + 0.1f};
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int u = 0; u < 64; ++u) {
      if (MODE == 0) acc[u % NACC] = __builtin_fmaf(acc[u % NACC], w, 1.0f);
      if (MODE == 1) pacc[u % NACC] = __builtin_elementwise_fma(pacc[u % NACC], pw, pw);
      if (MODE == 2) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_ror:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[u % NACC]) : "v"(w), "v"(w));
      if (MODE == 3) acc[u % NACC] = __builtin_amdgcn_exp2f(acc[u % NACC]);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i] + pacc[i].x + pacc[i].y;
  out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int MODE, int NACC>
int run(const char* name, float* out, long long* cyc) {
  for (int threads : {256, 512, 1024}) {
    hipLaunchKernelGGL((probe<MODE, NACC>), dim3(1), dim3(threads), 0, 0, out, cyc, 1.0f);
    CK(hipDeviceSynchronize());
    long long h[16];
    CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double mx = 0;
    for (int i = 0; i < threads / 64; ++i) mx = h[i] > mx ? h[i] : mx;
    printf("%-28s nacc=%d waves/SIMD=%d  memtime-ticks/instr = %.2f\n", name, NACC, threads / 256, mx / (64.0 * 64.0));
  }
  return 0;
}

int main() {
  float* out; long long* cyc;
  CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, 256));
  run<0, 8>("v_fma_f32 independent", out, cyc);
  run<0, 1>("v_fma_f32 dependent", out, cyc);
  run<0, 2>("v_fma_f32 2 chains", out, cyc);
  run<1, 8>("v_pk_fma_f32 independent", out, cyc);
  run<1, 1>("v_pk_fma_f32 dependent", out, cyc);
  run<1, 2>("v_pk_fma_f32 2 chains", out, cyc);
  run<2, 8>("v_fmac_dpp independent", out, cyc);
  run<2, 2>("v_fmac_dpp 2 chains", out, cyc);
  run<3, 8>("v_exp_f32 independent", out, cyc);
  run<3, 1>("v_exp_f32 dependent", out, cyc);
  int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0); printf("clockRate kHz %d\n", clk);
  int wc = 0; hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0); printf("wallClockRate kHz %d\n", wc);
  return 0;
}
