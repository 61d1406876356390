// Exchange-skeleton probe (gfx950): what one cross-wave exchange of the recurrent kernels costs, in isolation.
// A workgroup of NWV waves runs T dependent steps; every step each wave reduces a value over its lanes, the waves'
// partials meet in LDS (one s_barrier), and every lane gets the 16 totals back and folds them into the value of the
// next step - the skeleton of rec_fwd_kernel / rec_bwd_kernel without their arithmetic.  Variants differ in the LDS
// layout, the cross-wave sum and the way the totals are delivered to the lanes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vmlmf_amd/csrc tools/microbench/xchg_probe.hip -o tools/microbench/bin/xchg_probe
#include <cstdio>
#include <cstdlib>
#include "vmlmf_device.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: [wave][rank] partials, ds_read_b128 + DPP class sums + 16 v_readlane (the shipped form)
//      1: [rank][wave] partials, ds_read_b128 + 2 adds + 16 v_readlane
//      2: as 1 without the rowsum4 in front of the write
//      3: as 1 without the s_waitcnt in front of the barrier (timing only: not a valid protocol)
//      4: as 1, totals delivered by one v_mfma_f32_32x32x2_f32 instead of the readlanes
//      5: as 1, totals consumed by a 16-step DPP rotation (fmac_ror_x16) instead of readlanes + FMAs
//      6: no exchange at all: rowsum4 + readlanes inside the wave (no LDS, no barrier)
//      7: LDS round trip + barrier only (no sums, no readlanes)
//      8: as 1 with s_setprio 3
//      9: barrier only (no LDS)   10: LDS write + read only (no barrier)   11: nothing (loop + one multiply)
template <int MODE, int NWC>
__global__ void __launch_bounds__(512) xchg(float* out, const float* win, int T) {
  __shared__ float4 part4[2][16];   // [buf][rank] x 4 waves, or [buf][wave(4)][rank/4]
  float* part = reinterpret_cast<float*>(part4);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 128) part[tid] = 0.f;
  if (wave >= NWC) {   // a mover-like wave: barrier only
    __syncthreads();
    for (int t = 0; t < T; ++t) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    return;
  }
  float w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) w[i] = win[i * 64 + lane];
  float v = win[tid];
  if (MODE == 8) asm volatile("s_setprio 3");
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const int buf = t & 1;
    float s = v;
    if (MODE != 2) s = rowsum4(v);
    float tot16 = 0.f;
    float q[16];
    if (MODE == 6 || MODE == 11) {
      tot16 = s;
    } else if (MODE == 9) {
      asm volatile("s_barrier" ::: "memory");
      tot16 = s;
    } else if (MODE == 10) {
      if (lane < 16) part[buf * 64 + lane * 4 + wave] = s;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      tot16 = ld4(part + buf * 64 + (lane & 15) * 4).x;
    } else {
      if (MODE == 0) {
        if (lane < 16) part[buf * 64 + wave * 16 + lane] = s;
      } else {
        if (lane < 16) part[buf * 64 + lane * 4 + wave] = s;
      }
      if (MODE == 3) asm volatile("s_barrier" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (MODE == 0) {
        const int li = lane & 15, cc = li & 3, ww = li >> 2;
        const float4 p = ld4(part + buf * 64 + (ww < NWC ? ww : 3) * 16 + 4 * cc);
        const float sx = class_sum16<4>(p.x), sy = class_sum16<4>(p.y), sz = class_sum16<4>(p.z), sw = class_sum16<4>(p.w);
#pragma unroll
        for (int cc2 = 0; cc2 < 4; ++cc2) {
          q[4 * cc2 + 0] = bcast_lane(sx, cc2);
          q[4 * cc2 + 1] = bcast_lane(sy, cc2);
          q[4 * cc2 + 2] = bcast_lane(sz, cc2);
          q[4 * cc2 + 3] = bcast_lane(sw, cc2);
        }
      } else {
        const float4 p = ld4(part + buf * 64 + (lane & 15) * 4);
        tot16 = MODE == 7 ? p.x : (p.x + p.y) + (p.z + p.w);
      }
    }
    if (MODE == 1 || MODE == 2 || MODE == 3 || MODE == 6 || MODE == 8) {
#pragma unroll
      for (int r = 0; r < 16; ++r) q[r] = bcast_lane(tot16, r);
    }
    if (MODE == 4) {
      // A[i][k]: lane l supplies A[l % 32][l / 32]; B[k][j]: lane l supplies B[l / 32][l % 32].  B = (k == 0 ? 1 : 0):
      // D[i][j] = A[i][0]; lane l gets rows 8 (reg / 4) + 4 (l / 32) + reg % 4 in its 16 registers.
      f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      const float bsel = lane < 32 ? 1.f : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tot16, bsel, acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) q[r] = acc[r];
    }
    if (MODE == 5) {
      float a0 = 0.f, a1 = 0.f;
      dpp_fence(tot16);
      fmac_ror_x16(a0, a1, tot16, w);
      v = a0 + a1;
    } else if (MODE == 7 || MODE >= 9) {
      v = tot16 * w[0];
    } else {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        a0 = fmaf(q[r], w[r], a0);
        a1 = fmaf(q[r + 1], w[r + 1], a1);
      }
      v = a0 + a1;
    }
  }
  out[blockIdx.x * blockDim.x + tid] = v;
}

template <int MODE, int NWC>
void run(const char* name, float* out, const float* win, int nmov) {
  const int T = 2048, iters = 10;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((xchg<MODE, NWC>), dim3(64), dim3(64 * (NWC + nmov)), 0, 0, out, win, T);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((xchg<MODE, NWC>), dim3(64), dim3(64 * (NWC + nmov)), 0, 0, out, win, T);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-58s compute waves %d movers %d: %6.1f ns per step\n", name, NWC, nmov, ms * 1e6f / iters / T);
}

int main() {
  float *out, *win;
  CK(hipMalloc(&out, 64 * 512 * 4));
  CK(hipMalloc(&win, 16 * 64 * 4 + 4096));
  float h[16 * 64 + 1024];
  for (int i = 0; i < 16 * 64 + 1024; ++i) h[i] = 0.01f * ((float)rand() / RAND_MAX - 0.5f);
  CK(hipMemcpy(win, h, sizeof(h), hipMemcpyHostToDevice));
  run<0, 3>("0 [wave][rank] + DPP class sums + 16 readlanes (shipped)", out, win, 1);
  run<0, 3>("0 same", out, win, 2);
  run<1, 3>("1 [rank][wave] + 2 adds + 16 readlanes", out, win, 1);
  run<1, 3>("1 same", out, win, 2);
  run<1, 3>("1 same", out, win, 0);
  run<2, 3>("2 as 1 without rowsum4", out, win, 1);
  run<3, 3>("3 as 1 without waitcnt before the barrier (invalid)", out, win, 1);
  run<4, 3>("4 as 1 with one 32x32x2 MFMA instead of readlanes", out, win, 1);
  run<5, 3>("5 as 1 consumed by fmac_ror_x16 (no readlanes)", out, win, 1);
  run<6, 3>("6 no exchange: rowsum4 + readlanes inside the wave", out, win, 1);
  run<7, 3>("7 LDS round trip + barrier only", out, win, 1);
  run<7, 3>("7 same", out, win, 0);
  run<7, 1>("7 same, ONE compute wave", out, win, 0);
  run<8, 3>("8 as 1 with s_setprio 3", out, win, 1);
  run<9, 3>("9 rowsum4 + barrier only", out, win, 1);
  run<10, 3>("10 rowsum4 + LDS write/read only", out, win, 1);
  run<11, 3>("11 rowsum4 + multiply only", out, win, 1);
  return 0;
}
