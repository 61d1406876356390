// The weight-gradient products of a wavefront stack with four interleaved column tiles per wave (atb4_task, vmlmf_atb.inc): a
// translation unit of its own - the fully unrolled row-pair batches take hipcc minutes.
#include "vmlmf_launch.h"
#include <stdlib.h>
#include <string.h>
#include "vmlmf_atb.inc"

// 256 threads, one task per wave, no LDS; grid (tasks / 4, chunks, layers)
template <int NBT1, int ABL = 0>
__global__ void __launch_bounds__(256) wgrad4_stack_kernel(AtbStack S) {
  const VGeo& g = vg_karg_ref<VGeo>(offsetof(AtbStack, g) + (size_t)blockIdx.z * sizeof(VGeo));
  const AtbArgs a = atb_stack_args(S);
  const int lane = threadIdx.x & 63;
  const int task = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int MT1 = g.NT / 32, MT2 = (g.H + 127) / 128, MT3 = (g.I + 127) / 128;
  if (task < MT1) atb4_task<1, NBT1, ABL>(g, a, task, blockIdx.y, lane);
  else if (task < MT1 + MT2) atb4_task<2, 1, ABL>(g, a, task - MT1, blockIdx.y, lane);
  else if (task < MT1 + MT2 + MT3) atb4_task<3, 1, ABL>(g, a, task - MT1 - MT2, blockIdx.y, lane);
}
// VMLMF_WGRAD4=0: the stacks' weight gradients through wgrad_mfma_stack_kernel always (A/B)
static bool wgrad4_enabled() { const char* e = getenv("VMLMF_WGRAD4"); return e == nullptr || atoi(e) != 0; }
static const bool g_wgrad4 = wgrad4_enabled();
static bool wgrad4_geo_ok(const VGeo& g) {
  return g.G == 1 && !g.flat && !g.bf && !g.foldx && !g.generic && g.KH <= 32 && g.KX <= 32 && g.I <= g.NT && (g.H & 3) == 0 &&
         (g.B & 1) == 0 &&   // (a row pair never straddles two time steps)
         (g.syT & 3) == 0 && (g.syB & 3) == 0 && (g.PCH & 3) == 0;
}
static bool wgrad4_ok(const VGeo& g, const WghArgs& w) {
  const auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  return wgrad4_geo_ok(g) && (g.RC2 & 1) == 0 && al(w.y) && al(w.h0) && al(w.dpre) && al(w.wpart);
}
static int wgrad4_tasks(const VGeo& g) { return g.NT / 32 + (g.H + 127) / 128 + (g.I + 127) / 128; }
// Rows per chunk for a stack this kernel will take (0: not its stack): the kernel holds ONE workgroup per CU, so the rows are cut into
// as many chunks as give every CU at most one workgroup - one round - and never more chunks than the geometry already has.
int wgrad4_chunk_rows(int L, const VGeo* g, int cus) {
  if (!g_wgrad4) return 0;
  int tasks = 0;
  for (int l = 0; l < L; ++l) {
    if (!wgrad4_geo_ok(g[l])) return 0;
    const int t = wgrad4_tasks(g[l]);
    tasks = t > tasks ? t : tasks;
  }
  const int per = (tasks + 3) / 4;
  int chunks = cus / (L * per);
  if (chunks < 1) chunks = 1;
  const int TB = g[0].T * g[0].B;
  int rc2 = (TB + chunks - 1) / chunks;
  rc2 = (rc2 + 1) & ~1;
  return rc2 > g[0].RC2 ? rc2 : ((g[0].RC2 + 1) & ~1);
}


int launch_wgrad4_stack(int L, const VGeo* g, const WghArgs* w, const AtbStack& S, hipStream_t s) {
  bool four = g_wgrad4;
  int tasks4 = 0;
  const int n1 = (vg_nb1(g[0]) + 31) / 32;
  for (int l = 0; l < L; ++l) {
    four = four && wgrad4_ok(g[l], w[l]);
    const int t = wgrad4_tasks(g[l]);
    tasks4 = t > tasks4 ? t : tasks4;
  }
  if (!four) return -3;
  {
    const dim3 grid4((tasks4 + 3) / 4, g[0].nchunk, L);
    if (n1 == 1) hipLaunchKernelGGL((wgrad4_stack_kernel<1>), grid4, dim3(256), 0, s, S);
    else hipLaunchKernelGGL((wgrad4_stack_kernel<2>), grid4, dim3(256), 0, s, S);
    return (int)hipGetLastError();
  }
}
