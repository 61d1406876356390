// Wavefront kernels for stacked layers: parameter images with the half-pass rotation layout, and the dispatch over
// the per-rank translation units (vmlmf_wave_k*.hip).
#include "vmlmf_launch.h"

// Rotated images for the DPP rank reduce of vmlmf_wave.inc.  Register j of the thread in `slot`:
//   j <  16 NPF : full pass p = j / 16, rotation kk = j % 16: the lane multiplies what it receives from lane src = i + sgn kk
//                 (mod 16) of its row with the weight that couples that unit to rank 16 p + i
//   j >= 16 NPF : half pass, rotation kk = j - 16 NPF (0..7), rank 16 NPF + (i mod 8)
// Four images: UR (U_h), VR[4] (V_h per gate), URX (U_x, inputs by slot), VRX[4] (V_x per gate).
__global__ void __launch_bounds__(256) wf_pack_kernel(VGeo g, RefP p, WfPack W, int K, float* __restrict__ out) {
  const int NT = g.NT, NPF = K / 16;
  const int lane = threadIdx.x & 63;
  const int got = __builtin_amdgcn_update_dpp(0, lane, 0x121, 0xf, 0xf, true);  // row_ror:1 on lane ids
  const int sgn = (((got - lane) & 15) == 1) ? 1 : -1;
  const int total = (int)W.total;
  for (int e = (int)blockIdx.x * 256 + threadIdx.x; e < total; e += (int)gridDim.x * 256) {
    float v = 0.f;
    int le, kind;   // kind 0: UR, 1: VR, 2: URX, 3: VRX
    if (e < W.VR) le = e - (int)W.UR, kind = 0;
    else if (e < W.URX) le = e - (int)W.VR, kind = 1;
    else if (e < W.VRX) le = e - (int)W.URX, kind = 2;
    else le = e - (int)W.VRX, kind = 3;
    const int nreg = (kind & 1) ? 4 * K : K;
    if (le < nreg * NT) {
      const int jj = le / NT, slot = le - jj * NT, k = jj / K, j = jj - k * K, i = slot & 15;
      const bool full = j < 16 * NPF;
      const int kk = full ? (j & 15) : (j - 16 * NPF);
      const int rank = full ? (j >> 4) * 16 + i : 16 * NPF + (i & 7);
      const int src = (slot & ~15) | ((i + sgn * kk) & 15);
      int n;
      if (vg_slot_unit(g, src, n)) {
        if (kind == 0) v = ref_uc(g, p, n, rank);
        else if (kind == 1) v = ref_vc(g, p, n, k, rank);
        else if (kind == 2) v = n < g.I ? ref_ux(g, p, n, rank) : 0.f;
        else v = ref_vx(g, p, n, k, rank);
      }
    }
    out[e] = v;
  }
}

WfPack wf_pack_layout(const VGeo& g) {
  WfPack W;
  long long o = 0;
  auto take = [&](long long n) { long long r = o; o += (n + 63) / 64 * 64; return r; };
  const long long K = g.KH;
  W.UR = take(K * g.NT);
  W.VR = take(4 * K * g.NT);
  W.URX = take(K * g.NT);
  W.VRX = take(4 * K * g.NT);
  W.total = o;
  return W;
}

int launch_wf_pack(const VGeo& g, const RefP& p, const WfPack& W, float* img, hipStream_t s) {
  int nb = (int)((W.total + 255) / 256);
  if (nb > 512) nb = 512;
  hipLaunchKernelGGL(wf_pack_kernel, dim3(nb), dim3(256), 0, s, g, p, W, g.KH, img);
  return (int)hipGetLastError();
}

// one group, x rank padded like the h rank, at most four waves of units; rank 32 with four waves would need more than the
// 168 VGPRs a 10-wave workgroup leaves per lane
bool wf_supported(const VGeo& g) {
  if (g.generic || g.rb || g.bf || g.flat || g.G != 1 || g.R != 1) return false;
  if (g.KX != g.KH || g.I > g.H || g.NW > 4) return false;
  if (g.KH != 8 && g.KH != 16 && g.KH != 24 && g.KH != 32) return false;
  if (g.KH == 32 && g.NW == 4) return false;
  return true;
}

#define DECL(k)                                                              \
  int launch_wf_fwd_k##k(const VGeo& g, const WfFwdArgs& a, hipStream_t s); \
  int launch_wf_bwd_k##k(const VGeo& g, const WfBwdArgs& a, hipStream_t s);
DECL(8) DECL(16) DECL(24) DECL(32)
#undef DECL

int launch_wf_fwd(const VGeo& g, const WfFwdArgs& a, hipStream_t s) {
  switch (g.KH) {
    case 8: return launch_wf_fwd_k8(g, a, s);
    case 16: return launch_wf_fwd_k16(g, a, s);
    case 24: return launch_wf_fwd_k24(g, a, s);
    case 32: return launch_wf_fwd_k32(g, a, s);
  }
  return -3;
}
int launch_wf_bwd(const VGeo& g, const WfBwdArgs& a, hipStream_t s) {
  switch (g.KH) {
    case 8: return launch_wf_bwd_k8(g, a, s);
    case 16: return launch_wf_bwd_k16(g, a, s);
    case 24: return launch_wf_bwd_k24(g, a, s);
    case 32: return launch_wf_bwd_k32(g, a, s);
  }
  return -3;
}

bool wf_bwd_built() { return true; }
