// Wavefront kernels for stacked layers: layout of the rotated parameter images (produced by pack_stack_kernel, vmlmf_pack.hip),
// eligibility, and the dispatch over the per-rank translation units (vmlmf_wave_k*.hip).
#include "vmlmf_launch.h"

WfPack wf_pack_layout(const VGeo& g) {
  WfPack W;
  long long o = 0;
  auto take = [&](long long n) { long long r = o; o += (n + 63) / 64 * 64; return r; };
  const long long K = wf_width(g);
  const long long mixed = g.KH != g.KX ? 1 : 0;
  W.UR = take(K * g.NT);
  W.VR = take(4 * K * g.NT);
  W.URX = take(K * g.NT);
  W.VRX = take(4 * K * g.NT);
  W.VE = take(mixed * 4 * K * g.NT);
  W.UE = take(mixed * K * g.NT);
  W.VXK = take(mixed * 4 * K * g.H);
  W.UXK = take(mixed * K * g.NT);
  W.total = o;
  return W;
}

// one or two (wave-aligned) groups, at most four waves of units; rank 32 with four waves would need more than the
// 168 VGPRs a 10-wave workgroup leaves per lane
bool wf_supported(const VGeo& g) {
  if (g.generic || g.rb || g.bf || g.flat || g.G > 2 || g.R != 1) return false;   // (V4's flat layout: not covered)
  if (g.I > g.H || g.NW > 4) return false;
  const int K = wf_width(g);   // padded w_rank != padded u_rank: both sides run at the wider one (zero ranks behind the narrower)
  if (K != 8 && K != 16 && K != 24 && K != 32) return false;
  if (K == 32 && g.NW == 4) return false;
  return true;
}

#define DECL(k)                                                              \
  int launch_wf_fwd_k##k(const VGeo& g, const WfFwdArgs& a, hipStream_t s); \
  int launch_wf_bwd_k##k(const VGeo& g, const WfBwdArgs& a, hipStream_t s);
DECL(8) DECL(16) DECL(24) DECL(32)
#undef DECL

int launch_wf_fwd(const VGeo& g, const WfFwdArgs& a, hipStream_t s) {
  switch (wf_width(g)) {
    case 8: return launch_wf_fwd_k8(g, a, s);
    case 16: return launch_wf_fwd_k16(g, a, s);
    case 24: return launch_wf_fwd_k24(g, a, s);
    case 32: return launch_wf_fwd_k32(g, a, s);
  }
  return -3;
}
int launch_wf_bwd(const VGeo& g, const WfBwdArgs& a, hipStream_t s) {
  switch (wf_width(g)) {
    case 8: return launch_wf_bwd_k8(g, a, s);
    case 16: return launch_wf_bwd_k16(g, a, s);
    case 24: return launch_wf_bwd_k24(g, a, s);
    case 32: return launch_wf_bwd_k32(g, a, s);
  }
  return -3;
}
