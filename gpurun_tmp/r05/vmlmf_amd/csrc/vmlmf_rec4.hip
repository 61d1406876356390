// rec4_bwd_kernel instantiations (vmlmf_rec4.inc): the backward recurrence with the weight gradients formed in the row's workgroup
#include "vmlmf_rec4.inc"

bool rec4_bwd_supported(const VGeo& g) { return rec4_bwd_ok(g); }

int launch_rec4_bwd(const VGeo& g, const BwdArgs& a, hipStream_t s) {
  if (!rec4_bwd_ok(g)) return -3;
  return g.KH == 8 ? rec4_bwd_launch_kh<8>(g, a, s) : rec4_bwd_launch_kh<16>(g, a, s);
}
