// rec3_fwd_kernel / rec3_bwd_kernel instantiations (vmlmf_rec3.inc): one-group layers, padded hidden rank <= 16, <= 3 waves of units
#include "vmlmf_rec3.inc"

bool rec3_fwd_supported(const VGeo& g) { return rec3_fwd_ok(g); }
bool rec3_bwd_supported(const VGeo& g) { return rec3_bwd_ok(g); }

int launch_rec3_fwd(const VGeo& g, const FwdArgs& a, const XwArgs& xw, hipStream_t s) {
  if (!rec3_fwd_ok(g)) return -3;
  return g.KH == 8 ? rec3_fwd_launch_kh<8>(g, a, xw, s) : rec3_fwd_launch_kh<16>(g, a, xw, s);
}

int launch_rec3_bwd(const VGeo& g, const BwdArgs& a, hipStream_t s) {
  if (!rec3_bwd_ok(g)) return -3;
  return g.KH == 8 ? rec3_bwd_launch_kh<8>(g, a, s) : rec3_bwd_launch_kh<16>(g, a, s);
}
