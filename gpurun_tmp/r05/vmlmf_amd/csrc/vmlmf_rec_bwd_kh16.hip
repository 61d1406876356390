// rec_bwd_kernel instantiations for padded hidden rank 16
#include "vmlmf_rec_bwd.inc"
int launch_rec_bwd_kh16(const VGeo& g, const BwdArgs& a, hipStream_t s) { return bwd_launch_kh<16>(g, a, s); }
