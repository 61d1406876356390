// rec_bwd_kernel instantiations for padded hidden rank 24
#include "vmlmf_rec_bwd.inc"
int launch_rec_bwd_kh24(const VGeo& g, const BwdArgs& a, hipStream_t s) { return bwd_launch_kh<24>(g, a, s); }
