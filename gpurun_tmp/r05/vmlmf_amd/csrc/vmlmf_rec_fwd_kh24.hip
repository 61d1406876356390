// rec_fwd_kernel instantiations for padded hidden rank 24 (one translation unit per rank: parallel build)
#include "vmlmf_rec_fwd.inc"
int launch_rec_fwd_kh24(const VGeo& g, const FwdArgs& a, const XwArgs& xw, hipStream_t s) { return fwd_launch_kh<24>(g, a, xw, s); }
