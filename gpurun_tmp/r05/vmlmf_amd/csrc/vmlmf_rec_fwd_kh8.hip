// rec_fwd_kernel instantiations for padded hidden rank 8 (one translation unit per rank: parallel build)
#include "vmlmf_rec_fwd.inc"
int launch_rec_fwd_kh8(const VGeo& g, const FwdArgs& a, const XwArgs& xw, hipStream_t s) { return fwd_launch_kh<8>(g, a, xw, s); }
