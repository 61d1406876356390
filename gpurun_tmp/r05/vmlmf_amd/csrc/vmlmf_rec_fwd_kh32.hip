// rec_fwd_kernel instantiations for padded hidden rank 32 (one translation unit per rank: parallel build)
#include "vmlmf_rec_fwd.inc"
int launch_rec_fwd_kh32(const VGeo& g, const FwdArgs& a, const XwArgs& xw, hipStream_t s) { return fwd_launch_kh<32>(g, a, xw, s); }
