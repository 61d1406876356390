// wavefront kernel instantiations for padded rank 32 (one translation unit per rank: parallel build)
#include "vmlmf_wave.inc"
int launch_wf_fwd_k32(const VGeo& g, const WfFwdArgs& a, hipStream_t s) { return wf_fwd_dispatch<32>(g, a, s); }
int launch_wf_bwd_k32(const VGeo& g, const WfBwdArgs& a, hipStream_t s) { return wf_bwd_dispatch<32>(g, a, s); }
