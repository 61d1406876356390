// wavefront kernel instantiations for padded rank 8 (one translation unit per rank: parallel build)
#include "vmlmf_wave.inc"
int launch_wf_fwd_k8(const VGeo& g, const WfFwdArgs& a, hipStream_t s) { return wf_fwd_dispatch<8>(g, a, s); }
int launch_wf_bwd_k8(const VGeo& g, const WfBwdArgs& a, hipStream_t s) { return wf_bwd_dispatch<8>(g, a, s); }
