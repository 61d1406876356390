// Row-block recurrent kernels, bf16-MFMA variant (desc.dtype = VMLMF_DT_BF16): one family of instantiations (vmlmf_rb.inc).
#include "vmlmf_rb.inc"

int rb_dispatch_g1a_bf(const VGeo& g, const RbGeo& q, const RbIo& io, bool fwd, hipStream_t s) {
  const int KS = g.KH / 4;
  const bool isflat = g.flat != 0;
  RB_CASES_MT_BF(2, 1) RB_CASES_MT_BF(4, 1)
  return -3;
}
