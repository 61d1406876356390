// Row-block recurrent kernels, bf16-MFMA variant (desc.dtype = VMLMF_DT_BF16): one family of instantiations (vmlmf_rb.inc).
#include "vmlmf_rb.inc"

int rb_dispatch_g1b_bf(const VGeo& g, const RbGeo& q, const RbIo& io, bool fwd, hipStream_t s) {
  const int KS = g.KH / 4;
  const bool isflat = g.flat != 0;
  RB_CASES_MT_BF(6, 2) RB_CASES_MT_BF(8, 2)
  return -3;
}
