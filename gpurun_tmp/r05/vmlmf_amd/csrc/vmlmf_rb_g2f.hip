// Row-block recurrent kernels: one family of instantiations (vmlmf_rb.inc; dispatch in vmlmf_rb.hip).
#include "vmlmf_rb.inc"

int rb_dispatch_g2f(const VGeo& g, const RbGeo& q, const RbIo& io, bool fwd, hipStream_t s) {
  const int KS = g.KH / 4;
  const bool isflat = g.flat != 0;
  RB_CASES_MT(4, 2, true, 2) RB_CASES_MT(8, 2, true, 2)
  return -3;
}
