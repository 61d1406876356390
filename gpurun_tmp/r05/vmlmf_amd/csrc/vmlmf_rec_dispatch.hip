// Rank-width dispatch for the persistent recurrent kernels (instantiations live in vmlmf_rec_*_kh*.hip).
#include "vmlmf_launch.h"

#define DECL(kh)                                                                  \
  int launch_rec_fwd_kh##kh(const VGeo& g, const FwdArgs& a, const XwArgs& xw, hipStream_t s);      \
  int launch_rec_bwd_kh##kh(const VGeo& g, const BwdArgs& a, hipStream_t s);
DECL(8) DECL(16) DECL(24) DECL(32)
#undef DECL

int launch_rec_fwd(const VGeo& g, const FwdArgs& a, const XwArgs& xw, hipStream_t s) {
  switch (g.KH) {
    case 8: return launch_rec_fwd_kh8(g, a, xw, s);
    case 16: return launch_rec_fwd_kh16(g, a, xw, s);
    case 24: return launch_rec_fwd_kh24(g, a, xw, s);
    case 32: return launch_rec_fwd_kh32(g, a, xw, s);
  }
  return -3;
}

int launch_rec_bwd(const VGeo& g, const BwdArgs& a, hipStream_t s) {
  switch (g.KH) {
    case 8: return launch_rec_bwd_kh8(g, a, s);
    case 16: return launch_rec_bwd_kh16(g, a, s);
    case 24: return launch_rec_bwd_kh24(g, a, s);
    case 32: return launch_rec_bwd_kh32(g, a, s);
  }
  return -3;
}
