// rec4_bwd_kernel instantiations (vmlmf_rec4.inc): the backward recurrence with the weight gradients formed in the row's workgroup
#include "vmlmf_rec4.inc"
#include <stdlib.h>

bool rec4_bwd_supported(const VGeo& g) { return rec4_bwd_ok(g); }
int rec4_bwd_rows(const VGeo& g, int cus) { return rec4_rows_per_wg(g, cus); }

int launch_rec4_bwd(const VGeo& g, const BwdArgs& a, int rows, hipStream_t s) {
  if (!rec4_bwd_ok(g) || (rows != 1 && rows != 2)) return -3;
  if (rows == 2) return g.KH == 8 ? rec4_bwd_launch_kh<8, 2, 0>(g, a, s) : rec4_bwd_launch_kh<16, 2, 0>(g, a, s);
#ifdef VMLMF_EXPERIMENTS
  static const int abl = []() { const char* e = getenv("VMLMF_R4_ABL"); return e ? atoi(e) : 0; }();
  if (g.KH == 16) switch (abl) {
    case 1: return rec4_bwd_launch_kh<16, 1, 1>(g, a, s);
    case 2: return rec4_bwd_launch_kh<16, 1, 2>(g, a, s);
    case 3: return rec4_bwd_launch_kh<16, 1, 3>(g, a, s);
    case 4: return rec4_bwd_launch_kh<16, 1, 4>(g, a, s);
    case 5: return rec4_bwd_launch_kh<16, 1, 5>(g, a, s);
    case 8: return rec4_bwd_launch_kh<16, 1, 8>(g, a, s);
    case 9: return rec4_bwd_launch_kh<16, 1, 9>(g, a, s);
    case 12: return rec4_bwd_launch_kh<16, 1, 12>(g, a, s);
    case 13: return rec4_bwd_launch_kh<16, 1, 13>(g, a, s);
    case 34: return rec4_bwd_launch_kh<16, 1, 34>(g, a, s);
    case 64: return rec4_bwd_launch_kh<16, 1, 64>(g, a, s);
    case 96: return rec4_bwd_launch_kh<16, 1, 96>(g, a, s);
  }
#endif
  return g.KH == 8 ? rec4_bwd_launch_kh<8>(g, a, s) : rec4_bwd_launch_kh<16>(g, a, s);
}
