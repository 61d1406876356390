// Parameter-side and fully-parallel kernels of the VMLMF hot path (gfx950):
//   pack_kernel    reference-layout parameters -> per-thread register images (+ hoisted ex/eh vectors)
//   xproj_kernel   gx[t,b,n,k] = (x_t U_x) V_x^T + x .* ex + (b_x + b_h)   for all (t,b) at once
//   reduce_kernel  deterministic sum of the wgrad workgroups' partial accumulators
//   finish_kernel  canonical gradients -> reference-layout gradients (folds d(ex), d(eh) into dia/U/V)
#include "vmlmf_launch.h"
#include <stdlib.h>
#include <string.h>

// ---------------------------------------------------------------------------------------------------
// pack
// ---------------------------------------------------------------------------------------------------
// Image element (j, slot) is register j of the thread in slot `slot`.  Rotated images serve the DPP
// reductions: in pass p, step kk, lane i of a 16-lane row multiplies the value it RECEIVES from lane
// src = i + sgn*kk (mod 16) with the weight that couples that source unit to rank p*16 + i.
//
// Two kinds of workgroup share the launch.  "Copy" workgroups ([0, ncopy)) produce the elements that are one
// (re-indexed) parameter each.  "Dot" workgroups produce the elements that are a rank-long dot product (the hoisted
// diagonal-removal vectors ex / eh and the dense x-side matrix W_x): 32 lanes per element, one rank per lane, a
// shuffle reduction -- one memory latency per element and a few hundred bytes of code.  (Written as one thread per
// element with the 32 loads unrolled, this kernel was 107 KB of code, more than the instruction cache holds.)
struct PackDots {
  int nEH, nEXI, nEXT, nWXD;   // element counts of the four dot regions (0 when absent)
};

__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// The rank-space dot products whose results the recurrent kernels can also form themselves (direct mode, vmlmf_direct.inc: eh and
// the dense x-side matrix) are ONE arithmetic everywhere: the products in rank order through a chain of fused multiply-adds,
// (dr_chain: the even and the odd ranks through a chain each, added at the end).
template <class FU, class FV>
__device__ __forceinline__ float chain_dot(const int nr, FU&& u, FV&& v) {
  // lane r of the 32-lane group loads rank r (one memory latency per 32 ranks), then every lane walks the
  // chain over the group's registers (v_readlane: both groups of the wave at once, each keeps its own)
  const int l32 = threadIdx.x & 31;
  const bool hi = (threadIdx.x & 32) != 0;
  float acc = 0.f, acc1 = 0.f;   // the chains of the even and of the odd ranks (dr_chain)
  for (int r0 = 0; r0 < nr; r0 += 32) {   // (ranks beyond 32: the step-wise layers; another block of loads)
    const int rc = r0 + l32 < nr ? r0 + l32 : nr - 1;   // clamped: unconditional loads
    const float ul = u(rc), vl = v(rc);
    const int n32 = nr - r0 < 32 ? nr - r0 : 32;
    for (int r = 0; r < n32; ++r) {
      const float u0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ul), r)), u1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ul), 32 + r));
      const float v0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vl), r)), v1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vl), 32 + r));
      const float uu = hi ? u1 : u0, vv = hi ? v1 : v0;
      if (r & 1) acc1 = (r0 + r == 1) ? uu * vv : fmaf(uu, vv, acc1);
      else acc = (r0 + r == 0) ? uu * vv : fmaf(uu, vv, acc);
    }
  }
  acc += acc1;
  return acc;
}

// ex(n,k) = dia_x[n] - sum_r ux(n,r) vx(n,k,r), one rank per lane of a 32-lane group (vmlmf.py:102-106 hoisted)
__device__ __forceinline__ float dot_ex(const VGeo& g, const RefP& p, int n, int k, int l32) {
  float acc = 0.f;
  for (int r = l32; r < g.rw; r += 32) acc = fmaf(ref_ux(g, p, n, r), ref_vx(g, p, n, k, r), acc);
  return p.dia_x[n] - half_wave_sum(acc);
}

// PackRanges: the copy elements of up to three ranges [lo, hi) only (n == 0: all of them, up to L.RB).  The clustered stacks read of
// this kernel's images nothing but EH / EXT (dot elements) and BBT; the wavefront launches VE, UE, UXO, VXT, BBT (and the dot elements)
// beside their own rotated images - the rest (two thirds of the elements: the step-wise path's dense matrices, the rotated images of the
// per-layer kernels) is not produced for them.
struct PackRanges {
  int n, lo[3], hi[3], pad;
};
__device__ __forceinline__ void pack_body(const VGeo& g, const RefP& p, const VPack& L, const PackDots& D, const int ncopy,
                                          float* __restrict__ out, const int bid, const PackRanges& R = PackRanges{0, {0, 0, 0}, {0, 0, 0}, 0}) {
  const int NT = g.NT;
  if (bid >= ncopy) {   // ---- dot elements: 8 per workgroup ----
    const int l32 = threadIdx.x & 31;
    int d = (bid - ncopy) * 8 + (threadIdx.x >> 5);
    float v = 0.f;
    long long dst = -1;
    if (d < D.nEH) {            // EH[k][slot] = dia_h[n] - sum_r uc(n,r) vc(n,k,r)   (block 0 of the rank space)
      const int k = d / NT, slot = d - k * NT;
      int n;
      dst = L.EH + d;
      if (vg_slot_unit(g, slot, n) && !g.novm)
        v = p.dia_h[n] - chain_dot(g.ru0, [&](int r) { return ref_uc(g, p, n, r); }, [&](int r) { return ref_vc(g, p, n, k, r); });
    } else if ((d -= D.nEH) < D.nEXI) {   // EXI[k][slot]
      const int k = d / NT, slot = d - k * NT;
      int n;
      dst = L.EXI + d;
      if (vg_slot_unit(g, slot, n) && n < g.I && !g.novm) v = dot_ex(g, p, n, k, l32);
    } else if ((d -= D.nEXI) < D.nEXT) {  // EXT[k][n]
      const int k = d / g.H, n = d - k * g.H;
      dst = L.EXT + d;
      if (n < g.I && !g.novm) v = dot_ex(g, p, n, k, l32);
    } else if ((d -= D.nEXT) < D.nWXD) {  // WXD[(m*4+k)][slot] = W_x: x[m] -> pre-activation k of the unit in `slot`
      const int j = d / NT, slot = d - j * NT, m = j >> 2, kk = j & 3;
      int nn;
      dst = L.WXD + d;
      if (vg_slot_unit(g, slot, nn)) {
        if (nn == m && !g.novm) {
          v = p.dia_x[nn];      // diag(d_x) + (U V^T with its diagonal removed): the diagonal is d_x itself
        } else {
          v = chain_dot(g.rw, [&](int r) { return ref_ux(g, p, m, r); }, [&](int r) { return ref_vx(g, p, nn, kk, r); });
        }
      }
    }
    if (dst >= 0 && l32 == 0) out[dst] = v;
    return;
  }
  // ---- copy elements ----
  const int lane = threadIdx.x & 63;
  const int got = __builtin_amdgcn_update_dpp(0, lane, 0x121, 0xf, 0xf, true);  // row_ror:1 on lane ids
  const int sgn = (((got - lane) & 15) == 1) ? 1 : -1;
  const int stride = ncopy * 256;
  const int n0 = R.n > 0 ? R.hi[0] - R.lo[0] : (int)L.RB;   // (the RB region behind L.RB belongs to rb_pack_kernel)
  const int n1 = R.n > 1 ? n0 + R.hi[1] - R.lo[1] : n0, total = R.n > 2 ? n1 + R.hi[2] - R.lo[2] : n1;
  for (int ve = bid * 256 + threadIdx.x; ve < total; ve += stride) {
    const int e = R.n == 0 ? ve : (ve < n0 ? R.lo[0] + ve : (ve < n1 ? R.lo[1] + (ve - n0) : R.lo[2] + (ve - n1)));
    float v = 0.f;
    if (e < L.UR) {  // VE[(k*KH+rr)][slot] = vc(n,k,rr)
      const int le = e - (int)L.VE;
      if (le < 4 * g.KH * NT) {
        const int j = le / NT, slot = le - j * NT, k = j / g.KH, rr = j - k * g.KH;
        int n;
        if (vg_slot_unit(g, slot, n)) v = ref_vc(g, p, n, k, rr);
      }
    } else if (e < L.VR) {  // UR[(p*16+kk)][slot] = uc(n(src), p*16+i)
      const int le = e - (int)L.UR;
      if (le < g.KQ * NT) {
        const int j = le / NT, slot = le - j * NT, pp = j >> 4, kk = j & 15, i = slot & 15;
        const int src = (slot & ~15) | ((i + sgn * kk) & 15);
        int n;
        if (vg_slot_unit(g, src, n) && pp * 16 + i < g.KH) v = ref_uc(g, p, n, pp * 16 + i);
      }
    } else if (e < L.UE) {  // VR[(k*KQ + p*16+kk)][slot] = vc(n(src), k, p*16+i)
      const int le = e - (int)L.VR;
      if (le < 4 * g.KQ * NT) {
        const int j = le / NT, slot = le - j * NT, k = j / g.KQ, jj = j - k * g.KQ;
        const int pp = jj >> 4, kk = jj & 15, i = slot & 15;
        const int src = (slot & ~15) | ((i + sgn * kk) & 15);
        int n;
        if (vg_slot_unit(g, src, n) && pp * 16 + i < g.KH) v = ref_vc(g, p, n, k, pp * 16 + i);
      }
    } else if (e < L.EH) {  // UE[rr][slot] = uc(n, rr)
      const int le = e - (int)L.UE;
      if (le < g.KH * NT) {
        const int rr = le / NT, slot = le - rr * NT;
        int n;
        if (vg_slot_unit(g, slot, n)) v = ref_uc(g, p, n, rr);
      }
    } else if (e < L.VRX) {  // EH: dot workgroups; the alignment gap behind it is zeroed here
      if (e - (int)L.EH < D.nEH) continue;
    } else if (e < L.UXO) {  // VRX[(k*KQX + p*16+kk)][slot] = vx(n(src), k, p*16+i)
      const int le = e - (int)L.VRX;
      if (le < 4 * g.KQX * NT) {
        const int j = le / NT, slot = le - j * NT, k = j / g.KQX, jj = j - k * g.KQX;
        const int pp = jj >> 4, kk = jj & 15, i = slot & 15;
        const int src = (slot & ~15) | ((i + sgn * kk) & 15);
        int n;
        if (vg_slot_unit(g, src, n) && pp * 16 + i < g.KX) v = ref_vx(g, p, n, k, pp * 16 + i);
      }
    } else if (e < L.EXI) {  // UXO[r][slot] = ux(n, r) for x-units
      const int le = e - (int)L.UXO;
      if (le < g.KX * NT) {
        const int r = le / NT, slot = le - r * NT;
        int n;
        if (vg_slot_unit(g, slot, n) && n < g.I) v = ref_ux(g, p, n, r);
      }
    } else if (e < L.UXP) {  // EXI: dot workgroups
      if (e - (int)L.EXI < D.nEXI) continue;
    } else if (e < L.VXT) {  // UXP[m][r]  row-major, rank padded
      const int le = e - (int)L.UXP;
      if (le < g.I * g.KX) {
        const int m = le / g.KX;
        v = ref_ux(g, p, m, le - m * g.KX);
      }
    } else if (e < L.EXT) {  // VXT[(k*KX+r)][n]
      const int le = e - (int)L.VXT;
      if (le < 4 * g.KX * g.H) {
        const int j = le / g.H, n = le - j * g.H, k = j / g.KX;
        v = ref_vx(g, p, n, k, j - k * g.KX);
      }
    } else if (e < L.BBT) {  // EXT: dot workgroups
      if (e - (int)L.EXT < D.nEXT) continue;
    } else if (e < L.UD) {  // BBT[k][n]
      const int le = e - (int)L.BBT;
      if (le < 4 * g.H) {
        const int k = le / g.H;
        v = ref_bb(g, p, le - k * g.H, k);
      }
    } else {  // step-wise path: group structure written out densely
      const int GK = g.G * g.KH, N4 = 4 * NT;
      int n = -1, k = 0, rr = 0, j = 0, mode = 0;   // mode 1: U element (n, j, rr)   2: V element (n, k, j, rr)   3: Vx
      if (e < L.VD) {           // UD[n][j*KH+rr]
        const int le = e - (int)L.UD;
        if (le < g.H * GK) { n = le / GK; const int c = le - n * GK; j = c / g.KH; rr = c - j * g.KH; mode = 1; }
      } else if (e < L.UDT) {   // VD[j*KH+rr][slot*4+k]
        const int le = e - (int)L.VD;
        if (le < GK * N4) {
          const int c = le / N4, i = le - c * N4;
          j = c / g.KH; rr = c - j * g.KH; k = i & 3;
          if (vg_slot_unit(g, i >> 2, n)) mode = 2; else n = -1;
        }
      } else if (e < L.VDT) {   // UDT[j*KH+rr][n]
        const int le = e - (int)L.UDT;
        if (le < GK * g.H) { const int c = le / g.H; n = le - c * g.H; j = c / g.KH; rr = c - j * g.KH; mode = 1; }
      } else if (e < L.VXTT) {  // VDT[slot*4+k][j*KH+rr]
        const int le = e - (int)L.VDT;
        if (le < N4 * GK) {
          const int i = le / GK, c = le - i * GK;
          j = c / g.KH; rr = c - j * g.KH; k = i & 3;
          if (vg_slot_unit(g, i >> 2, n)) mode = 2; else n = -1;
        }
      } else if (e < L.TKT) {   // VXTT[slot][r][k]: V_x^T with the four gates of a slot side by side - the B operand of dqx = dpre VxT
                                //  as ONE 16-byte load per lane and four contraction steps (gemm_skinny_kernel, BMODE 2)
        const int le = e - (int)L.VXTT;
        if (le < N4 * g.KX) {
          const int sl = le / (4 * g.KX), rem = le - sl * 4 * g.KX;
          rr = rem >> 2; k = rem & 3;
          if (vg_slot_unit(g, sl, n)) mode = 3; else n = -1;
        }
      } else if (e >= L.WXD) {  // WXD: dot workgroups
        if (e - (int)L.WXD < D.nWXD) continue;
      } else if (e >= L.VXD) {  // VXD[r][slot*4+k]
        const int le = e - (int)L.VXD;
        if (le < g.KX * N4) {
          rr = le / N4;
          const int i = le - rr * N4;
          k = i & 3;
          if (vg_slot_unit(g, i >> 2, n)) mode = 3; else n = -1;
        }
      }                         // else TKT: the split-K tickets start at zero
      if (mode == 1) {          // unit n feeds destination (grp - s) mod G through block s
        const int sblk = (g.G == 2 && rr >= g.off1) ? 1 : 0;
        const int dest = (n / g.Hg - sblk + g.G) % g.G;
        if (dest == j) v = ref_uc(g, p, n, rr);
      } else if (mode == 2) {   // gate k of unit n reads Q[qsel]
        const int qsel = g.G == 1 ? 0 : (g.flat ? (k * g.H + n) / (4 * g.Hg) : n / g.Hg);
        if (qsel == j) v = ref_vc(g, p, n, k, rr);
      } else if (mode == 3) {
        v = ref_vx(g, p, n, k, rr);
      }
    }
    out[e] = v;
  }
}

__global__ void __launch_bounds__(256) pack_kernel(VGeo g, RefP p, VPack L, PackDots D, int ncopy, float* __restrict__ out) {
  pack_body(g, p, L, D, ncopy, out, (int)blockIdx.x);
}

static void pack_counts(const VGeo& g, const VPack& L, PackDots& D, int& ncopy, int& ndot) {
  D.nEH = 4 * g.NT;
  D.nEXI = (int)(L.UXP - L.EXI) > 0 ? 4 * g.NT : 0;
  D.nEXT = 4 * g.H;
  D.nWXD = (L.RB - L.WXD) > 0 ? 4 * g.I * g.NT : 0;
  ncopy = (int)((L.RB + 255) / 256);
  if (ncopy > 2048) ncopy = 2048;
  ndot = (D.nEH + D.nEXI + D.nEXT + D.nWXD + 7) / 8;
}

int launch_pack(const VGeo& g, const RefP& p, const VPack& L, float* pack, hipStream_t s) {
  if (L.total >= (1LL << 31)) return -3;
  PackDots D;
  int ncopy, ndot;
  pack_counts(g, L, D, ncopy, ndot);
  hipLaunchKernelGGL(pack_kernel, dim3(ncopy + ndot), dim3(256), 0, s, g, p, L, D, ncopy, pack);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// pack of a whole stack (wavefront launches, vmlmf_wave.inc): every layer's images - pack_kernel's and the rotated
// images with the half-pass layout - in ONE launch (grid.y = layer), which also zeroes the progress words of the two
// wavefront launches (no memset nodes).  Two launches per layer plus two memsets were 28 us of config C's 188.
// ---------------------------------------------------------------------------------------------------
// Rotated images for the DPP rank reduce of vmlmf_wave.inc.  Register j of the thread in `slot`:
//   j <  16 NPF : full pass p = j / 16, rotation kk = j % 16: the lane multiplies what it receives from lane src = i + sgn kk
//                 (mod 16) of its row with the weight that couples that unit to rank 16 p + i
//   j >= 16 NPF : half pass, rotation kk = j - 16 NPF (0..7), rank 16 NPF + (i mod 8)
// Four images: UR (U_h), VR[4] (V_h per gate), URX (U_x, inputs by slot), VRX[4] (V_x per gate).
__device__ __forceinline__ void wf_pack_body(const VGeo& g, const RefP& p, const WfPack& W, float* __restrict__ out, const int bid,
                                             const int nblk) {
  const int NT = g.NT, K = wf_width(g), NPF = K / 16;
  const int lane = threadIdx.x & 63;
  const int got = __builtin_amdgcn_update_dpp(0, lane, 0x121, 0xf, 0xf, true);  // row_ror:1 on lane ids
  const int sgn = (((got - lane) & 15) == 1) ? 1 : -1;
  const int total = (int)W.total;
  for (int e = bid * 256 + threadIdx.x; e < total; e += nblk * 256) {
    float v = 0.f;
    if (e >= W.VE) {   // (mixed widths only) plain re-layouts at the common width K: VE, UE by slot; VXK by unit; UXK by slot
      if (e < W.UE) {
        const int le = e - (int)W.VE;
        if (le < 4 * K * NT) {
          const int jj = le / NT, slot = le - jj * NT, k = jj / K, rr = jj - k * K;
          int n;
          if (vg_slot_unit(g, slot, n)) v = ref_vc(g, p, n, k, rr);
        }
      } else if (e < W.VXK) {
        const int le = e - (int)W.UE;
        if (le < K * NT) {
          const int rr = le / NT, slot = le - rr * NT;
          int n;
          if (vg_slot_unit(g, slot, n)) v = ref_uc(g, p, n, rr);
        }
      } else if (e < W.UXK) {
        const int le = e - (int)W.VXK;
        if (le < 4 * K * g.H) {
          const int jj = le / g.H, n = le - jj * g.H, k = jj / K;
          v = ref_vx(g, p, n, k, jj - k * K);
        }
      } else {
        const int le = e - (int)W.UXK;
        if (le < K * NT) {
          const int r = le / NT, slot = le - r * NT;
          int n;
          if (vg_slot_unit(g, slot, n) && n < g.I) v = ref_ux(g, p, n, r);
        }
      }
      out[e] = v;
      continue;
    }
    int le, kind;   // kind 0: UR, 1: VR, 2: URX, 3: VRX
    if (e < W.VR) le = e - (int)W.UR, kind = 0;
    else if (e < W.URX) le = e - (int)W.VR, kind = 1;
    else if (e < W.VRX) le = e - (int)W.URX, kind = 2;
    else le = e - (int)W.VRX, kind = 3;
    const int nreg = (kind & 1) ? 4 * K : K;
    if (le < nreg * NT) {
      const int jj = le / NT, slot = le - jj * NT, k = jj / K, j = jj - k * K, i = slot & 15;
      const bool full = j < 16 * NPF;
      const int kk = full ? (j & 15) : (j - 16 * NPF);
      const int rank = full ? (j >> 4) * 16 + i : 16 * NPF + (i & 7);
      const int src = (slot & ~15) | ((i + sgn * kk) & 15);
      int n;
      if (vg_slot_unit(g, src, n)) {
        if (kind == 0) v = ref_uc(g, p, n, rank);
        else if (kind == 1) v = ref_vc(g, p, n, k, rank);
        else if (kind == 2) v = n < g.I ? ref_ux(g, p, n, rank) : 0.f;
        else v = ref_vx(g, p, n, k, rank);
      }
    }
    out[e] = v;
  }
}

struct PackStackLayer {
  VGeo g;
  RefP p;
  VPack L;
  WfPack W;
  PackDots D;
  int ncopy, ndot, nwf, pad;
  PackRanges R;
  float* out;
};
struct PackStackArgs {
  PackStackLayer l[WF_MAXL];
  unsigned* zero[2];   // progress words of the forward / backward wavefront launch (or NULL)
  int nzero[2];
};

__global__ void __launch_bounds__(256) pack_stack_kernel(PackStackArgs a) {
  const int layer = blockIdx.y, bid = blockIdx.x;
  const PackStackLayer& ly = vg_karg_ref<PackStackLayer>((size_t)layer * sizeof(PackStackLayer));
  if (layer == 0) {
    const int nthr = (int)gridDim.x * 256;
#pragma unroll
    for (int z = 0; z < 2; ++z)
      if (a.zero[z] != nullptr)
        for (int i = bid * 256 + (int)threadIdx.x; i < a.nzero[z]; i += nthr) a.zero[z][i] = 0u;
  }
  const int npk = ly.ncopy + ly.ndot;
  if (bid < npk) pack_body(ly.g, ly.p, ly.L, ly.D, ly.ncopy, ly.out, bid, ly.R);
  else if (bid < npk + ly.nwf) wf_pack_body(ly.g, ly.p, ly.W, ly.out + ly.L.WF, bid - npk, ly.nwf);
}

int launch_pack_stack(int L, const VGeo* g, const RefP* p, const VPack* P, const WfPack& W, float* const* pack, unsigned* zero0,
                      int nzero0, unsigned* zero1, int nzero1, hipStream_t s, int images) {
  static_assert(sizeof(PackStackArgs) <= 4096, "kernel-argument segment");
  static_assert(sizeof(PackStackLayer) % 8 == 0, "layer blocks are read as dwords at a multiple of their size");
  PackStackArgs a;
  memset(&a, 0, sizeof(a));
  int gx = 0;
  for (int l = 0; l < L; ++l) {
    if (P[l].total >= (1LL << 31)) return -3;
    PackStackLayer& y = a.l[l];
    y.g = g[l], y.p = p[l], y.L = P[l], y.W = W, y.out = pack[l];
    pack_counts(g[l], P[l], y.D, y.ncopy, y.ndot);
    const VPack& Q = P[l];
    if (images == PACK_CLUSTERED) {   // EH / EXT come from the dot workgroups; of the copy elements only BBT is read (vmlmf_rbx.hip)
      y.R.n = 1, y.R.lo[0] = (int)Q.BBT, y.R.hi[0] = (int)Q.UD;
    } else if (images == PACK_WAVEFRONT) {   // VE | UE, the gap behind EH | UXO, the gap behind EXI | VXT, the gap behind EXT, BBT
      // (UXP - I x KX elements between EXI and VXT - goes along: cheaper than a fourth select per element)
      y.R.n = 3;
      y.R.lo[0] = (int)Q.VE, y.R.hi[0] = (int)Q.UR;
      y.R.lo[1] = (int)Q.UE, y.R.hi[1] = (int)Q.VRX;
      y.R.lo[2] = (int)Q.UXO, y.R.hi[2] = (int)Q.UD;
    }
    if (y.R.n > 0) {
      int n = 0;
      for (int i = 0; i < y.R.n; ++i) n += y.R.hi[i] - y.R.lo[i];
      y.ncopy = (n + 255) / 256;
      if (y.ncopy > 2048) y.ncopy = 2048;
    }
    y.nwf = (int)((W.total + 255) / 256);
    if (y.nwf > 256) y.nwf = 256;
    const int n = y.ncopy + y.ndot + y.nwf;
    gx = n > gx ? n : gx;
  }
  a.zero[0] = zero0, a.nzero[0] = nzero0, a.zero[1] = zero1, a.nzero[1] = nzero1;
  hipLaunchKernelGGL(pack_stack_kernel, dim3(gx, L), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// xproj: the non-recurrent half of the gate pre-activations for every (t,b) row
// ---------------------------------------------------------------------------------------------------
// XR: rows per workgroup.  A thread reloads its slots' V_x rows (4 KX floats per slot) once per workgroup, so the
// L2 traffic for them is rows / XR times the matrix: 8 is best at the headline shape (8192 rows, measured: 4 -> 19.1 us,
// 8 -> 15.5, 16 -> 16.7, 32 slower still: too few workgroups), 16 for the long LM shapes (8960 rows x 768 slots).
template <int KX, int XR>
__global__ void __launch_bounds__(256) xproj_kernel(VGeo g, const float* __restrict__ x,
                                                    const float* __restrict__ uxp,
                                                    const float* __restrict__ vxt,
                                                    const float* __restrict__ ext,
                                                    const float* __restrict__ bbt, float* __restrict__ gx,
                                                    float* __restrict__ qx) {
  // Rows are enumerated in the PADDED space rp = t*Bp + b; gx is slot-padded [T][Bp][NT][4].  Pad rows
  // (b >= B) and pad slots (lanes without a hidden unit) are written as zeros: the recurrent kernels read
  // every slot unconditionally and must only ever see finite values there.
  extern __shared__ float4 smem4[];
  float* xs = reinterpret_cast<float*>(smem4);  // [XR][I]
  float* qs = xs + XR * g.I;                    // [XR][KX]
  const int tid = threadIdx.x;
  const int TBp = g.T * g.Bp, I = g.I, H = g.H, NT = g.NT;
  const int row0 = blockIdx.x * XR;
  for (int idx = tid; idx < XR * I; idx += 256) {
    const int r = idx / I, m = idx - r * I, rp = row0 + r;
    float v = 0.f;
    if (rp < TBp) {
      const int t = rp / g.Bp, b = rp - t * g.Bp;
      if (b < g.B) v = x[t * g.sxT + b * g.sxB + m];
    }
    xs[idx] = v;
  }
  __syncthreads();
  // qx = x U_x for the XR rows.  Thread (j, c) owns rank j and the c-th of NC input chunks: a U_x element is loaded
  // once per workgroup and serves all XR rows, and a thread's loads are I / NC long (8 in flight at a time) -- one
  // thread per (row, rank) walking all of I was a chain of I / 8 L2 latencies (57 us of the 138 at I = 650).
  // Chunk partials meet in LDS and are summed in chunk order.
  {
    constexpr int NC = 256 / KX;
    float* part = qs + XR * KX;   // [NC][XR][KX]
    const int j = tid % KX, c = tid / KX;
    if (c < NC) {
      const int per = (I + NC - 1) / NC, m0 = c * per, m1 = m0 + per < I ? m0 + per : I;
      float acc[XR];
#pragma unroll
      for (int r = 0; r < XR; ++r) acc[r] = 0.f;
      int m = m0;
      for (; m + 8 <= m1; m += 8) {
        float u[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] = uxp[(m + i) * KX + j];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int r = 0; r < XR; ++r) acc[r] = fmaf(xs[r * I + m + i], u[i], acc[r]);
      }
      for (; m < m1; ++m) {
        const float u = uxp[m * KX + j];
#pragma unroll
        for (int r = 0; r < XR; ++r) acc[r] = fmaf(xs[r * I + m], u, acc[r]);
      }
#pragma unroll
      for (int r = 0; r < XR; ++r) part[(c * XR + r) * KX + j] = acc[r];
    }
    __syncthreads();
    for (int idx = tid; idx < XR * KX; idx += 256) {
      const int r = idx / KX, rp = row0 + r;
      float acc = 0.f;
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) acc += part[cc * XR * KX + idx];
      qs[idx] = acc;
      if (qx != nullptr && rp < TBp) {
        const int t = rp / g.Bp, b = rp - t * g.Bp;
        if (b < g.B) qx[(size_t)(t * g.B + b) * KX + (idx - r * KX)] = acc;
      }
    }
  }
  if (gx == nullptr) return;   // qx only: the expansion runs on the matrix cores (xexp_mfma_kernel, large layers)
  __syncthreads();
  for (int slot = tid; slot < NT; slot += 256) {
    int n;
    const bool valid = vg_slot_unit(g, slot, n);
    float v[4][KX], e[4], bb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int r = 0; r < KX; ++r) v[k][r] = valid ? vxt[(size_t)(k * KX + r) * H + n] : 0.f;
      e[k] = valid ? ext[k * H + n] : 0.f;
      bb[k] = valid ? bbt[k * H + n] : 0.f;
    }
    for (int r = 0; r < XR; ++r) {
      const int rp = row0 + r;
      if (rp >= TBp) break;
      const int b = rp % g.Bp;
      float4 out = f4zero();
      if (valid && b < g.B) {
        const float xv = (n < I) ? xs[r * I + n] : 0.f;
        float pre[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) pre[k] = fmaf(xv, e[k], bb[k]);
#pragma unroll
        for (int j4 = 0; j4 < KX / 4; ++j4) {
          const float4 q = ld4(qs + r * KX + 4 * j4);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            pre[k] = fmaf(q.x, v[k][4 * j4 + 0], pre[k]);
            pre[k] = fmaf(q.y, v[k][4 * j4 + 1], pre[k]);
            pre[k] = fmaf(q.z, v[k][4 * j4 + 2], pre[k]);
            pre[k] = fmaf(q.w, v[k][4 * j4 + 3], pre[k]);
          }
        }
        out = make_float4(pre[0], pre[1], pre[2], pre[3]);
      }
      if (g.bf) {   // bf16 tape: 8 bytes per slot
        uint2 pk;
        pk.x = (unsigned)vg_f2bf(out.x) | ((unsigned)vg_f2bf(out.y) << 16);
        pk.y = (unsigned)vg_f2bf(out.z) | ((unsigned)vg_f2bf(out.w) << 16);
        reinterpret_cast<uint2*>(gx)[(size_t)rp * NT + slot] = pk;
      } else {
        st4(gx + ((size_t)rp * NT + slot) * 4, out);
      }
    }
  }
}

// The expansion of the x side for LARGE layers (step-wise / clustered families: H = 650 ...) on fp32 MFMA:
//   gx[row][slot][k] = sum_r qx[row][r] VXD[r][slot*4+k] + x[row][n] ex[k][n] + bb[k][n]
// 64 x 64 tiles of the (T*B) x (4*slots) result, four waves with a 32 x 32 sub-tile each on v_mfma_f32_32x32x2_f32, K = the
// padded x rank (<= 32) staged through LDS in one go, the tile then goes through LDS into (row, slot) elements whose four
// gates are one 16-byte store.  As VALU FMAs in xproj_kernel (one thread per slot, 4 KX weights in registers, 8 rows per
// workgroup) this expansion was 120 us per PTB layer (8960 rows x 2816 columns); it is bound by the 100 MB it writes.
template <int KX>
__global__ void __launch_bounds__(256) xexp_mfma_kernel(VGeo g, const float* __restrict__ x, const float* __restrict__ qx,
                                                        const float* __restrict__ vxd, const float* __restrict__ ext,
                                                        const float* __restrict__ bbt, float* __restrict__ gx) {
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  constexpr int PADM = 68;
  __shared__ float As[KX][PADM];   // [k][row]
  __shared__ float Bs[KX][PADM];   // [k][column]
  __shared__ float Ct[64][PADM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lk = lane >> 5, wm = wave & 1, wn = wave >> 1;
  const int NT = g.NT, N = 4 * NT, TB = g.T * g.B, B = g.B, H = g.H;
  const int tiles_n = N / 64;   // NT is a multiple of 64
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int m0 = tm * 64, n0 = tn * 64;
  {
    const int ar = tid >> 2, aj = tid & 3;
    const bool rok = m0 + ar < TB;
    const float* ap = qx + (size_t)(rok ? m0 + ar : 0) * KX;
    float ra[KX / 4];
#pragma unroll
    for (int i = 0; i < KX / 4; ++i) ra[i] = ap[aj + 4 * i];   // unconditional loads, masked values
    const int bk = tid >> 4, bn = (tid & 15) * 4;
    float4 rb[(KX + 15) / 16];
#pragma unroll
    for (int i = 0; i < (KX + 15) / 16; ++i) {
      const int k = bk + 16 * i;
      rb[i] = ld4(vxd + (size_t)(k < KX ? k : 0) * N + n0 + bn);
    }
#pragma unroll
    for (int i = 0; i < KX / 4; ++i) As[aj + 4 * i][ar] = rok ? ra[i] : 0.f;
#pragma unroll
    for (int i = 0; i < (KX + 15) / 16; ++i)
      if (bk + 16 * i < KX) *reinterpret_cast<float4*>(&Bs[bk + 16 * i][bn]) = rb[i];
  }
  __syncthreads();
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int s2 = 0; s2 < KX / 2; ++s2)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[2 * s2 + lk][32 * wm + li], Bs[2 * s2 + lk][32 * wn + li], acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 16; ++r) Ct[32 * wm + (r & 3) + 8 * (r >> 2) + 4 * lk][32 * wn + li] = acc[r];
  __syncthreads();
  // thread -> slot (n0 / 4 + tid % 16) of rows tid / 16 + 16 i: every load of its four elements before the first store
  const int sl = tid & 15, slot = (n0 >> 2) + sl;
  int n;
  const bool valid = vg_slot_unit(g, slot, n);
  const int nc = valid ? n : 0;
  const float e0 = ext[0 * H + nc], e1 = ext[1 * H + nc], e2 = ext[2 * H + nc], e3 = ext[3 * H + nc];
  const float b0 = bbt[0 * H + nc], b1 = bbt[1 * H + nc], b2 = bbt[2 * H + nc], b3 = bbt[3 * H + nc];
  const bool hasx = valid && n < g.I;
  float xv[4];
  int rg[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    rg[i] = m0 + (tid >> 4) + 16 * i;
    const int rc = rg[i] < TB ? rg[i] : TB - 1;
    const int t = rc / B, b = rc - t * B;
    const float v = x[(size_t)t * g.sxT + (size_t)b * g.sxB + (hasx ? n : 0)];
    xv[i] = hasx ? v : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (rg[i] < TB) {
      const float4 c4 = *reinterpret_cast<const float4*>(&Ct[(tid >> 4) + 16 * i][4 * sl]);
      const float4 out = valid ? make_float4(fmaf(xv[i], e0, c4.x + b0), fmaf(xv[i], e1, c4.y + b1), fmaf(xv[i], e2, c4.z + b2),
                                             fmaf(xv[i], e3, c4.w + b3))
                               : f4zero();
      st4(gx + ((size_t)rg[i] * NT + slot) * 4, out);
    }
  }
}

int launch_xproj(const VGeo& g, const VPack& L, const float* pack, const float* x, float* gx, float* qx,
                 hipStream_t s) {
  const int TBp = g.T * g.Bp;
  // rows per workgroup: 4 for short sequences of rows (config C, 3072 rows: 0.2390 ms per step against 0.2426 with 8 and
  // 0.2565 with 16), 8 otherwise (8192 rows: 15.5 us against 19.1 with 4); VMLMF_XR = 4 / 8 / 16 overrides (A/B runs)
  static const int xr_env = []() { const char* e = getenv("VMLMF_XR"); return e ? atoi(e) : 0; }();
  const int xr = xr_env == 16 ? 16 : (xr_env == 8 ? 8 : (xr_env == 4 ? 4 : (TBp <= 4096 ? 4 : 8)));
  const dim3 grid((TBp + xr - 1) / xr), block(256);
  const size_t lds = sizeof(float) * ((size_t)xr * g.I + (size_t)xr * g.KX + (size_t)(256 / g.KX) * xr * g.KX);
  const float *uxp = pack + L.UXP, *vxt = pack + L.VXT, *ext = pack + L.EXT, *bbt = pack + L.BBT;
#define VX_CASE(K)                                                                                          \
  case K:                                                                                                   \
    if (xr == 16) {                                                                                         \
      if (lds > 48 * 1024) {                                                                                \
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(xproj_kernel<K, 16>),        \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
        if (e != hipSuccess) return (int)e;                                                                 \
      }                                                                                                     \
      hipLaunchKernelGGL((xproj_kernel<K, 16>), grid, block, lds, s, g, x, uxp, vxt, ext, bbt, gx, qx);     \
    } else if (xr == 4) {                                                                                   \
      hipLaunchKernelGGL((xproj_kernel<K, 4>), grid, block, lds, s, g, x, uxp, vxt, ext, bbt, gx, qx);      \
    } else {                                                                                                \
      hipLaunchKernelGGL((xproj_kernel<K, 8>), grid, block, lds, s, g, x, uxp, vxt, ext, bbt, gx, qx);      \
    }                                                                                                       \
    break;
  // large layers: qx by this kernel (gx = nullptr), the expansion on the matrix cores.  VMLMF_XEXP=0 keeps the VALU form (A/B)
  static const bool xexp_on = []() { const char* e = getenv("VMLMF_XEXP"); return e == nullptr || e[0] != '0'; }();
  const bool xexp = xexp_on && g.generic && !g.bf && g.Bp == g.B && qx != nullptr;
  float* const gx_final = gx;
  if (xexp) gx = nullptr;
  if (xexp && g.time_major && g.I >= 256 && g.KX % 16 == 0) {   // qx as a skinny MFMA product (rows of x contiguous in (t, b) order)
    const int rc = generic_qx(g, x, uxp, qx, s);
    if (rc != 0) return rc;
  } else
  switch (g.KX) {
    VX_CASE(8)
    VX_CASE(16)
    VX_CASE(24)
    VX_CASE(32)
    default:
      return -3;
  }
#undef VX_CASE
  if (xexp) {
    const hipError_t e0 = hipGetLastError();
    if (e0 != hipSuccess) return (int)e0;
    const dim3 grid2((unsigned)(((long long)g.T * g.B + 63) / 64 * (4 * g.NT / 64)));
    const float* vxd = pack + L.VXD;
    switch (g.KX) {
      case 8: hipLaunchKernelGGL((xexp_mfma_kernel<8>), grid2, block, 0, s, g, x, qx, vxd, ext, bbt, gx_final); break;
      case 16: hipLaunchKernelGGL((xexp_mfma_kernel<16>), grid2, block, 0, s, g, x, qx, vxd, ext, bbt, gx_final); break;
      case 24: hipLaunchKernelGGL((xexp_mfma_kernel<24>), grid2, block, 0, s, g, x, qx, vxd, ext, bbt, gx_final); break;
      case 32: hipLaunchKernelGGL((xexp_mfma_kernel<32>), grid2, block, 0, s, g, x, qx, vxd, ext, bbt, gx_final); break;
    }
  }
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// finish: canonical gradients -> reference layouts (oracle: uncanonicalize_grads)
// ---------------------------------------------------------------------------------------------------
// element e of the x-side V gradient -> (gate k, unit n, rank r) and where the reference keeps it
__device__ __forceinline__ float* vx_dest(const VGeo& g, const RefG& o, long long e, int& k, int& n, int& r) {
  if (g.pergate) {  // w{k+1}[r][n]
    k = (int)(e / ((long long)g.rw * g.H));
    const int rem = (int)(e % ((long long)g.rw * g.H));
    r = rem / g.H;
    n = rem % g.H;
    return vg_gate(o.wg, k) + rem;
  }
  const int row = (int)(e / g.rw), c = row / g.H;
  r = (int)(e % g.rw);
  k = vg_xchunk(g, c);   // chunk c of the reference's (4H, rw) matrix holds canonical gate c ^ xperm
  n = row % g.H;
  return o.v_x + e;
}

// `health` (device word, or NULL): set when a gradient written here is not finite - a launch that gave up a bounded wait leaves NaN
// partial products (VMLMF_E_PROTOCOL), and this kernel is where every parameter gradient of a layer is written.  The package's
// Adam reads the word in its tick launch and skips the step (vmlmf_optim.hip); the compare is free, the atomic never happens in a
// healthy step.
__device__ __forceinline__ void finish_put(float* dst, float v, unsigned* health) {
  dst[0] = v;
  if (!(fabsf(v) <= 3.4028234e38f) && health != nullptr) atomicOr(health, 1u);
}
// classifier gradients (HeadBwd), thread th of 16 x (C H + C): dW[c][n] = sum_b dlogits[b][c] hT[b][n], db[c] = sum_b dlogits[b][c],
// each a fixed-order sum over the batch: sixteen lanes per output, lane j takes the rows b = j, j + 16, ... (eight independent loads
// per pass), the sixteen partial sums meet in a fixed-order butterfly over the DPP row (as one thread per output the kernel took
// 37 us at B = 512)
__device__ __forceinline__ void finish_head(const VGeo& g, const HeadBwd& hd, const long long th, unsigned* health) {
  const int H = g.H;
  auto put = [&](float* dst, float v) { finish_put(dst, v, health); };
  {
    {
      const long long eh = th >> 4;
      const int j16 = (int)(th & 15);
      if (hd.C <= 0) return;
      const bool live = eh < (long long)hd.C * H + hd.C;   // (whole 16-lane rows stay together for the butterfly)
      const long long ec = live ? eh : 0;
      const bool isb = ec >= (long long)hd.C * H;
      const int c = isb ? (int)(ec - (long long)hd.C * H) : (int)(ec / H), n = isb ? 0 : (int)(ec % H);
      float s0 = 0.f, s1 = 0.f;
      for (int b0 = j16; b0 < g.B; b0 += 16 * 8) {
        float dv[8], hv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int b = b0 + 16 * j < g.B ? b0 + 16 * j : g.B - 1;
          dv[j] = hd.dl[(size_t)b * hd.C + c];
          hv[j] = hd.hlast[(size_t)b * hd.ldh + n];
        }
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          s0 = fmaf(b0 + 16 * j < g.B ? dv[j] : 0.f, isb ? 1.f : hv[j], s0);
          s1 = fmaf(b0 + 16 * (j + 1) < g.B ? dv[j + 1] : 0.f, isb ? 1.f : hv[j + 1], s1);
        }
      }
      float v = s0 + s1;
      v = add_ror16<8>(v);
      v = add_ror16<4>(v);
      v = add_ror16<2>(v);
      v = add_ror16<1>(v);
      if (live && j16 == 0) {
        if (isb) {
          if (hd.db != nullptr) put(&hd.db[c], v);
        } else if (hd.dW != nullptr) {
          put(&hd.dW[(size_t)c * H + n], v);
        }
      }
    }
  }
}
// Element (accumulator a, unit n) of the canonical gradients straight from the weight-gradient kernels' partial blocks: the inverse of
// reduce_cg_scatter (vmlmf_wgrad.hip) and the same fixed-order sum over the blocks (vg_block_sum) - reduce_cg_kernel and finish_kernel
// as ONE launch for the stacks, whose block counts are small (round 6).  Layers with I <= H and without the x-fold.
// where accumulator a of unit n sits inside a partial block, and how many blocks hold it
__device__ __forceinline__ long long finish_cg_locate(const VGeo& g, const ReduceCounts& wc, int a, const int n, int& cnt) {
  const int KX = g.KX, KH = g.KH, GK = g.G * KH, NT = g.NT, slot = vg_slot(g, n);
  const int MT2 = (g.H + 31) / 32, MT3 = (g.I + 31) / 32;
  const int NB1p = (vg_nb1(g) + 31) / 32 * 32, NB2p = (GK + 31) / 32 * 32, NB3p = (KX + 31) / 32 * 32;
  const long long o2 = (long long)NT * 4 * NB1p, o3 = o2 + (long long)MT2 * 32 * NB2p, oe = o3 + (long long)MT3 * 32 * NB3p;
  long long e;
  cnt = wc.c[0] == 0 ? g.nchunk : wc.c[0];
  if (a < 4 * KX) {                 // va_vx(k, j)
    const int k = a / KX, j = a - k * KX;
    e = (long long)(slot * 4 + k) * NB1p + j;
  } else if ((a -= 4 * KX) < 4 * KH) {   // va_vc(k, rr)
    const int k = a / KH, rr = a - k * KH;
    e = (long long)(slot * 4 + k) * NB1p + KX + (g.flat ? (k >= 2 ? KH : 0) : 0) + rr;
  } else if ((a -= 4 * KH) < KH) {  // va_uc(rr)
    const int rr = a, s_ = (g.G == 2 && rr >= g.off1) ? 1 : 0, dest = (n / g.Hg - s_ + g.G) % g.G;
    e = o2 + (long long)n * NB2p + dest * KH + rr;
    if (wc.c[0] != 0) cnt = wc.c[1];
  } else if ((a -= KH) < KX) {      // va_ux(r)
    e = o3 + (long long)n * NB3p + a;
    if (wc.c[0] != 0) cnt = wc.c[2];
  } else {                          // va_eh / va_ex / va_b (k)
    a -= KX;
    e = oe + (long long)(a >> 2) * NT * 4 + slot * 4 + (a & 3);
  }
  return e;
}
__device__ __forceinline__ float finish_cg_from_blocks(const VGeo& g, const float* __restrict__ P, const ReduceCounts& wc, int a, const int n) {
  int cnt;
  const long long e = finish_cg_locate(g, wc, a, n, cnt);
  return vg_block_sum(P, g.PCH, e, 0, cnt);
}

// cgu / e_in (finish_units_stack_kernel): the canonical gradients of ONE unit, [accumulator], summed over the blocks into LDS by the
// caller, and the element to finish (an element of that unit); by default the element is the thread's global index
__device__ __forceinline__ void finish_body(const VGeo& g, const RefP& p, const float* __restrict__ cg, const RefG& o, const HeadBwd& hd,
                                            const long long nbody, unsigned* health, const float* __restrict__ P = nullptr,
                                            const ReduceCounts wc = ReduceCounts{{0, 0, 0}}, const float* cgu = nullptr,
                                            const long long e_in = -1) {
  const int NT = g.NT, H = g.H, I = g.I, rw = g.rw, Hg = g.Hg;
  auto put = [&](float* dst, float v) { finish_put(dst, v, health); };
  const long long e_lin = e_in >= 0 ? e_in : (long long)blockIdx.x * blockDim.x + threadIdx.x;
  {
    const long long th = e_lin - nbody;
    if (th >= 0) {
      finish_head(g, hd, th, health);
      return;
    }
  }
  auto CG = [&](int a, int n) {
    return cgu != nullptr ? cgu[a] : (P != nullptr ? finish_cg_from_blocks(g, P, wc, a, n) : cg[(size_t)a * NT + vg_slot(g, n)]);
  };
  const long long n_ux = (long long)I * rw, n_vx = 4LL * H * rw, n_dx = I, n_dh = H, n_b = 4LL * H;
  const long long n_uh0 = (long long)H * g.ru0, n_vh0 = 4LL * H * g.ru0;
  const long long n_uh1 = g.G == 2 ? (long long)H * g.ru1 : 0, n_vh1 = g.G == 2 ? 4LL * H * g.ru1 : 0;
  long long e = e_lin;
  if (g.foldx) {
    // x-fold: the accumulator rows va_vx(k, m), m < I, hold G[k][m](n) = sum_rows dpre[row][n][k] x[row][m].
    // du_x[m][r] = sum_{n,k} G[k][m](n) vx(n,k,r) - ...: one wave per element, fixed-order butterfly
    if (e < n_ux * 64) {
      const long long eo = e >> 6;
      const int lane = (int)(e & 63), m = (int)(eo / rw), r = (int)(eo % rw);
      float v = 0.f;
      for (int n = lane; n < H; n += 64)
        for (int k = 0; k < 4; ++k) v = fmaf(CG(va_vx(g, k, m), n), ref_vx(g, p, n, k, r), v);
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0) {
        if (!g.novm)
          for (int k = 0; k < 4; ++k) v -= CG(va_ex(g, k), m) * ref_vx(g, p, m, k, r);
        put(&o.u_x[eo], v);
      }
      return;
    }
    e -= n_ux * 64;
    if (e < n_vx) {  // dv_x[k*H+n][r] = sum_m G[k][m](n) ux(m,r) - ...
      int k, n, r;
      float* dst = vx_dest(g, o, e, k, n, r);
      float v = 0.f;
      for (int m = 0; m < I; ++m) v = fmaf(CG(va_vx(g, k, m), n), ref_ux(g, p, m, r), v);
      if (n < I && !g.novm) v -= CG(va_ex(g, k), n) * ref_ux(g, p, n, r);
      put(dst, v);
      return;
    }
    e += n_ux;   // fall through to the branches below with the offsets they expect
  }
  if (e < n_ux) {  // du_x[m][r]
    if (g.foldx) return;
    const int m = (int)(e / rw), r = (int)(e % rw);
    float v = I > H ? cg[(size_t)g.NA * NT + (size_t)m * g.KX + r] : CG(va_ux(g, r), m);   // (I > H: reduce_cg_kernel's own block)
    if (!g.novm)
      for (int k = 0; k < 4; ++k) v -= CG(va_ex(g, k), m) * ref_vx(g, p, m, k, r);
    put(&o.u_x[e], v);
    return;
  }
  e -= n_ux;
  if (e < n_vx) {  // dv_x[k*H+n][r]
    if (g.foldx) return;
    int k, n, r;
    float* dst = vx_dest(g, o, e, k, n, r);
    float v = CG(va_vx(g, k, r), n);
    if (n < I && !g.novm) v -= CG(va_ex(g, k), n) * ref_ux(g, p, n, r);
    put(dst, v);
    return;
  }
  e -= n_vx;
  if (e < n_dx) {
    if (g.novm) return;
    const int m = (int)e;
    put(&o.dia_x[m], (CG(va_ex(g, 0), m) + CG(va_ex(g, 1), m)) + (CG(va_ex(g, 2), m) + CG(va_ex(g, 3), m)));
    return;
  }
  e -= n_dx;
  if (e < n_dh) {
    if (g.novm) return;
    const int n = (int)e;
    put(&o.dia_h[n], (CG(va_eh(g, 0), n) + CG(va_eh(g, 1), n)) + (CG(va_eh(g, 2), n) + CG(va_eh(g, 3), n)));
    return;
  }
  e -= n_dh;
  if (e < n_b) {
    const int k = (int)(e / H), n = (int)(e % H);
    const float v = CG(va_b(g, k), n);
    if (g.pergate) {
      put(&vg_gate(o.bg, k)[n], v);
    } else {
      put(&o.b_x[vg_xchunk(g, k) * H + n], v);
      put(&o.b_h[vg_hchunk(g, k) * H + n], v);
    }
    return;
  }
  e -= n_b;
  for (int s = 0; s < g.G; ++s) {
    const int rus = s ? g.ru1 : g.ru0, base = s ? g.off1 : 0;
    const long long n_u = s ? n_uh1 : n_uh0, n_v = s ? n_vh1 : n_vh0;
    float* ou = s ? o.u_h1 : o.u_h0;
    float* ov = s ? o.v_h1 : o.v_h0;
    if (e < n_u) {
      int n, r;
      if (g.G == 1) {
        n = (int)(e / rus);
        r = (int)(e % rus);
      } else {  // u_h[s][j][m][r]: the unit that feeds destination j through shift s
        const int j = (int)(e / ((long long)Hg * rus));
        const int rem = (int)(e % ((long long)Hg * rus));
        const int m = rem / rus;
        r = rem % rus;
        n = ((j + s) % g.G) * Hg + m;
      }
      float v = CG(va_uc(g, base + r), n);
      if (s == 0 && !g.novm)
        for (int k = 0; k < 4; ++k) v -= CG(va_eh(g, k), n) * ref_vc(g, p, n, k, r);
      put(&ou[e], v);
      return;
    }
    e -= n_u;
    if (e < n_v) {
      int n, k, r;
      float* dst = ov + e;
      if (g.pergate) {  // u{k+1}[r][n]
        k = (int)(e / ((long long)rus * H));
        const int rem = (int)(e % ((long long)rus * H));
        r = rem / H;
        n = rem % H;
        dst = vg_gate(o.ug, k) + rem;
      } else if (g.G == 1) {
        const int row = (int)(e / rus);
        r = (int)(e % rus);
        k = row / H;
        n = row % H;
      } else {  // v_h[s][q][r][col]
        const int q = (int)(e / ((long long)rus * 4 * Hg));
        const int rem = (int)(e % ((long long)rus * 4 * Hg));
        r = rem / (4 * Hg);
        const int col = rem % (4 * Hg);
        if (g.flat) {
          const int f = q * 4 * Hg + col;
          k = f / H;
          n = f % H;
        } else {
          const int c = col / Hg;
          k = g.hperm ? (c ^ 1) : c;
          n = q * Hg + (col % Hg);
        }
      }
      float v = CG(va_vc(g, k, base + r), n);
      if (s == 0 && !g.novm) v -= CG(va_eh(g, k), n) * ref_uc(g, p, n, r);
      put(dst, v);
      return;
    }
    e -= n_v;
  }
}

__global__ void __launch_bounds__(256) finish_kernel(VGeo g, RefP p, const float* __restrict__ cg, RefG o, HeadBwd hd,
                                                     long long nbody, unsigned* health) {
  finish_body(g, p, cg, o, hd, nbody, health);
}
// ---------------------------------------------------------------------------------------------------
// finish2: reduce_cg_kernel + finish_kernel of a backward whose weight-gradient workers rode on the recurrent launch, as ONE launch
// (verdict r4 item 1b: 5.2 + 5.6 us and a launch boundary at the headline shape).  Envelope: finish2_ok (one group, V1 / V3 layouts,
// x-fold, padded ranks <= 16: one 32-column tile per product).
// Three kinds of workgroup (384 threads):
//   [0, nU)            two hidden units each: the 344 partial sums they read (their eight (slot, gate) rows of C1, their two rows of
//                      C2, their element sums) are summed over the K blocks in reduce_cg_kernel's order (vg_block_sum) into LDS, then
//                      every gradient entry of the two units is one thread's work: dV_h, dV_x (through the x-fold), dU_h, the biases,
//                      d(dia)
//   [nU, nU + I)       d(u_x)[m][:] for one input m: the workers' K x NT/8 tile shares (WRide::dux), sixteen threads per rank each
//                      summing its part in index order, the parts in part order, minus the vm term
//   the rest           the classifier's dW / db (finish_kernel's code)
// Block 0 also puts the rows' progress words back to zero (reduce_cg_kernel did).
// ---------------------------------------------------------------------------------------------------
struct Finish2Args {
  const float* P;     // K partial blocks, PCH floats apart
  const float* dux;   // [K][NT / 8][256]
  unsigned* prog;
  unsigned* health;
  int K, nU;
};
constexpr int F2_T = 384;   // threads: one per partial sum of a pair of units (344)
__global__ void __launch_bounds__(F2_T) finish2_kernel(VGeo g, RefP p, RefG o, HeadBwd hd, Finish2Args fa) {
  __shared__ float red[352];       // the two units' reduced values: C1 [8 rows][32] | C2 [2][32] | E [3][8]
  __shared__ float part[16][16];   // d(u_x) workgroups: [part][rank]
  const int tid = threadIdx.x, H = g.H, NT = g.NT, I = g.I, KX = g.KX, KH = g.KH, rw = g.rw, ru = g.ru0;
  const int bid = blockIdx.x;
  if (bid == 0 && fa.prog != nullptr)
    for (int b = tid; b < g.B; b += F2_T) fa.prog[(size_t)b * WR_PROG_STRIDE] = 0u;
  auto put = [&](float* dst, float v) { finish_put(dst, v, fa.health); };
  const int MT2 = (H + 31) / 32, MT3 = (I + 31) / 32;
  const long long o2 = (long long)NT * 4 * 32, oe = o2 + (long long)MT2 * 32 * 32 + (long long)MT3 * 32 * 32;
  if (bid < fa.nU) {
    const int n0 = 2 * bid;                       // units n0, n0 + 1 (one group: unit == thread slot)
    // ---- 1. the K-sums
    for (int e = tid; e < 344; e += F2_T) {
      long long src;
      if (e < 256) src = (long long)n0 * 128 + e;                                        // C1 rows (n0, gate 0) .. (n0 + 1, gate 3)
      else if (e < 320) src = o2 + (long long)n0 * 32 + (e - 256);                        // C2 rows n0, n0 + 1
      else src = oe + (long long)((e - 320) >> 3) * NT * 4 + (long long)n0 * 4 + ((e - 320) & 7);   // E[which][(unit, gate)]
      red[e] = vg_block_sum(fa.P, g.PCH, src, 0, fa.K);
    }
    __syncthreads();
    // ---- 2. the two units' gradient entries, one per thread (finish_kernel's arithmetic on the same values)
    const float* C1 = red;           // [(u, k)][32]: columns m < KX the x-fold's G, columns KX + r the raw dV_h
    const float* C2 = red + 256;     // [u][32]
    const float* E = red + 320;      // [which: eh, ex, b][(u, k)]
    for (int t = tid; t < 2 * 154; t += F2_T) {
      const int u = t / 154, q = t - u * 154, n = n0 + u;
      if (n >= H) continue;
      if (q < 64) {                  // dv_h[k H + n][r]
        const int k = q >> 4, r = q & 15;
        if (r < ru) put(&o.v_h0[((size_t)k * H + n) * ru + r], C1[(u * 4 + k) * 32 + KX + r] - E[0 * 8 + u * 4 + k] * p.u_h0[(size_t)n * ru + r]);
      } else if (q < 128) {          // dv_x[k H + n][r] = sum_m G[k][m](n) u_x[m][r] - [n < I] ex(k)(n) u_x[n][r]
        const int k = (q - 64) >> 4, r = (q - 64) & 15;
        if (r < rw) {
          float v = 0.f;
          for (int m = 0; m < I; ++m) v = fmaf(C1[(u * 4 + k) * 32 + m], p.u_x[(size_t)m * rw + r], v);
          if (n < I) v -= E[1 * 8 + u * 4 + k] * p.u_x[(size_t)n * rw + r];
          put(&o.v_x[((size_t)k * H + n) * rw + r], v);
        }
      } else if (q < 144) {          // du_h[n][r] = raw - sum_k eh(k)(n) v_h[k H + n][r]
        const int r = q - 128;
        if (r < ru) {
          float v = C2[u * 32 + r];
          for (int k = 0; k < 4; ++k) v -= E[0 * 8 + u * 4 + k] * p.v_h0[((size_t)k * H + n) * ru + r];
          put(&o.u_h0[(size_t)n * ru + r], v);
        }
      } else if (q < 148) {          // biases (the reference keeps two copies of the same gradient)
        const int k = q - 144;
        const float v = E[2 * 8 + u * 4 + k];
        put(&o.b_x[(size_t)k * H + n], v);
        put(&o.b_h[(size_t)k * H + n], v);
      } else if (q == 148) {
        put(&o.dia_h[n], (E[0 * 8 + u * 4 + 0] + E[0 * 8 + u * 4 + 1]) + (E[0 * 8 + u * 4 + 2] + E[0 * 8 + u * 4 + 3]));
      } else if (q == 149) {
        if (n < I) put(&o.dia_x[n], (E[1 * 8 + u * 4 + 0] + E[1 * 8 + u * 4 + 1]) + (E[1 * 8 + u * 4 + 2] + E[1 * 8 + u * 4 + 3]));
      }
    }
    return;
  }
  if (bid < fa.nU + I) {
    // ---- d(u_x)[m][r]: K x MT1 tile shares, 16 parts x 16 ranks
    const int m = bid - fa.nU, r = tid & 15, pt = tid >> 4, MT1 = NT / 8;
    const int total = fa.K * MT1, per = (total + 15) / 16;
    const int c0 = pt * per < total ? pt * per : total, c1 = c0 + per < total ? c0 + per : total;
    if (tid < 256) part[pt][r] = vg_block_sum(fa.dux, 256, (long long)m * 16 + r, c0, c1);
    // the vm term needs ex(k)(m) summed over the K blocks (in the order the units' workgroups sum it)
    __shared__ float exs[4];
    if (tid >= 256 && tid < 260) exs[tid - 256] = vg_block_sum(fa.P, g.PCH, oe + (long long)NT * 4 + (long long)m * 4 + (tid - 256), 0, fa.K);
    __syncthreads();
    if (tid < 16 && r < rw) {
      float v = part[0][r];
#pragma unroll
      for (int q = 1; q < 16; ++q) v += part[q][r];
      for (int k = 0; k < 4; ++k) v -= exs[k] * p.v_x[((size_t)k * H + m) * rw + r];
      put(&o.u_x[(size_t)m * rw + r], v);
    }
    return;
  }
  // ---- the classifier's gradients
  const long long th = (long long)(bid - fa.nU - I) * F2_T + tid;   // (F2_T is a multiple of 16: whole DPP rows per output)
  finish_head(g, hd, th, fa.health);
}

struct FinishLayer {
  VGeo g;
  RefP p;
  RefG o;
  const float* cg;
  long long nbody;
  HeadBwd hd;   // classifier gradients ride with the top layer (C = 0 elsewhere)
  const float* P;    // or: the partial blocks themselves (cg unused): no reduce launch in front of this one
  ReduceCounts wc;
  int pad;
};
struct FinishStack {
  FinishLayer l[WF_MAXL];
  unsigned* health;
};
__global__ void __launch_bounds__(256) finish_stack_kernel(FinishStack S) {   // grid.y = layer (wavefront path; no classifier)
  const FinishLayer& f = vg_karg_ref<FinishLayer>((size_t)blockIdx.y * sizeof(FinishLayer));
  if ((long long)blockIdx.x * 256 >= f.nbody && f.hd.C <= 0) return;
  finish_body(f.g, f.p, f.cg, f.o, f.hd, f.nbody, vg_karg_ref<unsigned*>(offsetof(FinishStack, health)), f.P, f.wc);
}

static long long finish_elements(const VGeo& g);
// ---------------------------------------------------------------------------------------------------
// finish_units_stack_kernel (round 6): reduce_cg_stack_kernel + finish_stack_kernel as ONE launch.  A workgroup takes ONE hidden unit
// of one layer: its NA canonical gradients (5 KX + 5 KH + 12 accumulators) are summed over the partial blocks - one accumulator per
// thread, vg_block_sum's order: the bits of the two-launch path - into LDS, then every reference-layout gradient entry of that unit
// (d(u_x) row n, the four d(v_x) and d(v_h) rows, d(u_h) row n, the biases, d(dia)) is one thread's finish_body() on those values.
// (The first fused form - finish_stack_kernel reading the blocks itself, vmlmf_tune("ffb") - sums d(ex) / d(eh) once per rank: 25.5
// us against 8.1 + 6.2.)  One-group layers without the x-fold, input_size <= hidden_size.  Workgroups behind the widest layer's
// units: the classifier's gradients (top layer).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ long long finish_unit_element(const VGeo& g, const int n, int j) {
  const int H = g.H, I = g.I, rw = g.rw, ru = g.ru0;
  const long long n_ux = (long long)I * rw, n_vx = 4LL * H * rw, n_dx = I, n_dh = H, n_b = 4LL * H, n_uh0 = (long long)H * ru;
  if (j < rw) return n < I ? (long long)n * rw + j : -1;                                   // d(u_x)[n][r]
  j -= rw;
  if (j < 4 * rw) {                                                                          // d(v_x): chunk / gate c, rank r
    const int c = j / rw, r = j - c * rw;
    return n_ux + (g.pergate ? ((long long)c * rw + r) * H + n : ((long long)c * H + n) * rw + r);
  }
  j -= 4 * rw;
  if (j == 0) return n < I ? n_ux + n_vx + n : -1;                                           // d(dia_x)
  if (j == 1) return n_ux + n_vx + n_dx + n;                                                 // d(dia_h)
  j -= 2;
  if (j < 4) return n_ux + n_vx + n_dx + n_dh + (long long)j * H + n;                        // biases of gate j
  j -= 4;
  const long long b0 = n_ux + n_vx + n_dx + n_dh + n_b;
  if (j < ru) return b0 + (long long)n * ru + j;                                             // d(u_h)[n][r]
  j -= ru;
  if (j < 4 * ru) {
    const int c = j / ru, r = j - c * ru;
    return b0 + n_uh0 + (g.pergate ? ((long long)c * ru + r) * H + n : ((long long)c * H + n) * ru + r);
  }
  return -1;
}
constexpr int FU_T = 256;
__global__ void __launch_bounds__(FU_T) finish_units_stack_kernel(FinishStack S, int hmax) {
  __shared__ float cgs[5 * 32 + 5 * 32 + 12];
  const FinishLayer& f = vg_karg_ref<FinishLayer>((size_t)blockIdx.y * sizeof(FinishLayer));
  unsigned* health = vg_karg_ref<unsigned*>(offsetof(FinishStack, health));
  const int n = (int)blockIdx.x, tid = (int)threadIdx.x;
  if (n >= hmax) {   // the classifier's elements
    if (f.hd.C > 0) finish_body(f.g, f.p, f.cg, f.o, f.hd, f.nbody, health, nullptr, ReduceCounts{{0, 0, 0}}, nullptr,
                                f.nbody + (long long)(n - hmax) * FU_T + tid);
    return;
  }
  if (n >= f.g.H) return;
  // (one thread per accumulator, the blocks in reduce_cg_stack_kernel's order.  Measured at config C: 14.3 us against 8.2 + 7.3 for
  //  the two launches it replaces; with every accumulator's blocks in four runs on four threads - 1024 threads a unit - 15.9: the
  //  launch is not bound by the chain of a thread's loads but by reading 27 MB of partial blocks in 1 KB pieces)
  const int NA = f.g.NA;
  for (int a = tid; a < NA; a += FU_T) cgs[a] = finish_cg_from_blocks(f.g, f.P, f.wc, a, n);
  __syncthreads();
  const int nout = 5 * f.g.rw + 5 * f.g.ru0 + 6;
  for (int j = tid; j < nout; j += FU_T) {
    const long long e = finish_unit_element(f.g, n, j);
    if (e >= 0) finish_body(f.g, f.p, f.cg, f.o, HeadBwd{}, f.nbody, health, nullptr, ReduceCounts{{0, 0, 0}}, cgs, e);
  }
}
bool finish_units_ok(const VGeo& g) { return g.G == 1 && !g.foldx && g.I <= g.H && g.NA <= 332; }
int launch_finish_units_stack(int L, const VGeo* g, const RefP* p, const RefG* out, const HeadBwd& hd_top, hipStream_t s,
                              unsigned* health, const float* const* wpart, const ReduceCounts* wc) {
  FinishStack S;
  memset(&S, 0, sizeof(S));
  S.health = health;
  int hmax = 0;
  long long nhead = 0;
  for (int l = 0; l < L; ++l) {
    if (!finish_units_ok(g[l])) return -3;
    S.l[l].g = g[l], S.l[l].p = p[l], S.l[l].o = out[l], S.l[l].cg = nullptr, S.l[l].nbody = finish_elements(g[l]);
    S.l[l].P = wpart[l];
    if (wc != nullptr) S.l[l].wc = wc[l];
    hmax = g[l].H > hmax ? g[l].H : hmax;
    if (l == L - 1 && hd_top.C > 0) {
      S.l[l].hd = hd_top;
      nhead = 16 * ((long long)hd_top.C * g[l].H + hd_top.C);   // sixteen lanes per classifier output
    }
  }
  hipLaunchKernelGGL(finish_units_stack_kernel, dim3((unsigned)(hmax + (nhead + FU_T - 1) / FU_T), L), dim3(FU_T), 0, s, S, hmax);
  return (int)hipGetLastError();
}

static long long finish_elements(const VGeo& g) {
  long long n = (long long)g.I * g.rw * (g.foldx ? 64 : 1) + 4LL * g.H * g.rw + g.I + g.H + 4LL * g.H;
  n += (long long)g.H * g.ru0 + 4LL * g.H * g.ru0;
  if (g.G == 2) n += (long long)g.H * g.ru1 + 4LL * g.H * g.ru1;
  return (n + 255) / 256 * 256;   // the classifier's elements start on a workgroup boundary
}

bool finish_from_blocks_ok(const VGeo& g) { return !g.foldx && g.I <= g.H; }
int launch_finish_stack(int L, const VGeo* g, const RefP* p, const float* const* cgrad, const RefG* out, const HeadBwd& hd_top,
                        hipStream_t s, unsigned* health, const float* const* wpart, const ReduceCounts* wc) {
  static_assert(sizeof(FinishStack) <= 4096, "kernel-argument segment");
  FinishStack S;
  memset(&S, 0, sizeof(S));
  S.health = health;
  long long nmax = 0;
  for (int l = 0; l < L; ++l) {
    S.l[l].g = g[l], S.l[l].p = p[l], S.l[l].o = out[l], S.l[l].cg = cgrad[l], S.l[l].nbody = finish_elements(g[l]);
    if (wpart != nullptr) {
      if (!finish_from_blocks_ok(g[l])) return -3;
      S.l[l].P = wpart[l];
      if (wc != nullptr) S.l[l].wc = wc[l];
    }
    long long n = S.l[l].nbody;
    if (l == L - 1 && hd_top.C > 0) {
      S.l[l].hd = hd_top;
      n += (16 * ((long long)hd_top.C * g[l].H + hd_top.C) + 255) / 256 * 256;   // sixteen lanes per classifier output
    }
    nmax = n > nmax ? n : nmax;
  }
  hipLaunchKernelGGL(finish_stack_kernel, dim3((unsigned)(nmax / 256), L), dim3(256), 0, s, S);
  return (int)hipGetLastError();
}

// (variants 1 / 3: VMLMF_V1_CELL / VMLMF_V3_LM of include/vmlmf_hip.h - v_h as (4H, r), v_x as (4H, rw), vm vectors present)
bool finish2_ok(const VGeo& g) {
  return (g.variant == 1 || g.variant == 3) && g.G == 1 && !g.flat && !g.generic && !g.rb && !g.bf && g.foldx &&
         g.KH <= 16 && g.KX <= 16 && g.I <= 16 && vg_nb1(g) <= 32;
}
int launch_finish2(const VGeo& g, const RefP& p, const float* wpart, const float* dux, int K, const RefG& out, const HeadBwd& hd,
                   unsigned* prog, hipStream_t s, unsigned* health) {
  if (!finish2_ok(g) || K < 1) return -3;
  Finish2Args fa;
  fa.P = wpart, fa.dux = dux, fa.prog = prog, fa.health = health, fa.K = K, fa.nU = (g.H + 1) / 2;
  const long long nhead = hd.C > 0 ? 16 * ((long long)hd.C * g.H + hd.C) : 0;   // sixteen lanes per classifier output
  hipLaunchKernelGGL(finish2_kernel, dim3((unsigned)(fa.nU + g.I + (nhead + F2_T - 1) / F2_T)), dim3(F2_T), 0, s, g, p, out, hd, fa);
  return (int)hipGetLastError();
}

int launch_finish(const VGeo& g, const RefP& p, const float* cgrad, const RefG& out, const HeadBwd& hd, hipStream_t s, unsigned* health) {
  const long long nbody = finish_elements(g);
  const long long nhead = hd.C > 0 ? 16 * ((long long)hd.C * g.H + hd.C) : 0;   // sixteen lanes per classifier output
  hipLaunchKernelGGL(finish_kernel, dim3((unsigned)((nbody + nhead + 255) / 256)), dim3(256), 0, s, g, p, cgrad, out, hd, nbody, health);
  return (int)hipGetLastError();
}
