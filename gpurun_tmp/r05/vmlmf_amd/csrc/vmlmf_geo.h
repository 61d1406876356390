// Geometry shared by host launch code and device kernels (gfx950 only; wave = 64 lanes).
//
// Thread <-> hidden-unit map of the persistent kernels.  A workgroup has NT = G * W * 64 threads; the G
// groups of the hidden->hidden path are wave-aligned (W waves each), so that a wave's rank-space partial
// sums belong to exactly one source group:
//     grp = tid / (64 W),  m = tid % (64 W),  unit n = grp * Hg + m  (valid iff m < Hg).
//
// Rank space.  Shift s of the group path has rank ru[s]; blocks are padded to multiples of 8 and
// concatenated: block 0 = [0, off1), block 1 = [off1, KH).  KX = pad8(w_rank).  The reductions run in
// passes of 16 ranks (one rank per lane of a 16-lane DPP row): NP = ceil(KH/16), KQ = 16 NP.
#pragma once
#include <stdint.h>

struct VGeo {
  int variant, B, T, I, H, rw, G, Hg, W, NT, NW;
  int ru0, ru1, off1;
  int KX, KH, NP, KQ, NPX, KQX;
  int flat;   // V4: [B, g*4Hg] flattened then chunked (vmlmf_lm.py:135,155): gate k picks Q[k / 2]
  int hperm;  // V2, V6: h-side chunks are (f,i,n,o) (vmlmf_group.py:134,149-152)
  int xperm;  // V6: so are the x-side chunks (vmlmf_group.py:211)
  int novm;   // V5, V6: no dia_x / dia_h and no diagonal removal: ex = eh = 0, nothing folds back in finish_kernel
  int pergate;  // V5: V factors and biases come as four (rank, H) / (1, H) tensors, one per gate (vmlmf.py:159-186)
  int R;      // batch rows per workgroup
  int nwg;    // workgroups of the recurrent kernels
  int Bp;     // nwg * R: batch rows of the internal (slot-padded) buffers [T][Bp][NT]
  int RC;     // (t,b) rows per dqx_dx workgroup
  int nblk;   // dqx_dx workgroups
  int RC2;    // (t,b) rows per wgrad chunk
  int nchunk; // wgrad chunks (grid.y)
  long long PCH;  // floats of partial products per chunk (wgrad_mfma_kernel)
  int NA;     // accumulators per thread in wgrad = 5 KX + 5 KH + 12
  long long sxT, sxB, syT, syB;  // element strides of x/dx and y/dy
  int time_major, training;
  int foldx;    // 1 (I <= KX): the dV product contracts dpre with x instead of qx = x U_x (same 32-column tile), i.e. it
                //    yields G = dpre^T x; finish_kernel derives dV_x = G U_x and dU_x = G^T V_x from it, so neither dqx
                //    nor the x^T dqx product is needed unless the layer's input wants a gradient
  int generic;  // 1: step-wise path (vmlmf_generic.hip): factors do not fit the register-resident kernels
  int bf;       // 1: desc.dtype = bf16: bf16 MFMA in the recurrence, bf16 tapes (x-side pre-activations, gates, dpre)
  int bt;       // 1: the gate tape of the wavefront kernels is kept as bf16 (desc.dtype = bf16 on a stack below the batch where the
                // bf16-MFMA row blocks pay: fp32 arithmetic everywhere, 8 instead of 16 bytes per unit and step written and read back)
  int rb;       // > 0: the recurrence runs on the row-block MFMA kernels (vmlmf_rb.hip), 16 batch rows per workgroup;
                //      the value is S, the workgroups a row block's hidden units are split over (1 = no cluster)
};

// Geometry of the row-block kernels (vmlmf_rb.hip).  Valid 16-unit tiles of a group are dealt to "wave slots"
// (4 compute waves x S workgroups, split evenly over the groups), MT consecutive tiles each.
struct RbGeo {
  int S, MT, TPGV, WSG, NMT, nmu, nrb;
  int rbl;                      // live batch rows of a 16-column MFMA tile (16; 8 or 4 for clusters at small batches)
  int mlist[2][5];              // per group: the M-tiles of the padded rank space its units couple to
  unsigned tgcode;              // four bits per M-tile m: 1 + the one group whose units feed it in the forward reduce, 0: several / all
  long long UA, VA, VB, UB, total;   // float offsets of the A-operand images inside the RB region of PACK
  long long xq_floats, flag_words;   // cluster exchange scratch (S > 1)
};

// float offsets inside the PACK region (parameter images, produced by pack_kernel)
struct VPack {
  long long VE, UR, VR, UE, EH, VRX, UXO, EXI, UXP, VXT, EXT, BBT;
  long long UD, VD, UDT, VDT, VXTT;   // dense group factors + V_x^T, step-wise path only
  long long TKT;                      // split-K tickets of the step-wise GEMMs (ints; pack_kernel zeroes them)
  long long VXD;                      // V_x as a (rank x 4*slots) matrix: B operand of the MFMA x-side expansion (large layers)
  long long WXD;                      // dense x-side matrix W_x[m][k][slot] of the x-projection wave (I <= 16 only)
  long long RB;                       // A-operand images of the row-block kernels (RbGeo offsets are relative to RB)
  long long WF, total;                // rotated images of the wavefront kernels (WfPack offsets are relative to WF)
};

#ifdef __HIPCC__
#define VG_HD __host__ __device__ inline
#else
#define VG_HD inline
#endif

#define VG_GEMM_TICKETS 64   // tiles a split-K GEMM of the step-wise path may have
#define VG_GEMM_SPLIT 8     // partial copies its scratch holds

VG_HD int vg_pad8(int v) { return (v + 7) / 8 * 8; }

// Shapes whose x-projection rides inside rec_fwd_kernel (its wave NW): one row per workgroup, at most three
// compute waves, an input narrow enough that the dense x-side matrix fits that wave's registers.
VG_HD bool vg_xwave_ok(const VGeo& g) { return !g.generic && !g.rb && !g.flat && g.R == 1 && g.NT <= 192 && g.I <= 16; }

// rb_floats: RbGeo::total of the layer (0 without the row-block kernels)
VG_HD VPack vg_pack_layout(const VGeo& g, long long rb_floats = 0, long long wf_floats = 0) {
  VPack p;
  long long o = 0;
  auto take = [&](long long n) { long long r = o; o += (n + 63) / 64 * 64; return r; };
  const int pk = g.generic ? 0 : 1;   // register images exist only for the persistent kernels
  const bool dense = g.generic && !g.rb;   // dense group factors: step-wise recurrence only
  p.VE = take(pk * 4LL * g.KH * g.NT);
  p.UR = take(pk * 1LL * g.KQ * g.NT);
  p.VR = take(pk * 4LL * g.KQ * g.NT);
  p.UE = take(pk * 1LL * g.KH * g.NT);
  p.EH = take(4LL * g.NT);
  p.VRX = take(pk * 4LL * g.KQX * g.NT);
  p.UXO = take(pk * 1LL * g.KX * g.NT);
  p.EXI = take(pk * 4LL * g.NT);
  p.UXP = take(1LL * g.I * g.KX);
  p.VXT = take(4LL * g.KX * g.H);
  p.EXT = take(4LL * g.H);
  p.BBT = take(4LL * g.H);
  const long long GK = (long long)g.G * g.KH, N4 = 4LL * g.NT;
  p.UD = take(dense ? g.H * GK : 0);
  p.VD = take(dense ? GK * N4 : 0);
  p.UDT = take(dense ? GK * g.H : 0);
  p.VDT = take(dense ? N4 * GK : 0);
  p.VXTT = take(g.generic ? N4 * g.KX : 0);
  p.TKT = take(g.generic ? VG_GEMM_TICKETS : 0);
  p.VXD = take(g.generic ? N4 * g.KX : 0);
  p.WXD = take(vg_xwave_ok(g) ? 4LL * g.I * g.NT : 0);
  p.RB = take(g.rb ? rb_floats : 0);
  p.WF = take(wf_floats);
  p.total = o;
  return p;
}

// Columns of B in the dV product (wgrad mode 1): qx and the rank-space vectors a (slot, gate) column can pair with.  In
// the flat layout a slot's gates read both vectors; otherwise (V1-V3, V5, V6) a unit only ever reads the vector of its
// own group, and the slots of a task (8 consecutive ones) lie in one group, so the other vector's columns would only
// be computed to be thrown away by reduce_cg_kernel (they were: 80 instead of 48 columns for the group cell).
VG_HD int vg_nb1(const VGeo& g) { return g.KX + (g.flat ? g.G * g.KH : g.KH); }

// wgrad accumulator indices (per thread slot)
VG_HD int va_vx(const VGeo& g, int k, int r) { return k * g.KX + r; }
VG_HD int va_vc(const VGeo& g, int k, int rr) { return 4 * g.KX + k * g.KH + rr; }
VG_HD int va_uc(const VGeo& g, int rr) { return 4 * g.KX + 4 * g.KH + rr; }
VG_HD int va_ux(const VGeo& g, int r) { return 4 * g.KX + 5 * g.KH + r; }
VG_HD int va_eh(const VGeo& g, int k) { return 5 * g.KX + 5 * g.KH + k; }
VG_HD int va_ex(const VGeo& g, int k) { return 5 * g.KX + 5 * g.KH + 4 + k; }
VG_HD int va_b(const VGeo& g, int k) { return 5 * g.KX + 5 * g.KH + 8 + k; }

// thread slot of hidden unit n
VG_HD int vg_slot(const VGeo& g, int n) {
  int grp = n / g.Hg;
  return grp * 64 * g.W + (n - grp * g.Hg);
}
