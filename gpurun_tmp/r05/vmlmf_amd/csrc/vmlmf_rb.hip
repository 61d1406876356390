// Row-block recurrent kernels on the fp32 matrix cores (gfx950): one workgroup owns RB = 16 batch rows for all T
// timesteps and both skinny products of a step run on v_mfma_f32_16x16x4_f32 with the WEIGHTS as the A operand (resident
// in registers for the whole sequence) and the batch rows as the N dimension:
//
//   reduce   Q^T  (ranks x rows)  = U^T (ranks x units) . h^T (units x rows)           K = this wave's units
//   expand   pre^T(units x rows)  = V_k (units x ranks) . Q^T (ranks x rows)  + gx^T   K = ranks, per gate k
//
// The transposed formulation is what keeps the recurrence in registers: an MFMA result tile holds, in lane (c, kq)
// (c = lane & 15 = batch row, kq = lane >> 4), the four elements i = 4 kq + reg of column c.  With units as the M
// dimension of `expand`, lane (c, kq) ends a step holding h of units 4 kq + {0..3} of its tile for row c - which is exactly a
// B operand (B[k = kq][n = c]) of the next step's `reduce` when contraction step r pairs lane group kq with unit 4 kq + r.
// The weight images are built for that pairing once per call (rb_pack_kernel), so neither h nor the rank-space vector is ever
// transposed, shuffled or staged: registers -> MFMA -> registers.  The only exchange per step is the sum of the waves'
// K-partial Q tiles through LDS (one barrier), and, when a layer's factors do not fit one CU (H = 650), the same sum across
// the S workgroups of a cluster through L2 (write-through stores + one flag per workgroup and step).
//
// Replaces the same reference code as rec_fwd_kernel / rec_bwd_kernel (vmlmf.py:300-314 + 78-125, vmlmf_group.py:85-155,
// vmlmf_lm.py:272-280 + 222-269, 166-174 + 97-163, and autograd's replay of them); selected by make_geo for large batches
// and for layers beyond the register-resident VALU kernels.  Same tapes, so the weight-gradient kernels are shared.
#include "vmlmf_launch.h"
#include <cstring>

namespace {

constexpr int RB = 16;        // batch rows per workgroup (the N of the MFMA tiles)
constexpr int RB_WAVES = 4;   // compute waves per workgroup, one per SIMD

// D-row i of a result tile <-> rank inside its 16-rank M-tile: pi(i) = 4 (i % 4) + i / 4, so that register r of lane
// group kq holds rank 4 r + kq, i.e. contraction step r of a later product covers the CONTIGUOUS ranks 4 r .. 4 r + 3
// (steps beyond a rank width that is not a multiple of 16 can then be skipped).
__host__ __device__ inline int rb_pi(int i) { return 4 * (i & 3) + (i >> 2); }

// valid tile tv -> (group, first slot, first unit)
__device__ __forceinline__ void rb_tile(const VGeo& g, const RbGeo& q, int tv, int& grp, int& sb, int& n0) {
  grp = tv / q.TPGV;
  const int tg = tv - grp * q.TPGV;
  sb = grp * 64 * g.W + 16 * tg;
  n0 = grp * g.Hg + 16 * tg;
}

// weight of unit (slot) into row R = j * KQ + rank of the padded rank space (zero blocks of the group structure included)
__device__ inline float rb_udz(const VGeo& g, const RefP& p, int row, int slot) {
  int n;
  if (!vg_slot_unit(g, slot, n)) return 0.f;
  const int j = row / g.KQ, rr = row - j * g.KQ;
  if (rr >= g.KH) return 0.f;
  const int s = (g.G == 2 && rr >= g.off1) ? 1 : 0;
  const int dest = (n / g.Hg - s + g.G) % g.G;
  return dest == j ? ref_uc(g, p, n, rr) : 0.f;
}
__device__ inline float rb_vc(const VGeo& g, const RefP& p, int slot, int k, int rr) {
  int n;
  if (!vg_slot_unit(g, slot, n) || rr >= g.KH) return 0.f;
  return ref_vc(g, p, n, k, rr);
}

// ---------------------------------------------------------------------------------------------------
// A-operand images.  Element (.., lane) is what lane (c = lane & 15, kq = lane >> 4) feeds to the MFMA: A[i = c][k = kq].
//   UA[tv][m][r]      reduce (fwd):   row pi-position c of M-tile m  x  unit 4 kq + r of the tile
//   VA[tv][k][s4]     expand (fwd):   unit c of the tile, gate k     x  rank 4 s4 + kq
//   VB[tv][k][mv][r]  reduce (bwd):   rank 16 mv + pi(c)             x  (unit 4 kq + r, gate k)
//   UB[tv][m][r]      expand (bwd):   unit c of the tile             x  row 16 m + 4 r + kq of the padded rank space
// ---------------------------------------------------------------------------------------------------
// `flags` (or NULL): the cluster's epoch words of the forward launch behind this one, zeroed here instead of by a memset node of
// their own (a 5 us launch for 1 KB)
__global__ void __launch_bounds__(256) rb_pack_kernel(VGeo g, RbGeo q, RefP p, float* __restrict__ out, unsigned* __restrict__ flags) {
  if (flags != nullptr && blockIdx.x == 0)
    for (int i = threadIdx.x; i < (int)q.flag_words; i += 256) flags[i] = 0u;
  // 32-bit index arithmetic throughout (64-bit divisions by run-time values cost hundreds of cycles each: the first
  // version of this kernel took 290 us at the PTB shape, 0.5 M elements)
  const int NTV = g.G * q.TPGV, KS = g.KH / 4, NP = g.NP, NMT = q.NMT;
  const int nUA = NTV * NMT * 4 * 64, nVA = NTV * 4 * KS * 64, nVB = NTV * 4 * NP * 4 * 64;
  const int total = 2 * nUA + nVA + nVB;
  const int e = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (e >= total) return;
  const int lane = e & 63, c = lane & 15, kq = lane >> 4;
  int le = e;
  int grp, sb, n0;
  if (le < nUA) {                       // UA
    int j = le >> 6;
    const int r = j & 3;
    j >>= 2;
    const int tv = j / NMT, m = j - tv * NMT;
    rb_tile(g, q, tv, grp, sb, n0);
    out[q.UA + le] = rb_udz(g, p, 16 * m + rb_pi(c), sb + 4 * kq + r);
  } else if ((le -= nUA) < nVA) {       // VA
    int j = le >> 6;
    const int j2 = j / KS, s4 = j - j2 * KS;
    const int k = j2 & 3, tv = j2 >> 2;
    rb_tile(g, q, tv, grp, sb, n0);
    out[q.VA + le] = rb_vc(g, p, sb + c, k, 4 * s4 + kq);
  } else if ((le -= nVA) < nVB) {       // VB
    int j = le >> 6;
    const int r = j & 3;
    j >>= 2;
    const int j2 = j / NP, mv = j - j2 * NP;
    const int k = j2 & 3, tv = j2 >> 2;
    rb_tile(g, q, tv, grp, sb, n0);
    out[q.VB + le] = rb_vc(g, p, sb + 4 * kq + r, k, 16 * mv + rb_pi(c));
  } else {                              // UB
    le -= nVB;
    int j = le >> 6;
    const int r = j & 3;
    j >>= 2;
    const int tv = j / NMT, m = j - tv * NMT;
    rb_tile(g, q, tv, grp, sb, n0);
    out[q.UB + le] = rb_udz(g, p, 16 * m + 4 * r + kq, sb + c);
  }
}

// dpre of the slots no tile covers (wholly padded 16-slot tiles behind a group's last unit) must read as zero in the
// batched kernels that contract over slots (dqx = dpre V_x); rb_bwd_kernel never writes them.
// (block 0 also zeroes the backward launch's epoch words: no memset node)
__global__ void __launch_bounds__(256) rb_zero_pad_kernel(VGeo g, RbGeo q, float* __restrict__ dpre, unsigned* __restrict__ flags) {
  if (flags != nullptr && blockIdx.x == 0)
    for (int i = threadIdx.x; i < (int)q.flag_words; i += 256) flags[i] = 0u;
  const size_t row = blockIdx.x;   // (t, b)
  const int per = 64 * g.W - 16 * q.TPGV;   // uncovered slots per group
  for (int i = threadIdx.x; i < g.G * per; i += 256) {
    const int grp = i / per, slot = grp * 64 * g.W + 16 * q.TPGV + (i - grp * per);
    if (g.bf) reinterpret_cast<uint2*>(dpre)[row * g.NT + slot] = make_uint2(0u, 0u);
    else st4(dpre + (row * g.NT + slot) * 4, f4zero());
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// host interface
// ---------------------------------------------------------------------------------------------------
int rb_dispatch_g1a(const VGeo& g, const RbGeo& q, const RbIo& io, bool fwd, hipStream_t s);   // vmlmf_rb_*.hip
int rb_dispatch_g1b(const VGeo& g, const RbGeo& q, const RbIo& io, bool fwd, hipStream_t s);
int rb_dispatch_g2(const VGeo& g, const RbGeo& q, const RbIo& io, bool fwd, hipStream_t s);
int rb_dispatch_g2f(const VGeo& g, const RbGeo& q, const RbIo& io, bool fwd, hipStream_t s);
int rb_dispatch_g2e(const VGeo& g, const RbGeo& q, const RbIo& io, bool fwd, hipStream_t s);
int rb_dispatch_g1a_bf(const VGeo& g, const RbGeo& q, const RbIo& io, bool fwd, hipStream_t s);   // bf16-MFMA variant
int rb_dispatch_g1b_bf(const VGeo& g, const RbGeo& q, const RbIo& io, bool fwd, hipStream_t s);

// the instantiated (contraction steps, tiles per wave, M-tiles per group, flat, groups) combinations
static bool rb_has(int ks, int mt, int nmu, bool flat, int G, bool bf) {
  if (mt < 1 || mt > 4) return false;
  if (bf && G != 1) return false;   // the bf16 variant is instantiated for the plain (one-group) layers
  if (G == 1) return !flat && (ks == 2 || ks == 4 || ks == 6 || ks == 8) && nmu == (ks + 3) / 4;
  if (ks == 16) return flat && mt <= 3 && nmu == 4;
  return (ks == 4 || ks == 8) && nmu == 2;
}

bool rb_geometry(const VGeo& g, int S, RbGeo* out, int rows) {
  RbGeo q;
  memset(&q, 0, sizeof(q));
  if (g.KH % 4 != 0 || g.G > 2 || (RB_WAVES * S) % g.G != 0) return false;
  if (S != 1 && S != 2 && S != 4 && S != 8 && S != 16) return false;   // the cluster sum reads members in chunks of 2 / 4 / 8
  q.S = S;
  q.TPGV = (g.Hg + 15) / 16;
  q.WSG = RB_WAVES * S / g.G;
  q.MT = (q.TPGV + q.WSG - 1) / q.WSG;
  q.NMT = g.G * g.NP;
  // live rows per workgroup: 16, or fewer for a cluster at a small batch - the exchange volume and with it the step time
  // fall with the live rows (config E per GPU of an 8-GPU node, B = 32: 7.5 us per step with 16 rows), as long as all
  // clusters stay co-resident (one workgroup per CU, 256 CUs)
  q.rbl = RB;
  if (rows == 4 || rows == 8 || rows == 16) {
    q.rbl = rows;
  } else if (S > 1) {
    while (q.rbl > 4 && (long long)((g.B + q.rbl / 2 - 1) / (q.rbl / 2)) * S <= 256) q.rbl /= 2;
  }
  q.nrb = (g.B + q.rbl - 1) / q.rbl;
  // M-tiles of the padded rank space the units of group `grp` couple to (vmlmf_geo.h: block s of the rank space feeds
  // destination (grp - s) mod G)
  int nmu = 0;
  for (int grp = 0; grp < g.G; ++grp) {
    int n = 0;
    for (int m = 0; m < q.NMT; ++m) {
      const int j = m / g.NP, lo = (m - j * g.NP) * 16, hi = lo + 16;   // ranks [lo, hi) of vector j
      bool need = false;
      for (int s = 0; s < g.G; ++s) {
        const int b0 = s == 0 ? 0 : g.off1, b1 = (g.G == 2 && s == 0) ? g.off1 : g.KH;   // ranks of block s
        if ((grp - s + g.G) % g.G == j && b0 < hi && lo < b1) need = true;
      }
      if (need) {
        if (n >= 5) return false;
        q.mlist[grp][n++] = m;
      }
    }
    if (grp == 0) nmu = n;
    if (n != nmu) return false;
  }
  q.nmu = nmu;
  // forward reduce: which group's units feed M-tile m.  With two groups every tile has exactly one feeding group, so only
  // that group's members of a cluster hold a non-zero partial of it (the cluster sum then moves half the bytes)
  q.tgcode = 0;
  if (g.G == 2 && q.NMT <= 8) {
    for (int m = 0; m < q.NMT; ++m) {
      int owner = -1, cnt = 0;
      for (int grp = 0; grp < g.G; ++grp)
        for (int u = 0; u < nmu; ++u)
          if (q.mlist[grp][u] == m) owner = grp, ++cnt;
      if (cnt == 1) q.tgcode |= (unsigned)(owner + 1) << (4 * m);
    }
  }
  const long long NTV = (long long)g.G * q.TPGV, KS = g.KH / 4;
  long long o = 0;
  auto take = [&](long long n) { long long r = o; o += (n + 63) / 64 * 64; return r; };
  q.UA = take(NTV * q.NMT * 4 * 64);
  q.VA = take(NTV * 4 * KS * 64);
  q.VB = take(NTV * 4 * g.NP * 4 * 64);
  q.UB = take(NTV * q.NMT * 4 * 64);
  q.total = o;
  q.xq_floats = S > 1 ? (long long)q.nrb * 2 * S * q.NMT * 256 : 0;
  q.flag_words = S > 1 ? (long long)q.nrb * S * 32 + 64 : 0;   // (vmlmf_rb.inc: RB_FLAG_STRIDE)
  if (!rb_has(g.KH / 4, q.MT, q.nmu, g.flat != 0, g.G, g.bf != 0)) return false;
  *out = q;
  return true;
}

int launch_rb_pack(const VGeo& g, const RbGeo& q, const RefP& p, float* img, hipStream_t s, unsigned* zero_flags) {
  if (q.total >= (1LL << 30)) return -3;
  const long long blocks = (q.total + 255) / 256;
  hipLaunchKernelGGL(rb_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g, q, p, img, q.S > 1 ? zero_flags : nullptr);
  return (int)hipGetLastError();
}

static int rb_dispatch(const VGeo& g, const RbGeo& q, const RbIo& io, bool fwd, hipStream_t s) {
  if (g.bf) return g.KH / 4 <= 4 ? rb_dispatch_g1a_bf(g, q, io, fwd, s) : rb_dispatch_g1b_bf(g, q, io, fwd, s);
  if (g.G == 1) return g.KH / 4 <= 4 ? rb_dispatch_g1a(g, q, io, fwd, s) : rb_dispatch_g1b(g, q, io, fwd, s);
  if (g.KH / 4 == 16) return rb_dispatch_g2e(g, q, io, fwd, s);
  return g.flat ? rb_dispatch_g2f(g, q, io, fwd, s) : rb_dispatch_g2(g, q, io, fwd, s);
}

int launch_rb_fwd(const VGeo& g, const RbGeo& q, const RbIo& io, hipStream_t s) {
  if (q.S > 1 && !io.flags_zeroed) {
    const hipError_t e = hipMemsetAsync(io.flag, 0, sizeof(unsigned) * (size_t)q.flag_words, s);
    if (e != hipSuccess) return (int)e;
  }
  return rb_dispatch(g, q, io, true, s);
}

int launch_rb_bwd(const VGeo& g, const RbGeo& q, const RbIo& io, hipStream_t s) {
  const bool pad = 64 * g.W - 16 * q.TPGV > 0;
  if (q.S > 1 && !pad) {
    const hipError_t e = hipMemsetAsync(io.flag, 0, sizeof(unsigned) * (size_t)q.flag_words, s);
    if (e != hipSuccess) return (int)e;
  }
  if (pad) {
    hipLaunchKernelGGL(rb_zero_pad_kernel, dim3((unsigned)(g.T * g.B)), dim3(256), 0, s, g, q, io.dpre, q.S > 1 ? io.flag : nullptr);
    const int rc = (int)hipGetLastError();
    if (rc != 0) return rc;
  }
  return rb_dispatch(g, q, io, false, s);
}
