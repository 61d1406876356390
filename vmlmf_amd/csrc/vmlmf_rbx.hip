// Clustered layers stacked in one launch per direction (gfx950): the layer loop of the LM network (V/src/models/vmlmf_lm.py:437-439,
// layers :53-174 / :178-280) for layers whose factors need a cluster of workgroups (H = 650), at the batch sizes where one layer's
// clusters leave CUs idle (BASELINE configs[4] per GPU of an 8-GPU node: 32 rows; up to 128).  Today's chained form runs layer 2
// after layer 1 has finished all T steps: 2 T dependent cluster steps per direction.  Here every layer has its own clusters in the
// SAME launch and layer l + 1 follows layer l a few steps behind: T + 3 ticks.
//
// What makes that possible without a second exchange: a layer forms its x side ITSELF, in the same shape as its h side.
//   forward   qx^T (x ranks x rows) = U_x^T . x^T      K = this wave's 16 inputs      -> NPX more tiles of the step's cluster sum
//             pre^T += V_x,k (units x x ranks) . qx^T + x . ex_k + b_k                  (a third MFMA product on the accumulators)
//   backward  dqx^T = sum_k V_x,k^T . dpre_k^T         K = this wave's 16 units       -> NPX more tiles of the cluster sum
//             dx^T  = U_x (inputs x x ranks) . dqx^T + sum_k dpre_k . ex_k
// (input n <-> hidden unit n: these layers have input_size == hidden_size), so a member needs of the layer below only the 16 x 16
// block y[t][rows][units of its own tile] - written by the member with the SAME index of the layer below.  That member's epoch word
// (the one its own cluster polls) tells when: its y[t] stores are issued in the shadow of step t + 1's exchange and drained before
// the publication of step t + 2, i.e. y[t] is complete at epoch t + 3 (a last epoch T + 2 is published behind the loop).  The
// producer writes those rows through to memory (agent scope), the consumer reads them past its caches.  No producer ever waits for a
// consumer.  Every wait is bounded; a wait that gives up poisons the layer's results (NaN) and raises the status word.
//
// The weight-gradient kernels behind the backward launch find what they always find: dpre, the layer's input, y, Q / dQ and - stored
// here by member 0 of every cluster - qx / dqx.  No (T, B, 4H) x-side pre-activation tensor exists on this path.
#include "vmlmf_rb.inc"
#include <cstddef>

namespace {

constexpr unsigned RBX_SPIN = 1u << 22;
constexpr size_t rbx_align8(size_t v) { return (v + 7) & ~(size_t)7; }
// byte offset of the third kernel argument (VGeo, RbGeo, args) in the kernel-argument segment
constexpr size_t RBX_KARG_A = rbx_align8(rbx_align8(sizeof(VGeo)) + sizeof(RbGeo));

// Wait (bounded) until the producer member's epoch word has reached `need`.  `seen` caches the last value read.  false: gave up.
__device__ __forceinline__ bool rbx_wait(const unsigned* pf, unsigned need, unsigned& seen) {
  unsigned spins = 0;
  while (seen < need) {
    seen = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load((const gu32*)pf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (seen >= need) break;
    __builtin_amdgcn_s_sleep(2);
    if (++spins > RBX_SPIN) return false;
  }
  return true;
}
__device__ __forceinline__ float rbx_ld(const float* p) {   // past this CU's caches: a row another workgroup wrote during the launch
  return __hip_atomic_load((const gf32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void rbx_st(float* p, float v) {   // written through: another workgroup reads it during the launch
  __hip_atomic_store((gf32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
template <int KS, int NMU, bool FLAT, int G, int NPX>
__global__ void __launch_bounds__(256) rbx_fwd_kernel(VGeo g, RbGeo q, RbxFwdArgs args) {
  constexpr int NP = (KS + 3) / 4, NMT = G * NP, NQ = FLAT ? 2 * NP : NP, NMTX = NMT + NPX;
  constexpr int NRL = KS - 4 * (NP - 1);   // contraction steps of the last rank tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, kq = lane >> 4;
  const int S = q.S, NT = g.NT, B = g.B, T = g.T, H = g.H;
  const int layer = (int)blockIdx.x / args.bpl, bid = (int)blockIdx.x - layer * args.bpl;
  const RbxLayerF& a = vg_karg_ref<RbxLayerF>(RBX_KARG_A + offsetof(RbxFwdArgs, l) + (size_t)layer * sizeof(RbxLayerF));
  // (row block, member) as in rb_fwd_kernel: the members of a cluster sit 8 blocks apart (one XCD); a layer's blocks are a multiple
  // of 8, so the clusters of one row block in every layer share that XCD
  const int rbi = (bid / (8 * S)) * 8 + (bid & 7), sidx = (bid >> 3) % S;
  if (rbi >= q.nrb) return;
  const int ws = sidx * RB_WAVES + wave, grp = ws / q.WSG, wi = ws - grp * q.WSG;
  const bool on = wi < q.TPGV;          // (a spare wave otherwise: its operands are zero, it stores nothing)
  const int tg = on ? wi : 0, tv = grp * q.TPGV + tg;
  const int sbt = grp * 64 * g.W + 16 * tg;
  const int row = rbi * q.rbl + c;
  const bool rok = c < q.rbl && row < B;
  const int rowc = rok ? row : B - 1;
  const bool train = a.gates != nullptr;
  const size_t sstride = (size_t)B * NT;
  const unsigned* pflag = a.pflag != nullptr ? a.pflag + ((size_t)rbi * S + sidx) * RB_FLAG_STRIDE : nullptr;
  const bool cross = pflag != nullptr;
  const bool pub = a.pub != 0;
  RbXchg X;
  X.xq = a.xq, X.flag = a.flag, X.err = a.flag + (size_t)q.nrb * S * RB_FLAG_STRIDE, X.status = args.status;

  extern __shared__ float4 smem4[];
  float* part = reinterpret_cast<float*>(smem4);        // [2][RB_WAVES][NMTX][64][4]
  float* full = part + 2 * RB_WAVES * NMTX * 256;       // [NMTX][64][4]

  // ---- resident operands: h side as rb_fwd_kernel's, x side in the same pairings
  f32x4v ua[NMU], va[4][NP], uxa[NPX], vxa[4][NPX];
  f32x4v eh[4], exr[4], bbr[4];
  f32x4v h, cst;
  bool uval[4];
  int xun[4];
  const int KSX = g.KX / 4;
#pragma unroll
  for (int u = 0; u < NMU; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = a.img[q.UA + (((size_t)tv * NMT + q.mlist[grp][u]) * 4 + r) * 64 + lane];
      ua[u][r] = on ? v : 0.f;
    }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int mv = 0; mv < NP; ++mv)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int s4 = 4 * mv + r;
        const float v = a.img[q.VA + (((size_t)tv * 4 + k) * KS + (s4 < KS ? s4 : 0)) * 64 + lane];
        va[k][mv][r] = (on && s4 < KS) ? v : 0.f;
      }
#pragma unroll
  for (int mx = 0; mx < NPX; ++mx)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = a.img[q.UXA + (((size_t)tv * NPX + mx) * 4 + r) * 64 + lane];
      uxa[mx][r] = on ? v : 0.f;
    }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int mx = 0; mx < NPX; ++mx)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int s4 = 4 * mx + r;
        const float v = a.img[q.VXA + (((size_t)tv * 4 + k) * KSX + (s4 < KSX ? s4 : 0)) * 64 + lane];
        vxa[k][mx][r] = (on && s4 < KSX) ? v : 0.f;
      }
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int m = 16 * tg + 4 * kq + reg;          // unit index inside the group
    uval[reg] = on && m < g.Hg;
    const int slot = sbt + 4 * kq + reg, n = grp * g.Hg + (m < g.Hg ? m : 0);
    xun[reg] = n;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float e1 = a.EH[k * NT + slot], e2 = a.EXT[k * H + n], e3 = a.BBT[k * H + n];
      eh[k][reg] = on ? e1 : 0.f;
      exr[k][reg] = uval[reg] ? e2 : 0.f;
      bbr[k][reg] = uval[reg] ? e3 : 0.f;
    }
    const bool ld = uval[reg] && rok;
    h[reg] = (ld && a.h0 != nullptr) ? a.h0[(size_t)rowc * H + n] : 0.f;
    cst[reg] = (ld && a.c0 != nullptr) ? a.c0[(size_t)rowc * H + n] : 0.f;
  }
  if (train && on && rok) *reinterpret_cast<f32x4v*>(a.cs + (size_t)row * NT + sbt + 4 * kq) = cst;   // slice 0 = c_{-1}
  // partial slots this wave never writes must read as zero
  for (int i = threadIdx.x; i < 2 * RB_WAVES * NMTX * 64; i += 256) reinterpret_cast<float4*>(part)[i] = f4zero();
  __syncthreads();

  // ---- the layer's input of a step: this lane's four inputs of its row (= units of its tile)
  unsigned seen = 0;
  bool dead = false;
  f32x4v xn = {0.f, 0.f, 0.f, 0.f};
  auto fetch_x = [&](int t) {
    if (t >= T) return;
    if (cross) {
      const unsigned need = (unsigned)(t + 3 < T + 2 ? t + 3 : T + 2);
      if (!dead && !rbx_wait(pflag, need, seen)) {
        dead = true;
        if (lane == 0) {
          __hip_atomic_store((gu32*)X.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          vg_raise(X.status, VMLMF_ST_CLUSTER);
        }
      }
      if (dead) {   // the layer below never got there: this layer's results are NaN from here on, not plausible numbers
        xn = f32x4v{NAN, NAN, NAN, NAN};
        return;
      }
      const float* xr = a.x + (size_t)t * g.sxT + (size_t)rowc * g.sxB;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) xn[reg] = rbx_ld(xr + xun[reg]);
    } else {
      const float* xr = a.x + (size_t)t * g.sxT + (size_t)rowc * g.sxB;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) xn[reg] = xr[xun[reg]];
    }
  };
  fetch_x(0);

  const bool dropping = a.drop.state != nullptr;
  DropKey dkey = {0u, 0u, 0u, 0u};
  if (dropping) dkey = drop_key(a.drop);
  // the tape of a step (activated gates, c_t, h_t = y[t] and its dropped copy), from the registers that hold them until the next
  // step's gates overwrite them: issued inside the NEXT step's exchange (rb_fwd_kernel: DEFER)
  f32x4v pg[4];
  auto store_tape = [&](const int ts) {
    if (on && rok) {
      const size_t so = (size_t)ts * sstride + (size_t)row * NT + sbt + 4 * kq;
      if (train) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) tape_st4<false>(a.gates, so + reg, f32x4v{pg[0][reg], pg[1][reg], pg[2][reg], pg[3][reg]});
        *reinterpret_cast<f32x4v*>(a.cs + so + sstride) = cst;
      }
      const size_t yo = (size_t)ts * g.syT + (size_t)row * g.syB + grp * g.Hg + 16 * tg + 4 * kq;
      float* yt = a.y + yo;
      if (pub && !dropping) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
          if (uval[reg]) rbx_st(yt + reg, h[reg]);
      } else {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
          if (uval[reg]) yt[reg] = h[reg];
      }
      if (dropping) {   // the copy the layer above / the projection reads (vmlmf_lm.py:438-439), the lane's four units in one call
        float f[4];
        drop_factors(dkey, a.drop.thresh, a.drop.scale, (unsigned)(ts * B + row), (unsigned)(sbt >> 2) + kq, f);
        float* yd = a.drop.yd + yo;
        if (pub) {
#pragma unroll
          for (int reg = 0; reg < 4; ++reg)
            if (uval[reg]) rbx_st(yd + reg, h[reg] * f[reg]);
        } else {
#pragma unroll
          for (int reg = 0; reg < 4; ++reg)
            if (uval[reg]) yd[reg] = h[reg] * f[reg];
        }
      }
    }
  };

  for (int t = 0; t < T; ++t) {
    const int buf = t & 1;
    // ---- 1. reduce: this wave's K-partials of the rank-space tiles its units couple to, and of the x ranks
    f32x4v xc;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) xc[reg] = uval[reg] ? xn[reg] : 0.f;
    f32x4v qa[NMU], qxa[NPX];
#pragma unroll
    for (int u = 0; u < NMU; ++u) qa[u] = rb_dot<false>(ua[u], h, f32x4v{0.f, 0.f, 0.f, 0.f});
#pragma unroll
    for (int mx = 0; mx < NPX; ++mx) qxa[mx] = rb_dot<false>(uxa[mx], xc, f32x4v{0.f, 0.f, 0.f, 0.f});
    float* pw = part + ((size_t)(buf * RB_WAVES + wave) * NMTX) * 256;
#pragma unroll
    for (int u = 0; u < NMU; ++u) *reinterpret_cast<f32x4v*>(pw + (size_t)q.mlist[grp][u] * 256 + lane * 4) = qa[u];
#pragma unroll
    for (int mx = 0; mx < NPX; ++mx) *reinterpret_cast<f32x4v*>(pw + (size_t)(NMT + mx) * 256 + lane * 4) = qxa[mx];
    __syncthreads();
    // ---- 2. sum over the waves and over the cluster
    const float* pb = part + (size_t)buf * RB_WAVES * NMTX * 256;
    auto wave_sum = [&](int m) {
      f32x4v s = *reinterpret_cast<const f32x4v*>(pb + (size_t)m * 256 + lane * 4);
#pragma unroll
      for (int w = 1; w < RB_WAVES; ++w) s += *reinterpret_cast<const f32x4v*>(pb + ((size_t)w * NMTX + m) * 256 + lane * 4);
      return s;
    };
    // everything of a step that is not the exchange goes out between this member's publication and its wait for the others: the
    // tape of the step before, and the next step's input (with the wait for the layer below, which is ahead of this one)
    auto others = [&]() {
      if (t > 0) store_tape(t - 1);
      fetch_x(t + 1);
    };
#if defined(RBX_ABL) && (RBX_ABL & 1)
    rb_cluster_sum<NMT>(X, S, rbi, sidx, (unsigned)(t + 1), wave, lane, full, wave_sum, c < q.rbl, q.tgcode, grp, G, others);
#else
    rb_cluster_sum<NMTX>(X, S, rbi, sidx, (unsigned)(t + 1), wave, lane, full, wave_sum, c < q.rbl, q.tgcode, grp, G, others);
#endif
    f32x4v qs[NQ], qxs[NPX];
    const int mq0 = FLAT ? 0 : grp * NP;   // first M-tile of the vector(s) this wave's gates read
#pragma unroll
    for (int i = 0; i < NQ; ++i) qs[i] = *reinterpret_cast<const f32x4v*>(full + (size_t)(mq0 + i) * 256 + lane * 4);
#pragma unroll
    for (int mx = 0; mx < NPX; ++mx) qxs[mx] = *reinterpret_cast<const f32x4v*>(full + (size_t)(NMT + mx) * 256 + lane * 4);
#if defined(RBX_ABL) && (RBX_ABL & 1)
#pragma unroll
    for (int mx = 0; mx < NPX; ++mx) qxs[mx] = qxa[mx];
#endif
    // Q[t] and qx[t] for the weight gradients: wave w of member 0 stores the tiles m = w, w + 4, ...
    if (train && sidx == 0) {
      for (int m = wave; m < NMTX; m += RB_WAVES) {
        const f32x4v s = *reinterpret_cast<const f32x4v*>(full + (size_t)m * 256 + lane * 4);
        if (m < NMT) {
          const int j = m / NP, mv = m - j * NP;
          float* qt = a.Qs + (((size_t)t * B + rowc) * G + j) * g.KH;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rr = 16 * mv + 4 * r + kq;
            if (rok && rr < g.KH) qt[rr] = s[r];
          }
        } else {
          float* qt = a.qx + ((size_t)t * B + rowc) * g.KX;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rr = 16 * (m - NMT) + 4 * r + kq;
            if (rok && rr < g.KX) qt[rr] = s[r];
          }
        }
      }
    }
    // ---- 3. expand + gates
    f32x4v acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) acc[k][reg] = fmaf(h[reg], eh[k][reg], fmaf(xc[reg], exr[k][reg], bbr[k][reg]));
#pragma unroll
    for (int mx = 0; mx < NPX; ++mx)
      if (16 * mx < g.KX) {   // (wave-uniform)
#if defined(RBX_ABL) && (RBX_ABL & 2)
        acc[mx] += qxs[mx];
#else
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = rb_dot<false>(vxa[k][mx], qxs[mx], acc[k]);
#endif
      }
#pragma unroll
    for (int mv = 0; mv < NP; ++mv)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (mv < NP - 1) acc[k] = rb_dot<false>(va[k][mv], qs[(FLAT ? (k >> 1) * NP : 0) + mv], acc[k]);
        else acc[k] = rb_dot<false, NRL>(va[k][mv], qs[(FLAT ? (k >> 1) * NP : 0) + mv], acc[k]);
      }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const float ig = fast_sigmoid(acc[0][reg]), fg = fast_sigmoid(acc[1][reg]), og = fast_sigmoid(acc[2][reg]);
      const float ng = fast_tanh(acc[3][reg]);
      cst[reg] = fmaf(fg, cst[reg], ig * ng);
      h[reg] = og * fast_tanh(cst[reg]);
      pg[0][reg] = ig, pg[1][reg] = fg, pg[2][reg] = og, pg[3][reg] = ng;
    }
  }
  store_tape(T - 1);
  if (on && rok) {
    const size_t o = (size_t)row * H + grp * g.Hg + 16 * tg + 4 * kq;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      if (uval[reg]) {
        if (a.hT != nullptr) a.hT[o + reg] = h[reg];
        if (a.cT != nullptr) a.cT[o + reg] = cst[reg];
      }
    }
  }
  if (pub) {   // the last rows are complete: one more epoch for the layer above
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave == 0 && lane == 0)
      __hip_atomic_store((gu32*)(a.flag + ((size_t)rbi * S + sidx) * RB_FLAG_STRIDE), (unsigned)(T + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------
template <int KS, int NMU, bool FLAT, int G, int NPX>
__global__ void __launch_bounds__(256) rbx_bwd_kernel(VGeo g, RbGeo q, RbxBwdArgs args) {
  constexpr int NP = (KS + 3) / 4, NMT = G * NP, NJ = FLAT ? 2 : 1, NMTX = NMT + NPX;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, kq = lane >> 4;
  const int S = q.S, NT = g.NT, B = g.B, T = g.T, H = g.H;
  const int layer = (int)blockIdx.x / args.bpl, bid = (int)blockIdx.x - layer * args.bpl;
  const RbxLayerB& a = vg_karg_ref<RbxLayerB>(RBX_KARG_A + offsetof(RbxBwdArgs, l) + (size_t)layer * sizeof(RbxLayerB));
  const int rbi = (bid / (8 * S)) * 8 + (bid & 7), sidx = (bid >> 3) % S;
  if (rbi >= q.nrb) return;
  const int ws = sidx * RB_WAVES + wave, grp = ws / q.WSG, wi = ws - grp * q.WSG;
  const bool on = wi < q.TPGV;
  const int tg = on ? wi : 0, tv = grp * q.TPGV + tg;
  const int sbt = grp * 64 * g.W + 16 * tg;
  const int row = rbi * q.rbl + c;
  const bool rok = c < q.rbl && row < B;
  const int rowc = rok ? row : B - 1;
  const bool has_dy = a.dy != nullptr;
  const bool want_dx = a.dx != nullptr;
  const size_t sstride = (size_t)B * NT;
  const unsigned* pflag = a.pflag != nullptr ? a.pflag + ((size_t)rbi * S + sidx) * RB_FLAG_STRIDE : nullptr;
  const bool cross = pflag != nullptr;
  const bool pub = a.pub != 0;
  RbXchg X;
  X.xq = a.xq, X.flag = a.flag, X.err = a.flag + (size_t)q.nrb * S * RB_FLAG_STRIDE, X.status = args.status;

  extern __shared__ float4 smem4[];
  float* part = reinterpret_cast<float*>(smem4);        // [2][RB_WAVES][NMTX][64][4]
  float* full = part + 2 * RB_WAVES * NMTX * 256;

  f32x4v vb[4][NP], ub[NMU], vxb[4][NPX], uxb[NPX];
  f32x4v eh[4], exr[4];
  f32x4v dhrec, dcs, ccur;
  bool uval[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int mv = 0; mv < NP; ++mv)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = a.img[q.VB + ((((size_t)tv * 4 + k) * NP + mv) * 4 + r) * 64 + lane];
        vb[k][mv][r] = on ? v : 0.f;
      }
#pragma unroll
  for (int u = 0; u < NMU; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = a.img[q.UB + (((size_t)tv * NMT + q.mlist[grp][u]) * 4 + r) * 64 + lane];
      ub[u][r] = on ? v : 0.f;
    }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int mx = 0; mx < NPX; ++mx)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = a.img[q.VXB + ((((size_t)tv * 4 + k) * NPX + mx) * 4 + r) * 64 + lane];
        vxb[k][mx][r] = on ? v : 0.f;
      }
#pragma unroll
  for (int mx = 0; mx < NPX; ++mx)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = a.img[q.UXB + (((size_t)tv * NPX + mx) * 4 + r) * 64 + lane];
      uxb[mx][r] = on ? v : 0.f;
    }
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int m = 16 * tg + 4 * kq + reg;
    uval[reg] = on && m < g.Hg;
    const int slot = sbt + 4 * kq + reg, n = grp * g.Hg + (m < g.Hg ? m : 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float e1 = a.EH[k * NT + slot], e2 = a.EXT[k * H + n];
      eh[k][reg] = on ? e1 : 0.f;
      exr[k][reg] = uval[reg] ? e2 : 0.f;
    }
    const bool ld = uval[reg] && rok;
    dhrec[reg] = (ld && a.dhT != nullptr) ? a.dhT[(size_t)rowc * H + n] : 0.f;
    dcs[reg] = (ld && a.dcT != nullptr) ? a.dcT[(size_t)rowc * H + n] : 0.f;
  }
  ccur = *reinterpret_cast<const f32x4v*>(a.cs + (size_t)T * sstride + (size_t)rowc * NT + sbt + 4 * kq);
  for (int i = threadIdx.x; i < 2 * RB_WAVES * NMTX * 64; i += 256) reinterpret_cast<float4*>(part)[i] = f4zero();
  __syncthreads();

  // tape of a step: activated gates, c of the step before, upstream dy (from the layer above when it runs in this launch)
  f32x4v gtn[4], cpn, dyn = {0.f, 0.f, 0.f, 0.f}, dfn = {1.f, 1.f, 1.f, 1.f};
  const bool dropping = a.drop.state != nullptr && has_dy;   // dy is the gradient of the DROPPED copy of y
  DropKey dkey = {0u, 0u, 0u, 0u};
  if (dropping) dkey = drop_key(a.drop);
  unsigned seen = 0;
  bool dead = false;
  auto fetch_tape = [&](int t) {
    if (t < 0) return;
    const size_t so = (size_t)t * sstride + (size_t)rowc * NT + sbt + 4 * kq;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) gtn[reg] = tape_ld4<false>(a.gates, so + reg);
    cpn = *reinterpret_cast<const f32x4v*>(a.cs + so);   // slice t = c_{t-1}
    if (has_dy) {
      const float* dyt = a.dy + (size_t)t * g.syT + (size_t)rowc * g.syB + grp * g.Hg + 16 * tg + 4 * kq;
      if (cross) {
        const int s = T - 1 - t;   // the step of the layer above that formed dx[t]
        const unsigned need = (unsigned)(s + 3 < T + 2 ? s + 3 : T + 2);
        if (!dead && !rbx_wait(pflag, need, seen)) {
          dead = true;
          if (lane == 0) {
            __hip_atomic_store((gu32*)X.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            vg_raise(X.status, VMLMF_ST_CLUSTER);
          }
        }
        if (dead) {
          dyn = f32x4v{NAN, NAN, NAN, NAN};
        } else {
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const float v = rbx_ld(dyt + (uval[reg] ? reg : 0));
            dyn[reg] = uval[reg] ? v : 0.f;
          }
        }
      } else {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const float v = dyt[uval[reg] ? reg : 0];
          dyn[reg] = uval[reg] ? v : 0.f;
        }
      }
      if (dropping) {
        float f[4];
        drop_factors(dkey, a.drop.thresh, a.drop.scale, (unsigned)(t * B + rowc), (unsigned)(sbt >> 2) + kq, f);
        dfn = f32x4v{f[0], f[1], f[2], f[3]};
      }
    }
  };
  fetch_tape(T - 1);

  // dx of the step before (formed behind that step's exchange): stored inside this step's exchange
  f32x4v dxp = {0.f, 0.f, 0.f, 0.f};
  auto store_dx = [&](const int ts) {
    if (want_dx && on && rok) {
      float* dxt = a.dx + (size_t)ts * g.sxT + (size_t)row * g.sxB + grp * g.Hg + 16 * tg + 4 * kq;
      if (pub) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
          if (uval[reg]) rbx_st(dxt + reg, dxp[reg]);
      } else {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
          if (uval[reg]) dxt[reg] = dxp[reg];
      }
    }
  };

  for (int t = T - 1; t >= 0; --t) {
    const int buf = t & 1;
    f32x4v gt[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) gt[reg] = gtn[reg];
    const f32x4v cp = cpn, dyv = dropping ? dyn * dfn : dyn;
    // ---- 1. gate derivatives, K-partials of dQ and of dqx
    f32x4v dp[4], ehterm, exterm;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const float ig = gt[reg][0], fg = gt[reg][1], og = gt[reg][2], ng = gt[reg][3];
      const float tc = fast_tanh(ccur[reg]);
      const float dh = (rok ? dyv[reg] : 0.f) + dhrec[reg];
      const float dct = fmaf(dh, og * (1.f - tc * tc), dcs[reg]);
      dp[0][reg] = dct * (ng * ig * (1.f - ig));
      dp[1][reg] = dct * (cp[reg] * fg * (1.f - fg));
      dp[2][reg] = dh * (tc * og * (1.f - og));
      dp[3][reg] = dct * (ig * (1.f - ng * ng));
      dcs[reg] = dct * fg;
      ehterm[reg] = (dp[0][reg] * eh[0][reg] + dp[1][reg] * eh[1][reg]) + (dp[2][reg] * eh[2][reg] + dp[3][reg] * eh[3][reg]);
      exterm[reg] = (dp[0][reg] * exr[0][reg] + dp[1][reg] * exr[1][reg]) + (dp[2][reg] * exr[2][reg] + dp[3][reg] * exr[3][reg]);
    }
    ccur = cp;
    f32x4v qa[NJ][NP], qxa[NPX];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int mv = 0; mv < NP; ++mv) qa[j][mv] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mx = 0; mx < NPX; ++mx) qxa[mx] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int mv = 0; mv < NP; ++mv) qa[FLAT ? (k >> 1) : 0][mv] = rb_dot<false>(vb[k][mv], dp[k], qa[FLAT ? (k >> 1) : 0][mv]);
#if defined(RBX_ABL) && (RBX_ABL & 2)
      qxa[k & 1] += dp[k];
#else
#pragma unroll
      for (int mx = 0; mx < NPX; ++mx) qxa[mx] = rb_dot<false>(vxb[k][mx], dp[k], qxa[mx]);
#endif
    }
    float* pw = part + ((size_t)(buf * RB_WAVES + wave) * NMTX) * 256;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int mv = 0; mv < NP; ++mv)
        *reinterpret_cast<f32x4v*>(pw + (size_t)((FLAT ? j : grp) * NP + mv) * 256 + lane * 4) = qa[j][mv];
#pragma unroll
    for (int mx = 0; mx < NPX; ++mx) *reinterpret_cast<f32x4v*>(pw + (size_t)(NMT + mx) * 256 + lane * 4) = qxa[mx];
    __syncthreads();
    // ---- 2. sum over waves and cluster; the step's dpre, the dx of the step before and the next step's tape in the exchange's shadow
    const float* pb = part + (size_t)buf * RB_WAVES * NMTX * 256;
    auto wave_sum = [&](int m) {
      f32x4v s = *reinterpret_cast<const f32x4v*>(pb + (size_t)m * 256 + lane * 4);
#pragma unroll
      for (int w = 1; w < RB_WAVES; ++w) s += *reinterpret_cast<const f32x4v*>(pb + ((size_t)w * NMTX + m) * 256 + lane * 4);
      return s;
    };
    auto others = [&]() {
      if (on && rok) {
        const size_t de = (size_t)t * sstride + (size_t)row * NT + sbt + 4 * kq;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) tape_st4<false>(a.dpre, de + reg, f32x4v{dp[0][reg], dp[1][reg], dp[2][reg], dp[3][reg]});
      }
      if (t < T - 1) store_dx(t + 1);
      fetch_tape(t - 1);
    };
#if defined(RBX_ABL) && (RBX_ABL & 1)
    rb_cluster_sum<NMT>(X, S, rbi, sidx, (unsigned)(T - t), wave, lane, full, wave_sum, c < q.rbl, 0u, 0, 1, others);
#else
    rb_cluster_sum<NMTX>(X, S, rbi, sidx, (unsigned)(T - t), wave, lane, full, wave_sum, c < q.rbl, 0u, 0, 1, others);
#endif
    f32x4v qs[NMU], dqs[NPX];
#pragma unroll
    for (int u = 0; u < NMU; ++u) qs[u] = *reinterpret_cast<const f32x4v*>(full + (size_t)q.mlist[grp][u] * 256 + lane * 4);
#pragma unroll
    for (int mx = 0; mx < NPX; ++mx) dqs[mx] = *reinterpret_cast<const f32x4v*>(full + (size_t)(NMT + mx) * 256 + lane * 4);
#if defined(RBX_ABL) && (RBX_ABL & 1)
#pragma unroll
    for (int mx = 0; mx < NPX; ++mx) dqs[mx] = qxa[mx];
#endif
    if (sidx == 0) {   // dQ[t], dqx[t] for the weight gradients
      for (int m = wave; m < NMTX; m += RB_WAVES) {
        const f32x4v s = *reinterpret_cast<const f32x4v*>(full + (size_t)m * 256 + lane * 4);
        if (m < NMT) {
          const int j = m / NP, mv = m - j * NP;
          float* qt = a.dQs + (((size_t)t * B + rowc) * G + j) * g.KH;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rr = 16 * mv + 4 * r + kq;
            if (rok && rr < g.KH) qt[rr] = s[r];
          }
        } else {
          float* qt = a.dqx + ((size_t)t * B + rowc) * g.KX;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rr = 16 * (m - NMT) + 4 * r + kq;
            if (rok && rr < g.KX) qt[rr] = s[r];
          }
        }
      }
    }
    // ---- 3. expand: dh_{t-1} = dQ . U_h + dpre . eh;  dx_t = dqx . U_x + dpre . ex
    f32x4v acc = ehterm;
#pragma unroll
    for (int u = 0; u < NMU; ++u) acc = rb_dot<false>(ub[u], qs[u], acc);
    dhrec = acc;
    if (want_dx) {
      f32x4v ax = exterm;
#pragma unroll
      for (int mx = 0; mx < NPX; ++mx)
        if (16 * mx < g.KX) ax = rb_dot<false>(uxb[mx], dqs[mx], ax);
      dxp = ax;
    }
  }
  store_dx(0);
  if (on && rok) {
    const size_t o = (size_t)row * H + grp * g.Hg + 16 * tg + 4 * kq;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      if (uval[reg]) {
        if (a.dh0 != nullptr) a.dh0[o + reg] = dhrec[reg];
        if (a.dc0 != nullptr) a.dc0[o + reg] = dcs[reg];
      }
    }
  }
  if (pub) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave == 0 && lane == 0)
      __hip_atomic_store((gu32*)(a.flag + ((size_t)rbi * S + sidx) * RB_FLAG_STRIDE), (unsigned)(T + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// dpre of the slots no tile covers must read as zero in the batched kernels behind the backward launch (rb_zero_pad_kernel's job
// for one layer); block (0, l) also clears layer l's epoch words for the backward launch
__global__ void __launch_bounds__(256) rbx_zero_kernel(VGeo g, RbGeo q, RbxZeroArgs z) {
  const int l = blockIdx.y;
  float* dpre = l == 0 ? z.dpre[0] : l == 1 ? z.dpre[1] : l == 2 ? z.dpre[2] : z.dpre[3];
  unsigned* flags = l == 0 ? z.flags[0] : l == 1 ? z.flags[1] : l == 2 ? z.flags[2] : z.flags[3];
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < (int)q.flag_words; i += 256) flags[i] = 0u;
  const int per = 64 * g.W - 16 * q.TPGV;   // uncovered slots per group
  if (per <= 0) return;
  const size_t row = blockIdx.x;   // (t, b)
  for (int i = threadIdx.x; i < g.G * per; i += 256) {
    const int grp = i / per, slot = grp * 64 * g.W + 16 * q.TPGV + (i - grp * per);
    st4(dpre + (row * g.NT + slot) * 4, f4zero());
  }
}

template <int KS, int NMU, bool FLAT, int G, int NPX>
int rbx_launch(const VGeo& g, const RbGeo& q, const void* args, bool fwd, hipStream_t s) {
  constexpr int NP = (KS + 3) / 4, NMTX = G * NP + NPX;
  const size_t lds = sizeof(float) * ((size_t)2 * RB_WAVES * NMTX * 256 + (size_t)NMTX * 256);
  const int bpl = (q.nrb + 7) / 8 * 8 * q.S;
  if (fwd) {
    RbxFwdArgs a = *static_cast<const RbxFwdArgs*>(args);
    a.bpl = bpl;
    auto kern = rbx_fwd_kernel<KS, NMU, FLAT, G, NPX>;
    if (lds > 64 * 1024) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(bpl * a.L)), dim3(256), lds, s, g, q, a);
  } else {
    RbxBwdArgs a = *static_cast<const RbxBwdArgs*>(args);
    a.bpl = bpl;
    auto kern = rbx_bwd_kernel<KS, NMU, FLAT, G, NPX>;
    if (lds > 64 * 1024) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(bpl * a.L)), dim3(256), lds, s, g, q, a);
  }
  return (int)hipGetLastError();
}

// the instantiated layers: the PTB group layer (V4: two groups, ranks 32 + 32 -> KH = 64 per vector, flat) and the plain PTB layer
// (V3: rank 32), both with w_rank 32 (two x-rank tiles)
int rbx_dispatch(const VGeo& g, const RbGeo& q, const void* args, bool fwd, hipStream_t s) {
  const int KS = g.KH / 4;
  if (q.NPX == 2 && g.G == 2 && g.flat && KS == 16 && q.nmu == 4) return rbx_launch<16, 4, true, 2, 2>(g, q, args, fwd, s);
  if (q.NPX == 2 && g.G == 1 && !g.flat && KS == 8 && q.nmu == 2) return rbx_launch<8, 2, false, 1, 2>(g, q, args, fwd, s);
  return -3;
}

}  // namespace

bool rbx_supported(const VGeo& g, const RbGeo& q) {
  if (!q.xf || q.S < 2 || q.MT != 1 || g.bf) return false;
  const int KS = g.KH / 4;
  return q.NPX == 2 && ((g.G == 2 && g.flat && KS == 16 && q.nmu == 4) || (g.G == 1 && !g.flat && KS == 8 && q.nmu == 2));
}

int launch_rbx_fwd(const VGeo& g, const RbGeo& q, const RbxFwdArgs& a, hipStream_t s) { return rbx_dispatch(g, q, &a, true, s); }
int launch_rbx_bwd(const VGeo& g, const RbGeo& q, const RbxBwdArgs& a, hipStream_t s) { return rbx_dispatch(g, q, &a, false, s); }

int launch_rbx_zero(const VGeo& g, const RbGeo& q, int L, float* const* dpre, unsigned* const* flags, hipStream_t s) {
  RbxZeroArgs z;
  memset(&z, 0, sizeof(z));
  for (int l = 0; l < L; ++l) z.dpre[l] = dpre[l], z.flags[l] = flags[l];
  const bool pad = 64 * g.W - 16 * q.TPGV > 0;
  hipLaunchKernelGGL(rbx_zero_kernel, dim3(pad ? (unsigned)(g.T * g.B) : 1u, (unsigned)L), dim3(256), 0, s, g, q, z);
  return (int)hipGetLastError();
}
