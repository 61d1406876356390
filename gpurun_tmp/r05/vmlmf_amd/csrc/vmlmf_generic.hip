// Step-wise path for layers whose factors do not fit one CU's register file (padded hidden rank > 32 or
// more than 512 thread slots, e.g. BASELINE config E: H = 650, ranks 32 / [32,32]).
//
// The persistent kernels keep U_h/V_h in registers for all T steps; when that is impossible the recurrence is
// run one timestep at a time, cuDNN-style, with the batch as the GEMM M dimension:
//     Q_t   = H_{t-1} Ud            (B x H)(H x G*KH)         gemm_skinny_kernel (fp32 MFMA 16x16x4, K split in-workgroup)
//     P_t   = Q_t Vd                (B x G*KH)(G*KH x 4*slots) gemm_tile_kernel<1> (fp32 MFMA 32x32x2), whose epilogue is
//     gates, c_t, h_t               elementwise                the forward gate math (gates_fwd_one)
// and in reverse
//     dQ_t  = dpre_t VdT            (B x 4*slots)(4*slots x G*KH)   gemm_skinny_kernel
//     dH_{t-1} = dQ_t UdT           (B x G*KH)(G*KH x H)            gemm_rows16_kernel<2>, whose epilogue is
//     dpre_{t-1} from the tape      elementwise                     the gate derivatives (gates_bwd_kernel for step T-1)
// Ud/Vd are the group structure written out densely (zeros where a unit does not feed / read a rank-space
// vector); they are produced by pack_kernel.  The non-recurrent kernels (xproj, wgrad_mfma, reduce, finish)
// are shared with the persistent path; dqx / dx use the same GEMM kernel over all T*B rows.
// Same arithmetic, same tape layout ([t][B][slot]), so the parity tests cover both paths with one oracle.
#include "vmlmf_launch.h"
#include <stdlib.h>
#include <type_traits>
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------
// element-wise halves of a timestep, per (batch row b, thread slot): stand-alone kernels for the first / only step
// and epilogues of the wide GEMMs for the rest (gemm_tile_kernel<EPI>)
struct StepF {
  const float *gx, *P, *EH, *h0, *c0;
  float *y, *hT, *cT, *gates, *cs, *ccar;
  int t;
};

// inputs of one (b, slot) element of the forward gate math; loading is kept apart from computing so that a thread
// with several elements has all of its loads in flight before its first store (the pointers may alias for all
// the compiler knows, so it would not reorder them itself)
struct FwdIn {
  float4 gx4;
  float hp, cp, e0, e1, e2, e3;
  int n;
  bool valid;
};

__device__ __forceinline__ FwdIn gates_fwd_load(const VGeo& g, const StepF& a, int b, int slot) {
  FwdIn v;
  v.valid = vg_slot_unit(g, slot, v.n);
  const int t = a.t, NT = g.NT, H = g.H, n = v.n;
  const size_t so = (size_t)b * NT + slot;
  v.gx4 = ld4(a.gx + ((size_t)t * g.Bp * NT + so) * 4);
  v.hp = 0.f;
  if (v.valid) v.hp = t > 0 ? a.y[(size_t)(t - 1) * g.syT + (size_t)b * g.syB + n] : (a.h0 != nullptr ? a.h0[(size_t)b * H + n] : 0.f);
  if (t == 0)
    v.cp = (v.valid && a.c0 != nullptr) ? a.c0[(size_t)b * H + n] : 0.f;
  else
    v.cp = a.ccar[so];
  v.e0 = a.EH[0 * NT + slot], v.e1 = a.EH[1 * NT + slot], v.e2 = a.EH[2 * NT + slot], v.e3 = a.EH[3 * NT + slot];
  return v;
}

__device__ __forceinline__ void gates_fwd_finish(const VGeo& g, const StepF& a, int b, int slot, const FwdIn& v, float4 p4) {
  const int t = a.t, NT = g.NT, H = g.H, n = v.n;
  const size_t so = (size_t)b * NT + slot;
  const float ig = fast_sigmoid(v.gx4.x + p4.x + v.hp * v.e0);
  const float fg = fast_sigmoid(v.gx4.y + p4.y + v.hp * v.e1);
  const float og = fast_sigmoid(v.gx4.z + p4.z + v.hp * v.e2);
  const float ng = fast_tanh(v.gx4.w + p4.w + v.hp * v.e3);
  const float c = fmaf(fg, v.cp, ig * ng);
  const float h = og * fast_tanh(c);
  a.ccar[so] = c;
  if (v.valid) {
    a.y[(size_t)t * g.syT + (size_t)b * g.syB + n] = h;
    if (t == g.T - 1) {
      if (a.hT != nullptr) a.hT[(size_t)b * H + n] = h;
      if (a.cT != nullptr) a.cT[(size_t)b * H + n] = c;
    }
  }
  if (a.gates != nullptr) {
    const size_t sstride = (size_t)g.Bp * NT;
    st4(a.gates + ((size_t)t * sstride + so) * 4, make_float4(ig, fg, og, ng));
    if (t == 0) a.cs[so] = v.cp;
    a.cs[(size_t)(t + 1) * sstride + so] = c;
  }
}

__device__ __forceinline__ void gates_fwd_one(const VGeo& g, const StepF& a, int b, int slot, float4 p4) {
  gates_fwd_finish(g, a, b, slot, gates_fwd_load(g, a, b, slot), p4);
}

struct StepB {
  const float *gates, *cs, *dy, *EH;
  float *dpre, *dHrec, *ehterm, *dcar;
  int t;
};

struct BwdIn {
  float4 g4;
  float ccur, cprv, eht, dcar, dy;
};

__device__ __forceinline__ BwdIn gates_bwd_load(const VGeo& g, const StepB& a, int b, int slot, bool valid, int n) {
  BwdIn v;
  const int t = a.t, NT = g.NT;
  const size_t so = (size_t)b * NT + slot, sstride = (size_t)g.Bp * NT;
  v.g4 = ld4(a.gates + ((size_t)t * sstride + so) * 4);
  v.ccur = a.cs[(size_t)(t + 1) * sstride + so], v.cprv = a.cs[(size_t)t * sstride + so];
  v.eht = a.ehterm[so], v.dcar = a.dcar[so];
  v.dy = (valid && a.dy != nullptr) ? a.dy[(size_t)t * g.syT + (size_t)b * g.syB + n] : 0.f;
  return v;
}

// dhrec: the recurrent part of dh for this unit (dHrec[b][n]); pad slots (never valid) only keep dpre at zero.
// eh: EH[k][slot], k = 0..3
__device__ __forceinline__ void gates_bwd_finish(const VGeo& g, const StepB& a, int b, int slot, bool valid, const BwdIn& v,
                                                 float dhrec, float4 eh) {
  const int t = a.t, NT = g.NT;
  const size_t so = (size_t)b * NT + slot, sstride = (size_t)g.Bp * NT;
  float dh = v.eht;
  if (valid) {
    dh += dhrec;
    if (a.dy != nullptr) dh += v.dy;
  }
  const float ig = v.g4.x, fg = v.g4.y, og = v.g4.z, ng = v.g4.w;
  const float tc = fast_tanh(v.ccur);
  const float dct = fmaf(dh * og, 1.f - tc * tc, v.dcar);
  float dp0 = dct * ng * ig * (1.f - ig), dp1 = dct * v.cprv * fg * (1.f - fg);
  float dp2 = dh * tc * og * (1.f - og), dp3 = dct * ig * (1.f - ng * ng);
  if (!valid) dp0 = dp1 = dp2 = dp3 = 0.f;
  a.dcar[so] = dct * fg;
  st4(a.dpre + ((size_t)t * sstride + so) * 4, make_float4(dp0, dp1, dp2, dp3));
  a.ehterm[so] = (dp0 * eh.x + dp1 * eh.y) + (dp2 * eh.z + dp3 * eh.w);
}

__device__ __forceinline__ void gates_bwd_one(const VGeo& g, const StepB& a, int b, int slot, bool valid, int n,
                                              float dhrec) {
  const int NT = g.NT;
  const float4 eh = make_float4(a.EH[0 * NT + slot], a.EH[1 * NT + slot], a.EH[2 * NT + slot], a.EH[3 * NT + slot]);
  gates_bwd_finish(g, a, b, slot, valid, gates_bwd_load(g, a, b, slot, valid, n), dhrec, eh);
}

struct EpiArgs {
  VGeo g;
  StepF f;
  StepB b;
};


// C[M x N] = A[M x K] B[K x N], row-major, any sizes (masked).  64 x 64 tile of C per workgroup of four waves
// (each a 32 x 32 sub-tile on v_mfma_f32_32x32x2_f32).  K is staged through LDS 128 at a time: every thread
// issues all of its loads of a stage before the first LDS write, so a stage costs one memory latency, and the 64
// MFMAs of the stage then run back to back (the products of the step-wise path have K = 128 or are split to
// about that).  Skinny products with a long K (Q = H Ud, dQ = dpre VdT: 8 tiles, K = 650 / 3072) split K over
// gridDim.y workgroups: each writes its partial tile (write-through), takes a ticket, and the last one to arrive
// sums the partials in index order (deterministic whatever the arrival order) and resets the ticket.
constexpr int GBM = 64, GBN = 64, GBK = 128, GPAD = 4;
constexpr size_t GEMM_LDS = sizeof(float) * 2 * GBK * (GBM + GPAD);

// Workgroup b runs on XCD b % 8 (observed dispatch order; a speed matter only), each with an L2 of its own.  Logical tile
// ids are handed out so that an XCD gets a contiguous range of them: with ids running along the smaller operand's
// dimension first, the larger operand is then fetched into one L2 instead of all eight.
__device__ __forceinline__ int xcd_tile_id(int bid, int total) {
  return (total & 7) == 0 ? (bid & 7) * (total >> 3) + (bid >> 3) : bid;
}

struct GemmArgs {
  const float* A;
  long long lda;
  const float* B;
  long long ldb;
  float* C;
  long long ldc;
  int M, N, K;
  float* part;   // [gridDim.y][tiles][64 * 64] partial tiles in accumulator order (split K only)
  int* ticket;   // one per tile, zero on entry and on exit
  // K split over gridDim.y workgroups WITHOUT a reduction (gemm_skinny_kernel): workgroup z writes its partial C to
  // C + z * zstride; the consumer adds the copies while it loads them (gemm_rows16_kernel: na copies of A, astride
  // apart), and the launch boundary between the two is all the synchronisation there is
  long long zstride;
  int na;
  long long astride;
  float* Asum;   // gemm_rows16_kernel: where the column-tile-0 workgroups leave A summed over its copies (or nullptr)
};

// EPI 1: C = P_t is not stored; each (row, slot) of the tile goes straight through the forward gate math.
// (The backward counterpart lives in gemm_rows16_kernel: on 64 x 64 tiles dH_rec has 44 of them and they carried all
// of the element-wise work, 19.5 us against 8.9 + 5.2 unfused.)
template <int EPI>
__global__ void __launch_bounds__(256) gemm_tile_kernel(GemmArgs a, EpiArgs e) {
  extern __shared__ float4 gsm4[];
  float(*As)[GBM + GPAD] = reinterpret_cast<float(*)[GBM + GPAD]>(gsm4);               // [k][m]
  float(*Bs)[GBN + GPAD] = reinterpret_cast<float(*)[GBN + GPAD]>(As + GBK);            // [k][n]
  __shared__ int last_flag;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lk = lane >> 5, wm = wave & 1, wn = wave >> 1;
  const int tiles_n = (a.N + GBN - 1) / GBN, tiles_m = (a.M + GBM - 1) / GBM;
  const int lid = xcd_tile_id(blockIdx.x, tiles_m * tiles_n);
  int tm, tn;
  if (a.N > a.M) {   // B is the larger operand: an XCD owns a range of its columns, for every row tile
    tn = lid / tiles_m, tm = lid - tn * tiles_m;
  } else {
    tm = lid / tiles_n, tn = lid - tm * tiles_n;
  }
  const int m0 = tm * GBM, n0 = tn * GBN;
  const int nz = gridDim.y, kz = blockIdx.y;
  const int kper = ((a.K + nz - 1) / nz + 15) / 16 * 16;
  const int k0 = kz * kper, k1 = k0 + kper < a.K ? k0 + kper : a.K;
  // staging roles.  A tile 64 rows x 128 k: thread -> (row, k = j + 4 i), four lanes cover 16 contiguous bytes of
  // a row and their LDS writes land in four different banks.  B tile 128 k x 64 n: thread -> (k = kk + 16 i, 4 n).
  const int ar = tid >> 2, aj = tid & 3;
  const int bk = tid >> 4, bn = (tid & 15) * 4;
  const bool arow_ok = m0 + ar < a.M;
  const float* Ap = a.A + (long long)(arow_ok ? m0 + ar : 0) * a.lda;
  const bool bvec = (a.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.B) & 15) == 0) && (n0 + bn + 3 < a.N);

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int kb = k0; kb < k1; kb += GBK) {
    float ra[GBK / 4];
    float4 rb[GBK / 16];
#pragma unroll
    for (int i = 0; i < GBK / 4; ++i) {
      const int k = kb + aj + 4 * i;
      ra[i] = (arow_ok && k < k1) ? Ap[k] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < GBK / 16; ++i) {
      const int k = kb + bk + 16 * i;
      rb[i] = f4zero();
      if (k < k1) {
        const float* Bp = a.B + (long long)k * a.ldb + n0 + bn;
        if (bvec)
          rb[i] = ld4(Bp);
        else
          rb[i] = make_float4(n0 + bn + 0 < a.N ? Bp[0] : 0.f, n0 + bn + 1 < a.N ? Bp[1] : 0.f,
                              n0 + bn + 2 < a.N ? Bp[2] : 0.f, n0 + bn + 3 < a.N ? Bp[3] : 0.f);
      }
    }
    if (kb != k0) __syncthreads();   // the previous stage's MFMAs have read LDS
#pragma unroll
    for (int i = 0; i < GBK / 4; ++i) As[aj + 4 * i][ar] = ra[i];
#pragma unroll
    for (int i = 0; i < GBK / 16; ++i) *reinterpret_cast<float4*>(&Bs[bk + 16 * i][bn]) = rb[i];
    __syncthreads();
    const int ks = k1 - kb < GBK ? k1 - kb : GBK;   // multiple of 16 except for the tail of K (zero-filled)
    const int steps = (ks + 1) / 2;
#pragma unroll 8
    for (int s = 0; s < steps; ++s) {
      const float av = As[2 * s + lk][32 * wm + li];
      const float bv = Bs[2 * s + lk][32 * wn + li];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
  }
  const int col = n0 + 32 * wn + li;
  const bool cok = col < a.N;
  if (EPI != 0) {   // K is a single-workgroup product here (nz == 1): tile -> LDS -> element-wise continuation
    float(*Ct)[GBN + GPAD] = reinterpret_cast<float(*)[GBN + GPAD]>(gsm4);
    __syncthreads();   // every wave is done reading the staging buffers
#pragma unroll
    for (int r = 0; r < 16; ++r) Ct[32 * wm + (r & 3) + 8 * (r >> 2) + 4 * lk][32 * wn + li] = acc[r];
    __syncthreads();
    if (EPI == 1) {   // thread -> slot (n0/4 + tid % 16) of rows tid/16 + 16 i
      const int sl = tid & 15, slot = (n0 >> 2) + sl;
      if (slot < e.g.NT) {
        FwdIn in[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int b = m0 + (tid >> 4) + 16 * i;
          if (b < a.M) in[i] = gates_fwd_load(e.g, e.f, b, slot);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = (tid >> 4) + 16 * i, b = m0 + row;
          if (b < a.M) gates_fwd_finish(e.g, e.f, b, slot, in[i], *reinterpret_cast<const float4*>(&Ct[row][4 * sl]));
        }
      }
    }
    return;
  }
  if (nz == 1) {
    if (cok) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = m0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (i < a.M) a.C[(long long)i * a.ldc + col] = acc[r];
      }
    }
    return;
  }
  // Split K.  The partial tile goes out in accumulator order (thread-major: 16 consecutive floats per thread, so
  // both the stores here and the loads of the summing workgroup are full 64-byte accesses), written through to
  // agent scope so that a workgroup on another XCD sees it once vmcnt has counted the stores; then a ticket.
  const size_t tile_elems = (size_t)GBM * GBN;
  float* mine = a.part + ((size_t)kz * gridDim.x + blockIdx.x) * tile_elems + (size_t)tid * 16;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    st4g_agent((gf32*)(mine + 4 * q), make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const int t = __hip_atomic_fetch_add(a.ticket + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_flag = t == nz - 1;
  }
  __syncthreads();
  if (!last_flag) return;
  // last arrival: sum the nz partials in index order (agent-scope loads: the lines may never have been in this
  // XCD's L2, but they must not be served from a stale copy either)
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* base = a.part + (size_t)blockIdx.x * tile_elems + (size_t)tid * 16;
#pragma unroll 4
  for (int z = 0; z < nz; ++z) {
    const float* src = base + (size_t)z * gridDim.x * tile_elems;
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = __hip_atomic_load(src + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += v[r];
  }
  if (cok) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = m0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (i < a.M) a.C[(long long)i * a.ldc + col] = acc[r];
    }
  }
  if (tid == 0) __hip_atomic_store(a.ticket + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// C[M x N] = A[M x K] B[K x N] for a skinny C and a long K (Q = H Ud: 256 x 128, K = 650; dQ = dpre VdT: K = 3072).
// With 64 x 64 tiles such a product has eight tiles: either eight CUs grind through K on the slow fp32 MFMA
// (64 cycles per 32x32x2) or K is split across workgroups and pays a cross-XCD hand-over (write-through partials,
// ticket, agent-scope reads: 20-35 us measured).  Here a workgroup owns one 16 x 16 tile of C (128 of them for
// 256 x 128) and its NWV waves split K among themselves: every lane issues all loads of its slice up front (A as
// one 16-byte load per 16 k, B as 64-byte rows shared by 16 lanes), runs its v_mfma_f32_16x16x4_f32 chain from
// registers -- no LDS staging, one memory latency -- and the partial tiles meet in LDS, summed in wave order.
typedef float f32x4v __attribute__((ext_vector_type(4)));

// BT: `B` points at B^T (N x K row-major, ldb its row stride): both operands are then read along k, 16 bytes per lane
// and load (the step-wise path keeps every factor in both orientations); otherwise B is K x N and a lane fetches its
// column element row by row (four 64-byte segments per load instruction).  Reading both along k measured slower
// (rows 12 KB apart): no BT instantiation is built any more.
// NSUB: 16-column sub-tiles per workgroup (2: a 16 x 32 tile whose two MFMA chains share the A operand -- for tall
// products such as dqx = dpre VxT (8960 x 32, K = 3072), where A is the 110 MB operand and should be read once).
// BMODE 0: B is K x N row-major; 1 (BT): `B` points at B^T; 2: B "quad-interleaved" along k, [k / 4][n][k % 4] (ldb = N): the four
// contraction steps a lane feeds from one 16-byte load of A get their B values from one 16-byte load too, sixteen lanes' loads
// contiguous (256 bytes) - with the row-major B every one of them was a dword load of its own, 96 of 108 load instructions of a
// batch (dqx = dpre VxT of the H = 650 layers: 58 -> 50.5 us; batches of 6 or 4 blocks and four instead of
// eight waves measured the same within 4 us: what is left is the 64-byte row segments of A, rows 12 KB apart)
template <int NWV, int BMODE, int SK_CH, int NSUB>
__global__ void __launch_bounds__(NWV * 64) gemm_skinny_kernel(GemmArgs a) {
  constexpr bool BT = BMODE == 1;
  __shared__ float4 red[NSUB][NWV - 1][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  constexpr int TN = 16 * NSUB;
  const int tiles_n = (a.N + TN - 1) / TN, tiles_m = (a.M + 15) / 16;
  const int lid = xcd_tile_id(blockIdx.x, tiles_m * tiles_n);
  int tm, tn;
  if (a.N > a.M) {
    tn = lid / tiles_m, tm = lid - tn * tiles_m;
  } else {
    tm = lid / tiles_n, tn = lid - tm * tiles_n;
  }
  const int m0 = tm * 16, n0 = tn * TN;
  const int nz = gridDim.y, parts = NWV * nz;   // K slices: one per (workgroup z, wave)
  const int nblk = (a.K + 15) / 16, per = (nblk + parts - 1) / parts;
  const int kb0 = ((int)blockIdx.y * NWV + wave) * per * 16;
  const int kb1 = kb0 + per * 16 < a.K ? kb0 + per * 16 : a.K;
  float* const Cz = a.C + (long long)blockIdx.y * a.zstride;
  const bool row_ok = m0 + r < a.M;
  bool col_ok[NSUB];
  const float* Bp[NSUB];
#pragma unroll
  for (int u = 0; u < NSUB; ++u) {
    col_ok[u] = n0 + 16 * u + r < a.N;
    const int c = col_ok[u] ? n0 + 16 * u + r : 0;
    Bp[u] = BT ? a.B + (long long)c * a.ldb : (BMODE == 2 ? a.B + (long long)c * 4 : a.B + c);
  }
  const float* Ap = a.A + (long long)(row_ok ? m0 + r : 0) * a.lda;
  f32x4v acc[NSUB];
#pragma unroll
  for (int u = 0; u < NSUB; ++u) acc[u] = f32x4v{0.f, 0.f, 0.f, 0.f};
  // Operand fetches are branch-free and in two phases: every load of a batch goes out from a clamped (always valid)
  // address, and masking happens where the values are consumed -- with the loads under `if`, or masked right where they
  // are issued, the compiler put a vmcnt(0) wait in front of most of them.  AV: 16-byte loads of whole quads of A (rows
  // 16-byte aligned, K a multiple of four: a quad is then entirely inside or outside the slice), else element loads.
  const bool avec_ok = a.lda % 4 == 0 && (reinterpret_cast<uintptr_t>(a.A) & 15) == 0 && a.K % 4 == 0;
  const bool bvec_ok = BT && a.ldb % 4 == 0 && (reinterpret_cast<uintptr_t>(a.B) & 15) == 0 && a.K % 4 == 0;
  auto run = [&](auto av_c, auto bv_c) {
    constexpr int AV = decltype(av_c)::value;
    constexpr bool BV = decltype(bv_c)::value;
    for (int kb = kb0; kb < kb1; kb += 16 * SK_CH) {
      float av[SK_CH][4], bv[NSUB][SK_CH][4];
#pragma unroll
      for (int c = 0; c < SK_CH; ++c) {
        const int k = kb + 16 * c + 4 * q;
        if (AV == 4) {
          const float4 t = ld4(Ap + (k < kb1 ? k : 0));
          av[c][0] = t.x, av[c][1] = t.y, av[c][2] = t.z, av[c][3] = t.w;
        } else if (AV == 2) {   // rows 8-byte aligned, K even (H = 650): pairs are entirely inside or outside
          const float2 t0 = *reinterpret_cast<const float2*>(Ap + (k < kb1 ? k : 0));
          const float2 t1 = *reinterpret_cast<const float2*>(Ap + (k + 2 < kb1 ? k + 2 : 0));
          av[c][0] = t0.x, av[c][1] = t0.y, av[c][2] = t1.x, av[c][3] = t1.y;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) av[c][j] = Ap[k + j < kb1 ? k + j : 0];
        }
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
          if (BMODE == 2) {   // (K a multiple of four, k too: a quad is entirely inside or outside the slice)
            const float4 t = ld4(Bp[u] + (long long)((k < kb1 ? k : 0) >> 2) * a.ldb * 4);
            bv[u][c][0] = t.x, bv[u][c][1] = t.y, bv[u][c][2] = t.z, bv[u][c][3] = t.w;
          } else if (BT && BV) {
            const float4 t = ld4(Bp[u] + (k < kb1 ? k : 0));
            bv[u][c][0] = t.x, bv[u][c][1] = t.y, bv[u][c][2] = t.z, bv[u][c][3] = t.w;
          } else if (BT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[u][c][j] = Bp[u][k + j < kb1 ? k + j : 0];
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[u][c][j] = Bp[u][(long long)(k + j < kb1 ? k + j : 0) * a.ldb];
          }
        }
      }
#pragma unroll
      for (int c = 0; c < SK_CH; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool in = kb + 16 * c + 4 * q + j < kb1;
          const float av_m = (in && row_ok) ? av[c][j] : 0.f;
#pragma unroll
          for (int u = 0; u < NSUB; ++u)
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_m, (in && col_ok[u]) ? bv[u][c][j] : 0.f, acc[u], 0, 0, 0);
        }
    }
  };
  const bool apair_ok = a.lda % 2 == 0 && (reinterpret_cast<uintptr_t>(a.A) & 7) == 0 && a.K % 2 == 0;
  if (avec_ok && bvec_ok)
    run(std::integral_constant<int, 4>{}, std::true_type{});
  else if (avec_ok)
    run(std::integral_constant<int, 4>{}, std::false_type{});
  else if (apair_ok)
    run(std::integral_constant<int, 2>{}, std::false_type{});
  else
    run(std::integral_constant<int, 1>{}, std::false_type{});
  if (wave > 0) {
#pragma unroll
    for (int u = 0; u < NSUB; ++u) red[u][wave - 1][lane] = make_float4(acc[u][0], acc[u][1], acc[u][2], acc[u][3]);
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int u = 0; u < NSUB; ++u) {
#pragma unroll
      for (int w = 0; w < NWV - 1; ++w) {
        const float4 v = red[u][w][lane];
        acc[u][0] += v.x, acc[u][1] += v.y, acc[u][2] += v.z, acc[u][3] += v.w;
      }
      if (col_ok[u]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = m0 + 4 * q + i;
          if (row < a.M) Cz[(long long)row * a.ldc + n0 + 16 * u + r] = acc[u][i];
        }
      }
    }
  }
}

// C[M x N] = A[M x K] B[K x N] for a short K (the rank space, K = G*KH <= 128) and a wide N, in 16 x 64 tiles: the four
// waves of a workgroup own a 16 x 16 sub-tile each, read their operands straight into MFMA layout (as the skinny
// kernel does) and never meet.  dH_rec = dQ_t UdT (256 x 650, K = 128) is 176 workgroups this way instead of 44 tiles
// of 64 x 64, and the accumulator layout (lane = column, four consecutive rows) is already one (row, unit) element
// per register: with EPI == 2 the gate derivatives of the previous timestep continue from the registers
// (gates_bwd_finish), four elements per lane, all their loads issued before the first store; C is not stored then.
// (The forward product P_t = Q_t Vd with its gate epilogue was tried on these tiles too -- a 4 x 4 quad transpose
// brings a slot's four gates into one lane -- and measured 2 % slower than the 64 x 64 tiles it keeps.)
template <int EPI, int NA>
__global__ void __launch_bounds__(256) gemm_rows16_kernel(GemmArgs a, EpiArgs e) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int tiles_n = (a.N + 63) / 64, tiles_m = (a.M + 15) / 16;
  const int lid = xcd_tile_id(blockIdx.x, tiles_m * tiles_n);
  const int tn = lid / tiles_m, tm = lid - tn * tiles_m;   // an XCD owns a range of B's columns
  const int m0 = tm * 16, col = tn * 64 + 16 * wave + r;
  const bool row_ok = m0 + r < a.M, col_ok = col < a.N;
  const float* Ap = a.A + (long long)(row_ok ? m0 + r : 0) * a.lda;
  const float* Bp = a.B + (col_ok ? col : 0);
  constexpr int CH = 8;   // 16-wide k blocks per batch (K = 128 in one)
  const bool keep_sum = NA > 1 && a.Asum != nullptr && tn == 0 && wave == 0 && row_ok;
  f32x4v acc = {0.f, 0.f, 0.f, 0.f};
  for (int kb = 0; kb < a.K; kb += 16 * CH) {
    // every load of the batch is issued before anything consumes one: the NA copies of A (a K-split producer leaves
    // its partial sums side by side, see GemmArgs) and the B rows
    float av[NA][CH][4], bv[CH][4];
    // branch-free: K is a multiple of four here (the padded rank space) and the rows are 16-byte aligned (checked by the
    // launcher), so a quad of k is entirely inside or outside; outside quads load from k = 0 and are masked afterwards
    float4 a4[NA][CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int k = kb + 16 * c + 4 * q;
      const int kk = k < a.K ? k : 0;
#pragma unroll
      for (int z = 0; z < NA; ++z) a4[z][c] = ld4(Ap + z * a.astride + kk);
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[c][j] = Bp[(long long)(kk + j) * a.ldb];
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const bool in = kb + 16 * c + 4 * q < a.K;
      float4 t = a4[0][c];
#pragma unroll
      for (int z = 1; z < NA; ++z) t = f4add(t, a4[z][c]);   // copy order
      av[0][c][0] = (in && row_ok) ? t.x : 0.f, av[0][c][1] = (in && row_ok) ? t.y : 0.f;
      av[0][c][2] = (in && row_ok) ? t.z : 0.f, av[0][c][3] = (in && row_ok) ? t.w : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float b = (in && col_ok) ? bv[c][j] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][c][j], b, acc, 0, 0, 0);
      }
    }
    if (keep_sum) {   // the summed A tile, for whoever needs it after this launch (dQ_t for the weight gradients)
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int k = kb + 16 * c + 4 * q;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (k + j < a.K) a.Asum[(long long)(m0 + r) * a.lda + k + j] = av[0][c][j];
      }
    }
  }
  if (!col_ok) return;
  if (EPI == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + 4 * q + i;
      if (row < a.M) a.C[(long long)row * a.ldc + col] = acc[i];
    }
  } else {   // column = hidden unit n, rows = batch rows
    const int n = col, slot = vg_slot(e.g, n), NT = e.g.NT;
    const float4 eh = make_float4(e.b.EH[0 * NT + slot], e.b.EH[1 * NT + slot], e.b.EH[2 * NT + slot], e.b.EH[3 * NT + slot]);
    BwdIn in[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int b = m0 + 4 * q + i;
      if (b < a.M) in[i] = gates_bwd_load(e.g, e.b, b, slot, true, n);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int b = m0 + 4 * q + i;
      if (b < a.M) gates_bwd_finish(e.g, e.b, b, slot, true, in[i], acc[i], eh);
    }
  }
}

// The backward pair of a timestep on the step-wise path:
//   launch_dq_split   dQ partials = dpre_t VdT with K split over ZS workgroups per tile (no reduction, see GemmArgs)
//   launch_dhrec      dH_rec = (sum of the partials) UdT on 16 x 64 tiles; leaves the summed dQ_t in dQsum for the weight
//                     gradients; epi 2 continues into the gate derivatives of the previous step
constexpr int DQ_ZS = 2;

static int launch_dq_split(const float* dpre_t, long long lda, const float* VdT, int GK, float* part, int B, int K,
                           hipStream_t s) {
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = dpre_t, a.lda = lda, a.B = VdT, a.ldb = GK, a.C = part, a.ldc = GK, a.M = B, a.N = GK, a.K = K;
  a.zstride = (long long)B * GK;
  const int t16 = ((B + 15) / 16) * ((GK + 15) / 16);
  if (K / DQ_ZS >= 1536)
    hipLaunchKernelGGL((gemm_skinny_kernel<8, 0, 12, 1>), dim3(t16, DQ_ZS), dim3(512), 0, s, a);
  else
    hipLaunchKernelGGL((gemm_skinny_kernel<8, 0, 8, 1>), dim3(t16, DQ_ZS), dim3(512), 0, s, a);
  return (int)hipGetLastError();
}

static int launch_dhrec(const float* part, int GK, const float* UdT, int H, float* dHrec, float* dQsum, int B,
                        const EpiArgs* ea, hipStream_t s);

// split-K scratch of one layer call (GenericBuf::part / ticket): room for GEMM_MAX_SPLIT partial copies of the
// largest skinny product (B x G*KH) and one ticket per tile of it
constexpr int GEMM_MAX_SPLIT = VG_GEMM_SPLIT;
// VMLMF_SKINNY=0 keeps the split-K tiles for those products (A/B measurements)
static const int g_skinny_mode = []() {
  const char* e = getenv("VMLMF_SKINNY");
  return e == nullptr ? 1 : atoi(e);
}();
static const bool g_skinny = g_skinny_mode != 0;
// VMLMF_DQ_SPLIT=0: dQ_t as one product per tile (A/B measurements)
static const bool g_dq_split = []() {
  const char* e = getenv("VMLMF_DQ_SPLIT");
  return (e == nullptr || e[0] != '0') && g_skinny_mode == 1;
}();
// VMLMF_FUSE_GATES=0: element-wise halves of a step as kernels of their own (A/B measurements)
static const int g_fuse_mode = []() {
  const char* e = getenv("VMLMF_FUSE_GATES");
  return e == nullptr ? 1 : atoi(e);
}();
static const bool g_fuse = g_fuse_mode != 0;       // forward: gate math as the epilogue of P_t = Q_t Vd
static const bool g_fuse_bwd = g_fuse_mode != 3 && g_fuse_mode != 0;   // backward: gate derivatives as the epilogue of dH_rec
                                                                       // (3: 64 x 64 tiles and a gates kernel, for A/B runs)

static int launch_dhrec(const float* part, int GK, const float* UdT, int H, float* dHrec, float* dQsum, int B,
                        const EpiArgs* ea, hipStream_t s) {
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = part, a.lda = GK, a.B = UdT, a.ldb = H, a.C = dHrec, a.ldc = H, a.M = B, a.N = H, a.K = GK;
  a.na = DQ_ZS, a.astride = (long long)B * GK, a.Asum = dQsum;
  const int t16 = ((B + 15) / 16) * ((H + 63) / 64);
  static_assert(DQ_ZS == 2, "gemm_rows16_kernel is instantiated for two partial copies");
  if (ea != nullptr) {
    hipLaunchKernelGGL((gemm_rows16_kernel<2, 2>), dim3(t16), dim3(256), 0, s, a, *ea);
  } else {
    EpiArgs none;
    memset(&none, 0, sizeof(none));
    hipLaunchKernelGGL((gemm_rows16_kernel<0, 2>), dim3(t16), dim3(256), 0, s, a, none);
  }
  return (int)hipGetLastError();
}

// Bt / ldbt: the same factor stored transposed (N x K), or nullptr
static int gemm(const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc, int M, int N,
                int K, float* part, long long part_cap, int* ticket, int ticket_cap, hipStream_t s,
                const float* Bt = nullptr, long long ldbt = 0, int epi = 0, const EpiArgs* ea = nullptr, bool bquad = false) {
  GemmArgs a{A, lda, B, ldb, C, ldc, M, N, K, part, ticket};
  static bool raised = false;
  if (!raised) {   // 69 KB of dynamic LDS
    for (const void* f : {reinterpret_cast<const void*>(gemm_tile_kernel<0>), reinterpret_cast<const void*>(gemm_tile_kernel<1>)}) {
      const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_LDS);
      if (e != hipSuccess) return (int)e;
    }
    raised = true;
  }
  if (bquad) {   // B is [k / 4][n][k % 4] (the caller's image; K and lda multiples of four): always the skinny kernel (VMLMF_SKINNY has
                 // no row-major image to fall back to)
    if (N > 128 || K % 4 != 0 || lda % 4 != 0) return -3;
    const int t16 = ((M + 15) / 16) * ((N + 15) / 16);
    if (N % 32 == 0 && t16 >= 1024)
      hipLaunchKernelGGL((gemm_skinny_kernel<8, 2, 12, 2>), dim3(t16 / 2), dim3(512), 0, s, a);
    else
      hipLaunchKernelGGL((gemm_skinny_kernel<8, 2, 12, 1>), dim3(t16), dim3(512), 0, s, a);
    return (int)hipGetLastError();
  }
  if (epi == 2 || epi == 3) {   // 16 x 64 tiles, operands straight into MFMA layout; 2: gate derivatives as the epilogue
    const int t16 = ((M + 15) / 16) * ((N + 63) / 64);
    if (epi == 2) {
      hipLaunchKernelGGL((gemm_rows16_kernel<2, 1>), dim3(t16), dim3(256), 0, s, a, *ea);
    } else {
      EpiArgs none;
      memset(&none, 0, sizeof(none));
      hipLaunchKernelGGL((gemm_rows16_kernel<0, 1>), dim3(t16), dim3(256), 0, s, a, none);
    }
    return (int)hipGetLastError();
  }
  if (epi == 1) {   // fused forward gate math: one workgroup per 64 x 64 tile, whole K (<= a few stages)
    const int tiles = ((M + GBM - 1) / GBM) * ((N + GBN - 1) / GBN);
    hipLaunchKernelGGL(gemm_tile_kernel<1>, dim3(tiles, 1), dim3(256), GEMM_LDS, s, a, *ea);
    return (int)hipGetLastError();
  }
  if (N <= 128 && K >= 256 && g_skinny) {   // skinny output, long K: 16 x 16 tiles, K split inside the workgroup
    const int t16 = ((M + 15) / 16) * ((N + 15) / 16);
    // (B^T with both operands read along k - VMLMF_SKINNY=4 - measured slower, rows 12 KB apart, and left the library in round 5
    //  together with the 16 x 32 tile form of the row-major B, whose one user - dqx - reads the quad image now)
    if (K >= 1536)
      hipLaunchKernelGGL((gemm_skinny_kernel<8, 0, 12, 1>), dim3(t16), dim3(512), 0, s, a);
    else
      hipLaunchKernelGGL((gemm_skinny_kernel<8, 0, 8, 1>), dim3(t16), dim3(512), 0, s, a);
    return (int)hipGetLastError();
  }
  const int tiles = ((M + GBM - 1) / GBM) * ((N + GBN - 1) / GBN);
  int nz = 1;
  if (tiles < 64 && K >= 256 && part != nullptr && tiles <= ticket_cap) {
    nz = (K + 127) / 128;   // one 128-wide stage per workgroup where the scratch allows
    if (nz > GEMM_MAX_SPLIT) nz = GEMM_MAX_SPLIT;
    while (nz > 1 && (long long)nz * tiles * GBM * GBN > part_cap) --nz;
    if (nz < 1) nz = 1;
  }
  EpiArgs none;
  memset(&none, 0, sizeof(none));
  hipLaunchKernelGGL(gemm_tile_kernel<0>, dim3(tiles, nz), dim3(256), GEMM_LDS, s, a, none);
  return (int)hipGetLastError();
}

// one thread per (batch row, thread slot)
__global__ void __launch_bounds__(256) gates_fwd_kernel(VGeo g, StepF a) {
  const int slot = blockIdx.y * 256 + threadIdx.x, b = blockIdx.x;
  if (slot >= g.NT) return;
  gates_fwd_one(g, a, b, slot, ld4(a.P + ((size_t)b * g.NT + slot) * 4));
}

__global__ void __launch_bounds__(256) gates_bwd_kernel(VGeo g, StepB a) {
  const int slot = blockIdx.y * 256 + threadIdx.x, b = blockIdx.x;
  if (slot >= g.NT) return;
  int n;
  const bool valid = vg_slot_unit(g, slot, n);
  gates_bwd_one(g, a, b, slot, valid, n, valid ? a.dHrec[(size_t)b * g.H + n] : 0.f);
}

// dpre of the pad slots (thread slots without a hidden unit) for every timestep: the fused backward epilogue only
// visits real units, and the products that contract over slots must not meet stale workspace contents there
__global__ void __launch_bounds__(256) zero_pad_dpre_kernel(VGeo g, float* __restrict__ dpre) {
  const size_t row = blockIdx.x;   // (t, b)
  for (int slot = threadIdx.x; slot < g.NT; slot += 256) {
    int n;
    if (!vg_slot_unit(g, slot, n)) st4(dpre + (row * g.NT + slot) * 4, f4zero());
  }
}

// mode 0: initialise the backward carries from (dhT, dcT);  mode 1: emit dh0 = dHrec + ehterm, dc0 = dcar
__global__ void __launch_bounds__(256) carry_kernel(VGeo g, int mode, const float* dhT, const float* dcT, float* dHrec,
                                                    float* ehterm, float* dcar, float* dh0, float* dc0) {
  const int slot = blockIdx.y * 256 + threadIdx.x, b = blockIdx.x;
  if (slot >= g.NT) return;
  int n;
  const bool valid = vg_slot_unit(g, slot, n);
  const size_t so = (size_t)b * g.NT + slot;
  if (mode == 0) {
    ehterm[so] = 0.f;
    dcar[so] = (valid && dcT != nullptr) ? dcT[(size_t)b * g.H + n] : 0.f;
    if (valid) dHrec[(size_t)b * g.H + n] = dhT != nullptr ? dhT[(size_t)b * g.H + n] : 0.f;
  } else if (valid) {
    if (dh0 != nullptr) dh0[(size_t)b * g.H + n] = dHrec[(size_t)b * g.H + n] + ehterm[so];
    if (dc0 != nullptr) dc0[(size_t)b * g.H + n] = dcar[so];
  }
}

// dx[row][m] = dqx[row] . ux[m] + sum_k dpre[row][slot(m)][k] * ex[m][k]
// A workgroup owns DXR consecutive (t, b) rows: their dqx vectors sit in LDS (broadcast reads), a thread keeps the U_x
// row of its input m in registers and walks the rows, so U_x is fetched once per DXR rows and the stores run along m.
// (One thread per output with both factors read from memory took 570 us at T*B = 8960, I = 650.)
constexpr int DXR = 16;

template <int KX>
__global__ void __launch_bounds__(256) dx_kernel(VGeo g, const float* __restrict__ dqx, const float* __restrict__ dpre,
                                                 const float* __restrict__ uxp, const float* __restrict__ ext,
                                                 float* __restrict__ dx) {
  __shared__ __attribute__((aligned(16))) float sq[DXR][KX];
  const int row0 = blockIdx.x * DXR, nrows = g.T * g.B;
  for (int e = threadIdx.x; e < DXR * KX; e += 256) {
    const int r = e / KX, row = row0 + r;
    sq[r][e - r * KX] = row < nrows ? dqx[(size_t)row * KX + (e - r * KX)] : 0.f;
  }
  __syncthreads();
  for (int m = threadIdx.x; m < g.I; m += 256) {
    float u[KX];
#pragma unroll
    for (int q = 0; q < KX / 4; ++q) {
      const float4 v = ld4(uxp + (size_t)m * KX + 4 * q);
      u[4 * q] = v.x, u[4 * q + 1] = v.y, u[4 * q + 2] = v.z, u[4 * q + 3] = v.w;
    }
    // inputs beyond the last unit (cells without vm only) have no x .* ex term and no slot
    const int mh = m < g.H ? m : 0;
    const float em = m < g.H ? 1.f : 0.f;
    const float e0 = em * ext[0 * g.H + mh], e1 = em * ext[1 * g.H + mh], e2 = em * ext[2 * g.H + mh], e3 = em * ext[3 * g.H + mh];
    const int slot = vg_slot(g, mh);
    float4 d[DXR];
#pragma unroll
    for (int r = 0; r < DXR; ++r) {
      const int row = row0 + r, t = row / g.B, b = row - t * g.B;
      d[r] = row < nrows ? ld4(dpre + ((size_t)(t * g.Bp + b) * g.NT + slot) * 4) : f4zero();
    }
#pragma unroll
    for (int r = 0; r < DXR; ++r) {
      const int row = row0 + r;
      if (row >= nrows) break;
      float acc = (d[r].x * e0 + d[r].y * e1) + (d[r].z * e2 + d[r].w * e3);
#pragma unroll
      for (int q = 0; q < KX / 4; ++q) {
        const float4 s4 = *reinterpret_cast<const float4*>(&sq[r][4 * q]);
        acc = fmaf(s4.x, u[4 * q], acc);
        acc = fmaf(s4.y, u[4 * q + 1], acc);
        acc = fmaf(s4.z, u[4 * q + 2], acc);
        acc = fmaf(s4.w, u[4 * q + 3], acc);
      }
      const int t = row / g.B, b = row - t * g.B;
      dx[(size_t)t * g.sxT + (size_t)b * g.sxB + m] = acc;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
int generic_forward(const VGeo& g, const GenericBuf& w, hipStream_t s) {
  const int B = g.B, H = g.H, NT = g.NT, GK = g.G * g.KH, T = g.T;
  const dim3 egrid(B, (NT + 255) / 256), eblock(256);
  int rc;
  for (int t = 0; t < T; ++t) {
    const float* A;
    long long lda;
    if (t > 0) {
      A = w.y + (size_t)(t - 1) * g.syT, lda = g.syB;
    } else if (w.h0 != nullptr) {
      A = w.h0, lda = H;
    } else {
      A = w.zeros, lda = H;
    }
    float* Qt = w.Qs != nullptr ? w.Qs + (size_t)t * B * GK : w.Qtmp;
    if ((rc = gemm(A, lda, w.Ud, GK, Qt, GK, B, GK, H, w.part, w.part_cap, w.ticket, w.ticket_cap, s, w.UdT, H)) != 0)
      return rc;
    EpiArgs ea;
    memset(&ea, 0, sizeof(ea));
    ea.g = g;
    StepF& a = ea.f;
    a.gx = w.gx, a.P = w.P, a.EH = w.EH, a.h0 = w.h0, a.c0 = w.c0, a.y = w.y, a.hT = w.hT, a.cT = w.cT;
    a.gates = w.gates, a.cs = w.cs, a.ccar = w.ccar, a.t = t;
    if (g_fuse) {   // P_t = Q_t Vd never reaches memory: the gate math is the epilogue of its tiles
      if ((rc = gemm(Qt, GK, w.Vd, (long long)NT * 4, nullptr, 0, B, NT * 4, GK, nullptr, 0, nullptr, 0, s, nullptr, 0, 1,
                     &ea)) != 0)
        return rc;
      continue;
    }
    if ((rc = gemm(Qt, GK, w.Vd, (long long)NT * 4, w.P, (long long)NT * 4, B, NT * 4, GK, w.part, w.part_cap, w.ticket,
                   w.ticket_cap, s)) != 0)
      return rc;
    hipLaunchKernelGGL(gates_fwd_kernel, egrid, eblock, 0, s, g, a);
    if ((rc = (int)hipGetLastError()) != 0) return rc;
  }
  return 0;
}

int generic_backward(const VGeo& g, const GenericBuf& w, hipStream_t s) {
  const int B = g.B, H = g.H, NT = g.NT, GK = g.G * g.KH, T = g.T;
  const dim3 egrid(B, (NT + 255) / 256), eblock(256);
  int rc;
  hipLaunchKernelGGL(carry_kernel, egrid, eblock, 0, s, g, 0, w.dhT, w.dcT, w.dHrec, w.ehterm, w.dcar,
                     (float*)nullptr, (float*)nullptr);
  if ((rc = (int)hipGetLastError()) != 0) return rc;
  const size_t sstride = (size_t)g.Bp * NT;
  if (g_fuse_bwd && NT > H) {
    hipLaunchKernelGGL(zero_pad_dpre_kernel, dim3((unsigned)(T * g.Bp)), dim3(256), 0, s, g, w.dpre);
    if ((rc = (int)hipGetLastError()) != 0) return rc;
  }
  for (int t = T - 1; t >= 0; --t) {
    EpiArgs ea;
    memset(&ea, 0, sizeof(ea));
    ea.g = g;
    StepB& a = ea.b;
    a.gates = w.gates, a.cs = w.cs, a.dy = w.dy, a.EH = w.EH, a.dpre = w.dpre, a.dHrec = w.dHrec;
    a.ehterm = w.ehterm, a.dcar = w.dcar, a.t = t;
    if (!g_fuse_bwd || t == T - 1) {   // fused: dpre_t of every later step comes out of the previous dH_rec product
      hipLaunchKernelGGL(gates_bwd_kernel, egrid, eblock, 0, s, g, a);
      if ((rc = (int)hipGetLastError()) != 0) return rc;
    }
    float* dQt = w.dQs + (size_t)t * B * GK;
    if (g_dq_split && (long long)DQ_ZS * B * GK <= w.part_cap) {
      // dQ_t in DQ_ZS partial copies (K = 4 * slots split over workgroups, no reduction); the dH_rec product adds them
      // as it loads them and leaves their sum in dQs[t] for the weight gradients
      if ((rc = launch_dq_split(w.dpre + (size_t)t * sstride * 4, (long long)NT * 4, w.VdT, GK, w.part, B, NT * 4, s)) != 0)
        return rc;
      a.t = t - 1;   // the step whose gate derivatives the epilogue computes
      if ((rc = launch_dhrec(w.part, GK, w.UdT, H, w.dHrec, dQt, B, (g_fuse_bwd && t > 0) ? &ea : nullptr, s)) != 0) return rc;
      continue;
    }
    if ((rc = gemm(w.dpre + (size_t)t * sstride * 4, (long long)NT * 4, w.VdT, GK, dQt, GK, B, GK, NT * 4, w.part,
                   w.part_cap, w.ticket, w.ticket_cap, s, w.Vd, (long long)NT * 4)) != 0)
      return rc;
    if (g_fuse_bwd && t > 0) {   // dH_rec = dQ_t UdT continues into the gate derivatives of step t - 1 inside the tiles
      a.t = t - 1;
      if ((rc = gemm(dQt, GK, w.UdT, H, nullptr, 0, B, H, GK, nullptr, 0, nullptr, 0, s, nullptr, 0, 2, &ea)) != 0) return rc;
      continue;
    }
    if ((rc = gemm(dQt, GK, w.UdT, H, w.dHrec, H, B, H, GK, w.part, w.part_cap, w.ticket, w.ticket_cap, s, nullptr, 0,
                   g_fuse_mode == 3 ? 0 : 3, nullptr)) != 0)
      return rc;
  }
  hipLaunchKernelGGL(carry_kernel, egrid, eblock, 0, s, g, 1, (const float*)nullptr, (const float*)nullptr, w.dHrec,
                     w.ehterm, w.dcar, w.dh0, w.dc0);
  if ((rc = (int)hipGetLastError()) != 0) return rc;
  return generic_dqx_dx(g, w, s);
}

int generic_dqx_dx(const VGeo& g, const GenericBuf& w, hipStream_t s) {
  const int B = g.B, NT = g.NT, T = g.T;
  int rc;
  // dqx over all rows, then dx
  // (w.VxT is the quad-interleaved image [slot][r][gate]: pack_kernel)
  if ((rc = gemm(w.dpre, (long long)NT * 4, w.VxT, g.KX, w.dqx, g.KX, T * B, g.KX, NT * 4, w.part, w.part_cap, w.ticket,
                 w.ticket_cap, s, nullptr, 0, 0, nullptr, true)) != 0)
    return rc;
  if (w.dx != nullptr) {
    const dim3 dgrid((unsigned)((T * B + DXR - 1) / DXR));
    switch (g.KX) {
      case 8: hipLaunchKernelGGL(dx_kernel<8>, dgrid, dim3(256), 0, s, g, w.dqx, w.dpre, w.UXP, w.EXT, w.dx); break;
      case 16: hipLaunchKernelGGL(dx_kernel<16>, dgrid, dim3(256), 0, s, g, w.dqx, w.dpre, w.UXP, w.EXT, w.dx); break;
      case 24: hipLaunchKernelGGL(dx_kernel<24>, dgrid, dim3(256), 0, s, g, w.dqx, w.dpre, w.UXP, w.EXT, w.dx); break;
      case 32: hipLaunchKernelGGL(dx_kernel<32>, dgrid, dim3(256), 0, s, g, w.dqx, w.dpre, w.UXP, w.EXT, w.dx); break;
      default: return -3;
    }
    if ((rc = (int)hipGetLastError()) != 0) return rc;
  }
  return 0;
}

// qx = x U_x over all rows of a large time-major layer (rows of x contiguous in (t, b) order): a skinny product with K = the
// input width, on the 16 x 16 MFMA tiles of gemm_skinny_kernel.  (xproj_kernel's own form of it took 50 us at H = 650.)
int generic_qx(const VGeo& g, const float* x, const float* UXP, float* qx, hipStream_t s) {
  return gemm(x, g.I, UXP, g.KX, qx, g.KX, g.T * g.B, g.KX, g.I, nullptr, 0, nullptr, 0, s);
}
