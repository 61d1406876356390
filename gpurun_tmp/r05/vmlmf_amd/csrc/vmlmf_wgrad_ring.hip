// Weight gradients of LARGE layers (gfx950): the same three products as wgrad_mfma_kernel (vmlmf_atb.inc) -
//     mode 1  dpre^T [qx | Q]   (+ the column sums d(eh), d(ex), db)      mode 2  h_{t-1}^T dQ      mode 3  x^T dqx
// - with the operands streamed through a ring of LDS stages by LDS-DMA (global_load_lds), instead of per-lane dword loads
// into registers.  Why: at the PTB layer (H 650, ranks 32 / [32,32], 8960 rows) wgrad_mfma_kernel runs at 0.27 of the fp32
// MFMA rate; every wave fetches its own B rows (the rank-space vectors: the same 1.2 KB per row for all 100 tasks of a chunk)
// and a quarter of each dpre line, 1.3 dword loads per MFMA, and it needs 64 row chunks (154 MB of partial blocks written and
// read back by reduce_cg) to put enough waves in flight to hide their latency.
//
// One workgroup = 8 waves = 256 A columns (mode 1: 64 thread slots x 4 gates, wave w = gate w & 3 of slot block w >> 2; modes
// 2 / 3: eight 32-column tiles) for one chunk of rows.  A stage is 16 rows: the A rows (1 KB each, one 16-byte-per-lane DMA
// instruction per row), the B rows (qx, Q / dQ / dqx, re-strided to a compile-time row pitch), and for mode 1 the h_{t-1} and x values of the
// tile's 64 units.  Four stages (136 KB): while stage s is multiplied, s + 1 .. s + 3 are in flight; every wave waits for ITS
// pieces of stage s with a counted s_waitcnt (the compiler does not see the asm loads), then ONE s_barrier per stage both
// publishes the stage and frees the slot stage s - 1 used.  The MFMA operands are ds_read_b32 in the layout the instruction
// wants (lane l: A[i = l & 31][k = l >> 5]), v_mfma_f32_32x32x2_f32 as before: the same fp32 FMA chains per partial block, only
// fewer, longer blocks.  Chunks per mode are chosen so that all workgroups take about the same time and fill the chip once
// (PTB layer: 15 / 20 / 5 chunks of 12 / 3 / 3 tiles = 255 workgroups); reduce_cg sums each region over its own count.
#include "vmlmf_launch.h"
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

// (the diagnosis builds behind the numbers of docs/design/wgrad_ring.md - no LDS reads, no DMA, constant operands, ...:
//  tools/experiments/ablation_switches.patch, -DWR_ABL=n)
constexpr int WR_KR = 16;       // rows per stage
constexpr int WR_NSTG = 4;      // ring slots
constexpr int WR_AW = 256;      // A columns per workgroup
constexpr int WR_QX = WR_KR * WR_AW;          // float offsets inside a stage: A | qx (or the B rows of modes 2 / 3) | Q | h | x
constexpr int WR_Q = WR_QX + 512;
constexpr int WR_H = WR_Q + 2048;
constexpr int WR_X = WR_H + WR_KR * 64;
constexpr int WR_STAGE = WR_X + WR_KR * 64;   // 8704 floats = 34 KB

struct RingArgs {
  AtbArgs a;
  int nc[3], rc[3], tiles[3], first[3];   // per mode: chunks, rows per chunk (a multiple of 16), 256-column tiles, first workgroup
};

// one LDS-DMA instruction: lane l's 16 (4) bytes at gsrc land at LDS byte address dst + 16 l (4 l); dst is wave-uniform.  M0 is
// written in the statement that reads it (it is compiler-reserved and not preserved).
__device__ __forceinline__ void dma16(const float* gsrc, unsigned dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"((const gf32*)gsrc), "s"(dst) : "memory");
}
__device__ __forceinline__ void dma4(const float* gsrc, unsigned dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"((const gf32*)gsrc), "s"(dst) : "memory");
}
// the same with a wave-uniform base address in SGPRs and a 32-bit per-lane byte offset that is constant over the stages: the
// address arithmetic of a piece is scalar (a wave's vector ALU instructions queue behind the fp32 MFMAs of its SIMD)
__device__ __forceinline__ void dma16s(const void* sbase, unsigned voff, unsigned dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
}
__device__ __forceinline__ void dma4s(const void* sbase, unsigned voff, unsigned dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_stage_and_meet() {   // all but the N youngest DMA pieces of this wave have landed; then the barrier
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// QS: floats between two rows of the Q / dQ stage image (32, 64 or 128: the smallest that holds G * KH).  Every LDS row stride is a
// compile-time constant, so an operand read is one base register + an immediate offset.
template <int MODE, int NB, int QS>
__device__ __forceinline__ void ring_run(const VGeo& g, const AtbArgs& a, const int tile, const int chunk, const int row0,
                                         const int row1, float* smem) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int li = lane & 31, lk = lane >> 5;
  const int B = g.B, H = g.H, KX = g.KX, KH = g.KH, GK = g.G * g.KH, NT4 = g.NT * 4;
  const int TB = g.T * B;
  const unsigned lds0 = (unsigned)(size_t)smem;   // LDS byte address of the ring (the only __shared__ object)
  const bool has_h0 = a.h0 != nullptr;
  const int Bnoh = has_h0 ? 0 : B;                // rows below this have h_{t-1} = 0 (t = 0 without an initial state)

  // ---- this wave's share of a stage's DMA pieces (the same number NJ for every wave: counted waits)
  // mode 1: A rows wave, wave + 8 | h rows wave, wave + 8 | x rows wave, wave + 8 | two pieces of the rank-space rows (Q, then qx)
  // modes 2, 3: A rows 2 wave, 2 wave + 1 as four 64-column pieces each | one piece of the B rows
  constexpr int NJ = MODE == 1 ? 8 : 9;
  constexpr int NQP = QS / 16;                                // pieces (256 floats) of 16 Q rows
  const int s0 = tile * 64;                                   // mode 1: first thread slot of the tile
  const int grp1 = MODE == 1 ? s0 / (64 * g.W) : 0;
  const int m1 = s0 - grp1 * 64 * g.W + lane;                 // this lane's unit inside the group (h, x pieces: 64 units per row)
  const int un1 = grp1 * g.Hg + (m1 < g.Hg ? m1 : g.Hg - 1);  // clamped: the pad slots' sums are dropped at the end
  const int ux1 = un1 < g.I ? un1 : g.I - 1;
  auto hrow = [&](const int r) -> const float* {              // h_{t-1} of row r (t = 0 without h0: any valid row, masked at use)
    return r >= B ? a.y + (size_t)(r - B) * g.syB : (has_h0 ? a.h0 + (size_t)r * H : a.y + (size_t)r * g.syB);
  };
  // per-lane byte offsets of the pieces (constant over the stages; the row of a piece goes into its scalar base):
  // a piece of sixteen rows of a row-major (rows x NC) matrix into an image with row stride ST floats - lane l fetches four floats
  // of row l / (ST / 4) of the piece; columns past NC are clamped (never read back)
  auto rows_voff = [&](const int NC, const int ST) -> unsigned {
    const int c4 = (lane % (ST / 4)) * 4;
    return (unsigned)((lane / (ST / 4)) * NC + (c4 < NC - 4 ? c4 : NC - 4)) * 4u;
  };
  const unsigned vo_q = rows_voff(GK, QS), vo_qx = rows_voff(KX, 32);
  const unsigned vo_h = (unsigned)un1 * 4u, vo_x = (unsigned)ux1 * 4u, vo_a = (unsigned)lane * 16u;
  unsigned vo_a23[4];
#pragma unroll
  for (int pq = 0; pq < 4; ++pq) {
    const int NA = MODE == 2 ? H : g.I, col = tile * WR_AW + pq * 64 + lane;
    vo_a23[pq] = (unsigned)(col < NA ? col : NA - 1) * 4u;
  }
  const int nst_i = (row1 - row0 + WR_KR - 1) / WR_KR;
  // Rows of the caller's tensors (x, y, h0) and of dpre are clamped to the last row; the rank-space buffers (qx, Q, dQ, dqx) are
  // read up to 15 rows past theirs in the last stage of the last chunk (vmlmf_api.hip's layout keeps 16 spare rows behind each) -
  // masked at use either way.
  auto issue = [&](const int st, const int slot) __attribute__((always_inline)) {
    const int r0 = st >= nst_i ? row0 : row0 + st * WR_KR;   // (past the last stage: any rows)
    const unsigned sb = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)slot * (WR_STAGE * 4));
    if constexpr (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int i = wave + 8 * j, r = r0 + i < TB ? r0 + i : TB - 1;
        dma16s(a.dpre + (size_t)r * NT4 + tile * WR_AW, vo_a, sb + (unsigned)(i * WR_AW) * 4);
        dma4s(hrow(r), vo_h, sb + (unsigned)(WR_H + i * 64) * 4);
        dma4s(a.x + (size_t)r * g.sxB, vo_x, sb + (unsigned)(WR_X + i * 64) * 4);
        // rank-space rows: pieces 0 .. NQP - 1 are Q, the next two qx; a wave without a piece of its own repeats piece 0
        int e = wave + 8 * j;
        if (e >= NQP + 2) e = 0;
        if (e < NQP) dma16s(a.Qs + (size_t)(r0 + e * (256 / QS)) * GK, vo_q, sb + (unsigned)(WR_Q + e * 256) * 4);
        else dma16s(a.qx + (size_t)(r0 + (e - NQP) * 8) * KX, vo_qx, sb + (unsigned)(WR_QX + (e - NQP) * 256) * 4);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int i = 2 * wave + j, r = r0 + i < TB ? r0 + i : TB - 1;
        const float* src = MODE == 2 ? hrow(r) : a.x + (size_t)r * g.sxB;
#pragma unroll
        for (int pq = 0; pq < 4; ++pq) dma4s(src, vo_a23[pq], sb + (unsigned)(i * WR_AW + pq * 64) * 4);
      }
      if constexpr (MODE == 2) {
        const int e = wave < NQP ? wave : 0;
        dma16s(a.dQs + (size_t)(r0 + e * (256 / QS)) * GK, vo_q, sb + (unsigned)(WR_Q + e * 256) * 4);
      } else {
        const int e = wave < 2 ? wave : 0;
        dma16s(a.dqx + (size_t)(r0 + e * 8) * KX, vo_qx, sb + (unsigned)(WR_QX + e * 256) * 4);
      }
    }
  };

  // ---- MFMA operand addresses: per operand one lane offset (floats inside a stage); row pair u adds a compile-time constant
  const int gk = wave & 3, sblk = wave >> 2;
  const int acol = MODE == 1 ? (sblk * 32 + li) * 4 + gk : wave * 32 + li;
  const int hcol = sblk * 32 + li;
  // mode 1: this (tile, gate)'s own rank-space vector inside a Q row
  const int qoff = MODE != 1 ? 0 : (g.flat ? (gk >= 2 ? KH : 0) : grp1 * KH);
  const int aoff = lk * WR_AW + acol, hoff = WR_H + lk * 64 + hcol, xoff = WR_X + lk * 64 + hcol;
  int boff[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    if (MODE == 1) {
      const int c = (j - 1) * 32 + li;
      boff[j] = j == 0 ? WR_QX + lk * 32 + (li < KX ? li : KX - 1) : WR_Q + lk * QS + qoff + (c < KH ? c : KH - 1);
    } else if (MODE == 2) {
      const int c = j * 32 + li;
      boff[j] = WR_Q + lk * QS + (c < GK ? c : GK - 1);
    } else {
      boff[j] = WR_QX + lk * 32 + (li < KX ? li : KX - 1);
    }
  }

  f32x16 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float e_h = 0.f, e_x = 0.f, e_b = 0.f;

  auto compute = [&](const float* S, const int r0, auto special) __attribute__((always_inline)) {
    constexpr bool SP = decltype(special)::value;   // a stage with rows past the chunk's end, or rows of t = 0 without h0
    float av[8], hv[8], xv[8], bv[8][NB];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      av[u] = S[aoff + u * 2 * WR_AW];
      if (MODE == 1) {
        hv[u] = S[hoff + u * 128];
        xv[u] = S[xoff + u * 128];
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) bv[u][j] = S[boff[j] + u * 2 * ((MODE == 3 || (MODE == 1 && j == 0)) ? 32 : QS)];
      if constexpr (SP) {
        const int r = r0 + 2 * u + lk;
        if (r >= row1) av[u] = 0.f;
        if (r < Bnoh) {
          if (MODE == 1) hv[u] = 0.f;
          if (MODE == 2) av[u] = 0.f;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u][j], acc[j], 0, 0, 0);
      if (MODE == 1) {
        e_h = fmaf(av[u], hv[u], e_h);
        e_x = fmaf(av[u], xv[u], e_x);
        e_b += av[u];
      }
    }
  };

  // ---- the ring.  Past the last stage the pieces keep being issued (rows clamped, into slots nothing reads any more): every
  // wait then counts the same number of younger pieces.  Three loops, so that the plain stages carry no masks and the
  // accumulators stay in place: stages with rows of t = 0 (no initial state), full stages, the last partial one.
  const int nst = (row1 - row0 + WR_KR - 1) / WR_KR;
  const int nfull = (row1 - row0) / WR_KR;
  int nhead = (MODE == 3 || row0 >= Bnoh) ? 0 : (Bnoh - row0 + WR_KR - 1) / WR_KR;
  nhead = nhead < nst ? nhead : nst;
  auto stage = [&](const int st, auto special) __attribute__((always_inline)) {
    wait_stage_and_meet<NJ * (WR_NSTG - 2)>();
    issue(st + WR_NSTG - 1, (st + WR_NSTG - 1) % WR_NSTG);
    compute(smem + (size_t)(st % WR_NSTG) * WR_STAGE, row0 + st * WR_KR, special);
  };
#pragma unroll
  for (int st = 0; st < WR_NSTG - 1; ++st) issue(st, st);
  int st = 0;
  for (; st < nhead; ++st) stage(st, std::true_type{});
  for (; st < nfull; ++st) stage(st, std::false_type{});
  for (; st < nst; ++st) stage(st, std::true_type{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing of this workgroup's DMA is left in flight when it ends

  // ---- partial block of this chunk: C1 [NT*4][NB1p] | C2 [M2p][NB2p] | C3 [M3p][NB3p] | E [3][NT*4]  (vmlmf_atb.inc)
  const int MT2 = (H + 31) / 32, MT3 = (g.I + 31) / 32;
  const int NB1p = (vg_nb1(g) + 31) / 32 * 32, NB2p = (GK + 31) / 32 * 32, NB3p = (KX + 31) / 32 * 32;
  const size_t o2 = (size_t)NT4 * NB1p, o3 = o2 + (size_t)MT2 * 32 * NB2p, oe = o3 + (size_t)MT3 * 32 * NB3p;
  float* P = a.P + (size_t)chunk * g.PCH;
  float* C = MODE == 1 ? P : (MODE == 2 ? P + o2 : P + o3);
  const int ldc = MODE == 1 ? NB1p : (MODE == 2 ? NB2p : NB3p);
  const int NBC = MODE == 1 ? KX + KH : (MODE == 2 ? GK : KX);        // B columns this wave formed
  const int rows_c = MODE == 1 ? NT4 : (MODE == 2 ? MT2 * 32 : MT3 * 32);
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    // column of lane li in B tile j, and where it sits in the partial row (flat layout: behind qx both vectors have columns)
    const int c = MODE == 1 ? (j == 0 ? li : KX + (j - 1) * 32 + li) : j * 32 + li;
    const bool okc = MODE == 1 ? (j == 0 ? li < KX : (j - 1) * 32 + li < KH) : c < NBC;
    const int cc = (MODE == 1 && g.flat && j > 0) ? c + qoff : c;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * lk;
      const int arow = MODE == 1 ? (s0 + sblk * 32 + i) * 4 + gk : tile * WR_AW + wave * 32 + i;
      if (okc && arow < rows_c) C[(size_t)arow * ldc + cc] = acc[j][r];
    }
  }
  if (MODE == 1) {   // column sums: the two half-waves hold the even / odd rows
    e_h += __shfl_xor(e_h, 32);
    e_x += __shfl_xor(e_x, 32);
    e_b += __shfl_xor(e_b, 32);
    const int mm = s0 - grp1 * 64 * g.W + hcol;          // unit of this lane's slot inside its group
    const bool v1 = mm < g.Hg;
    const bool xm = v1 && grp1 * g.Hg + mm < g.I;
    const int col = (s0 + hcol) * 4 + gk;
    if (lk == 0) {
      P[oe + 0 * (size_t)NT4 + col] = v1 ? e_h : 0.f;
      P[oe + 1 * (size_t)NT4 + col] = xm ? e_x : 0.f;
      P[oe + 2 * (size_t)NT4 + col] = e_b;
    }
  }
}

template <int NB1, int NB2>
__global__ void __launch_bounds__(512) wgrad_ring_kernel(VGeo g, RingArgs q) {
  extern __shared__ float4 smem4[];
  float* smem = reinterpret_cast<float*>(smem4);
  const int b = (int)blockIdx.x;
  const int mode = b >= q.first[2] ? 3 : (b >= q.first[1] ? 2 : 1);
  const int idx = b - q.first[mode - 1];
  const int chunk = idx / q.tiles[mode - 1], tile = idx - chunk * q.tiles[mode - 1];
  const int TB = g.T * g.B;
  const int row0 = chunk * q.rc[mode - 1];
  const int row1 = row0 + q.rc[mode - 1] < TB ? row0 + q.rc[mode - 1] : TB;
  constexpr int QS = NB2 == 1 ? 32 : (NB2 == 2 ? 64 : 128);
  if (mode == 1) ring_run<1, NB1, QS>(g, q.a, tile, chunk, row0, row1, smem);
  else if (mode == 2) ring_run<2, NB2, QS>(g, q.a, tile, chunk, row0, row1, smem);
  else ring_run<3, 1, QS>(g, q.a, tile, chunk, row0, row1, smem);
}

}  // namespace

// Layers this kernel takes: fp32 tapes, one tape row per batch row, contiguous (t, b) rows of x and y, 16-byte aligned rank
// rows, at least one full tile of columns.  (Whether it PAYS - enough rows for a chip-wide launch of long chunks - is the
// caller's question: vmlmf_api.hip.)
bool wgrad_ring_ok(const VGeo& g) {
  return !g.bf && !g.foldx && g.Bp == g.B && g.NT % 64 == 0 && g.NT >= 256 && g.I <= g.H && g.KX % 4 == 0 &&
         (g.G * g.KH) % 4 == 0 && g.sxT == (long long)g.B * g.sxB && g.syT == (long long)g.B * g.syB && g.KX <= 32 && g.G * g.KH <= 128;
}

// chunk counts per mode (nc_out[3] -> launch_reduce's ReduceCounts): workgroups of about equal MFMA time, one per CU
int launch_wgrad_ring(const VGeo& g, const WghArgs& w, int cus, int nc_out[3], hipStream_t s) {
  RingArgs q;
  memset(&q, 0, sizeof(q));
  q.a.dpre = w.dpre, q.a.x = w.x, q.a.y = w.y, q.a.h0 = w.h0, q.a.qx = w.qx, q.a.dqx = w.dqx, q.a.Qs = w.Qs, q.a.dQs = w.dQs;
  q.a.P = w.wpart;
  const int GK = g.G * g.KH, TB = g.T * g.B;
  const int n1 = 1 + (g.KH + 31) / 32, n2 = (GK + 31) / 32;
  q.tiles[0] = g.NT / 64, q.tiles[1] = (g.H + WR_AW - 1) / WR_AW, q.tiles[2] = (g.I + WR_AW - 1) / WR_AW;
  const int cost[3] = {n1, n2, 1};
  double sum = 0;
  for (int m = 0; m < 3; ++m) sum += (double)q.tiles[m] * cost[m];
  const double k = (double)(cus > 0 ? cus : 256) / sum;
  int first = 0;
  for (int m = 0; m < 3; ++m) {
    int nc = (int)(k * cost[m]);                      // rounded down: the launch stays within one workgroup per CU
    const int cap = g.nchunk < TB / (4 * WR_KR) ? g.nchunk : TB / (4 * WR_KR);   // the workspace holds g.nchunk blocks; >= 4 stages per chunk
    nc = nc > cap ? cap : nc;
    nc = nc < 1 ? 1 : nc;
    int rc = ((TB + nc - 1) / nc + WR_KR - 1) / WR_KR * WR_KR;
    nc = (TB + rc - 1) / rc;
    q.nc[m] = nc, q.rc[m] = rc, q.first[m] = first;
    first += nc * q.tiles[m];
    nc_out[m] = nc;
  }
  const size_t lds = sizeof(float) * (size_t)WR_NSTG * WR_STAGE;
#define WR_CASE(A, Bv)                                                                                                   \
  if (n1 == A && n2 == Bv) {                                                                                             \
    /* (set on every launch, as rec4_bwd_launch_kh does: HIP keeps the attribute per device, a process-wide flag would skip   \
       it on the second GPU of a process) */                                                                             \
    if (hipFuncSetAttribute((const void*)wgrad_ring_kernel<A, Bv>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
      (void)hipGetLastError();                                                                                           \
      return -3;                                                                                                         \
    }                                                                                                                    \
    hipLaunchKernelGGL((wgrad_ring_kernel<A, Bv>), dim3(first), dim3(512), lds, s, g, q);                                \
    return (int)hipGetLastError();                                                                                       \
  }
  WR_CASE(2, 1)
  WR_CASE(2, 2)
  WR_CASE(3, 2)
  WR_CASE(3, 3)
  WR_CASE(3, 4)
  WR_CASE(4, 3)
  WR_CASE(4, 4)
  WR_CASE(5, 4)
#undef WR_CASE
  return -3;
}
