// Weight-gradient kernels (gfx950): the batched, non-recurrent half of the backward pass.
// Every (t,b) row is independent here, so the whole chip works on it after rec_bwd_kernel has produced
// dpre[t,b,n,k].  Thread <-> hidden unit as in the recurrent kernels.
//
//   dqx_dx_kernel   per row: dqx = dpre V_x (v_fmac_f32_dpp reduce, rows in groups of 8 per barrier),
//                   dx = dqx U_x^T + dpre .* ex.
//   wgrad_mfma_kernel  every weight gradient as A^T B over the rows on fp32 MFMA (see below).
//   reduce_cg_kernel   fixed-order sum of the per-chunk partial products (deterministic, no float atomics).
#include "vmlmf_launch.h"
#include <string.h>

constexpr int RG = 8;  // rows per barrier group in dqx_dx_kernel

template <int KX, int MAXT>
__global__ void __launch_bounds__(MAXT) dqx_dx_kernel(VGeo g, WgxArgs a) {
  constexpr int NPX = (KX + 15) / 16, KQX = NPX * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NT = g.NT, NW = g.NW, W = g.W, B = g.B;
  const int TB = g.T * B;
  const int grp = tid / (64 * W);
  const int m = tid - grp * 64 * W;
  const bool valid = m < g.Hg;
  const int n = grp * g.Hg + (valid ? m : 0);
  const bool has_x = valid && n < g.I;
  const bool wave_x = __ballot(has_x) != 0ull;

  extern __shared__ float4 smem4[];
  float* partx = reinterpret_cast<float*>(smem4);  // [RG][NW][KQX]
  float* dqs = partx + RG * NW * KQX;              // [RG][KQX]

  float vrx[4][KQX], uxo[KX], exi[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int j = 0; j < KQX; ++j) vrx[k][j] = a.VRX[(size_t)(k * KQX + j) * NT + tid];
    exi[k] = a.EXI[k * NT + tid];
  }
#pragma unroll
  for (int r = 0; r < KX; ++r) uxo[r] = a.UXO[(size_t)r * NT + tid];

  const int row_end = (blockIdx.x + 1) * g.RC < TB ? (blockIdx.x + 1) * g.RC : TB;
  for (int row0 = blockIdx.x * g.RC; row0 < row_end; row0 += RG) {
    float4 d[RG];
    if (g.bf) {   // bf16 tape: 8 bytes per slot (wave-uniform branch around the whole batch of loads)
      uint2 raw[RG];
#pragma unroll
      for (int rl = 0; rl < RG; ++rl) {
        const int row = row0 + rl < row_end ? row0 + rl : row_end - 1;
        const int t = row / B, b = row - t * B;
        raw[rl] = reinterpret_cast<const uint2*>(a.dpre)[(size_t)(t * g.Bp + b) * NT + tid];
      }
#pragma unroll
      for (int rl = 0; rl < RG; ++rl)
        d[rl] = make_float4(__uint_as_float(raw[rl].x << 16), __uint_as_float(raw[rl].x & 0xffff0000u),
                            __uint_as_float(raw[rl].y << 16), __uint_as_float(raw[rl].y & 0xffff0000u));
    } else {
#pragma unroll
      for (int rl = 0; rl < RG; ++rl) {
        const int row = row0 + rl < row_end ? row0 + rl : row_end - 1;
        const int t = row / B, b = row - t * B;
        d[rl] = ld4(a.dpre + ((size_t)(t * g.Bp + b) * NT + tid) * 4);  // slot-padded: zeros in pad slots
      }
    }
#pragma unroll
    for (int rl = 0; rl < RG; ++rl) {
      float dp[4] = {d[rl].x, d[rl].y, d[rl].z, d[rl].w};
      dpp_fence(dp[0]);
      dpp_fence(dp[1]);
      dpp_fence(dp[2]);
      dpp_fence(dp[3]);
#pragma unroll
      for (int p = 0; p < NPX; ++p) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        sfor<16>([&](auto K) {
          fmac_ror<K>(acc[0], dp[0], vrx[0][p * 16 + K]);
          fmac_ror<K>(acc[1], dp[1], vrx[1][p * 16 + K]);
          fmac_ror<K>(acc[2], dp[2], vrx[2][p * 16 + K]);
          fmac_ror<K>(acc[3], dp[3], vrx[3][p * 16 + K]);
        });
        const float s = rowsum4((acc[0] + acc[1]) + (acc[2] + acc[3]));
        if (lane < 16) partx[(size_t)(rl * NW + wave) * KQX + p * 16 + lane] = s;
      }
    }
    __syncthreads();
    for (int e = tid; e < RG * KQX; e += NT) {   // finish the cross-wave sum: one thread per (row, rank)
      const int rl = e / KQX, r = e - rl * KQX;
      float s = 0.f;
      for (int w = 0; w < NW; ++w) s += partx[(size_t)(rl * NW + w) * KQX + r];
      dqs[e] = s;
      if (row0 + rl < row_end && r < KX) a.dqx[(size_t)(row0 + rl) * KX + r] = s;
    }
    __syncthreads();
    if (wave_x && a.dx != nullptr) {  // wave-uniform: only waves holding x-units produce dx
#pragma unroll
      for (int rl = 0; rl < RG; ++rl) {
        const int row = row0 + rl;
        float dxv = (d[rl].x * exi[0] + d[rl].y * exi[1]) + (d[rl].z * exi[2] + d[rl].w * exi[3]);
#pragma unroll
        for (int r4 = 0; r4 < KX / 4; ++r4) {
          const float4 q = ld4(dqs + rl * KQX + 4 * r4);
          dxv = fmaf(q.x, uxo[4 * r4 + 0], dxv);
          dxv = fmaf(q.y, uxo[4 * r4 + 1], dxv);
          dxv = fmaf(q.z, uxo[4 * r4 + 2], dxv);
          dxv = fmaf(q.w, uxo[4 * r4 + 3], dxv);
        }
        if (has_x && row < row_end) {
          const int t = row / B, b = row - t * B;
          a.dx[t * g.sxT + b * g.sxB + n] = dxv;
        }
      }
    }
  }
}

#include "vmlmf_atb.inc"

// NBT1 / NBT2: 32-wide tiles of B for mode 1 (KX + G*KH columns) and mode 2 (G*KH columns); mode 3 has one.
// NS: 4 NS waves per workgroup, waves w, w + 4, ... share task w (a part of the rows each): the kernel is bound by
// the load -> MFMA latency of each wave's row loop, so shortening the loop shortens the kernel.
template <int NBT1, int NBT2, int NS>
__device__ __forceinline__ void wgrad_body(const VGeo& g, const AtbArgs& a) {
  extern __shared__ float4 smem4[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = wave & 3, half = wave >> 2;
  const int task = blockIdx.x * 4 + slot;
  const int MT1 = g.NT / 8, MT2 = (g.H + 31) / 32, MT3 = g.foldx ? 0 : (g.I + 31) / 32;
  constexpr int NBTM = NBT1 > NBT2 ? NBT1 : NBT2;
  float* comb = reinterpret_cast<float*>(smem4) + (size_t)slot * (NS - 1) * (16 * NBTM + 3) * 64;
  // every wave of the workgroup takes the same number of barriers: tasks past the end run an empty mode
  if (task < MT1)
    atb_task<1, NBT1, NS>(g, a, task, blockIdx.y, lane, half, comb);
  else if (task < MT1 + MT2)
    atb_task<2, NBT2, NS>(g, a, task - MT1, blockIdx.y, lane, half, comb);
  else if (task < MT1 + MT2 + MT3)
    atb_task<3, 1, NS>(g, a, task - MT1 - MT2, blockIdx.y, lane, half, comb);
  else if (NS > 1)
    __syncthreads();
}
template <int NBT1, int NBT2, int NS>
__global__ void __launch_bounds__(256 * NS) wgrad_mfma_kernel(VGeo g, AtbArgs a) {
  wgrad_body<NBT1, NBT2, NS>(g, a);
}
template <int NBT1, int NBT2, int NS>
__global__ void __launch_bounds__(256 * NS) wgrad_mfma_stack_kernel(AtbStack S) {
  const VGeo& g = vg_karg_ref<VGeo>(offsetof(AtbStack, g) + (size_t)blockIdx.z * sizeof(VGeo));
  const AtbArgs a = atb_stack_args(S);
  wgrad_body<NBT1, NBT2, NS>(g, a);
}

// One thread per element of a chunk's partial block P (coalesced over chunks), fixed-order sum over the
// chunks (deterministic), then scatter into cgrad[accumulator][slot] (layout finish_kernel reads).
// fixed-order sum over the blocks of element e: four interleaved accumulators, eight loads in flight
__device__ __forceinline__ float reduce_cg_sum(const float* __restrict__ Pall, const long long PCH, const long long e, int c, const int c1) {
  return vg_block_sum(Pall, PCH, e, c, c1);   // (vmlmf_device.h: shared with finish2_kernel - the same order, the same bits)
}

__device__ __forceinline__ void reduce_cg_scatter(const VGeo& g, const long long e, const float total, float* __restrict__ cgrad);

// blocks that hold element e: all of them, or (wgrad_ring_kernel: each product has its own chunking) the count of e's region
__device__ __forceinline__ int reduce_cg_count(const VGeo& g, const ReduceCounts& wc, const long long e) {
  if (wc.c[0] == 0) return g.nchunk;
  const int GK = g.G * g.KH, MT2 = (g.H + 31) / 32, MT3 = (g.I + 31) / 32;
  const int NB1p = (vg_nb1(g) + 31) / 32 * 32, NB2p = (GK + 31) / 32 * 32, NB3p = (g.KX + 31) / 32 * 32;
  const long long o2 = (long long)g.NT * 4 * NB1p, o3 = o2 + (long long)MT2 * 32 * NB2p, oe = o3 + (long long)MT3 * 32 * NB3p;
  return e < o2 ? wc.c[0] : (e < o3 ? wc.c[1] : (e < oe ? wc.c[2] : wc.c[0]));
}

__device__ __forceinline__ void reduce_cg_body(const VGeo& g, const float* __restrict__ Pall, float* __restrict__ cgrad,
                                               const ReduceCounts& wc = ReduceCounts{{0, 0, 0}}) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= g.PCH) return;
  reduce_cg_scatter(g, e, reduce_cg_sum(Pall, g.PCH, e, 0, reduce_cg_count(g, wc, e)), cgrad);
}

// The same sum when there are many blocks (one per workgroup of rec4_bwd_kernel: up to the batch size): as one thread per
// element the loop over 256 blocks was a chain of 32 memory round trips (18 us at B = 256).  Here a workgroup takes 32 elements,
// its eight 32-lane groups an eighth of the blocks each, and the eight partial sums meet in LDS in group order - still a fixed
// order for a given block count.
__global__ void __launch_bounds__(256) reduce_cg_many_kernel(VGeo g, const float* __restrict__ Pall, float* __restrict__ cgrad) {
  __shared__ float red[8][32];
  const int li = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const long long e = (long long)blockIdx.x * 32 + li;
  const long long ec = e < g.PCH ? e : g.PCH - 1;
  const int per = (g.nchunk + 7) / 8;
  const int c0 = grp * per < g.nchunk ? grp * per : g.nchunk, c1 = c0 + per < g.nchunk ? c0 + per : g.nchunk;
  red[grp][li] = reduce_cg_sum(Pall, g.PCH, ec, c0, c1);
  __syncthreads();
  if (grp == 0 && e < g.PCH) {
    float total = red[0][li];
#pragma unroll
    for (int q = 1; q < 8; ++q) total += red[q][li];
    reduce_cg_scatter(g, e, total, cgrad);
  }
}

__device__ __forceinline__ void reduce_cg_scatter(const VGeo& g, const long long e, const float total, float* __restrict__ cgrad) {
  const int KX = g.KX, KH = g.KH, GK = g.G * KH, NT = g.NT;
  const int MT2 = (g.H + 31) / 32, MT3 = (g.I + 31) / 32;
  const int NB1p = (vg_nb1(g) + 31) / 32 * 32, NB2p = (GK + 31) / 32 * 32, NB3p = (KX + 31) / 32 * 32;
  const long long o2 = (long long)NT * 4 * NB1p, o3 = o2 + (long long)MT2 * 32 * NB2p,
                  oe = o3 + (long long)MT3 * 32 * NB3p;
  if (e < o2) {                       // C1[(slot,k)][j]
    const int i = (int)(e / NB1p), j = (int)(e - (long long)i * NB1p), slot = i >> 2, k = i & 3;
    if (j < KX) {
      cgrad[(size_t)va_vx(g, k, j) * NT + slot] = total;
    } else if (j < vg_nb1(g)) {
      if (g.flat) {   // both vectors' columns are there; gates i, f pair with vector 0, gates o, n with vector 1
        const int q = (j - KX) / KH, rr = (j - KX) - q * KH;
        if (q == (k >= 2 ? 1 : 0)) cgrad[(size_t)va_vc(g, k, rr) * NT + slot] = total;
      } else {        // only the columns of the slot's own group were computed
        cgrad[(size_t)va_vc(g, k, j - KX) * NT + slot] = total;
      }
    }
  } else if (e < o3) {                // C2[n][dest*KH + rr]
    const long long e2 = e - o2;
    const int n = (int)(e2 / NB2p), j = (int)(e2 - (long long)n * NB2p);
    if (n < g.H && j < GK) {
      const int dest = j / KH, rr = j - dest * KH;
      const int s_ = (g.G == 2 && rr >= g.off1) ? 1 : 0;
      const int grp = n / g.Hg;
      if (dest == (grp - s_ + g.G) % g.G) cgrad[(size_t)va_uc(g, rr) * NT + vg_slot(g, n)] = total;
    }
  } else if (e < oe) {                // C3[m][r]
    const long long e3 = e - o3;
    const int m = (int)(e3 / NB3p), r = (int)(e3 - (long long)m * NB3p);
    // dU_x lives in the slot of unit m; with more inputs than units (cells without vm only) in a block of its own behind the
    // per-slot accumulators
    if (!g.foldx && m < g.I && r < KX)
      cgrad[g.I > g.H ? (size_t)g.NA * NT + (size_t)m * KX + r : (size_t)va_ux(g, r) * NT + vg_slot(g, m)] = total;
  } else {                            // E[which][(slot,k)]
    const long long e4 = e - oe;
    const int which = (int)(e4 / (NT * 4)), i = (int)(e4 - (long long)which * NT * 4), slot = i >> 2, k = i & 3;
    const int acc = which == 0 ? va_eh(g, k) : (which == 1 ? va_ex(g, k) : va_b(g, k));
    cgrad[(size_t)acc * NT + slot] = total;
  }
}

__global__ void __launch_bounds__(256) reduce_cg_kernel(VGeo g, const float* __restrict__ Pall,
                                                        float* __restrict__ cgrad, unsigned* __restrict__ prog, ReduceCounts wc) {
  // after a launch with riding workers: the rows' progress words back to zero (nothing reads them any more)
  if (prog != nullptr && blockIdx.x == 0)
    for (int b = threadIdx.x; b < g.B; b += 256) prog[(size_t)b * WR_PROG_STRIDE] = 0u;
  reduce_cg_body(g, Pall, cgrad, wc);
}
struct ReduceStack {
  VGeo g[WF_MAXL];
  const float* P[WF_MAXL];
  float* cg[WF_MAXL];
  ReduceCounts wc[WF_MAXL];   // (all zero: g.nchunk blocks hold every region - the wavefront stacks; else wgrad_ring_kernel's per-product counts)
};
__global__ void __launch_bounds__(256) reduce_cg_stack_kernel(ReduceStack S) {   // grid.y = layer
  const VGeo& g = vg_karg_ref<VGeo>(offsetof(ReduceStack, g) + (size_t)blockIdx.y * sizeof(VGeo));
  const float* P = vg_karg_ref<const float*>(offsetof(ReduceStack, P) + (size_t)blockIdx.y * sizeof(float*));
  float* cg = vg_karg_ref<float*>(offsetof(ReduceStack, cg) + (size_t)blockIdx.y * sizeof(float*));
  const ReduceCounts wc = vg_karg_ref<ReduceCounts>(offsetof(ReduceStack, wc) + (size_t)blockIdx.y * sizeof(ReduceCounts));
  reduce_cg_body(g, P, cg, wc);
}

// ---------------------------------------------------------------------------------------------------
template <int MAXT>
static int launch_x_t(const VGeo& g, const WgxArgs& a, hipStream_t s) {
  const size_t lds = sizeof(float) * (size_t)(RG * g.NW * g.KQX + RG * g.KQX);
  const dim3 grid(g.nblk), block(g.NT);
  switch (g.KX) {
    case 8: hipLaunchKernelGGL((dqx_dx_kernel<8, MAXT>), grid, block, lds, s, g, a); break;
    case 16: hipLaunchKernelGGL((dqx_dx_kernel<16, MAXT>), grid, block, lds, s, g, a); break;
    case 24: hipLaunchKernelGGL((dqx_dx_kernel<24, MAXT>), grid, block, lds, s, g, a); break;
    case 32: hipLaunchKernelGGL((dqx_dx_kernel<32, MAXT>), grid, block, lds, s, g, a); break;
    default: return -3;
  }
  return (int)hipGetLastError();
}

int launch_wgrad_x(const VGeo& g, const WgxArgs& a, hipStream_t s) {
  if (g.NT <= 256) return launch_x_t<256>(g, a, s);
  if (g.NT <= 512) return launch_x_t<512>(g, a, s);
  return -3;
}

int launch_wgrad_h(const VGeo& g, const WghArgs& w, hipStream_t s) {
  AtbArgs a;
  a.dpre = w.dpre, a.x = w.x, a.y = w.y, a.h0 = w.h0, a.qx = w.qx, a.dqx = w.dqx, a.Qs = w.Qs, a.dQs = w.dQs;
  a.P = w.wpart;
  a.pad0 = 0, a.pad = 0;
  const int tasks = g.NT / 8 + (g.H + 31) / 32 + (g.foldx ? 0 : (g.I + 31) / 32);
  const int GK = g.G * g.KH, n1 = (g.KX + g.KH + 31) / 32, n2 = (GK + 31) / 32;   // mode 1: [qx | own vector] per (slot block, gate) task
  const dim3 grid((tasks + 3) / 4, g.nchunk);
  // two waves per task while the hand-over buffer stays small (four measured slower at the headline shape:
  // 0.1990 vs 0.1966 ms per step)
  if (n1 <= 2 && n2 <= 2) {
    const int nm = n1 > n2 ? n1 : n2;
    const size_t lds = sizeof(float) * 4 * (16 * (size_t)nm + 3) * 64;
    if (n1 == 1 && n2 == 1) hipLaunchKernelGGL((wgrad_mfma_kernel<1, 1, 2>), grid, dim3(512), lds, s, g, a);
    else if (n1 == 1) hipLaunchKernelGGL((wgrad_mfma_kernel<1, 2, 2>), grid, dim3(512), lds, s, g, a);
    else if (n2 == 1) hipLaunchKernelGGL((wgrad_mfma_kernel<2, 1, 2>), grid, dim3(512), lds, s, g, a);
    else hipLaunchKernelGGL((wgrad_mfma_kernel<2, 2, 2>), grid, dim3(512), lds, s, g, a);
    return (int)hipGetLastError();
  }
  const dim3 block(256);
#define WG_CASE(A, Bv) \
  if (n1 == A && n2 == Bv) hipLaunchKernelGGL((wgrad_mfma_kernel<A, Bv, 1>), grid, block, 0, s, g, a)
  // n1 = tiles of KX + (own vector: KH, flat layout: G*KH) (<= 160 columns), n2 = tiles of G*KH (<= 128 columns); with two
  // groups n2 can exceed n1
  WG_CASE(2, 3);
  else WG_CASE(2, 4);
  else WG_CASE(1, 3);
  else WG_CASE(1, 4);
  else WG_CASE(3, 1);
  else WG_CASE(3, 2);
  else WG_CASE(3, 3);
  else WG_CASE(3, 4);
  else WG_CASE(4, 2);
  else WG_CASE(4, 3);
  else WG_CASE(4, 4);
  else WG_CASE(5, 3);
  else WG_CASE(5, 4);
  else return -3;
#undef WG_CASE
  return (int)hipGetLastError();
}

int launch_reduce(const VGeo& g, const float* wpart, float* cgrad, unsigned* prog, hipStream_t s, ReduceCounts wc) {
  if (g.nchunk > 96 && prog == nullptr && wc.c[0] == 0) {
    hipLaunchKernelGGL(reduce_cg_many_kernel, dim3((unsigned)((g.PCH + 31) / 32)), dim3(256), 0, s, g, wpart, cgrad);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(reduce_cg_kernel, dim3((unsigned)((g.PCH + 255) / 256)), dim3(256), 0, s, g, wpart, cgrad, prog, wc);
  return (int)hipGetLastError();
}

// ---- the layers of a stack in one launch each (same ranks, rows and chunking in every layer; the input width may differ)
int launch_wgrad_h_stack(int L, const VGeo* g, const WghArgs* w, hipStream_t s) {
  static_assert(sizeof(AtbStack) <= 4096, "kernel-argument segment");
  AtbStack S;
  memset(&S, 0, sizeof(S));
  int tasks = 0;
  for (int l = 0; l < L; ++l) {
    S.g[l] = g[l];
    AtbArgs& a = S.a[l];
    a.dpre = w[l].dpre, a.x = w[l].x, a.y = w[l].y, a.h0 = w[l].h0, a.qx = w[l].qx, a.dqx = w[l].dqx, a.Qs = w[l].Qs, a.dQs = w[l].dQs;
    a.P = w[l].wpart;
    a.pad0 = 0, a.pad = 0;
    const int t = g[l].NT / 8 + (g[l].H + 31) / 32 + (g[l].foldx ? 0 : (g[l].I + 31) / 32);
    tasks = t > tasks ? t : tasks;
    if (g[l].nchunk != g[0].nchunk || g[l].KX != g[0].KX || g[l].KH != g[0].KH || g[l].G != g[0].G || g[l].flat != g[0].flat) return -3;
  }
  const int GK = g[0].G * g[0].KH, n1 = (vg_nb1(g[0]) + 31) / 32, n2 = (GK + 31) / 32;
  if (n1 > 2 || n2 > 2) return -3;   // (the stacks of the wavefront kernels: ranks <= 32, one group)
  {
    const int rc4 = launch_wgrad4_stack(L, g, w, S, s);   // four interleaved column tiles per wave (vmlmf_wgrad4.hip); -3: not this stack
    if (rc4 != -3) return rc4;
  }
  const dim3 grid((tasks + 3) / 4, g[0].nchunk, L);
  const int nm = n1 > n2 ? n1 : n2;
  const size_t lds = sizeof(float) * 4 * (16 * (size_t)nm + 3) * 64;
  if (n1 == 1 && n2 == 1) hipLaunchKernelGGL((wgrad_mfma_stack_kernel<1, 1, 2>), grid, dim3(512), lds, s, S);
  else if (n1 == 1) hipLaunchKernelGGL((wgrad_mfma_stack_kernel<1, 2, 2>), grid, dim3(512), lds, s, S);
  else if (n2 == 1) hipLaunchKernelGGL((wgrad_mfma_stack_kernel<2, 1, 2>), grid, dim3(512), lds, s, S);
  else hipLaunchKernelGGL((wgrad_mfma_stack_kernel<2, 2, 2>), grid, dim3(512), lds, s, S);
  return (int)hipGetLastError();
}

int launch_reduce_stack(int L, const VGeo* g, const float* const* wpart, float* const* cgrad, hipStream_t s, const ReduceCounts* wc) {
  static_assert(sizeof(ReduceStack) <= 4096, "kernel-argument segment");
  ReduceStack S;
  memset(&S, 0, sizeof(S));
  long long pch = 0;
  for (int l = 0; l < L; ++l) {
    S.g[l] = g[l], S.P[l] = wpart[l], S.cg[l] = cgrad[l];
    if (wc != nullptr) S.wc[l] = wc[l];
    pch = g[l].PCH > pch ? g[l].PCH : pch;
  }
  hipLaunchKernelGGL(reduce_cg_stack_kernel, dim3((unsigned)((pch + 255) / 256), L), dim3(256), 0, s, S);
  return (int)hipGetLastError();
}
