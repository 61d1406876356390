// Step-wise path for layers whose factors do not fit one CU's register file (padded hidden rank > 32 or
// more than 512 thread slots, e.g. BASELINE config E: H = 650, ranks 32 / [32,32]).
//
// The persistent kernels keep U_h/V_h in registers for all T steps; when that is impossible the recurrence is
// run one timestep at a time, cuDNN-style, with the batch as the GEMM M dimension:
//     Q_t   = H_{t-1} Ud            (B x H)(H x G*KH)         gemm_tile_kernel (fp32 MFMA 32x32x2)
//     P_t   = Q_t Vd                (B x G*KH)(G*KH x 4*slots) gemm_tile_kernel
//     gates, c_t, h_t               elementwise                gates_fwd_kernel
// and in reverse
//     dpre_t from the tape          elementwise                gates_bwd_kernel
//     dQ_t  = dpre_t VdT            (B x 4*slots)(4*slots x G*KH)
//     dH_{t-1} = dQ_t UdT           (B x G*KH)(G*KH x H)
// Ud/Vd are the group structure written out densely (zeros where a unit does not feed / read a rank-space
// vector); they are produced by pack_kernel.  The non-recurrent kernels (xproj, wgrad_mfma, reduce, finish)
// are shared with the persistent path; dqx / dx use the same GEMM kernel over all T*B rows.
// Same arithmetic, same tape layout ([t][B][slot]), so the parity tests cover both paths with one oracle.
#include "vmlmf_launch.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// C[M x N] = A[M x K] B[K x N], row-major, any sizes (masked).  64 x 64 tile of C per workgroup of four waves
// (each a 32 x 32 sub-tile on v_mfma_f32_32x32x2_f32).  K is staged through LDS 128 at a time: every thread
// issues all of its loads of a stage before the first LDS write, so a stage costs one memory latency, and the 64
// MFMAs of the stage then run back to back (the products of the step-wise path have K = 128 or are split to
// about that).  Skinny products with a long K (Q = H Ud, dQ = dpre VdT: 8 tiles, K = 650 / 3072) split K over
// gridDim.y workgroups: each writes its partial tile (write-through), takes a ticket, and the last one to arrive
// sums the partials in index order (deterministic whatever the arrival order) and resets the ticket.
constexpr int GBM = 64, GBN = 64, GBK = 128, GPAD = 4;
constexpr size_t GEMM_LDS = sizeof(float) * 2 * GBK * (GBM + GPAD);

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
};

__global__ void __launch_bounds__(256) gemm_tile_kernel(GemmArgs a) {
  extern __shared__ float4 gsm4[];
  float(*As)[GBM + GPAD] = reinterpret_cast<float(*)[GBM + GPAD]>(gsm4);               // [k][m]
  float(*Bs)[GBN + GPAD] = reinterpret_cast<float(*)[GBN + GPAD]>(As + GBK);            // [k][n]
  __shared__ int last_flag;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lk = lane >> 5, wm = wave & 1, wn = wave >> 1;
  const int tiles_n = (a.N + GBN - 1) / GBN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
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

// split-K scratch of one layer call (GenericBuf::part / ticket): room for GEMM_MAX_SPLIT partial copies of the
// largest skinny product (B x G*KH) and one ticket per tile of it
constexpr int GEMM_MAX_SPLIT = VG_GEMM_SPLIT;

static int gemm(const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc, int M, int N,
                int K, float* part, long long part_cap, int* ticket, int ticket_cap, hipStream_t s) {
  GemmArgs a{A, lda, B, ldb, C, ldc, M, N, K, part, ticket};
  const int tiles = ((M + GBM - 1) / GBM) * ((N + GBN - 1) / GBN);
  int nz = 1;
  if (tiles < 64 && K >= 256 && part != nullptr && tiles <= ticket_cap) {
    nz = (K + 127) / 128;   // one 128-wide stage per workgroup where the scratch allows
    if (nz > GEMM_MAX_SPLIT) nz = GEMM_MAX_SPLIT;
    while (nz > 1 && (long long)nz * tiles * GBM * GBN > part_cap) --nz;
    if (nz < 1) nz = 1;
  }
  static bool raised = false;
  if (!raised) {   // 69 KB of dynamic LDS
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tile_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_LDS);
    if (e != hipSuccess) return (int)e;
    raised = true;
  }
  hipLaunchKernelGGL(gemm_tile_kernel, dim3(tiles, nz), dim3(256), GEMM_LDS, s, a);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
struct StepF {
  const float *gx, *P, *EH, *h0, *c0;
  float *y, *hT, *cT, *gates, *cs, *ccar;
  int t;
};

// one thread per (batch row, thread slot)
__global__ void __launch_bounds__(256) gates_fwd_kernel(VGeo g, StepF a) {
  const int slot = blockIdx.y * 256 + threadIdx.x, b = blockIdx.x;
  if (slot >= g.NT) return;
  int n;
  const bool valid = vg_slot_unit(g, slot, n);
  const int t = a.t, NT = g.NT, H = g.H;
  const size_t so = (size_t)b * NT + slot;
  const float4 gx4 = ld4(a.gx + ((size_t)t * g.Bp * NT + so) * 4);
  const float4 p4 = ld4(a.P + so * 4);
  float hp = 0.f;
  if (valid) hp = t > 0 ? a.y[(size_t)(t - 1) * g.syT + (size_t)b * g.syB + n] : (a.h0 != nullptr ? a.h0[(size_t)b * H + n] : 0.f);
  float cp;
  if (t == 0)
    cp = (valid && a.c0 != nullptr) ? a.c0[(size_t)b * H + n] : 0.f;
  else
    cp = a.ccar[so];
  const float e0 = a.EH[0 * NT + slot], e1 = a.EH[1 * NT + slot], e2 = a.EH[2 * NT + slot], e3 = a.EH[3 * NT + slot];
  const float ig = fast_sigmoid(gx4.x + p4.x + hp * e0);
  const float fg = fast_sigmoid(gx4.y + p4.y + hp * e1);
  const float og = fast_sigmoid(gx4.z + p4.z + hp * e2);
  const float ng = fast_tanh(gx4.w + p4.w + hp * e3);
  const float c = fmaf(fg, cp, ig * ng);
  const float h = og * fast_tanh(c);
  a.ccar[so] = c;
  if (valid) {
    a.y[(size_t)t * g.syT + (size_t)b * g.syB + n] = h;
    if (t == g.T - 1) {
      if (a.hT != nullptr) a.hT[(size_t)b * H + n] = h;
      if (a.cT != nullptr) a.cT[(size_t)b * H + n] = c;
    }
  }
  if (a.gates != nullptr) {
    const size_t sstride = (size_t)g.Bp * NT;
    st4(a.gates + ((size_t)t * sstride + so) * 4, make_float4(ig, fg, og, ng));
    if (t == 0) a.cs[so] = cp;
    a.cs[(size_t)(t + 1) * sstride + so] = c;
  }
}

struct StepB {
  const float *gates, *cs, *dy, *EH;
  float *dpre, *dHrec, *ehterm, *dcar;
  int t;
};

__global__ void __launch_bounds__(256) gates_bwd_kernel(VGeo g, StepB a) {
  const int slot = blockIdx.y * 256 + threadIdx.x, b = blockIdx.x;
  if (slot >= g.NT) return;
  int n;
  const bool valid = vg_slot_unit(g, slot, n);
  const int t = a.t, NT = g.NT, H = g.H;
  const size_t so = (size_t)b * NT + slot, sstride = (size_t)g.Bp * NT;
  const float4 g4 = ld4(a.gates + ((size_t)t * sstride + so) * 4);
  const float ccur = a.cs[(size_t)(t + 1) * sstride + so], cprv = a.cs[(size_t)t * sstride + so];
  float dh = a.ehterm[so];
  if (valid) {
    dh += a.dHrec[(size_t)b * H + n];
    if (a.dy != nullptr) dh += a.dy[(size_t)t * g.syT + (size_t)b * g.syB + n];
  }
  const float ig = g4.x, fg = g4.y, og = g4.z, ng = g4.w;
  const float tc = fast_tanh(ccur);
  const float dct = fmaf(dh * og, 1.f - tc * tc, a.dcar[so]);
  float dp0 = dct * ng * ig * (1.f - ig), dp1 = dct * cprv * fg * (1.f - fg);
  float dp2 = dh * tc * og * (1.f - og), dp3 = dct * ig * (1.f - ng * ng);
  if (!valid) dp0 = dp1 = dp2 = dp3 = 0.f;
  a.dcar[so] = dct * fg;
  st4(a.dpre + ((size_t)t * sstride + so) * 4, make_float4(dp0, dp1, dp2, dp3));
  a.ehterm[so] = (dp0 * a.EH[0 * NT + slot] + dp1 * a.EH[1 * NT + slot]) +
                 (dp2 * a.EH[2 * NT + slot] + dp3 * a.EH[3 * NT + slot]);
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
__global__ void __launch_bounds__(256) dx_kernel(VGeo g, const float* __restrict__ dqx, const float* __restrict__ dpre,
                                                 const float* __restrict__ uxp, const float* __restrict__ ext,
                                                 float* __restrict__ dx) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)g.T * g.B * g.I;
  if (e >= total) return;
  const int row = (int)(e / g.I), m = (int)(e - (long long)row * g.I);
  const int t = row / g.B, b = row - t * g.B;
  const float4 d = ld4(dpre + ((size_t)(t * g.Bp + b) * g.NT + vg_slot(g, m)) * 4);
  float acc = (d.x * ext[0 * g.H + m] + d.y * ext[1 * g.H + m]) + (d.z * ext[2 * g.H + m] + d.w * ext[3 * g.H + m]);
  for (int r = 0; r < g.KX; ++r) acc = fmaf(dqx[(size_t)row * g.KX + r], uxp[(size_t)m * g.KX + r], acc);
  dx[(size_t)t * g.sxT + (size_t)b * g.sxB + m] = acc;
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
    if ((rc = gemm(A, lda, w.Ud, GK, Qt, GK, B, GK, H, w.part, w.part_cap, w.ticket, w.ticket_cap, s)) != 0) return rc;
    if ((rc = gemm(Qt, GK, w.Vd, (long long)NT * 4, w.P, (long long)NT * 4, B, NT * 4, GK, w.part, w.part_cap, w.ticket,
                   w.ticket_cap, s)) != 0)
      return rc;
    StepF a;
    a.gx = w.gx, a.P = w.P, a.EH = w.EH, a.h0 = w.h0, a.c0 = w.c0, a.y = w.y, a.hT = w.hT, a.cT = w.cT;
    a.gates = w.gates, a.cs = w.cs, a.ccar = w.ccar, a.t = t;
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
  for (int t = T - 1; t >= 0; --t) {
    StepB a;
    a.gates = w.gates, a.cs = w.cs, a.dy = w.dy, a.EH = w.EH, a.dpre = w.dpre, a.dHrec = w.dHrec;
    a.ehterm = w.ehterm, a.dcar = w.dcar, a.t = t;
    hipLaunchKernelGGL(gates_bwd_kernel, egrid, eblock, 0, s, g, a);
    if ((rc = (int)hipGetLastError()) != 0) return rc;
    float* dQt = w.dQs + (size_t)t * B * GK;
    if ((rc = gemm(w.dpre + (size_t)t * sstride * 4, (long long)NT * 4, w.VdT, GK, dQt, GK, B, GK, NT * 4, w.part,
                   w.part_cap, w.ticket, w.ticket_cap, s)) != 0)
      return rc;
    if ((rc = gemm(dQt, GK, w.UdT, H, w.dHrec, H, B, H, GK, w.part, w.part_cap, w.ticket, w.ticket_cap, s)) != 0) return rc;
  }
  hipLaunchKernelGGL(carry_kernel, egrid, eblock, 0, s, g, 1, (const float*)nullptr, (const float*)nullptr, w.dHrec,
                     w.ehterm, w.dcar, w.dh0, w.dc0);
  if ((rc = (int)hipGetLastError()) != 0) return rc;
  // dqx over all rows, then dx
  if ((rc = gemm(w.dpre, (long long)NT * 4, w.VxT, g.KX, w.dqx, g.KX, T * B, g.KX, NT * 4, w.part, w.part_cap, w.ticket,
                 w.ticket_cap, s)) != 0)
    return rc;
  if (w.dx != nullptr) {
    const long long total = (long long)T * B * g.I;
    hipLaunchKernelGGL(dx_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g, w.dqx, w.dpre, w.UXP, w.EXT,
                       w.dx);
    if ((rc = (int)hipGetLastError()) != 0) return rc;
  }
  return 0;
}
