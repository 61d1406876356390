// Step-wise path for layers whose factors do not fit one CU's register file (padded hidden rank > 32 or
// more than 512 thread slots, e.g. BASELINE config E: H = 650, ranks 32 / [32,32]).
//
// The persistent kernels keep U_h/V_h in registers for all T steps; when that is impossible the recurrence is
// run one timestep at a time, cuDNN-style, with the batch as the GEMM M dimension:
//     Q_t   = H_{t-1} Ud            (B x H)(H x G*KH)         gemm_nn_kernel  (fp32 MFMA 32x32x2)
//     P_t   = Q_t Vd                (B x G*KH)(G*KH x 4*slots) gemm_nn_kernel
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

struct GemmArgs {
  const float* A;
  long long lda;
  const float* B;
  long long ldb;
  float* C;
  long long ldc;
  int M, N, K;
};

// C[M x N] = A[M x K] B[K x N], row-major, any sizes (masked).  One 32x32 tile of C per workgroup of SPLIT
// waves: the waves split K between them (skinny products such as dQ = dpre VdT have few tiles but a long K)
// and their partial tiles are summed through LDS in wave order (deterministic).
// Lane l = (lk = l>>5, li = l&31) supplies A[row li][k] and B[k][col li]; within an 8-wide k block MFMA step s
// contracts k = kb + 4*lk + s, so a lane reads 4 consecutive floats of its A row per block.
template <int SPLIT, bool VEC>
__global__ void __launch_bounds__(64 * SPLIT) gemm_nn_kernel(GemmArgs a) {
  __shared__ float red[SPLIT > 1 ? SPLIT * 1024 : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles_n = (a.N + 31) / 32;
  const int tile = blockIdx.x;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int li = lane & 31, lk = lane >> 5;
  const int row = tm * 32 + li, col = tn * 32 + li;
  const bool rok = row < a.M, cok = col < a.N;
  const float* Ar = a.A + (long long)(rok ? row : 0) * a.lda;
  const float* Bc = a.B + (cok ? col : 0);
  // this wave's K range, in multiples of 16
  const int kper = ((a.K + SPLIT - 1) / SPLIT + 15) / 16 * 16;
  const int k0 = wave * kper, k1 = k0 + kper < a.K ? k0 + kper : a.K;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  constexpr int UB = 4;  // 8-wide k blocks per batch of loads (all loads of a batch are issued together)
  for (int kb = k0; kb < k1; kb += 8 * UB) {
    float av[UB][4], bv[UB][4];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (VEC) {   // rows of A are 16-byte aligned and K % 4 == 0: one 16-byte load per lane and block
        const int k = kb + 8 * u + 4 * lk;
        const bool kok = k < k1;
        const float4 x = ld4(Ar + (kok ? k : 0));
        const float m = (kok && rok) ? 1.f : 0.f;
        av[u][0] = m * x.x, av[u][1] = m * x.y, av[u][2] = m * x.z, av[u][3] = m * x.w;
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int k = kb + 8 * u + 4 * lk + s;
        const bool kok = k < k1;
        const int kc = kok ? k : 0;
        if (!VEC) {
          const float x = Ar[kc];
          av[u][s] = (kok && rok) ? x : 0.f;
        }
        const float y = Bc[(long long)kc * a.ldb];
        bv[u][s] = (kok && cok) ? y : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < UB; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][s], bv[u][s], acc, 0, 0, 0);
  }
  if (SPLIT > 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = acc[r];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float sum = red[r * 64 + lane];
      for (int w = 1; w < SPLIT; ++w) sum += red[w * 1024 + r * 64 + lane];
      acc[r] = sum;
    }
  }
  if (cok) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (i < a.M) a.C[(long long)i * a.ldc + col] = acc[r];
    }
  }
}

static int gemm(const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc, int M, int N,
                int K, hipStream_t s) {
  GemmArgs a{A, lda, B, ldb, C, ldc, M, N, K};
  const int tiles = ((M + 31) / 32) * ((N + 31) / 32);
  // few tiles and a long K: split K over up to 16 waves of the tile's workgroup
  const bool vec = (lda % 4 == 0) && (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
#define GEMM_GO(SP)                                                                              \
  do {                                                                                           \
    if (vec)                                                                                     \
      hipLaunchKernelGGL((gemm_nn_kernel<SP, true>), dim3(tiles), dim3(64 * SP), 0, s, a);       \
    else                                                                                         \
      hipLaunchKernelGGL((gemm_nn_kernel<SP, false>), dim3(tiles), dim3(64 * SP), 0, s, a);      \
  } while (0)
  if (tiles <= 128 && K >= 256)
    GEMM_GO(16);
  else if (tiles <= 512 && K >= 128)
    GEMM_GO(4);
  else
    GEMM_GO(1);
#undef GEMM_GO
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
    if ((rc = gemm(A, lda, w.Ud, GK, Qt, GK, B, GK, H, s)) != 0) return rc;
    if ((rc = gemm(Qt, GK, w.Vd, (long long)NT * 4, w.P, (long long)NT * 4, B, NT * 4, GK, s)) != 0) return rc;
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
    if ((rc = gemm(w.dpre + (size_t)t * sstride * 4, (long long)NT * 4, w.VdT, GK, dQt, GK, B, GK, NT * 4, s)) != 0)
      return rc;
    if ((rc = gemm(dQt, GK, w.UdT, H, w.dHrec, H, B, H, GK, s)) != 0) return rc;
  }
  hipLaunchKernelGGL(carry_kernel, egrid, eblock, 0, s, g, 1, (const float*)nullptr, (const float*)nullptr, w.dHrec,
                     w.ehterm, w.dcar, w.dh0, w.dc0);
  if ((rc = (int)hipGetLastError()) != 0) return rc;
  // dqx over all rows, then dx
  if ((rc = gemm(w.dpre, (long long)NT * 4, w.VxT, g.KX, w.dqx, g.KX, T * B, g.KX, NT * 4, s)) != 0) return rc;
  if (w.dx != nullptr) {
    const long long total = (long long)T * B * g.I;
    hipLaunchKernelGGL(dx_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g, w.dqx, w.dpre, w.UXP, w.EXT,
                       w.dx);
    if ((rc = (int)hipGetLastError()) != 0) return rc;
  }
  return 0;
}
