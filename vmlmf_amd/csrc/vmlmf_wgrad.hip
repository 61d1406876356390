// Weight-gradient kernels (gfx950): the batched, non-recurrent half of the backward pass.
// Every (t,b) row is independent here, so the whole chip works on it after rec_bwd_kernel has produced
// dpre[t,b,n,k].  Thread <-> hidden unit as in the recurrent kernels; each thread keeps the gradient
// accumulators of ITS unit's weight rows in registers across the workgroup's RC rows, then writes them as
// one partial (summed in fixed order by reduce_kernel -> deterministic, no float atomics).
//
//   wgrad_x:  dqx = dpre V_x (DPP reduce), dx = dqx U_x^T + dpre .* ex,
//             dV_x += dpre^T qx, dU_x += x^T dqx, d(ex) += dpre .* x
//   wgrad_h:  dV_h += dpre^T Q, dU_h += h_{t-1}^T dQ, d(eh) += dpre .* h_{t-1}, db += dpre
// (Q, dQ and qx are the rank-space vectors the forward / backward recurrent kernels already computed.)
#include "vmlmf_launch.h"

template <int KX, int MAXT>
__global__ void __launch_bounds__(MAXT) wgrad_x_kernel(VGeo g, WgxArgs a) {
  constexpr int NPX = (KX + 15) / 16, KQX = NPX * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NT = g.NT, NW = g.NW, W = g.W, H = g.H, B = g.B;
  const int TB = g.T * B;
  const int grp = tid / (64 * W);
  const int m = tid - grp * 64 * W;
  const bool valid = m < g.Hg;
  const int n = grp * g.Hg + (valid ? m : 0);
  const bool has_x = valid && n < g.I;
  const bool wave_x = __ballot(has_x) != 0ull;
  const int row0 = blockIdx.x * g.RC;
  const int nrows = (TB - row0 < g.RC) ? (TB - row0) : g.RC;

  extern __shared__ float4 smem4[];
  float* partx = reinterpret_cast<float*>(smem4);  // [RC][NW][KQX]

  {  // phase 1: dqx partials for every row of the chunk
    float vrx[4][KQX];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < KQX; ++j) vrx[k][j] = a.VRX[(size_t)(k * KQX + j) * NT + tid];
    for (int rl = 0; rl < nrows; ++rl) {
      const int row = row0 + rl, t = row / B, b = row - t * B;
      const float4 d = ld4(a.dpre + ((size_t)(t * g.Bp + b) * NT + tid) * 4);   // slot-padded, zeros in pad slots
      const float dp[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
      for (int p = 0; p < NPX; ++p) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        sfor<16>([&](auto K) {
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k] = fmaf(ror16<K>(dp[k]), vrx[k][p * 16 + K], acc[k]);
        });
        const float s = rowsum4((acc[0] + acc[1]) + (acc[2] + acc[3]));
        if (lane < 16) partx[(size_t)(rl * NW + wave) * KQX + p * 16 + lane] = s;
      }
    }
  }
  __syncthreads();

  float avx[4][KX], aux[KX], aex[4], uxo[KX], exi[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int r = 0; r < KX; ++r) avx[k][r] = 0.f;
    aex[k] = 0.f;
    exi[k] = a.EXI[k * NT + tid];
  }
#pragma unroll
  for (int r = 0; r < KX; ++r) {
    aux[r] = 0.f;
    uxo[r] = a.UXO[(size_t)r * NT + tid];
  }

  for (int rl = 0; rl < nrows; ++rl) {
    const int row = row0 + rl;
    const int t = row / B, b = row - t * B;
    const float4 d = ld4(a.dpre + ((size_t)(t * g.Bp + b) * NT + tid) * 4);
    const float dp[4] = {d.x, d.y, d.z, d.w};
    const float* qxr = a.qx + (size_t)row * KX;
#pragma unroll
    for (int r4 = 0; r4 < KX / 4; ++r4) {
      const float4 q = ld4(qxr + 4 * r4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        avx[k][4 * r4 + 0] = fmaf(dp[k], q.x, avx[k][4 * r4 + 0]);
        avx[k][4 * r4 + 1] = fmaf(dp[k], q.y, avx[k][4 * r4 + 1]);
        avx[k][4 * r4 + 2] = fmaf(dp[k], q.z, avx[k][4 * r4 + 2]);
        avx[k][4 * r4 + 3] = fmaf(dp[k], q.w, avx[k][4 * r4 + 3]);
      }
    }
    if (wave_x) {  // wave-uniform: only waves that hold x-units need dqx
      const float xv = has_x ? a.x[t * g.sxT + b * g.sxB + n] : 0.f;
      float dxv = (dp[0] * exi[0] + dp[1] * exi[1]) + (dp[2] * exi[2] + dp[3] * exi[3]);
#pragma unroll
      for (int k = 0; k < 4; ++k) aex[k] = fmaf(dp[k], xv, aex[k]);
#pragma unroll
      for (int r4 = 0; r4 < KX / 4; ++r4) {
        float4 q = f4zero();
        for (int w = 0; w < NW; ++w) q = f4add(q, ld4(partx + (size_t)(rl * NW + w) * KQX + 4 * r4));
        aux[4 * r4 + 0] = fmaf(xv, q.x, aux[4 * r4 + 0]);
        aux[4 * r4 + 1] = fmaf(xv, q.y, aux[4 * r4 + 1]);
        aux[4 * r4 + 2] = fmaf(xv, q.z, aux[4 * r4 + 2]);
        aux[4 * r4 + 3] = fmaf(xv, q.w, aux[4 * r4 + 3]);
        dxv = fmaf(q.x, uxo[4 * r4 + 0], dxv);
        dxv = fmaf(q.y, uxo[4 * r4 + 1], dxv);
        dxv = fmaf(q.z, uxo[4 * r4 + 2], dxv);
        dxv = fmaf(q.w, uxo[4 * r4 + 3], dxv);
      }
      if (has_x && a.dx != nullptr) a.dx[t * g.sxT + b * g.sxB + n] = dxv;
    }
  }
  float* wp = a.wpart + (size_t)blockIdx.x * g.NA * NT + tid;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int r = 0; r < KX; ++r) wp[(size_t)va_vx(g, k, r) * NT] = avx[k][r];
    wp[(size_t)va_ex(g, k) * NT] = aex[k];
  }
#pragma unroll
  for (int r = 0; r < KX; ++r) wp[(size_t)va_ux(g, r) * NT] = aux[r];
}

template <int KH, bool FLAT, int MAXT>
__global__ void __launch_bounds__(MAXT) wgrad_h_kernel(VGeo g, WghArgs a) {
  const int tid = threadIdx.x;
  const int NT = g.NT, W = g.W, H = g.H, B = g.B;
  const int TB = g.T * B;
  const int grp = tid / (64 * W);
  const int m = tid - grp * 64 * W;
  const bool valid = m < g.Hg;
  const int n = grp * g.Hg + (valid ? m : 0);
  const int ugrp = __builtin_amdgcn_readfirstlane(grp);  // groups are wave-aligned
  const int row0 = blockIdx.x * g.RC;
  const int nrows = (TB - row0 < g.RC) ? (TB - row0) : g.RC;
  const int GK = g.G * KH;
  // offsets (floats) inside a row's G*KH rank-space record
  const int qoff01 = FLAT ? 0 : ugrp * KH;                 // Q used by gates i,f
  const int qoff23 = FLAT ? KH : ugrp * KH;                // Q used by gates o,n
  const int d0 = ugrp * KH;                                // dQ[dest] for block 0: dest = grp
  const int d1 = ((ugrp - 1 + g.G) % g.G) * KH;            // block 1: dest = grp - 1

  float avc[4][KH], auc[KH], aeh[4], ab[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int rr = 0; rr < KH; ++rr) avc[k][rr] = 0.f;
    aeh[k] = 0.f;
    ab[k] = 0.f;
  }
#pragma unroll
  for (int rr = 0; rr < KH; ++rr) auc[rr] = 0.f;

  for (int rl = 0; rl < nrows; ++rl) {
    const int row = row0 + rl;
    const int t = row / B, b = row - t * B;
    const float4 d = ld4(a.dpre + ((size_t)(t * g.Bp + b) * NT + tid) * 4);
    const float dp[4] = {d.x, d.y, d.z, d.w};
    float hp = 0.f;
    if (valid) {
      if (t > 0)
        hp = a.y[(t - 1) * g.syT + b * g.syB + n];
      else if (a.h0 != nullptr)
        hp = a.h0[(size_t)b * H + n];
    }
    const float* Qr = a.Qs + (size_t)row * GK;
    const float* dQr = a.dQs + (size_t)row * GK;
#pragma unroll
    for (int c = 0; c < KH / 4; ++c) {
      const float4 qa = ld4(Qr + qoff01 + 4 * c);
      const float4 qb = FLAT ? ld4(Qr + qoff23 + 4 * c) : qa;
      const float4 dq = ld4(dQr + ((g.G == 2 && 4 * c >= g.off1) ? d1 : d0) + 4 * c);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4 q = (k >= 2) ? qb : qa;
        avc[k][4 * c + 0] = fmaf(dp[k], q.x, avc[k][4 * c + 0]);
        avc[k][4 * c + 1] = fmaf(dp[k], q.y, avc[k][4 * c + 1]);
        avc[k][4 * c + 2] = fmaf(dp[k], q.z, avc[k][4 * c + 2]);
        avc[k][4 * c + 3] = fmaf(dp[k], q.w, avc[k][4 * c + 3]);
      }
      auc[4 * c + 0] = fmaf(hp, dq.x, auc[4 * c + 0]);
      auc[4 * c + 1] = fmaf(hp, dq.y, auc[4 * c + 1]);
      auc[4 * c + 2] = fmaf(hp, dq.z, auc[4 * c + 2]);
      auc[4 * c + 3] = fmaf(hp, dq.w, auc[4 * c + 3]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      aeh[k] = fmaf(dp[k], hp, aeh[k]);
      ab[k] += dp[k];
    }
  }
  float* wp = a.wpart + (size_t)blockIdx.x * g.NA * NT + tid;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int rr = 0; rr < KH; ++rr) wp[(size_t)va_vc(g, k, rr) * NT] = avc[k][rr];
    wp[(size_t)va_eh(g, k) * NT] = aeh[k];
    wp[(size_t)va_b(g, k) * NT] = ab[k];
  }
#pragma unroll
  for (int rr = 0; rr < KH; ++rr) wp[(size_t)va_uc(g, rr) * NT] = auc[rr];
}

template <int MAXT>
static int launch_x_t(const VGeo& g, const WgxArgs& a, hipStream_t s) {
  const size_t lds = sizeof(float) * (size_t)g.RC * g.NW * g.KQX;
  const dim3 grid(g.nblk), block(g.NT);
  switch (g.KX) {
    case 8:
      hipLaunchKernelGGL((wgrad_x_kernel<8, MAXT>), grid, block, lds, s, g, a);
      break;
    case 16:
      hipLaunchKernelGGL((wgrad_x_kernel<16, MAXT>), grid, block, lds, s, g, a);
      break;
    case 24:
      hipLaunchKernelGGL((wgrad_x_kernel<24, MAXT>), grid, block, lds, s, g, a);
      break;
    case 32:
      hipLaunchKernelGGL((wgrad_x_kernel<32, MAXT>), grid, block, lds, s, g, a);
      break;
    default:
      return -3;
  }
  return (int)hipGetLastError();
}

int launch_wgrad_x(const VGeo& g, const WgxArgs& a, hipStream_t s) {
  if (g.NT <= 256) return launch_x_t<256>(g, a, s);
  if (g.NT <= 512) return launch_x_t<512>(g, a, s);
  return -3;
}

template <int KH, int MAXT>
static int launch_h_kh(const VGeo& g, const WghArgs& a, hipStream_t s) {
  const dim3 grid(g.nblk), block(g.NT);
  if (g.flat)
    hipLaunchKernelGGL((wgrad_h_kernel<KH, true, MAXT>), grid, block, 0, s, g, a);
  else
    hipLaunchKernelGGL((wgrad_h_kernel<KH, false, MAXT>), grid, block, 0, s, g, a);
  return (int)hipGetLastError();
}

template <int MAXT>
static int launch_h_t(const VGeo& g, const WghArgs& a, hipStream_t s) {
  switch (g.KH) {
    case 8:
      return launch_h_kh<8, MAXT>(g, a, s);
    case 16:
      return launch_h_kh<16, MAXT>(g, a, s);
    case 24:
      return launch_h_kh<24, MAXT>(g, a, s);
    case 32:
      return launch_h_kh<32, MAXT>(g, a, s);
    default:
      return -3;
  }
}

int launch_wgrad_h(const VGeo& g, const WghArgs& a, hipStream_t s) {
  if (g.NT <= 256) return launch_h_t<256>(g, a, s);
  if (g.NT <= 512) return launch_h_t<512>(g, a, s);
  return -3;
}
