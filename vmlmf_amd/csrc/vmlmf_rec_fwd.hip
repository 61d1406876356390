// Persistent recurrent forward kernel (gfx950).
//
// One workgroup owns R batch rows for all T timesteps; thread <-> hidden unit (vmlmf_geo.h).  The
// hidden->hidden factors never leave the register file:
//   ve[4][KH]  this unit's rows of V_h (all four gates)      -> expansion  pre[k] += Q . ve[k]
//   ur[KQ]     rotated image of U_h for the 16-lane DPP reduce -> Q = h_{t-1} U_h
// Per timestep (ONE workgroup barrier):
//   1. reduce   each lane multiplies the h it receives through row_ror:kk with its rotated weight; after
//               16 steps lane i of a row holds rank (16p+i) summed over the row's 16 units; permlane swaps
//               add the 4 rows; lanes 0-15 of every wave store the wave partial to LDS (double-buffered).
//   2. barrier
//   3. expand   sum the partials of the waves of the source group (broadcast ds_read_b128), 4*KH FMAs
//               against ve, add the precomputed x-side pre-activation gx[t] (prefetched a step ahead) and
//               h .* eh, LSTM gates (v_exp/v_rcp), write y[t] and the tape (gates, c, Q).
// Replaces the time loop + cell forward: vmlmf.py:300-314 + 78-125, vmlmf_group.py:85-155,
// vmlmf_lm.py:272-280 + 222-269, 166-174 + 97-163.
#include "vmlmf_launch.h"

template <int KH, int R, bool FLAT, int MAXT>
__global__ void __launch_bounds__(MAXT) rec_fwd_kernel(VGeo g, FwdArgs a) {
  constexpr int NP = (KH + 15) / 16, KQ = NP * 16, NC = KH / 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NT = g.NT, NW = g.NW, W = g.W, H = g.H, B = g.B, T = g.T;
  const int grp = tid / (64 * W);
  const int m = tid - grp * 64 * W;
  const bool valid = m < g.Hg;
  const int n = grp * g.Hg + (valid ? m : 0);

  float ve[4][KH], ur[KQ], eh[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int rr = 0; rr < KH; ++rr) ve[k][rr] = a.VE[(size_t)(k * KH + rr) * NT + tid];
    eh[k] = a.EH[k * NT + tid];
  }
#pragma unroll
  for (int j = 0; j < KQ; ++j) ur[j] = a.UR[(size_t)j * NT + tid];

  extern __shared__ float4 smem4[];
  float* part = reinterpret_cast<float*>(smem4);  // [2][R][NW][KQ]

  // LDS offset (floats) of the first contributing wave's partial for each 4-rank chunk
  int pb[NC];
  int pb2[FLAT ? NC : 1];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int s = (g.G == 2 && 4 * c >= g.off1) ? 1 : 0;
    const int q0 = FLAT ? 0 : grp;
    pb[c] = (((q0 + s) % g.G) * W) * KQ + 4 * c;
    if (FLAT) pb2[c] = (((1 + s) % g.G) * W) * KQ + 4 * c;
  }
  const bool qwriter = FLAT ? (wave == 0) : (wave == grp * W);

  int row[R];
  bool ok[R];
  float h[R], c[R];
  float4 gxr[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    row[r] = blockIdx.x * R + r;
    ok[r] = valid && row[r] < B;
    h[r] = (ok[r] && a.h0 != nullptr) ? a.h0[(size_t)row[r] * H + n] : 0.f;
    c[r] = (ok[r] && a.c0 != nullptr) ? a.c0[(size_t)row[r] * H + n] : 0.f;
    gxr[r] = ok[r] ? ld4(a.gx + ((size_t)row[r] * H + n) * 4) : f4zero();
  }

  for (int t = 0; t < T; ++t) {
    const int buf = t & 1;
    // ---- 1. rank-space reduce of h_{t-1}
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float hv = h[r];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        float a0 = 0.f, a1 = 0.f;
        sfor<8>([&](auto K) {
          a0 = fmaf(ror16<2 * K>(hv), ur[p * 16 + 2 * K], a0);
          a1 = fmaf(ror16<2 * K + 1>(hv), ur[p * 16 + 2 * K + 1], a1);
        });
        const float s = rowsum4(a0 + a1);
        if (lane < 16) part[((buf * R + r) * NW + wave) * KQ + p * 16 + lane] = s;
      }
    }
    // prefetch next step's x-side pre-activations while the partials land
    float4 gxn[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
      gxn[r] = (ok[r] && t + 1 < T) ? ld4(a.gx + ((size_t)((t + 1) * B + row[r]) * H + n) * 4) : f4zero();
    __syncthreads();
    // ---- 3. expand + gates
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float pre[4];
      pre[0] = fmaf(h[r], eh[0], gxr[r].x);
      pre[1] = fmaf(h[r], eh[1], gxr[r].y);
      pre[2] = fmaf(h[r], eh[2], gxr[r].z);
      pre[3] = fmaf(h[r], eh[3], gxr[r].w);
      const float* src = part + (size_t)(buf * R + r) * NW * KQ;
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) {
        float4 q = f4zero();
        for (int w = 0; w < W; ++w) q = f4add(q, ld4(src + pb[cc] + w * KQ));
        float4 q2 = q;
        if (FLAT) {
          q2 = f4zero();
          for (int w = 0; w < W; ++w) q2 = f4add(q2, ld4(src + pb2[cc] + w * KQ));
        }
        if (a.Qs != nullptr && qwriter && lane == cc && row[r] < B) {
          float* qdst = a.Qs + ((size_t)(t * B + row[r]) * g.G + (FLAT ? 0 : grp)) * KH + 4 * cc;
          st4(qdst, q);
          if (FLAT) st4(qdst + KH, q2);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 qq = (FLAT && k >= 2) ? q2 : q;
          pre[k] = fmaf(qq.x, ve[k][4 * cc + 0], pre[k]);
          pre[k] = fmaf(qq.y, ve[k][4 * cc + 1], pre[k]);
          pre[k] = fmaf(qq.z, ve[k][4 * cc + 2], pre[k]);
          pre[k] = fmaf(qq.w, ve[k][4 * cc + 3], pre[k]);
        }
      }
      const float ig = fast_sigmoid(pre[0]);
      const float fg = fast_sigmoid(pre[1]);
      const float og = fast_sigmoid(pre[2]);
      const float ng = fast_tanh(pre[3]);
      c[r] = fmaf(fg, c[r], ig * ng);
      h[r] = og * fast_tanh(c[r]);
      if (ok[r]) {
        const int b = row[r];
        a.y[t * g.syT + b * g.syB + n] = h[r];
        if (a.gates != nullptr) {
          const size_t e = (size_t)(t * B + b) * H + n;
          st4(a.gates + e * 4, make_float4(ig, fg, og, ng));
          a.cs[e] = c[r];
        }
      }
      gxr[r] = gxn[r];
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (ok[r]) {
      if (a.hT != nullptr) a.hT[(size_t)row[r] * H + n] = h[r];
      if (a.cT != nullptr) a.cT[(size_t)row[r] * H + n] = c[r];
    }
  }
}

template <int KH, int R, bool FLAT, int MAXT>
static int launch_one(const VGeo& g, const FwdArgs& a, hipStream_t s) {
  constexpr int KQ = (KH + 15) / 16 * 16;
  const size_t lds = sizeof(float) * 2 * R * g.NW * KQ;
  hipLaunchKernelGGL((rec_fwd_kernel<KH, R, FLAT, MAXT>), dim3(g.nwg), dim3(g.NT), lds, s, g, a);
  return (int)hipGetLastError();
}

template <int KH, int MAXT>
static int launch_kh(const VGeo& g, const FwdArgs& a, hipStream_t s) {
  if (g.flat) {
    if (g.R == 1) return launch_one<KH, 1, true, MAXT>(g, a, s);
    return -3;
  }
  if (g.R == 1) return launch_one<KH, 1, false, MAXT>(g, a, s);
  if (g.R == 2) return launch_one<KH, 2, false, MAXT>(g, a, s);
  return -3;
}

template <int MAXT>
static int launch_t(const VGeo& g, const FwdArgs& a, hipStream_t s) {
  switch (g.KH) {
    case 8:
      return launch_kh<8, MAXT>(g, a, s);
    case 16:
      return launch_kh<16, MAXT>(g, a, s);
    case 24:
      return launch_kh<24, MAXT>(g, a, s);
    case 32:
      return launch_kh<32, MAXT>(g, a, s);
    default:
      return -3;
  }
}

int launch_rec_fwd(const VGeo& g, const FwdArgs& a, hipStream_t s) {
  if (g.NT <= 256) return launch_t<256>(g, a, s);
  if (g.NT <= 512) return launch_t<512>(g, a, s);
  return -3;
}
