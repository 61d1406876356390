// Persistent recurrent backward kernel (gfx950).  Mirror image of rec_fwd_kernel:
//   vr[4][KQ]  rotated image of this row's V_h for the DPP reduce   dQ = sum_{k,n} dpre[k][n] V_h[k][n][:]
//   ue[KH]     this unit's own U_h weights                          dh_{t-1} = dQ . ue + dpre . eh
// Per timestep (one barrier): gate derivatives from the tape (gates, c) -> dpre (stored for the
// weight-gradient kernels) -> rank-space reduce -> barrier -> expand to dh_{t-1}.  Weight gradients are
// NOT accumulated here: they are batched GEMM-shaped sums over all (t,b) and run on the whole chip in
// wgrad_x/wgrad_h afterwards, off the serial chain.
// Replaces autograd's replay of the time loop.
#include "vmlmf_launch.h"

template <int KH, int R, bool FLAT, int MAXT>
__global__ void __launch_bounds__(MAXT) rec_bwd_kernel(VGeo g, BwdArgs a) {
  constexpr int NP = (KH + 15) / 16, KQ = NP * 16, NC = KH / 4;
  constexpr int NJ = FLAT ? 2 : 1;  // rank-space vectors a wave contributes to
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NT = g.NT, NW = g.NW, W = g.W, H = g.H, B = g.B, T = g.T;
  const int grp = tid / (64 * W);
  const int m = tid - grp * 64 * W;
  const bool valid = m < g.Hg;
  const int n = grp * g.Hg + (valid ? m : 0);

  float vr[4][KQ], ue[KH], eh[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int j = 0; j < KQ; ++j) vr[k][j] = a.VR[(size_t)(k * KQ + j) * NT + tid];
    eh[k] = a.EH[k * NT + tid];
  }
#pragma unroll
  for (int rr = 0; rr < KH; ++rr) ue[rr] = a.UE[(size_t)rr * NT + tid];

  extern __shared__ float4 smem4[];
  float* part = reinterpret_cast<float*>(smem4);  // [2][R][NW][NJ][KQ]

  // chunk c of block s: this unit needs dQ[dest][chunk], dest = (grp - s) mod G.
  //   non-FLAT: contributors are the W waves of group dest (qsel = group of the unit)
  //   FLAT:     every wave contributes to both vectors; slot [wave][dest]
  int pb[NC], dst[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int s = (g.G == 2 && 4 * c >= g.off1) ? 1 : 0;
    dst[c] = (grp - s + g.G) % g.G;
    pb[c] = FLAT ? (dst[c] * KQ + 4 * c) : (dst[c] * W * KQ + 4 * c);
  }
  const int cnt = FLAT ? NW : W;
  constexpr int wstride = NJ * KQ;
  const bool qwriter = (wave == grp * W);

  int row[R];
  bool ok[R];
  float dhrec[R], dcs[R], ccur[R], cprv[R], dyv[R];
  float4 g4[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    row[r] = blockIdx.x * R + r;
    ok[r] = valid && row[r] < B;
    const size_t bh = (size_t)row[r] * H + n;
    dhrec[r] = (ok[r] && a.dhT != nullptr) ? a.dhT[bh] : 0.f;
    dcs[r] = (ok[r] && a.dcT != nullptr) ? a.dcT[bh] : 0.f;
    const size_t e = (size_t)((T - 1) * B + row[r]) * H + n;
    ccur[r] = ok[r] ? a.cs[e] : 0.f;
    cprv[r] = !ok[r] ? 0.f : (T > 1 ? a.cs[e - (size_t)B * H] : (a.c0 != nullptr ? a.c0[bh] : 0.f));
    g4[r] = ok[r] ? ld4(a.gates + e * 4) : f4zero();
    dyv[r] = (ok[r] && a.dy != nullptr) ? a.dy[(T - 1) * g.syT + row[r] * g.syB + n] : 0.f;
  }

  for (int t = T - 1; t >= 0; --t) {
    const int buf = t & 1;
    // prefetch the tape of step t-1
    float4 g4n[R];
    float dyn[R], cpn[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      g4n[r] = f4zero();
      dyn[r] = 0.f;
      cpn[r] = 0.f;
      if (ok[r] && t > 0) {
        const size_t e = (size_t)((t - 1) * B + row[r]) * H + n;
        g4n[r] = ld4(a.gates + e * 4);
        if (a.dy != nullptr) dyn[r] = a.dy[(t - 1) * g.syT + row[r] * g.syB + n];
        cpn[r] = (t > 1) ? a.cs[e - (size_t)B * H] : (a.c0 != nullptr ? a.c0[(size_t)row[r] * H + n] : 0.f);
      }
    }
    float ehterm[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float ig = g4[r].x, fg = g4[r].y, og = g4[r].z, ng = g4[r].w;
      const float dh = dyv[r] + dhrec[r];
      const float tc = fast_tanh(ccur[r]);
      const float dct = fmaf(dh * og, 1.f - tc * tc, dcs[r]);
      float dp[4];
      dp[0] = dct * ng * ig * (1.f - ig);
      dp[1] = dct * cprv[r] * fg * (1.f - fg);
      dp[2] = dh * tc * og * (1.f - og);
      dp[3] = dct * ig * (1.f - ng * ng);
      dcs[r] = dct * fg;
      if (ok[r]) st4(a.dpre + ((size_t)(t * B + row[r]) * H + n) * 4, make_float4(dp[0], dp[1], dp[2], dp[3]));
      ehterm[r] = (dp[0] * eh[0] + dp[1] * eh[1]) + (dp[2] * eh[2] + dp[3] * eh[3]);
      // rank-space reduce of dpre
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        sfor<16>([&](auto K) {
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k] = fmaf(ror16<K>(dp[k]), vr[k][p * 16 + K], acc[k]);
        });
        float* wdst = part + ((size_t)((buf * R + r) * NW + wave) * NJ) * KQ + p * 16 + lane;
        if (FLAT) {
          const float s0 = rowsum4(acc[0] + acc[1]);
          const float s1 = rowsum4(acc[2] + acc[3]);
          if (lane < 16) {
            wdst[0] = s0;
            wdst[KQ] = s1;
          }
        } else {
          const float s0 = rowsum4((acc[0] + acc[1]) + (acc[2] + acc[3]));
          if (lane < 16) wdst[0] = s0;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float dhn = ehterm[r];
      const float* src = part + (size_t)(buf * R + r) * NW * NJ * KQ;
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) {
        float4 q = f4zero();
        for (int w = 0; w < cnt; ++w) q = f4add(q, ld4(src + pb[cc] + w * wstride));
        if (a.dQs != nullptr && qwriter && lane == cc && row[r] < B)
          st4(a.dQs + ((size_t)(t * B + row[r]) * g.G + dst[cc]) * KH + 4 * cc, q);
        dhn = fmaf(q.x, ue[4 * cc + 0], dhn);
        dhn = fmaf(q.y, ue[4 * cc + 1], dhn);
        dhn = fmaf(q.z, ue[4 * cc + 2], dhn);
        dhn = fmaf(q.w, ue[4 * cc + 3], dhn);
      }
      dhrec[r] = dhn;
      ccur[r] = cprv[r];
      cprv[r] = cpn[r];
      g4[r] = g4n[r];
      dyv[r] = dyn[r];
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (ok[r]) {
      if (a.dh0 != nullptr) a.dh0[(size_t)row[r] * H + n] = dhrec[r];
      if (a.dc0 != nullptr) a.dc0[(size_t)row[r] * H + n] = dcs[r];
    }
  }
}

template <int KH, int R, bool FLAT, int MAXT>
static int launch_one(const VGeo& g, const BwdArgs& a, hipStream_t s) {
  constexpr int KQ = (KH + 15) / 16 * 16;
  const size_t lds = sizeof(float) * 2 * R * g.NW * (FLAT ? 2 : 1) * KQ;
  hipLaunchKernelGGL((rec_bwd_kernel<KH, R, FLAT, MAXT>), dim3(g.nwg), dim3(g.NT), lds, s, g, a);
  return (int)hipGetLastError();
}

template <int KH, int MAXT>
static int launch_kh(const VGeo& g, const BwdArgs& a, hipStream_t s) {
  if (g.flat) {
    if (g.R == 1) return launch_one<KH, 1, true, MAXT>(g, a, s);
    return -3;
  }
  if (g.R == 1) return launch_one<KH, 1, false, MAXT>(g, a, s);
  if (g.R == 2) return launch_one<KH, 2, false, MAXT>(g, a, s);
  return -3;
}

template <int MAXT>
static int launch_t(const VGeo& g, const BwdArgs& a, hipStream_t s) {
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

int launch_rec_bwd(const VGeo& g, const BwdArgs& a, hipStream_t s) {
  if (g.NT <= 256) return launch_t<256>(g, a, s);
  if (g.NT <= 512) return launch_t<512>(g, a, s);
  return -3;
}
