// Classifier head of Net: logits = h_last W^T + bias (nn.Linear(H, 18), V/src/models/vmlmf.py:345,353-355)
// and its backward.  B x H x C is tiny (64 x 180 x 18 at the headline shape): a library GEMM costs three
// launches of 8-10 us plus a bias-gradient reduction; these two kernels are latency-sized instead.
// Summation orders are fixed (no atomics): results are run-to-run identical.
#include <hip/hip_runtime.h>

#include "vmlmf_launch.h"

namespace {

constexpr int HEAD_CMAX = 32;   // classes held in registers by the weight-gradient blocks

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// grid B, block 256: wave w owns classes w, w+4, ... (<= 8 of them, accumulated together so that every load
// of the feature loop is independent of the others); lanes stride the H features.
__global__ __launch_bounds__(256) void head_fwd_kernel(int H, int C, const float* __restrict__ h, long long ldh,
                                                       const float* __restrict__ W,
                                                       const float* __restrict__ bias, float* __restrict__ out) {
  const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* hb = h + (size_t)b * ldh;
  float acc[HEAD_CMAX / 4];
#pragma unroll
  for (int j = 0; j < HEAD_CMAX / 4; ++j) acc[j] = 0.f;
  // Loads go out from clamped (always valid) indices and are masked where they are consumed, four feature strides
  // per pass: under `c < C ? W[..] : 0` hipcc waits for each load before issuing the next (vmcnt(0) after every
  // one), which made this kernel a chain of memory latencies.
  for (int n0 = lane; n0 < H; n0 += 256) {
    float hv[4], wv[4][HEAD_CMAX / 4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int n = n0 + 64 * u < H ? n0 + 64 * u : lane;
      hv[u] = hb[n];
#pragma unroll
      for (int j = 0; j < HEAD_CMAX / 4; ++j) wv[u][j] = W[(size_t)(w + 4 * j < C ? w + 4 * j : 0) * H + n];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float hm = n0 + 64 * u < H ? hv[u] : 0.f;
#pragma unroll
      for (int j = 0; j < HEAD_CMAX / 4; ++j) acc[j] = fmaf(hm, w + 4 * j < C ? wv[u][j] : 0.f, acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < HEAD_CMAX / 4; ++j) {
    const int c = w + 4 * j;
    const float v = wave_sum(acc[j]);
    if (lane == 0 && c < C) out[(size_t)b * C + c] = v + (bias != nullptr ? bias[c] : 0.f);
  }
}

// One launch, three kinds of workgroup (256 threads each):
//   [0, B)               dh[b][:]  = dl[b][:] W                      (threads stride n)
//   [B, B + nH)          dW[:][n]  = sum_b dl[b][:] h[b][n]          (64 features per workgroup; batch rows staged
//                                                                     through LDS 64 at a time; lane <-> n,
//                                                                     wave <-> 8 classes)
//   B + nH               db[:]     = sum_b dl[b][:]
__global__ __launch_bounds__(256) void head_bwd_kernel(int B, int H, int C, const float* __restrict__ h,
                                                       long long ldh, const float* __restrict__ W,
                                                       const float* __restrict__ dl, float* __restrict__ dh,
                                                       float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float sh[64][64];                       // h tile  [b][n]
  __shared__ __attribute__((aligned(16))) float sdl[64][HEAD_CMAX];   // dl tile [b][wave * 8 + j]  (class = wave + 4 j)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nH = (H + 63) / 64;
  int blk = blockIdx.x;
  if (blk < B) {
    if (dh == nullptr) return;
    if (threadIdx.x < HEAD_CMAX) sdl[0][threadIdx.x] = (int)threadIdx.x < C ? dl[(size_t)blk * C + threadIdx.x] : 0.f;
    __syncthreads();
    for (int n = threadIdx.x; n < H; n += 256) {
      float wv[HEAD_CMAX];   // every weight load of the column is issued before the first FMA: one latency
#pragma unroll
      for (int c = 0; c < HEAD_CMAX; ++c) wv[c] = W[(size_t)(c < C ? c : 0) * H + n];   // sdl masks the classes past C
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < HEAD_CMAX; ++c) acc = fmaf(sdl[0][c], wv[c], acc);   // sdl is zero past C
      dh[(size_t)blk * H + n] = acc;
    }
    return;
  }
  blk -= B;
  if (blk < nH) {
    if (dW == nullptr) return;
    const int n = blk * 64 + lane;
    const bool ok = n < H;
    float acc[HEAD_CMAX / 4];
#pragma unroll
    for (int j = 0; j < HEAD_CMAX / 4; ++j) acc[j] = 0.f;
    for (int b0 = 0; b0 < B; b0 += 64) {
      if (b0 > 0) __syncthreads();
      // 24 independent loads per thread, all issued (clamped indices) before the first LDS write; masked on the way
      // into LDS.  Written as `cond ? load : 0` straight into LDS they went out one at a time.
      float th[16], td[8];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int b = b0 + i * 4 + w;
        th[i] = h[(size_t)(b < B ? b : 0) * ldh + (ok ? n : 0)];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {                  // 64 x 32 dl tile: thread -> (row, class slot)
        const int e = i * 256 + threadIdx.x, r = e >> 5, slot = e & 31, c = (slot >> 3) + 4 * (slot & 7);
        const int b = b0 + r;
        td[i] = dl[(size_t)(b < B ? b : 0) * C + (c < C ? c : 0)];
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) sh[i * 4 + w][lane] = (ok && b0 + i * 4 + w < B) ? th[i] : 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = i * 256 + threadIdx.x, r = e >> 5, slot = e & 31, c = (slot >> 3) + 4 * (slot & 7);
        sdl[r][slot] = (b0 + r < B && c < C) ? td[i] : 0.f;
      }
      __syncthreads();
#pragma unroll 4
      for (int r = 0; r < 64; ++r) {
        const float hv = sh[r][lane];
        const float4 d0 = *reinterpret_cast<const float4*>(&sdl[r][w * 8]);
        const float4 d1 = *reinterpret_cast<const float4*>(&sdl[r][w * 8 + 4]);
        acc[0] = fmaf(d0.x, hv, acc[0]);
        acc[1] = fmaf(d0.y, hv, acc[1]);
        acc[2] = fmaf(d0.z, hv, acc[2]);
        acc[3] = fmaf(d0.w, hv, acc[3]);
        acc[4] = fmaf(d1.x, hv, acc[4]);
        acc[5] = fmaf(d1.y, hv, acc[5]);
        acc[6] = fmaf(d1.z, hv, acc[6]);
        acc[7] = fmaf(d1.w, hv, acc[7]);
      }
    }
    if (ok) {
#pragma unroll
      for (int j = 0; j < HEAD_CMAX / 4; ++j) {
        const int c = w + 4 * j;
        if (c < C) dW[(size_t)c * H + n] = acc[j];
      }
    }
    return;
  }
  if (db == nullptr) return;
  // bias gradient: thread -> (class, one of 8 batch stripes); stripes are summed in a fixed order
  const int c = threadIdx.x & 31, st = threadIdx.x >> 5;
  float s = 0.f;
  if (c < C) {
#pragma unroll 8
    for (int b = st; b < B; b += 8) s += dl[(size_t)b * C + c];
  }
  sdl[st][c] = s;
  __syncthreads();
  if (threadIdx.x < C) {
    float t = sdl[0][c];
#pragma unroll
    for (int q = 1; q < 8; ++q) t += sdl[q][c];
    db[c] = t;
  }
}

}  // namespace

int head_max_classes() { return HEAD_CMAX; }

hipError_t launch_head_fwd(int B, int H, int C, const float* h, long long ldh, const float* W, const float* bias,
                           float* out, hipStream_t s) {
  hipLaunchKernelGGL(head_fwd_kernel, dim3(B), dim3(256), 0, s, H, C, h, ldh, W, bias, out);
  return hipGetLastError();
}

hipError_t launch_head_bwd(int B, int H, int C, const float* h, long long ldh, const float* W, const float* dl,
                           float* dh, float* dW, float* db, hipStream_t s) {
  const int nH = (H + 63) / 64;
  hipLaunchKernelGGL(head_bwd_kernel, dim3(B + nH + 1), dim3(256), 0, s, B, H, C, h, ldh, W, dl, dh, dW, db);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Cross-entropy of the classifier logits (mean over the rows whose target is not ignore_index), the loss the
// reference's training loop applies to Net's output (nn.CrossEntropyLoss, V/src/train_test/train.py:58-65).
// Stock PyTorch spends six launches on it (log_softmax, nll_loss, two fills and their two backward kernels);
// at B x C = 64 x 18 each of them is pure launch latency.  One workgroup, fixed summation order.
// ---------------------------------------------------------------------------------------------------
namespace {

// row statistics: lse[b] = log sum_c exp(z[b][c]);  loss = -(1/N) sum_valid (z[b][t_b] - lse[b]).
// Four lanes share a row (classes c = q, q + 4, ...: the loads of a row are independent and issued together, the row
// maximum and sum meet through two quad shuffles), 64 rows per pass of the single workgroup.  CQ = classes per lane
// held in registers (rows up to 4 CQ wide); wider rows re-read z (L1-resident) instead.
constexpr int CE_T = 1024;   // one workgroup of 16 waves: 256 rows per pass (the row loop is a chain of memory latencies:
                             // with 256 threads, 64 rows per pass, the kernel took 32 us at B = 512)
template <int CQ>
__global__ __launch_bounds__(CE_T) void ce_fwd_kernel(int B, int C, const float* __restrict__ z,
                                                     const long long* __restrict__ tgt, long long ignore_index,
                                                     float* __restrict__ loss, float* __restrict__ lse,
                                                     float* __restrict__ nvalid, float* __restrict__ dz_unit) {
  __shared__ float ssum[CE_T];
  __shared__ float scnt[CE_T];
  const int q = threadIdx.x & 3;
  float part = 0.f, cnt = 0.f;
  for (int b0 = 0; b0 < B; b0 += CE_T / 4) {
    const int b = b0 + (threadIdx.x >> 2);
    const bool rok = b < B;
    const float* zb = z + (size_t)(rok ? b : 0) * C;
    float v[CQ > 0 ? CQ : 1];
    float m = -INFINITY;
    if (CQ > 0) {
#pragma unroll
      for (int i = 0; i < CQ; ++i) {
        const int c = q + 4 * i;
        v[i] = (rok && c < C) ? zb[c] : -INFINITY;
        m = fmaxf(m, v[i]);
      }
    } else {
      for (int c = q; c < C; c += 4) m = fmaxf(m, rok ? zb[c] : -INFINITY);
    }
    m = fmaxf(m, __shfl_xor(m, 1, 64));
    m = fmaxf(m, __shfl_xor(m, 2, 64));
    float s = 0.f;
    if (CQ > 0) {
#pragma unroll
      for (int i = 0; i < CQ; ++i) s += expf(v[i] - m);   // exp(-inf) = 0 for the padding
    } else {
      for (int c = q; c < C; c += 4) s += rok ? expf(zb[c] - m) : 0.f;
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    const float l = m + logf(s);
    if (rok && q == 0) {
      lse[b] = l;
      const long long t = tgt[b];
      if (t != ignore_index) {
        // a class index outside [0, C) (PyTorch: device-side assert) poisons the loss instead of reading out of bounds
        const bool inr = t >= 0 && t < C;
        part += inr ? l - zb[inr ? t : 0] : NAN;
        cnt += 1.f;
      }
    }
  }
  ssum[threadIdx.x] = part;
  scnt[threadIdx.x] = cnt;
  __syncthreads();
  for (int o = CE_T / 2; o >= 1; o >>= 1) {   // fixed-shape tree: run-to-run identical
    if ((int)threadIdx.x < o) {
      ssum[threadIdx.x] += ssum[threadIdx.x + o];
      scnt[threadIdx.x] += scnt[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *nvalid = scnt[0];
    *loss = ssum[0] / scnt[0];   // 0/0 = NaN when every target is ignored, as PyTorch returns
  }
  // the gradient for d(loss) = 1, while the rows are still in cache: a backward whose incoming gradient is known to be
  // one needs no launch of its own (ce_bwd_kernel's arithmetic, same operation order)
  if (dz_unit != nullptr) {
    const float scale = 1.f / scnt[0];
    for (int e = threadIdx.x; e < B * C; e += CE_T) {
      const int b = e / C, c = e - b * C;
      const long long t = tgt[b];
      const float p = expf(z[e] - lse[b]);   // lse[b] was written by this workgroup before the barriers above
      dz_unit[e] = t == ignore_index ? 0.f : scale * (p - (c == (int)t ? 1.f : 0.f));
    }
  }
}

// dz[b][c] = (softmax(z[b])[c] - [c == t_b]) * dloss / N   (0 for ignored rows)
__global__ __launch_bounds__(256) void ce_bwd_kernel(int B, int C, const float* __restrict__ z,
                                                     const long long* __restrict__ tgt, long long ignore_index,
                                                     const float* __restrict__ lse,
                                                     const float* __restrict__ nvalid,
                                                     const float* __restrict__ dloss, float* __restrict__ dz) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)B * C) return;
  const int b = (int)(e / C), c = (int)(e - (long long)b * C);
  const long long t = tgt[b];
  const float scale = dloss[0] / nvalid[0];
  const float p = expf(z[e] - lse[b]);
  dz[e] = t == ignore_index ? 0.f : scale * (p - (c == (int)t ? 1.f : 0.f));
}

}  // namespace

hipError_t launch_ce_fwd(int B, int C, const float* z, const long long* tgt, long long ignore_index, float* loss,
                         float* lse, float* nvalid, float* dz_unit, hipStream_t s) {
  if (C <= 32)
    hipLaunchKernelGGL(ce_fwd_kernel<8>, dim3(1), dim3(CE_T), 0, s, B, C, z, tgt, ignore_index, loss, lse, nvalid, dz_unit);
  else
    hipLaunchKernelGGL(ce_fwd_kernel<0>, dim3(1), dim3(CE_T), 0, s, B, C, z, tgt, ignore_index, loss, lse, nvalid, dz_unit);
  return hipGetLastError();
}

hipError_t launch_ce_bwd(int B, int C, const float* z, const long long* tgt, long long ignore_index,
                         const float* lse, const float* nvalid, const float* dloss, float* dz, hipStream_t s) {
  const long long n = (long long)B * C;
  hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, B, C, z, tgt, ignore_index,
                     lse, nvalid, dloss, dz);
  return hipGetLastError();
}
