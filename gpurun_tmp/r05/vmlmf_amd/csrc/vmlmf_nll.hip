// Softmax negative log-likelihood over the vocabulary, the loss of the reference's language-model loop
// (nll_loss, V/src/train_test/lm_test.py:140-153: exp -> row sum -> divide -> gather -> log -> mean * batch_size;
// SURVEY section 8f rank 3).  scores is (R, V) with R = T*B rows (8960 x 10000 = 358 MB at BASELINE config E), so
// the loss is an HBM pass, not arithmetic: the reference's formulation reads or writes that matrix about eight times
// forward + backward.  Here forward reads it once (a row lives in its workgroup's registers between the max, the
// sum and the target pick), backward reads it once and writes the gradient once.  The log-sum-exp is taken around
// the row maximum, so rows the reference overflows on (a score above 88) stay finite here; everywhere else the
// results agree to fp32 rounding.  Fixed summation orders, no atomics.
#include <hip/hip_runtime.h>

#include "vmlmf_launch.h"

namespace {

constexpr int NLL_Q = 16;   // float4 per thread held in registers: rows up to 256 * 16 * 4 = 16384 wide

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// workgroup-wide reduction of one value per thread (256 threads), result broadcast; `red` has 4 floats
template <bool MAX>
__device__ __forceinline__ float block_reduce(float v, float* red) {
  v = MAX ? wave_max(v) : wave_sum(v);
  __syncthreads();   // red may still be read from the previous reduction
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return MAX ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
}

// one workgroup per row: lse[row] = log sum_v exp(scores[row][v]),  rowloss[row] = lse[row] - scores[row][y[row]]
template <bool VEC>
__global__ __launch_bounds__(256) void nll_rows_kernel(int V, const float* __restrict__ scores,
                                                       const long long* __restrict__ y, float* __restrict__ lse,
                                                       float* __restrict__ rowloss) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* z = scores + (size_t)row * V;
  float m = -INFINITY, s = 0.f;
  if (VEC) {   // V % 4 == 0, 16-byte aligned rows, V <= 16384: the row stays in registers
    const int nq = V >> 2;
    float4 v[NLL_Q];
#pragma unroll
    for (int i = 0; i < NLL_Q; ++i) {
      const int q = tid + 256 * i;
      const float4 t = reinterpret_cast<const float4*>(z)[q < nq ? q : 0];
      v[i] = q < nq ? t : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    }
#pragma unroll
    for (int i = 0; i < NLL_Q; ++i) m = fmaxf(m, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
    m = block_reduce<true>(m, red);
#pragma unroll
    for (int i = 0; i < NLL_Q; ++i)   // exp(-inf - m) = 0 for the padding
      s += (__expf(v[i].x - m) + __expf(v[i].y - m)) + (__expf(v[i].z - m) + __expf(v[i].w - m));
  } else {
    for (int c = tid; c < V; c += 256) m = fmaxf(m, z[c]);
    m = block_reduce<true>(m, red);
    for (int c = tid; c < V; c += 256) s += __expf(z[c] - m);
  }
  s = block_reduce<false>(s, red);
  if (tid == 0) {
    const float l = m + __logf(s);
    lse[row] = l;
    const long long t = y[row];   // outside [0, V): NaN loss, no out-of-bounds read (the reference's indexing asserts)
    const bool inr = t >= 0 && t < V;
    rowloss[row] = inr ? l - z[inr ? t : 0] : NAN;
  }
}

// loss = scale * sum_rows rowloss, summed in a fixed order by one workgroup
__global__ __launch_bounds__(256) void nll_sum_kernel(int R, float scale, const float* __restrict__ rowloss,
                                                      float* __restrict__ loss) {
  __shared__ float red[4];
  float part = 0.f;
  for (int r = threadIdx.x; r < R; r += 256) part += rowloss[r];
  const float total = block_reduce<false>(part, red);
  if (threadIdx.x == 0) *loss = scale * total;
}

// dscores[row][v] = dloss * scale * (exp(scores[row][v] - lse[row]) - [v == y[row]])
template <bool VEC>
__global__ __launch_bounds__(256) void nll_bwd_kernel(int V, float scale, const float* __restrict__ scores,
                                                      const long long* __restrict__ y, const float* __restrict__ lse,
                                                      const float* __restrict__ dloss, float* __restrict__ dscores) {
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* z = scores + (size_t)row * V;
  float* dz = dscores + (size_t)row * V;
  const float l = lse[row], gsc = dloss[0] * scale;
  const int t = (int)y[row];
  if (VEC) {
    const int nq = V >> 2;
    for (int q0 = 0; q0 < nq; q0 += 256 * 4) {   // four 16-byte loads in flight per thread
      float4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {   // clamped index, no condition: a load under `if` waits for the one before it
        const int q = q0 + tid + 256 * i;
        v[i] = reinterpret_cast<const float4*>(z)[q < nq ? q : 0];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = q0 + tid + 256 * i;
        if (q < nq) {
          float4 o;
          o.x = gsc * (__expf(v[i].x - l) - (4 * q + 0 == t ? 1.f : 0.f));
          o.y = gsc * (__expf(v[i].y - l) - (4 * q + 1 == t ? 1.f : 0.f));
          o.z = gsc * (__expf(v[i].z - l) - (4 * q + 2 == t ? 1.f : 0.f));
          o.w = gsc * (__expf(v[i].w - l) - (4 * q + 3 == t ? 1.f : 0.f));
          reinterpret_cast<float4*>(dz)[q] = o;
        }
      }
    }
  } else {
    for (int c = tid; c < V; c += 256) dz[c] = gsc * (__expf(z[c] - l) - (c == t ? 1.f : 0.f));
  }
}

// ---- training form (ABI 9): loss AND the gradient of the scores in ONE pass over the matrix, in place ----------------------
// The LM head's backward needs dscores = scale (softmax - onehot) and the bias gradient (its column sums); the two-kernel
// form above reads the scores twice, writes a second 358 MB matrix and leaves the column sums to a third pass.  Here a
// workgroup walks rows w, w + NWG, ...: a row (plus the bias, which the GEMM in front then need not add) lives in its
// registers between max, sum and target pick, is written back IN PLACE as its own gradient (for d(loss) = 1; the caller
// scales otherwise), and every thread keeps the column sums of the columns it owns over all its rows.  One read + one write
// of the matrix; NWG x V column partials (fixed-order sum in nll_finish_kernel).
constexpr int NLL_GQ = 12;   // float4 per thread: rows up to 256 * 12 * 4 = 12288 wide (192 VGPRs of row + bias + column sums)
__global__ __launch_bounds__(256, 2) void nll_grad_kernel(int R, int V, float scale, float* __restrict__ scores,
                                                          const float* __restrict__ bias, const long long* __restrict__ y,
                                                          float* __restrict__ rowloss, float* __restrict__ dbpart) {
  __shared__ float red[4];
  const int tid = threadIdx.x, nq = V >> 2;
  float4 bv[NLL_GQ], cs[NLL_GQ];
#pragma unroll
  for (int i = 0; i < NLL_GQ; ++i) {
    const int q = tid + 256 * i;
    const float4 t = bias != nullptr ? reinterpret_cast<const float4*>(bias)[q < nq ? q : 0] : make_float4(0.f, 0.f, 0.f, 0.f);
    bv[i] = q < nq ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    cs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int row = blockIdx.x; row < R; row += gridDim.x) {
    float* z = scores + (size_t)row * V;
    float4 v[NLL_GQ];
#pragma unroll
    for (int i = 0; i < NLL_GQ; ++i) {   // clamped index: every load unconditional
      const int q = tid + 256 * i;
      v[i] = reinterpret_cast<const float4*>(z)[q < nq ? q : 0];
    }
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NLL_GQ; ++i) {
      const int q = tid + 256 * i;
      v[i].x += bv[i].x, v[i].y += bv[i].y, v[i].z += bv[i].z, v[i].w += bv[i].w;
      if (q >= nq) v[i] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      m = fmaxf(m, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
    }
    m = block_reduce<true>(m, red);
    const long long t = y[row];
    const bool inr = t >= 0 && t < V;
    const int tc = inr ? (int)t : -1;
    float s = 0.f, ztm = 0.f;   // ztm: (target's score - m), held by the thread that owns the target's column
#pragma unroll
    for (int i = 0; i < NLL_GQ; ++i) {   // keep the exponentials: they are the softmax numerators
      const int q = tid + 256 * i;
      if ((tc >> 2) == q) ztm = ((tc & 3) == 0 ? v[i].x : (tc & 3) == 1 ? v[i].y : (tc & 3) == 2 ? v[i].z : v[i].w) - m;
      v[i].x = __expf(v[i].x - m), v[i].y = __expf(v[i].y - m), v[i].z = __expf(v[i].z - m), v[i].w = __expf(v[i].w - m);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    s = block_reduce<false>(s, red);
    const float inv = scale / s;
#pragma unroll
    for (int i = 0; i < NLL_GQ; ++i) {
      const int q = tid + 256 * i;
      if (q < nq) {
        if ((tc >> 2) == q) rowloss[row] = __logf(s) - ztm;   // lse - z_t
        float4 o;
        o.x = v[i].x * inv - (4 * q + 0 == tc ? scale : 0.f);
        o.y = v[i].y * inv - (4 * q + 1 == tc ? scale : 0.f);
        o.z = v[i].z * inv - (4 * q + 2 == tc ? scale : 0.f);
        o.w = v[i].w * inv - (4 * q + 3 == tc ? scale : 0.f);
        reinterpret_cast<float4*>(z)[q] = o;
        cs[i].x += o.x, cs[i].y += o.y, cs[i].z += o.z, cs[i].w += o.w;
      }
    }
    if (!inr && tid == 0) rowloss[row] = NAN;   // a target outside [0, V): NaN loss (the reference's indexing raises)
  }
#pragma unroll
  for (int i = 0; i < NLL_GQ; ++i) {
    const int q = tid + 256 * i;
    if (q < nq) reinterpret_cast<float4*>(dbpart + (size_t)blockIdx.x * V)[q] = cs[i];
  }
}
// dbias[v] = sum over the workgroups' column partials (fixed order); block 0 also finishes the loss.  A workgroup takes 32
// columns, its eight 32-lane groups an eighth of the partial rows each (eight loads in flight), the eight sums meet in LDS in
// group order.  (One thread per column walking all 512 partial rows was a chain of 64 round trips on 40 workgroups: 34 us.)
__global__ __launch_bounds__(256) void nll_finish_kernel(int R, int V, int nwg, float scale, const float* __restrict__ rowloss,
                                                         const float* __restrict__ dbpart, float* __restrict__ dbias,
                                                         float* __restrict__ loss) {
  __shared__ float red[4];
  __shared__ float col[8][32];
  const int li = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + li, cc = c < V ? c : V - 1;
  if (dbias != nullptr) {
    const int per = (nwg + 7) / 8;
    const int w0 = grp * per < nwg ? grp * per : nwg, w1 = w0 + per < nwg ? w0 + per : nwg;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    int w = w0;
    for (; w + 7 < w1; w += 8) {
      float t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = dbpart[(size_t)(w + i) * V + cc];
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i & 3] += t[i];
    }
    for (; w < w1; ++w) a[0] += dbpart[(size_t)w * V + cc];
    col[grp][li] = (a[0] + a[1]) + (a[2] + a[3]);
    __syncthreads();
    if (grp == 0 && c < V) {
      float total = col[0][li];
#pragma unroll
      for (int q = 1; q < 8; ++q) total += col[q][li];
      dbias[c] = total;
    }
  }
  if (blockIdx.x == 0) {
    float part = 0.f;
    for (int r = threadIdx.x; r < R; r += 256) part += rowloss[r];
    const float total = block_reduce<false>(part, red);
    if (threadIdx.x == 0) *loss = scale * total;
  }
}

bool vec_ok(int V, const float* a, const float* b) {
  return V % 4 == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0 && (b == nullptr || (reinterpret_cast<uintptr_t>(b) & 15) == 0);
}

}  // namespace

hipError_t launch_nll_fwd(int R, int V, const float* scores, const long long* y, float scale, float* loss, float* lse,
                          float* rowloss, hipStream_t s) {
  if (vec_ok(V, scores, nullptr) && V <= 256 * NLL_Q * 4)
    hipLaunchKernelGGL(nll_rows_kernel<true>, dim3(R), dim3(256), 0, s, V, scores, y, lse, rowloss);
  else
    hipLaunchKernelGGL(nll_rows_kernel<false>, dim3(R), dim3(256), 0, s, V, scores, y, lse, rowloss);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(nll_sum_kernel, dim3(1), dim3(256), 0, s, R, scale, rowloss, loss);
  return hipGetLastError();
}

hipError_t launch_nll_bwd(int R, int V, const float* scores, const long long* y, float scale, const float* lse,
                          const float* dloss, float* dscores, hipStream_t s) {
  if (vec_ok(V, scores, dscores))
    hipLaunchKernelGGL(nll_bwd_kernel<true>, dim3(R), dim3(256), 0, s, V, scale, scores, y, lse, dloss, dscores);
  else
    hipLaunchKernelGGL(nll_bwd_kernel<false>, dim3(R), dim3(256), 0, s, V, scale, scores, y, lse, dloss, dscores);
  return hipGetLastError();
}

// workgroups of nll_grad_kernel (two per CU: 192 + VGPRs each); the column partials need NWG x V floats of scratch
int nll_grad_workgroups(int R) { return R < 512 ? R : 512; }

// VMLMF_E_UNSUPPORTED (-3) when the in-register form does not cover the row width
int launch_nll_fwd_grad(int R, int V, float* scores, const float* bias, const long long* y, float scale, float* loss,
                        float* rowloss, float* dbias, float* scratch, hipStream_t s) {
  if (!vec_ok(V, scores, bias) || V > 256 * NLL_GQ * 4) return -3;
  const int nwg = nll_grad_workgroups(R);
  hipLaunchKernelGGL(nll_grad_kernel, dim3(nwg), dim3(256), 0, s, R, V, scale, scores, bias, y, rowloss, scratch);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(nll_finish_kernel, dim3((V + 31) / 32), dim3(256), 0, s, R, V, nwg, scale, rowloss, scratch, dbias, loss);
  return (int)hipGetLastError();
}
