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
