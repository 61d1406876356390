// Dropout launches of the LM network for the places no layer kernel covers (vmlmf_dropout.h has the scheme): the generator's
// per-forward snapshot, the embedding gather with its dropout in the same pass (vmlmf_lm.py:434-435), the stand-alone form for
// activations of layers whose kernels do not take DropArgs (y = x * factor: the forward and, with the same snapshot, the backward),
// and the factors themselves for the parity tests (what a fused kernel applied, as a tensor an oracle can multiply by).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vmlmf_dropout.h"
#include "vmlmf_launch.h"

namespace {

__global__ void drop_advance_kernel(unsigned long long* state, unsigned long long* snap) {
  const unsigned long long seed = state[0], off = state[1];
  snap[0] = seed, snap[1] = off;
  state[1] = off + 1ull;
}

// one thread per four columns of a position; MODE 0: y = x * factor, MODE 1: y = factor, MODE 2: y = w[tokens[position]] * factor
template <int MODE>
__global__ void __launch_bounds__(256) drop_rows_kernel(long long R, int H, int V, DropArgs d, DropCols cm, const float* __restrict__ x,
                                                        const long long* __restrict__ tokens, float* __restrict__ y) {
  const int quads = (H + 3) >> 2;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= R * quads) return;
  const long long r = i / quads;
  const int n0 = (int)(i - r * quads) * 4;
  const DropKey k = drop_key(d);
  const float* src = x;
  if (MODE == 2) {
    long long t = tokens[r];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);   // (the stock gather faults on such a token; here it reads a valid row)
    src = x + (size_t)t * H;
  } else if (MODE == 0) {
    src = x + (size_t)r * H;
  }
  float* dst = y + (size_t)r * H;
  const bool ident = cm.gstride == 0 || cm.Hg >= H;
  if (ident) {   // one call for the thread's four columns; 16-byte accesses where the rows allow them
    float f[4];
    drop_factors(k, d.thresh, d.scale, (unsigned)r, (unsigned)(n0 >> 2), f);
    if ((H & 3) == 0) {
      float4 v = make_float4(1.f, 1.f, 1.f, 1.f);
      if (MODE != 1) v = *reinterpret_cast<const float4*>(src + n0);
      *reinterpret_cast<float4*>(dst + n0) = make_float4(v.x * f[0], v.y * f[1], v.z * f[2], v.w * f[3]);
    } else {   // (H = 650 of the PTB network: rows start 8 bytes off)
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (MODE != 1 && n0 + e < H) ? src[n0 + e] : 1.f;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n0 + e < H) dst[n0 + e] = v[e] * f[e];
    }
    return;
  }
  // mapped columns (the factors a row-block layer's kernels apply, as a tensor): a call per element
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int n = n0 + e;
    if (n >= H) break;
    const int col = (n / cm.Hg) * cm.gstride + n % cm.Hg;
    float f[4];
    drop_factors(k, d.thresh, d.scale, (unsigned)r, (unsigned)(col >> 2), f);
    const float fe = f[col & 3];
    dst[n] = MODE == 1 ? fe : src[n] * fe;
  }
}

}  // namespace

int launch_drop_advance(unsigned long long* state, unsigned long long* snap, hipStream_t s) {
  hipLaunchKernelGGL(drop_advance_kernel, dim3(1), dim3(1), 0, s, state, snap);
  return (int)hipGetLastError();
}

int launch_drop_rows(int mode, long long R, int H, int V, const DropArgs& d, const DropCols& cm, const float* x, const long long* tokens, float* y,
                     hipStream_t s) {
  const long long n = R * ((H + 3) / 4);
  if (n <= 0) return 0;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (mode == 0) hipLaunchKernelGGL(drop_rows_kernel<0>, grid, block, 0, s, R, H, V, d, cm, x, tokens, y);
  else if (mode == 1) hipLaunchKernelGGL(drop_rows_kernel<1>, grid, block, 0, s, R, H, V, d, cm, x, tokens, y);
  else hipLaunchKernelGGL(drop_rows_kernel<2>, grid, block, 0, s, R, H, V, d, cm, x, tokens, y);
  return (int)hipGetLastError();
}
