// Embedding-table gradient of the LM network (Embed, V/src/models/vmlmf_lm.py:46-48: x = w[tokens]; autograd's backward is an
// index scatter-add of dx into a (V, H) zero matrix) - SURVEY section 8f rank 3, verdict r3 item 5.  Deterministic and
// without float atomics:
//   1. embed_mark_kernel   one bit per (vocabulary row, position): bits[v][p] = 1 iff tokens[p] == v (integer atomicOr:
//                          the RESULT does not depend on the order)
//   2. embed_bwd_kernel    one wave per vocabulary row: walks its bit row in ascending position order and sums the dx rows
//                          of exactly those positions - so every gradient row is summed in position order, and rows no token
//                          hit are written as zeros (the zero fill of the 26 MB matrix is part of the same pass).
// Traffic at config E (R = 8960 positions, V = 10000, H = 650): 11 MB of bits written + read, dx read once (23 MB), dW written
// once (26 MB).  The stock path sorts the 8960 keys (ten launches), zero-fills and scatters.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vmlmf_dropout.h"
#include "vmlmf_launch.h"

namespace {

__global__ __launch_bounds__(256) void embed_mark_kernel(int R, int V, int words, const long long* __restrict__ tokens,
                                                         unsigned* __restrict__ bits) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= R) return;
  const long long t = tokens[p];
  if (t < 0 || t >= V) return;   // (the forward's gather would have faulted; nothing to add)
  atomicOr(bits + (size_t)t * words + (p >> 5), 1u << (p & 31));
}

// one wave per vocabulary row; lane l owns columns l, l + 64, ... (H <= 64 * EMB_C)
constexpr int EMB_C = 16;
__global__ __launch_bounds__(256) void embed_bwd_kernel(int R, int H, int V, int words, const unsigned* __restrict__ bits,
                                                        const float* __restrict__ dy, float* __restrict__ dW) {
  const int lane = threadIdx.x & 63;
  const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (v >= V) return;
  const unsigned* brow = bits + (size_t)v * words;
  float acc[EMB_C];
#pragma unroll
  for (int c = 0; c < EMB_C; ++c) acc[c] = 0.f;
  for (int w0 = 0; w0 < words; w0 += 64) {
    const int wi = w0 + lane;
    const unsigned mine = wi < words ? brow[wi] : 0u;
    unsigned long long any = __ballot(mine != 0u);
    while (any != 0ull) {                       // lanes holding set bits, in ascending word order
      const int src = __ffsll((long long)any) - 1;
      any &= any - 1ull;
      unsigned word = (unsigned)__shfl((int)mine, src, 64);
      while (word != 0u) {                      // positions in ascending order
        const int bit = __ffs((int)word) - 1;
        word &= word - 1u;
        const int p = (w0 + src) * 32 + bit;
        const float* row = dy + (size_t)p * H;
#pragma unroll
        for (int c = 0; c < EMB_C; ++c) {
          const int col = lane + 64 * c;
          if (col < H) acc[c] += row[col];
        }
      }
    }
  }
  float* out = dW + (size_t)v * H;
#pragma unroll
  for (int c = 0; c < EMB_C; ++c) {
    const int col = lane + 64 * c;
    if (col < H) out[col] = acc[c];
  }
}

// the same sum when the embedding's output went through dropout (vmlmf_lm.py:434-435): dy is the gradient of the DROPPED copy, and
// the factor of every (position, column) is regenerated here (vmlmf_dropout.h) - no mask tensor, no pass of its own.  A lane owns
// four neighbouring columns (one generator call), 256 columns per round: H <= 1024 (16-byte accesses when H is a multiple of four).
__global__ __launch_bounds__(256) void embed_bwd_drop_kernel(int R, int H, int V, int words, const unsigned* __restrict__ bits,
                                                             const float* __restrict__ dy, float* __restrict__ dW, DropArgs d) {
  const int lane = threadIdx.x & 63;
  const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (v >= V) return;
  const DropKey key = drop_key(d);
  const bool vec = (H & 3) == 0;
  const unsigned* brow = bits + (size_t)v * words;
  float4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int w0 = 0; w0 < words; w0 += 64) {
    const int wi = w0 + lane;
    const unsigned mine = wi < words ? brow[wi] : 0u;
    unsigned long long any = __ballot(mine != 0u);
    while (any != 0ull) {
      const int src = __ffsll((long long)any) - 1;
      any &= any - 1ull;
      unsigned word = (unsigned)__shfl((int)mine, src, 64);
      while (word != 0u) {
        const int bit = __ffs((int)word) - 1;
        word &= word - 1u;
        const int p = (w0 + src) * 32 + bit;
        const float* row = dy + (size_t)p * H;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int col = 4 * lane + 256 * c;
          if (col < H) {
            float f[4];
            drop_factors(key, d.thresh, d.scale, (unsigned)p, (unsigned)(col >> 2), f);
            float4 g;
            if (vec) g = *reinterpret_cast<const float4*>(row + col);
            else g = make_float4(row[col], col + 1 < H ? row[col + 1] : 0.f, col + 2 < H ? row[col + 2] : 0.f, col + 3 < H ? row[col + 3] : 0.f);
            acc[c].x += g.x * f[0], acc[c].y += g.y * f[1], acc[c].z += g.z * f[2], acc[c].w += g.w * f[3];
          }
        }
      }
    }
  }
  float* out = dW + (size_t)v * H;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int col = 4 * lane + 256 * c;
    if (col >= H) continue;
    if (vec) {
      *reinterpret_cast<float4*>(out + col) = acc[c];
    } else {
      out[col] = acc[c].x;
      if (col + 1 < H) out[col + 1] = acc[c].y;
      if (col + 2 < H) out[col + 2] = acc[c].z;
      if (col + 3 < H) out[col + 3] = acc[c].w;
    }
  }
}

}  // namespace

size_t embed_bwd_scratch_bytes(int R, int V) { return (size_t)V * (size_t)((R + 31) / 32) * sizeof(unsigned); }

int launch_embed_bwd(int R, int H, int V, const long long* tokens, const float* dy, float* dW, void* scratch, size_t scratch_bytes,
                     hipStream_t s, const DropArgs* drop) {
  if (H > 64 * EMB_C) return -3;
  const int words = (R + 31) / 32;
  const size_t need = embed_bwd_scratch_bytes(R, V);
  if (scratch == nullptr || scratch_bytes < need) return -4;
  hipError_t e = hipMemsetAsync(scratch, 0, need, s);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(embed_mark_kernel, dim3((R + 255) / 256), dim3(256), 0, s, R, V, words, tokens, (unsigned*)scratch);
  if (drop != nullptr)
    hipLaunchKernelGGL(embed_bwd_drop_kernel, dim3((V + 3) / 4), dim3(256), 0, s, R, H, V, words, (const unsigned*)scratch, dy, dW, *drop);
  else
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((V + 3) / 4), dim3(256), 0, s, R, H, V, words, (const unsigned*)scratch, dy, dW);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// dst (cols x rows) = src (rows x cols)^T, fp32, through 64 x 64 LDS tiles (both sides in whole 256-byte row segments).  The LM
// head's weight gradient is fastest as the library GEMM that yields dW^T (H x V); the strided copy that turned it into dW took
// 43 us for 26 MB (tools/sessions/r04p.sh) - this takes the bytes' time.
__global__ void __launch_bounds__(256) transpose_kernel(int rows, int cols, const float* __restrict__ src, float* __restrict__ dst) {
  __shared__ float tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty + 4 * i, c = c0 + tx;
    if (r < rows && c < cols) tile[ty + 4 * i][tx] = src[(size_t)r * cols + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + 4 * i, r = r0 + tx;
    if (r < rows && c < cols) dst[(size_t)c * rows + r] = tile[tx][ty + 4 * i];
  }
}

int launch_transpose(int rows, int cols, const float* src, float* dst, hipStream_t s) {
  hipLaunchKernelGGL(transpose_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, s, rows, cols, src, dst);
  return (int)hipGetLastError();
}
