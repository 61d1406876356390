// Optimizer steps of the reference's two training loops as one launch over every parameter tensor (gfx950):
//   Adam            torch.optim.Adam(model.parameters(), lr)  V/src/train_test/train.py:47,65
//   clip + SGD      clip_grad_norm_ then  param -= lr * grad   V/src/train_test/lm_test.py:203-209
// The models have 10-20 small tensors (113 K parameters at the headline shape): the stock optimizer is pure
// launch / dispatcher latency (2 ms per step against 0.24 ms for forward + backward), the kernels here are
// bandwidth-trivial.  Pointers travel in the kernel-argument segment (no device-side table to maintain).
// SURVEY.md section 8f ("next" row).
#include <hip/hip_runtime.h>

#include "../../include/vmlmf_hip.h"

// the library's per-device gradient-health word and the guard mode (vmlmf_api.hip)
unsigned* vmlmf_health_word_if_any();
int vmlmf_adam_guard_mode();

namespace {

constexpr int GUARD_GO_W = 64, GUARD_SKIPPED_W = 66;   // words of the caller's guard block (VMLMF_GUARD_GO / _SKIPPED)
struct TensorList {
  float* p[VMLMF_MAX_TENSORS];
  const float* g[VMLMF_MAX_TENSORS];
  long long n[VMLMF_MAX_TENSORS];
  long long off[VMLMF_MAX_TENSORS];   // offset of the tensor's optimizer state inside the flat state buffers
  int sidx[VMLMF_MAX_TENSORS];        // which step counter is the tensor's
};

// torch.optim.Adam counts steps per parameter (one that receives its first gradient late starts at 1)
__global__ void tick_kernel(TensorList t, int count, float* steps) {
  if ((int)threadIdx.x < count) steps[t.sidx[threadIdx.x]] += 1.f;
}

// Guarded tick, default form (ABI 9): finish_kernel - the launch that writes every parameter gradient of a layer - sets the
// library's gradient-health word when one of them is not finite (the NaN partial products of a launch that gave up a bounded
// wait, VMLMF_E_PROTOCOL).  The tick reads the word: set -> nothing ticks, guard[GO] = 0 (adam_kernel returns at once), the step
// is counted as skipped and the word is cleared; clear -> the ordinary tick.  One extra load in a launch that exists anyway.
// ABI 10: the verdict is taken ONCE per optimizer step.  An optimizer step can be several launches (tensor lists of
// VMLMF_MAX_TENSORS, parameter groups): the FIRST one reads the health word and leaves the verdict in guard[GO], the others read
// guard[GO], and only the LAST one clears the health word - the first form cleared it at once, so the second list of a failed
// step saw a clean word and applied its NaN gradients.
__global__ void tick_health_kernel(TensorList t, int count, float* steps, unsigned* guard, unsigned* health, int flags) {
  const bool first = (flags & VMLMF_ADAM_FIRST) != 0, last = (flags & VMLMF_ADAM_LAST) != 0;
  const bool go = first ? (health != nullptr ? *health : 0u) == 0u : guard[GUARD_GO_W] != 0u;
  if (go && (int)threadIdx.x < count) steps[t.sidx[threadIdx.x]] += 1.f;
  if (threadIdx.x == 0) {
    if (first) {
      guard[GUARD_GO_W] = go ? 1u : 0u;
      if (!go) guard[GUARD_SKIPPED_W] += 1u;
    }
    if (last && health != nullptr) *health = 0u;
  }
}

// Guarded form of the tick, scanning form (vmlmf_tune("adam_guard", 2): gradients that did not come out of this library): a launch that gave up a bounded wait leaves NaN parameter gradients (vmlmf_hip.h:
// VMLMF_E_PROTOCOL), and the host only learns of it at its next call - inside a replayed hipGraph never.  An optimizer
// step over such gradients would poison the model for good, so the update is gated ON THE DEVICE: every workgroup scans a
// slice of every gradient for non-finite values and leaves a flag; the last one to arrive (ticket) decides - all clear: the
// step counters tick and guard[GO] = 1; otherwise nothing ticks, guard[GO] = 0 and guard[SKIPPED] counts the skipped step.
// adam_kernel returns at once when guard[GO] is 0: parameters and moments keep their values.
constexpr int GUARD_GO = 64, GUARD_TICKET = 65, GUARD_SKIPPED = 66;   // words of the guard block
// (The workgroups meet in ONE returning atomic: low half = arrivals, high half = workgroups that saw a non-finite value.  The
// last arriver reads the whole verdict from the value the atomic returns - no flag array, no release / acquire fences, which
// cost 1.7 - 3.5 us each on this chip, more than the scan itself.)
__global__ __launch_bounds__(256) void adam_gate_kernel(TensorList t, int count, float* steps, unsigned* guard) {
  __shared__ unsigned bad_s;
  __shared__ unsigned verdict_s;
  if (threadIdx.x == 0) bad_s = 0;
  __syncthreads();
  unsigned bad = 0;
  // sixteen tensors at a time, one element of each per pass: the loads of a pass are in flight together (tensor by tensor the
  // scan of the HAR net's ten small gradients was ten dependent memory round trips, 6 us)
  const long long stride = (long long)gridDim.x * 256, i0 = (long long)blockIdx.x * 256 + threadIdx.x;
  for (int t0 = 0; t0 < count; t0 += 16) {
    long long nmax = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const long long n = t0 + j < count ? t.n[t0 + j] : 0;
      nmax = n > nmax ? n : nmax;
    }
    for (long long i = i0; i < nmax; i += stride) {
      unsigned v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int tj = t0 + j < count ? t0 + j : t0;
        const long long n = t.n[tj];
        v[j] = reinterpret_cast<const unsigned*>(t.g[tj])[i < n ? i : 0];     // clamped: a repeated element changes nothing
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) bad |= ((v[j] & 0x7f800000u) == 0x7f800000u) ? 1u : 0u;   // exponent all ones: Inf or NaN
    }
  }
  if (bad) atomicOr(&bad_s, 1u);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned mine = 1u + (bad_s ? 0x10000u : 0u);
    const unsigned seen = atomicAdd(&guard[GUARD_TICKET], mine) + mine;     // arrivals and bad workgroups including this one
    verdict_s = (seen & 0xffffu) == gridDim.x ? (0x80000000u | (seen >> 16)) : 0u;
  }
  __syncthreads();
  const unsigned verdict = verdict_s;
  if (verdict == 0) return;                     // not the last workgroup
  const bool go = (verdict & 0x7fffffffu) == 0;
  if (go && (int)threadIdx.x < count) steps[t.sidx[threadIdx.x]] += 1.f;
  if (threadIdx.x == 0) {
    guard[GUARD_GO] = go ? 1u : 0u;
    guard[GUARD_TICKET] = 0;
    if (!go) guard[GUARD_SKIPPED] += 1u;
  }
}

// Same operation order as torch.optim.Adam's reference implementation (lerp for the first moment, mul + addcmul
// for the second, sqrt / sqrt(bias_correction2) + eps, addcdiv), so that results agree to rounding.
__global__ __launch_bounds__(256) void adam_kernel(TensorList t, float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ step, float lr, float b1, float b2,
                                                   float eps, float wd, const unsigned* __restrict__ guard) {
  if (guard != nullptr && guard[GUARD_GO] == 0) return;      // non-finite gradients: this step is skipped (adam_gate_kernel)
  const int ti = blockIdx.y;
  const long long n = t.n[ti];
  float* __restrict__ p = t.p[ti];
  const float* __restrict__ g = t.g[ti];
  float* __restrict__ mt = m + t.off[ti];
  float* __restrict__ vt = v + t.off[ti];
  const float s = step[t.sidx[ti]];
  const float bc1 = 1.f - powf(b1, s), bc2 = 1.f - powf(b2, s);
  const float step_size = lr / bc1, bc2s = sqrtf(bc2);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float gi = g[i];
    const float pi = p[i];
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    float mi = mt[i], vi = vt[i];
    mi = mi + (gi - mi) * (1.f - b1);
    vi = vi * b2 + (1.f - b2) * gi * gi;
    mt[i] = mi;
    vt[i] = vi;
    const float denom = sqrtf(vi) / bc2s + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}

// Tick, verdict and update in ONE launch (lists whose largest tensor has at most ADAM_FUSED_MAX elements - the HAR net's largest is
// 11 520 -: a tick launch in front of the update is pure latency there).  Every workgroup reads the step counter of its tensor
// and the verdict (FIRST: the health word; else guard[GO]) and nobody writes either before the LAST workgroup to arrive at a ticket
// does - a workgroup takes its ticket behind all its reads -: that workgroup ticks the counters, publishes the verdict and, on the
// LAST launch of the step, clears the health word.
constexpr long long ADAM_FUSED_MAX = 1 << 20;
constexpr int GUARD_TICKET_W = 65;
__global__ __launch_bounds__(256) void adam_fused_kernel(TensorList t, int count, float* __restrict__ m, float* __restrict__ v,
                                                         float* __restrict__ steps, float lr, float b1, float b2, float eps, float wd,
                                                         unsigned* __restrict__ guard, unsigned* __restrict__ health, int flags) {
  const int ti = blockIdx.y, tid = threadIdx.x;
  const bool first = (flags & VMLMF_ADAM_FIRST) != 0, last = (flags & VMLMF_ADAM_LAST) != 0;
  bool go = true;
  if (guard != nullptr) go = first ? (health != nullptr ? *health : 0u) == 0u : guard[GUARD_GO_W] != 0u;
  if (go) {
    const long long n = t.n[ti];
    float* __restrict__ p = t.p[ti];
    const float* __restrict__ g = t.g[ti];
    float* __restrict__ mt = m + t.off[ti];
    float* __restrict__ vt = v + t.off[ti];
    const float s = steps[t.sidx[ti]] + 1.f;
    const float bc1 = 1.f - powf(b1, s), bc2 = 1.f - powf(b2, s);
    const float step_size = lr / bc1, bc2s = sqrtf(bc2);
    for (long long i = (long long)blockIdx.x * 256 + tid; i < n; i += (long long)gridDim.x * 256) {   // adam_kernel's operation order
      float gi = g[i];
      const float pi = p[i];
      if (wd != 0.f) gi = fmaf(wd, pi, gi);
      float mi = mt[i], vi = vt[i];
      mi = mi + (gi - mi) * (1.f - b1);
      vi = vi * b2 + (1.f - b2) * gi * gi;
      mt[i] = mi;
      vt[i] = vi;
      p[i] = pi - step_size * (mi / (sqrtf(vi) / bc2s + eps));
    }
  }
  __shared__ unsigned last_s;
  if (tid == 0) {
    unsigned* tk = &guard[GUARD_TICKET_W];
    last_s = atomicAdd(tk, 1u) == gridDim.x * gridDim.y - 1 ? 1u : 0u;   // every other workgroup has read what it reads
    if (last_s) *tk = 0u;
  }
  __syncthreads();
  if (last_s == 0) return;
  if (go && tid < count) steps[t.sidx[tid]] += 1.f;
  if (guard != nullptr && tid == 0) {
    if (first) {
      guard[GUARD_GO_W] = go ? 1u : 0u;
      if (!go) guard[GUARD_SKIPPED_W] += 1u;
    }
    if (last && health != nullptr) *health = 0u;
  }
}

// sum of squares of every gradient: per-workgroup partials (fixed order), then one workgroup finishes
// (256 threads per workgroup, or 1024 for models with a tensor of a million elements: at most 64 workgroups walk a tensor - the
// scratch holds 64 partials for each - and the PTB network's 13 M gradient elements sit in two of them: 128 workgroups of four
// waves streamed 52 MB at 2.4 TB/s)
__global__ __launch_bounds__(1024) void sqsum_kernel(TensorList t, float* __restrict__ partial) {
  __shared__ float red[1024];
  const int NTH = (int)blockDim.x;
  const int ti = blockIdx.y;
  const long long n = t.n[ti];
  const float* __restrict__ g = t.g[ti];
  float s = 0.f;
  if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
    // 16-byte loads, four of them in flight per thread with an accumulator each: at most 64 workgroups walk a tensor
    // (the scratch holds 64 partials per tensor), so a 6.5 M-element vocabulary matrix needs every one of them to
    // stream (one dword load per iteration into one FMA chain ran at 0.35 TB/s)
    const long long n4 = n >> 2, stride = (long long)gridDim.x * NTH;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (long long i = (long long)blockIdx.x * NTH + threadIdx.x; i < n4; i += 4 * stride) {
      const long long i1 = i + stride, i2 = i + 2 * stride, i3 = i + 3 * stride;
      const float4 v0 = g4[i], v1 = g4[i1 < n4 ? i1 : i], v2 = g4[i2 < n4 ? i2 : i], v3 = g4[i3 < n4 ? i3 : i];
      a0 += (v0.x * v0.x + v0.y * v0.y) + (v0.z * v0.z + v0.w * v0.w);
      a1 += i1 < n4 ? (v1.x * v1.x + v1.y * v1.y) + (v1.z * v1.z + v1.w * v1.w) : 0.f;
      a2 += i2 < n4 ? (v2.x * v2.x + v2.y * v2.y) + (v2.z * v2.z + v2.w * v2.w) : 0.f;
      a3 += i3 < n4 ? (v3.x * v3.x + v3.y * v3.y) + (v3.z * v3.z + v3.w * v3.w) : 0.f;
    }
    s = (a0 + a1) + (a2 + a3);
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {   // the last n % 4 elements
      const float v = g[(n4 << 2) + threadIdx.x];
      s = fmaf(v, v, s);
    }
  } else {
    for (long long i = (long long)blockIdx.x * NTH + threadIdx.x; i < n; i += (long long)gridDim.x * NTH) s = fmaf(g[i], g[i], s);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = NTH / 2; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void norm_kernel(const float* __restrict__ partial, int count, float* __restrict__ norm) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < count; i += 256) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) norm[0] = sqrtf(red[0]);
}

// clip_grad_norm_ semantics: clip_coef = max_norm / (norm + 1e-6), clamped to 1; the gradients are scaled in
// place (the reference's loop reads param.grad afterwards) and the parameters take the plain SGD step.
__global__ __launch_bounds__(256) void sgd_clip_kernel(TensorList t, const float* __restrict__ norm, float lr,
                                                       float max_norm) {
  const int ti = blockIdx.y;
  const long long n = t.n[ti];
  float* __restrict__ p = t.p[ti];
  float* __restrict__ g = const_cast<float*>(t.g[ti]);
  // a non-finite norm (NaN gradients of a launch that gave up a bounded wait, VMLMF_E_PROTOCOL; an overflow): the step is
  // skipped - parameters AND gradients keep their values, `norm` tells the caller
  if (!isfinite(norm[0])) return;
  float coef = 1.f;
  if (max_norm > 0.f) {
    coef = max_norm / (norm[0] + 1e-6f);
    coef = coef < 1.f ? coef : 1.f;
  }
  // 16-byte accesses where both tensors allow them; an unclipped step (coef == 1) leaves the gradients as they are instead of
  // writing the same bits back (a third of the bytes: 156 MB -> 104 MB for the PTB network)
  const bool keep_g = coef == 1.f;
  long long done = 0;
  if (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g)) & 15) == 0) {
    const long long n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    float4* g4 = reinterpret_cast<float4*>(g);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
      float4 gv = g4[i], pv = p4[i];
      gv.x *= coef, gv.y *= coef, gv.z *= coef, gv.w *= coef;
      if (!keep_g) g4[i] = gv;
      pv.x = pv.x - lr * gv.x, pv.y = pv.y - lr * gv.y, pv.z = pv.z - lr * gv.z, pv.w = pv.w - lr * gv.w;
      p4[i] = pv;
    }
    done = n4 << 2;
  }
  for (long long i = done + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gi = g[i] * coef;
    if (!keep_g) g[i] = gi;
    p[i] = p[i] - lr * gi;
  }
}

int fill(const vmlmf_tensor_list* in, TensorList* t, long long* maxn) {
  if (in == nullptr || in->count < 1 || in->count > VMLMF_MAX_TENSORS) return VMLMF_E_BADARG;
  *maxn = 0;
  for (int i = 0; i < VMLMF_MAX_TENSORS; ++i) {
    const bool live = i < in->count;
    if (live && (in->param[i] == nullptr || in->grad[i] == nullptr || in->numel[i] < 0)) return VMLMF_E_BADARG;
    t->p[i] = live ? (float*)in->param[i] : nullptr;
    t->g[i] = live ? (const float*)in->grad[i] : nullptr;
    t->n[i] = live ? in->numel[i] : 0;
    t->off[i] = live ? in->state_offset[i] : 0;
    t->sidx[i] = live ? in->step_index[i] : 0;
    if (live && in->numel[i] > *maxn) *maxn = in->numel[i];
  }
  return 0;
}

unsigned blocks_for(long long maxn) {
  long long b = (maxn + 1023) / 1024;   // four elements per thread before the grid-stride loop wraps
  return (unsigned)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

}  // namespace

extern "C" {

int vmlmf_adam_step_ex(const vmlmf_tensor_list* tensors, float* exp_avg, float* exp_avg_sq, float* steps, float lr,
                       float beta1, float beta2, float eps, float weight_decay, void* guard, int flags, void* stream) {
  TensorList t;
  long long maxn = 0;
  const int rc = fill(tensors, &t, &maxn);
  if (rc != 0) return rc;
  if (exp_avg == nullptr || exp_avg_sq == nullptr || steps == nullptr) return VMLMF_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  const int mode = guard != nullptr ? vmlmf_adam_guard_mode() : 0;
  if (mode == 1 && maxn <= ADAM_FUSED_MAX) {   // one launch: tick, verdict and update (adam_fused_kernel; its ticket lives in the guard block)
    hipLaunchKernelGGL(adam_fused_kernel, dim3(blocks_for(maxn), tensors->count), dim3(256), 0, s, t, tensors->count, exp_avg, exp_avg_sq,
                       steps, lr, beta1, beta2, eps, weight_decay, (unsigned*)guard, vmlmf_health_word_if_any(), flags);
    return (int)hipGetLastError();
  }
  if (mode == 1) {
    hipLaunchKernelGGL(tick_health_kernel, dim3(1), dim3(64), 0, s, t, tensors->count, steps, (unsigned*)guard, vmlmf_health_word_if_any(), flags);
  } else if (mode == 2) {
    // enough workgroups that the largest tensor is one pass (the HAR net: 45 workgroups, ONE round of loads), at most 1024
    // (the arrivals' half of the ticket word holds 16 bits)
    long long nb = (maxn + 255) / 256;
    nb = nb < 1 ? 1 : (nb > 1024 ? 1024 : nb);
    hipLaunchKernelGGL(adam_gate_kernel, dim3((unsigned)nb), dim3(256), 0, s, t, tensors->count, steps, (unsigned*)guard);
  } else {
    hipLaunchKernelGGL(tick_kernel, dim3(1), dim3(64), 0, s, t, tensors->count, steps);
  }
  hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(maxn), tensors->count), dim3(256), 0, s, t, exp_avg, exp_avg_sq,
                     steps, lr, beta1, beta2, eps, weight_decay, mode != 0 ? (const unsigned*)guard : nullptr);
  return (int)hipGetLastError();
}

// (one call = one whole optimizer step)
int vmlmf_adam_step_guarded(const vmlmf_tensor_list* tensors, float* exp_avg, float* exp_avg_sq, float* steps, float lr,
                            float beta1, float beta2, float eps, float weight_decay, void* guard, void* stream) {
  return vmlmf_adam_step_ex(tensors, exp_avg, exp_avg_sq, steps, lr, beta1, beta2, eps, weight_decay, guard,
                            VMLMF_ADAM_FIRST | VMLMF_ADAM_LAST, stream);
}

int vmlmf_adam_step(const vmlmf_tensor_list* tensors, float* exp_avg, float* exp_avg_sq, float* steps, float lr,
                    float beta1, float beta2, float eps, float weight_decay, void* stream) {
  return vmlmf_adam_step_ex(tensors, exp_avg, exp_avg_sq, steps, lr, beta1, beta2, eps, weight_decay, nullptr,
                            VMLMF_ADAM_FIRST | VMLMF_ADAM_LAST, stream);
}

int vmlmf_sgd_clip_step(const vmlmf_tensor_list* tensors, float lr, float max_norm, float* norm, float* scratch,
                        void* stream) {
  TensorList t;
  long long maxn = 0;
  const int rc = fill(tensors, &t, &maxn);
  if (rc != 0) return rc;
  if (norm == nullptr || scratch == nullptr) return VMLMF_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  unsigned nb = blocks_for(maxn);
  if (nb > 64) nb = 64;   // scratch holds VMLMF_MAX_TENSORS * 64 partial sums
  hipLaunchKernelGGL(sqsum_kernel, dim3(nb, tensors->count), dim3(maxn >= (1LL << 20) ? 1024 : 256), 0, s, t, scratch);
  hipLaunchKernelGGL(norm_kernel, dim3(1), dim3(256), 0, s, scratch, (int)(nb * tensors->count), norm);
  hipLaunchKernelGGL(sgd_clip_kernel, dim3(blocks_for(maxn), tensors->count), dim3(256), 0, s, t, norm, lr, max_norm);
  return (int)hipGetLastError();
}

}  // extern "C"
