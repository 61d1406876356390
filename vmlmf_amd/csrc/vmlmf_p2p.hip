// One-shot peer-to-peer all-reduce of a SMALL flat gradient buffer (ABI 13; SURVEY.md section 8e: the HAR network's 121 KiB,
// train.py:64-65 - the exchange sits between loss.backward() and optimizer.step()).  The reference has no distributed code.
//
// Why beside RCCL: the buffer is latency-bound (a ring all-reduce of 121 KiB over 8 ranks is 14 dependent hops, ~30 us, fully
// exposed behind the finishing launch).  Here every rank WRITES its buffer into a slot of every peer's staging area over xGMI (one
// hop, all peers at once), raises one epoch word per peer, and then sums the `world` slots of its OWN staging area in rank order:
// two launches on the caller's stream, no host involvement, the same bits on every rank.
//
//   staging (per rank, hipMalloc'ed, exported with hipIpcGetMemHandle, mapped by every peer):
//     data  [2 parities][world slots][cap floats]     slot r of parity p: rank r's buffer of an exchange with (epoch & 1) == p
//     flags [2][world][32 words]                      word 0 of line (p, r): the epoch rank r has completely written into slot (p, r)
//   Two parities: a rank can only start exchange e + 2 after it has finished e + 1, which needed every peer's push of e + 1, which a
//   peer enqueues behind its own sum of e - so nobody overwrites a slot that is still being read.
//   The epoch lives in device memory and is advanced by the sum launch: a captured pair of launches replays correctly.
//   Every wait is bounded: a peer that never arrives leaves NaN in the buffer and VMLMF_ST_P2P in the status word
//   (the next C-ABI call returns VMLMF_E_PROTOCOL).
// Never exercised across GPUs (one GPU per lease): the test runs two processes on ONE device over the same IPC path
// (tests/test_gpu_rehearsal.py); torch.distributed stays the default transport of vmlmf_amd.dp.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>

#include "../../include/vmlmf_hip.h"
#include "vmlmf_launch.h"
#include "vmlmf_device.h"

int vmlmf_set_error(int code, const std::string& msg);   // vmlmf_api.hip
unsigned* vmlmf_status_word(void* stream);                // vmlmf_api.hip (mapped host word, or NULL)

namespace {

constexpr int P2P_LINE = 32;              // unsigned words per flag line (128 bytes)
constexpr unsigned P2P_SPIN = 1u << 22;   // looks at a peer's word before giving up (with s_sleep between: seconds)

struct P2P {
  int rank, world;
  size_t cap;                             // floats per slot
  float* mine;                            // this rank's staging area
  float* peer[VMLMF_P2P_MAX_RANKS];       // every rank's staging area as mapped here (peer[rank] == mine)
  unsigned* dev;                          // device words: [0] epoch of the last finished exchange, [1 + r] push tickets, [16] sum ticket
  bool connected;
};

__host__ __device__ inline size_t p2p_data_floats(int world, size_t cap) { return (size_t)2 * world * cap; }
__host__ __device__ inline size_t p2p_total_bytes(int world, size_t cap) {
  return p2p_data_floats(world, cap) * sizeof(float) + (size_t)2 * world * P2P_LINE * sizeof(unsigned);
}

struct P2PArgs {
  float* peer[VMLMF_P2P_MAX_RANKS];
  float* buf;
  unsigned* dev;
  unsigned* status;
  size_t cap, n;
  int rank, world, nblk, avg;
};

__device__ __forceinline__ float4 p2p_ld4_sys(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return make_float4(v[0], v[1], v[2], v[3]);
}
// system-scope write-through stores to a per-lane address (the peer's memory): as asm so that the scope bits are what they say;
// the wide store keeps two wait states behind it (its data registers: vmlmf_device.h, hazard (1))
__device__ __forceinline__ void p2p_st4_sys(float* p, const float4 v) {
  const f32x4 t = f32x4{v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ void p2p_st1_sys(float* p, const float v) {
  asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ float p2p_ld1_sys(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// grid (nblk, world): block (x, r) writes this rank's buffer into slot (parity, rank) of peer r; the last block per peer raises the word
__global__ void __launch_bounds__(256) p2p_push_kernel(P2PArgs a) {
  const int r = (int)blockIdx.y, tid = (int)threadIdx.x;
  const unsigned epoch = __hip_atomic_load(a.dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  const unsigned par = epoch & 1u;
  float* dst = a.peer[r] + ((size_t)par * a.world + a.rank) * a.cap;
  const size_t n4 = a.n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n4; i += (size_t)a.nblk * 256) {
    const float4 v = *reinterpret_cast<const float4*>(a.buf + 4 * i);
    p2p_st4_sys(dst + 4 * i, v);
  }
  if (blockIdx.x == 0 && tid < (int)(a.n - 4 * n4)) p2p_st1_sys(dst + 4 * n4 + tid, a.buf[4 * n4 + tid]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the write-through stores above are inline asm: drained here, then published
  __threadfence_system();
  __syncthreads();
  if (tid == 0) {
    const unsigned t = atomicAdd(a.dev + 1 + r, 1u);
    if (t == (unsigned)a.nblk - 1u) {
      a.dev[1 + r] = 0u;
      unsigned* flags = reinterpret_cast<unsigned*>(a.peer[r] + p2p_data_floats(a.world, a.cap));
      __hip_atomic_store(flags + ((size_t)par * a.world + a.rank) * P2P_LINE, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// grid nblk: waits for every rank's word of this exchange, sums the slots of this rank's staging area in rank order into buf
__global__ void __launch_bounds__(256) p2p_sum_kernel(P2PArgs a) {
  const int tid = (int)threadIdx.x;
  const unsigned epoch = __hip_atomic_load(a.dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  const unsigned par = epoch & 1u;
  const float* mine = a.peer[a.rank];
  const unsigned* flags = reinterpret_cast<const unsigned*>(mine + p2p_data_floats(a.world, a.cap));
  int ok = 1;
  if (tid < a.world) {
    unsigned spins = 0;
    while (__hip_atomic_load(flags + ((size_t)par * a.world + tid) * P2P_LINE, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
      __builtin_amdgcn_s_sleep(32);
      if (++spins > P2P_SPIN) {   // bounded: a peer that never arrives must not hang the GPU
        ok = 0;
        break;
      }
    }
  }
  ok = __syncthreads_and(ok);
  if (!ok && tid == 0 && blockIdx.x == 0) vg_raise(a.status, VMLMF_ST_P2P);
  const float scale = a.avg ? 1.f / (float)a.world : 1.f;
  const float bad = __int_as_float(0x7fc00000);
  const size_t n4 = a.n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n4; i += (size_t)a.nblk * 256) {
    float4 s = p2p_ld4_sys(mine + ((size_t)par * a.world) * a.cap + 4 * i);
    for (int r = 1; r < a.world; ++r) {
      const float4 v = p2p_ld4_sys(mine + ((size_t)par * a.world + r) * a.cap + 4 * i);
      s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
    if (!ok) s = make_float4(bad, bad, bad, bad);
    *reinterpret_cast<float4*>(a.buf + 4 * i) = make_float4(s.x * scale, s.y * scale, s.z * scale, s.w * scale);
  }
  if (blockIdx.x == 0 && tid < (int)(a.n - 4 * n4)) {
    const size_t i = 4 * n4 + tid;
    float s = p2p_ld1_sys(mine + ((size_t)par * a.world) * a.cap + i);
    for (int r = 1; r < a.world; ++r) s += p2p_ld1_sys(mine + ((size_t)par * a.world + r) * a.cap + i);
    a.buf[i] = ok ? s * scale : bad;
  }
  // the exchange is over for this rank when every block has read its share: the last one advances the epoch
  __syncthreads();
  if (tid == 0) {
    const unsigned t = atomicAdd(a.dev + 16, 1u);
    if (t == (unsigned)a.nblk - 1u) {
      a.dev[16] = 0u;
      __hip_atomic_store(a.dev, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

int hipfail(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  (void)hipGetLastError();
  return vmlmf_set_error((int)e, std::string("p2p: ") + what + ": " + hipGetErrorString(e));
}

}  // namespace

extern "C" {

int vmlmf_p2p_create(void** p2p, int rank, int world, size_t max_floats, unsigned char* handle_out) {
  if (p2p == nullptr || handle_out == nullptr) return vmlmf_set_error(VMLMF_E_BADARG, "p2p_create: null pointer");
  if (world < 1 || world > VMLMF_P2P_MAX_RANKS || rank < 0 || rank >= world)
    return vmlmf_set_error(VMLMF_E_BADARG, "p2p_create: 1 <= world <= 8 ranks, 0 <= rank < world");
  if (max_floats < 1 || max_floats > ((size_t)1 << 22))
    return vmlmf_set_error(VMLMF_E_UNSUPPORTED, "p2p_create: a buffer of 1 .. 4 Mi floats (the exchange is for small, latency-bound buffers; RCCL for the rest)");
  static_assert(sizeof(hipIpcMemHandle_t) <= VMLMF_P2P_HANDLE_BYTES, "handle size");
  P2P* h = new P2P();
  memset(h, 0, sizeof(*h));
  h->rank = rank, h->world = world, h->cap = (max_floats + 3) / 4 * 4;
  const size_t bytes = p2p_total_bytes(world, h->cap);
  int rc = hipfail(hipMalloc((void**)&h->mine, bytes), "hipMalloc(staging)");
  if (rc == 0) rc = hipfail(hipMemset(h->mine, 0, bytes), "hipMemset(staging)");
  if (rc == 0) rc = hipfail(hipMalloc((void**)&h->dev, 32 * sizeof(unsigned)), "hipMalloc(words)");
  if (rc == 0) rc = hipfail(hipMemset(h->dev, 0, 32 * sizeof(unsigned)), "hipMemset(words)");
  hipIpcMemHandle_t ipc;
  if (rc == 0) rc = hipfail(hipIpcGetMemHandle(&ipc, h->mine), "hipIpcGetMemHandle");
  if (rc == 0) rc = hipfail(hipDeviceSynchronize(), "hipDeviceSynchronize");
  if (rc != 0) {
    if (h->mine) (void)hipFree(h->mine);
    if (h->dev) (void)hipFree(h->dev);
    delete h;
    return rc;
  }
  memset(handle_out, 0, VMLMF_P2P_HANDLE_BYTES);
  memcpy(handle_out, &ipc, sizeof(ipc));
  h->peer[rank] = h->mine;
  *p2p = h;
  return 0;
}

int vmlmf_p2p_connect(void* p2p, const unsigned char* handles) {
  P2P* h = (P2P*)p2p;
  if (h == nullptr || handles == nullptr) return vmlmf_set_error(VMLMF_E_BADARG, "p2p_connect: null pointer");
  for (int r = 0; r < h->world; ++r) {
    if (r == h->rank || h->peer[r] != nullptr) continue;
    hipIpcMemHandle_t ipc;
    memcpy(&ipc, handles + (size_t)r * VMLMF_P2P_HANDLE_BYTES, sizeof(ipc));
    void* p = nullptr;
    const int rc = hipfail(hipIpcOpenMemHandle(&p, ipc, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle");
    if (rc != 0) return rc;
    h->peer[r] = (float*)p;
  }
  h->connected = true;
  return 0;
}

int vmlmf_p2p_allreduce(void* p2p, float* buf, size_t n, int op, void* stream) {
  P2P* h = (P2P*)p2p;
  if (h == nullptr || buf == nullptr) return vmlmf_set_error(VMLMF_E_BADARG, "p2p_allreduce: null pointer");
  if (!h->connected) return vmlmf_set_error(VMLMF_E_BADARG, "p2p_allreduce: vmlmf_p2p_connect() first");
  if (n < 1 || n > h->cap) return vmlmf_set_error(VMLMF_E_BADARG, "p2p_allreduce: more floats than vmlmf_p2p_create() was told");
  if ((reinterpret_cast<uintptr_t>(buf) & 15u) != 0) return vmlmf_set_error(VMLMF_E_BADARG, "p2p_allreduce: the buffer must be 16-byte aligned");
  if (op != VMLMF_SUM && op != VMLMF_AVG) return vmlmf_set_error(VMLMF_E_BADARG, "p2p_allreduce: op");
  P2PArgs a;
  memset(&a, 0, sizeof(a));
  for (int r = 0; r < h->world; ++r) a.peer[r] = h->peer[r];
  a.buf = buf, a.dev = h->dev, a.status = vmlmf_status_word(stream), a.cap = h->cap, a.n = n;
  a.rank = h->rank, a.world = h->world, a.avg = op == VMLMF_AVG ? 1 : 0;
  const size_t n4 = n / 4;
  a.nblk = (int)((n4 + 255) / 256);
  a.nblk = a.nblk < 1 ? 1 : (a.nblk > 64 ? 64 : a.nblk);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(p2p_push_kernel, dim3(a.nblk, h->world), dim3(256), 0, s, a);
  int rc = hipfail(hipGetLastError(), "push launch");
  if (rc != 0) return rc;
  hipLaunchKernelGGL(p2p_sum_kernel, dim3(a.nblk), dim3(256), 0, s, a);
  return hipfail(hipGetLastError(), "sum launch");
}

int vmlmf_p2p_destroy(void* p2p) {
  P2P* h = (P2P*)p2p;
  if (h == nullptr) return 0;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < h->world; ++r)
    if (r != h->rank && h->peer[r] != nullptr) (void)hipIpcCloseMemHandle(h->peer[r]);
  if (h->mine) (void)hipFree(h->mine);
  if (h->dev) (void)hipFree(h->dev);
  delete h;
  return 0;
}

}  // extern "C"
