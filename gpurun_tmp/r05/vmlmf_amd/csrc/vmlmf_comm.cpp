// Gradient exchange of the data-parallel path behind the C ABI (include/vmlmf_hip.h):
// one in-place RCCL all-reduce per flat fp32 gradient buffer, all buffers of a step under ONE group call, on the
// caller's stream.  The reference has no distributed code; this is what a maintainer's DP wrapper would bind.
//
// RCCL is bound at run time (dlopen), not at link time: a process that already holds librccl.so.1 (PyTorch-ROCm
// ships its own copy) must keep using THAT copy - two RCCL instances in one process would each run their own
// bootstrap and proxy threads - and a single-GPU process that never calls these entry points needs no RCCL at all.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>

#include "../../include/vmlmf_hip.h"

int vmlmf_set_error(int code, const std::string& msg);   // vmlmf_api.hip

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, []() {
    // 1. the copy this process already has (RTLD_NOLOAD), 2. the loader's search path, 3. the ROCm install
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    r.handle = dlopen(names[0], RTLD_NOW | RTLD_NOLOAD);
    for (int i = 0; r.handle == nullptr && i < 3; ++i) r.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (r.handle == nullptr) {
      r.why = std::string("librccl.so.1 not found: ") + dlerror();
      return;
    }
    auto sym = [&](const char* n) {
      void* p = dlsym(r.handle, n);
      if (p == nullptr && r.why.empty()) r.why = std::string("librccl lacks ") + n;
      return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  });
  return &r;
}

int no_rccl() { return vmlmf_set_error(VMLMF_E_UNSUPPORTED, "RCCL unavailable: " + rccl()->why); }

int nccl_fail(Rccl* r, ncclResult_t e, const char* what) {
  return vmlmf_set_error(VMLMF_E_COMM, std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(e) : "RCCL error"));
}

}  // namespace

extern "C" {

int vmlmf_comm_unique_id(void* id128) {
  Rccl* r = rccl();
  if (!r->why.empty()) return no_rccl();
  if (id128 == nullptr) return vmlmf_set_error(VMLMF_E_BADARG, "comm: null id buffer");
  static_assert(sizeof(ncclUniqueId) == VMLMF_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  const ncclResult_t e = r->GetUniqueId(&id);
  if (e != ncclSuccess) return nccl_fail(r, e, "ncclGetUniqueId");
  memcpy(id128, &id, sizeof(id));
  return 0;
}

int vmlmf_comm_init(void** comm, int world, int rank, const void* id128) {
  Rccl* r = rccl();
  if (!r->why.empty()) return no_rccl();
  if (comm == nullptr || id128 == nullptr || world < 1 || rank < 0 || rank >= world)
    return vmlmf_set_error(VMLMF_E_BADARG, "comm: bad world / rank / id");
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  const ncclResult_t e = r->CommInitRank(&c, world, id, rank);   // binds to the CURRENT HIP device
  if (e != ncclSuccess) return nccl_fail(r, e, "ncclCommInitRank");
  *comm = (void*)c;
  return 0;
}

int vmlmf_comm_count(void* comm, int* ranks) {
  Rccl* r = rccl();
  if (!r->why.empty()) return no_rccl();
  if (comm == nullptr || ranks == nullptr) return vmlmf_set_error(VMLMF_E_BADARG, "comm_count: null communicator / result");
  const ncclResult_t e = r->CommCount((ncclComm_t)comm, ranks);
  return e == ncclSuccess ? 0 : nccl_fail(r, e, "ncclCommCount");
}

int vmlmf_comm_destroy(void* comm) {
  Rccl* r = rccl();
  if (!r->why.empty()) return no_rccl();
  if (comm == nullptr) return 0;
  const ncclResult_t e = r->CommDestroy((ncclComm_t)comm);
  return e == ncclSuccess ? 0 : nccl_fail(r, e, "ncclCommDestroy");
}

int vmlmf_flat_allreduce_group(int nbuf, void* const* bufs, const size_t* counts, int op, void* comm, void* stream) {
  Rccl* r = rccl();
  if (!r->why.empty()) return no_rccl();
  if (comm == nullptr || nbuf < 0 || (nbuf > 0 && (bufs == nullptr || counts == nullptr)))
    return vmlmf_set_error(VMLMF_E_BADARG, "allreduce: null communicator / buffers");
  if (op != VMLMF_SUM && op != VMLMF_AVG) return vmlmf_set_error(VMLMF_E_BADARG, "allreduce: op must be VMLMF_SUM or VMLMF_AVG");
  const ncclRedOp_t rop = op == VMLMF_AVG ? ncclAvg : ncclSum;
  ncclResult_t e = r->GroupStart();
  if (e != ncclSuccess) return nccl_fail(r, e, "ncclGroupStart");
  for (int i = 0; i < nbuf; ++i) {
    if (bufs[i] == nullptr || counts[i] == 0) continue;
    e = r->AllReduce(bufs[i], bufs[i], counts[i], ncclFloat32, rop, (ncclComm_t)comm, (hipStream_t)stream);
    if (e != ncclSuccess) {
      r->GroupEnd();
      return nccl_fail(r, e, "ncclAllReduce");
    }
  }
  e = r->GroupEnd();
  return e == ncclSuccess ? 0 : nccl_fail(r, e, "ncclGroupEnd");
}

int vmlmf_flat_allreduce(void* buf, size_t n, int op, void* comm, void* stream) {
  void* bufs[1] = {buf};
  const size_t counts[1] = {n};
  return vmlmf_flat_allreduce_group(1, bufs, counts, op, comm, stream);
}

}  // extern "C"
