"""Data-parallel gradient exchange for the VMLMF hot path: ONE flat fp32 buffer, ONE collective per step.

The reference has no distributed code.  Batch rows are independent through the
whole forward/backward, so the only exchange is the sum over ranks of the parameter gradients.  The
payload is tiny (HAR Net: 30 951 floats = 121 KiB), i.e. latency-bound on xGMI: bucketing would only add
launches.  The kernels already write a layer's gradients AND those of the classifier riding on it into one flat
allocation (functional.VmlmfSeqFn.backward), so the exchange is ONE in-place all-reduce on the compute stream,
with no staging copies (torch.distributed backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests).

Reduction op must reproduce single-process semantics:
  HAR  loss = mean CE over the local batch  (train.py:63)      -> AVG over ranks
  LM   loss = mean token NLL * B_local      (lm_test.py:147-153) -> SUM over ranks
"""
from __future__ import annotations

import ctypes

import torch
import torch.distributed as dist

from . import _lib


class CabiComm:
    """An RCCL communicator made and used through the C ABI (include/vmlmf_hip.h: vmlmf_comm_*, vmlmf_flat_allreduce_group):
    what a non-PyTorch host would bind.  torch.distributed only carries the 128-byte id from rank 0 to the others.
    The calling process must have its HIP device current (torch.cuda.set_device)."""

    def __init__(self, device, group=None):
        """EVERY rank of the group takes the same sequence of torch.distributed collectives whatever fails locally (a rank
        that raised before a collective the others are already in would hang the job): rank 0 always broadcasts 1 + 128
        bytes - a flag saying whether it could make the id, and the id - every rank calls vmlmf_comm_init only when the flag
        is set (ncclCommInitRank is itself collective), and a MIN all-reduce of "my communicator exists" decides for all.
        self.handle is None afterwards when any rank failed; self.error holds the local reason."""
        self.lib = _lib.lib()
        self.device = torch.device(device)
        self.handle, self.error = None, None
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        ident = (ctypes.c_ubyte * _lib.COMM_ID_BYTES)()
        have_id = 0
        if rank == 0:
            try:
                _lib.check(self.lib.vmlmf_comm_unique_id(ident))
                have_id = 1
            except Exception as e:      # noqa: BLE001 - reported to every rank through the flag byte
                self.error = e
        t = torch.tensor([have_id] + list(ident), dtype=torch.uint8, device=self.device)
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        got = t.cpu().tolist()
        handle, ok = ctypes.c_void_p(), 0
        if got[0] == 1:
            ident = (ctypes.c_ubyte * _lib.COMM_ID_BYTES)(*got[1:])
            try:
                with _lib.on_device(self.device):
                    _lib.check(self.lib.vmlmf_comm_init(ctypes.byref(handle), world, rank, ident))
                ok = 1
            except Exception as e:      # noqa: BLE001
                self.error = e
        flag = torch.tensor([ok], device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 1:
            self.handle = handle
        elif ok:
            self.lib.vmlmf_comm_destroy(handle)

    def ranks(self):
        """Ranks RCCL reports for the communicator (ncclCommCount)."""
        n = ctypes.c_int(0)
        _lib.check(self.lib.vmlmf_comm_count(self.handle, ctypes.byref(n)))
        return n.value

    def all_reduce(self, tensors, op):
        """In place, every tensor of the list under one RCCL group call, on torch's current stream."""
        n = len(tensors)
        bufs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tensors])
        counts = (ctypes.c_size_t * n)(*[t.numel() for t in tensors])
        with _lib.on_device(self.device):
            _lib.check(self.lib.vmlmf_flat_allreduce_group(n, bufs, counts, _lib.AVG if op == "avg" else _lib.SUM,
                                                            self.handle, _lib.raw_stream(self.device)))

    def close(self):
        if self.handle:
            self.lib.vmlmf_comm_destroy(self.handle)
            self.handle = None


class FlatGradAllReduce:
    """Owns a flat buffer covering the gradients of `params` (only those that can receive one)."""

    def __init__(self, params, op="avg", group=None, transport="torch"):
        """transport: "torch" = torch.distributed collectives (RCCL under backend "nccl", gloo in the CPU tests);
        "cabi" = the package's own RCCL entry points behind the C ABI (HIP tensors, backend "nccl" only; when the
        communicator cannot be made on EVERY rank, all ranks fall back to "torch" together)."""
        assert op in ("avg", "sum") and transport in ("torch", "cabi")
        self.op = op
        self.group = group
        self.transport = transport
        self._comm = None           # CabiComm, made at the first reduce()
        self._comm_tried = False
        self.params = [p for p in params if p.requires_grad]
        self.always = False   # run the collectives even in a group of one (bench self-test of the RCCL path)
        self.last_collectives = 0   # all-reduce operations the last reduce() issued (HAR Net: 1 - layer and classifier share a buffer)
        self.flat = None   # staging buffer for gradients that do not already share a flat allocation

    # parameters that never receive a gradient (e.g. Net.cell, the reference's unused duplicate,
    # vmlmf.py:349-350) are left out; every rank sees the same set because the model is replicated
    def numel(self):
        return sum(p.grad.numel() for p in self.params if p.grad is not None)

    @staticmethod
    def _spans(grads):
        """Group gradient tensors by the allocation they live in.  The VMLMF layer (functional.VmlmfSeqFn) and
        the classifier head hand autograd views of ONE flat buffer each, so a group usually tiles a contiguous
        range that can be reduced in place; a gradient alone in its allocation goes through the staging buffer with
        the other loners (one collective for all of them).  Returns [(flat_view_or_None, [grads])]."""
        groups = {}
        for g in grads:
            groups.setdefault(g.untyped_storage().data_ptr(), []).append(g)
        out = []
        for gs in groups.values():
            gs = sorted(gs, key=lambda t: t.storage_offset())
            tiled = len(gs) > 1 and all(t.is_contiguous() and t.dtype == torch.float32 for t in gs) and all(
                a.storage_offset() + a.numel() == b.storage_offset() for a, b in zip(gs, gs[1:]))
            if tiled:
                n = gs[-1].storage_offset() + gs[-1].numel() - gs[0].storage_offset()
                out.append((torch.as_strided(gs[0], (n,), (1,), gs[0].storage_offset()), gs))
            else:
                out.append((None, gs))
        return out

    def rccl_ranks(self):
        """Ranks of the communicator the exchange runs on, as RCCL reports them (C-ABI transport), else the group's size."""
        if self._comm is not None:
            return self._comm.ranks()
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def transport_used(self):
        return "cabi:vmlmf_flat_allreduce_group(rccl)" if self._comm is not None else f"torch.distributed:{dist.get_backend(self.group) if dist.is_initialized() else 'none'}"

    def _cabi(self, device, backend):
        """The C-ABI communicator, or None (wrong backend / CPU tensors / creation failed somewhere)."""
        if self.transport != "cabi" or backend != "nccl" or device.type != "cuda":
            return None
        if not self._comm_tried:
            self._comm_tried = True
            comm = CabiComm(device, self.group)     # never raises before its collectives are done; decides collectively
            self._comm = comm if comm.handle else None
            self._comm_error = comm.error
        return self._comm

    def _all_reduce(self, tensors, op, backend):
        """One collective launch for all `tensors`: on RCCL several all-reduces issued under the coalescing manager
        become one group call (the exchange is latency-bound: two launches would cost twice one)."""
        comm = self._cabi(tensors[0].device, backend)
        if comm is not None:
            comm.all_reduce(tensors, "avg" if op == dist.ReduceOp.AVG else "sum")
            return
        if len(tensors) > 1 and backend == "nccl" and hasattr(dist, "_coalescing_manager"):
            try:
                with dist._coalescing_manager(group=self.group, device=tensors[0].device, async_ops=False):
                    for t in tensors:
                        dist.all_reduce(t, op=op, group=self.group)
                return
            except (TypeError, RuntimeError):   # private API moved: plain back-to-back collectives
                pass
        for t in tensors:
            dist.all_reduce(t, op=op, group=self.group)

    def reduce(self):
        """Call after backward().  In-place on the parameters' .grad tensors: gradients that already tile a flat
        allocation are reduced where they are (no staging copies); the rest go through the staging buffer."""
        params = [p for p in self.params if p.grad is not None]
        if not params:
            return
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        backend = dist.get_backend(self.group) if dist.is_initialized() else ""
        native_avg = self.op == "avg" and backend == "nccl"      # RCCL averages in the collective itself
        op = dist.ReduceOp.AVG if native_avg else dist.ReduceOp.SUM
        scale = None if (native_avg or self.op == "sum") else 1.0 / world
        loose, flats = [], []
        self.last_collectives = 0
        for flat, gs in self._spans([p.grad for p in params]):
            if flat is None:
                loose += gs
            else:
                flats.append(flat)
        if flats and (world > 1 or self.always):
            self._all_reduce(flats, op, backend)
            self.last_collectives += len(flats)
            if scale is not None:
                for flat in flats:
                    flat.mul_(scale)
        if loose:
            if self.flat is None or self.flat.numel() != sum(g.numel() for g in loose):
                self.flat = torch.empty(sum(g.numel() for g in loose), dtype=torch.float32, device=loose[0].device)
            views, o = [], 0
            for g in loose:
                views.append(self.flat[o:o + g.numel()].view_as(g))
                o += g.numel()
            torch._foreach_copy_(views, loose)
            if world > 1 or self.always:
                self._all_reduce([self.flat], op, backend)
                self.last_collectives += 1
                if scale is not None:
                    self.flat.mul_(scale)
            torch._foreach_copy_(loose, views)


def broadcast_parameters(module, src=0, group=None):
    """Make every replica start from rank `src`'s parameters (one flat broadcast)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    ps = list(module.parameters())
    flat = torch.cat([p.detach().reshape(-1) for p in ps])
    dist.broadcast(flat, src=src, group=group)
    o = 0
    with torch.no_grad():
        for p in ps:
            p.copy_(flat[o:o + p.numel()].view_as(p))
            o += p.numel()


def shard_batch(x, rank, world, dim=0):
    """Contiguous split of the global minibatch by rank."""
    n = x.shape[dim]
    per = (n + world - 1) // world
    return x.narrow(dim, min(rank * per, n), max(0, min(per, n - rank * per)))
