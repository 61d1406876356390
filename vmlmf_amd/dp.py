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


class P2PExchange:
    """The one-shot peer-to-peer all-reduce of the C ABI (include/vmlmf_hip.h, ABI 13: vmlmf_p2p_*; csrc/vmlmf_p2p.hip) for SMALL
    flat buffers (the HAR network's 121 KiB between loss.backward() and optimizer.step(), train.py:64-65): every rank writes its
    buffer into every peer's hipIpc-mapped staging area and sums its own area's slots in rank order - two launches on the compute
    stream, no collective library.  torch.distributed only carries the 64-byte IPC handles, once (any backend: gloo works).
    Like CabiComm, every rank takes the same sequence of collectives whatever fails locally, and a MIN all-reduce of "my side
    is connected" decides for all: self.handle is None afterwards when any rank failed (self.error: the local reason)."""

    def __init__(self, device, max_floats, group=None):
        self.lib = _lib.lib()
        self.device = torch.device(device)
        self.handle, self.error, self.max_floats = None, None, int(max_floats)
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        self.world = world
        h = ctypes.c_void_p()
        mine = (ctypes.c_ubyte * _lib.P2P_HANDLE_BYTES)()
        made = 0
        try:
            if world > _lib.P2P_MAX_RANKS:
                raise RuntimeError(f"peer-to-peer exchange: at most {_lib.P2P_MAX_RANKS} ranks")
            with _lib.on_device(self.device):
                _lib.check(self.lib.vmlmf_p2p_create(ctypes.byref(h), rank, world, self.max_floats, mine))
            made = 1
        except Exception as e:      # noqa: BLE001 - decided collectively below
            self.error = e
        gathered = [None] * world
        dist.all_gather_object(gathered, (made, bytes(mine)), group=group)
        ok = 0
        if made and all(g[0] == 1 for g in gathered):
            blob = b"".join(g[1] for g in gathered)
            try:
                with _lib.on_device(self.device):
                    _lib.check(self.lib.vmlmf_p2p_connect(h, (ctypes.c_ubyte * len(blob)).from_buffer_copy(blob)))
                ok = 1
            except Exception as e:      # noqa: BLE001
                self.error = e
        flag = torch.tensor([ok], device=self.device if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 1:
            self.handle = h
        elif made:
            self.lib.vmlmf_p2p_destroy(h)

    def ranks(self):
        return self.world

    def all_reduce(self, tensors, op):
        """In place, tensor by tensor (each <= max_floats floats, fp32, contiguous, 16-byte aligned), on torch's current stream."""
        with _lib.on_device(self.device):
            for t in tensors:
                _lib.check(self.lib.vmlmf_p2p_allreduce(self.handle, t.data_ptr(), t.numel(), _lib.AVG if op == "avg" else _lib.SUM,
                                                        _lib.raw_stream(self.device)))

    def close(self):
        if self.handle:
            self.lib.vmlmf_p2p_destroy(self.handle)
            self.handle = None


class FlatGradAllReduce:
    """Owns a flat buffer covering the gradients of `params` (only those that can receive one)."""

    def __init__(self, params, op="avg", group=None, transport="torch"):
        """transport: "torch" = torch.distributed collectives (RCCL under backend "nccl", gloo in the CPU tests);
        "cabi" = the package's own RCCL entry points behind the C ABI (HIP tensors, backend "nccl" only; when the
        communicator cannot be made on EVERY rank, all ranks fall back to "torch" together); "p2p" = the one-shot peer-to-peer
        exchange of the C ABI (P2PExchange: HIP tensors of at most 4 Mi floats each, any backend for the handle exchange; the same
        collective fallback).  "torch" stays the default: the peer-to-peer path has never run across GPUs."""
        assert op in ("avg", "sum") and transport in ("torch", "cabi", "p2p")
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
        return self.exchange_ranks()[0]

    def exchange_ranks(self):
        """(ranks, who counted them): RCCL's own ncclCommCount on the C-ABI transport; otherwise only the size of the
        torch.distributed group, labelled with its backend - over gloo no RCCL communicator exists at all."""
        if isinstance(self._comm, P2PExchange):
            return self._comm.ranks(), "vmlmf_p2p_connect (staging areas mapped: one per rank)"
        if self._comm is not None:
            return self._comm.ranks(), "ncclCommCount (RCCL communicator behind the C ABI)"
        if not dist.is_initialized():
            return 1, "no process group"
        return dist.get_world_size(self.group), f"torch.distributed.get_world_size (backend {dist.get_backend(self.group)})"

    def transport_used(self):
        if isinstance(self._comm, P2PExchange):
            return "p2p:vmlmf_p2p_allreduce(hipIpc staging, rank-order sum)"
        return "cabi:vmlmf_flat_allreduce_group(rccl)" if self._comm is not None else f"torch.distributed:{dist.get_backend(self.group) if dist.is_initialized() else 'none'}"

    def _p2p(self, tensors):
        """The peer-to-peer exchange, or None (CPU tensors / no process group / a buffer it does not take / creation failed somewhere).
        Sized at the first reduce() for the largest buffer of that step."""
        if self.transport != "p2p" or not dist.is_initialized() or tensors[0].device.type != "cuda":
            return None
        if not self._comm_tried:
            self._comm_tried = True
            ex = P2PExchange(tensors[0].device, max(t.numel() for t in tensors), self.group)
            self._comm = ex if ex.handle else None
            self._comm_error = ex.error
        c = self._comm
        if c is None or any(t.numel() > c.max_floats or t.dtype != torch.float32 or not t.is_contiguous() or t.data_ptr() % 16 for t in tensors):
            return None
        return c

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
        comm = self._p2p(tensors) if self.transport == "p2p" else self._cabi(tensors[0].device, backend)
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


class BucketedGradAllReduce:
    """Gradient exchange for a model whose gradient BYTES matter (the LM network: 13.44 M parameters = 53.8 MB, 97 % of
    them `fc.w` and `embed.w`): the parameters are cut into buckets in the order the backward pass
    finishes their gradients, and a bucket's all-reduce is started - asynchronously, on the communication stream - the
    moment its last gradient has been accumulated, so the vocabulary projection's 26 MB travel over xGMI while the
    recurrent layers' backward still runs.  wait() joins them all; only then may the caller look at the gradients
    (clip_grad_norm_ over the REDUCED gradients, lm_test.py:204).

    buckets: [[parameters]], in backward order.  op "sum" (the LM loss, lm_test.py:147-153: Σ over ranks of the local losses
    is the global-batch loss) or "avg".  transport "torch": torch.distributed (backend nccl = RCCL; gloo on CPU tensors and
    in the one-GPU rehearsal), "cabi": the package's own RCCL entry points on a side stream.
    Every rank issues the same collectives in the same order: a bucket is launched from the hook of its LAST outstanding
    gradient, buckets that never completed (a parameter without a gradient this step) are launched by wait() in bucket
    order, and launches from hooks are forced into bucket order too (bucket k waits for buckets < k to have been issued)."""

    IN_PLACE_BYTES = 256 * 1024     # a lone gradient at least this big is reduced where it is; smaller ones share a staging buffer

    def __init__(self, buckets, op="sum", group=None, transport="torch"):
        assert op in ("sum", "avg") and transport in ("torch", "cabi")
        self.op, self.group, self.transport = op, group, transport
        self.buckets = [[p for p in b if p.requires_grad] for b in buckets]
        self.buckets = [b for b in self.buckets if b]
        self._of = {}
        for bi, b in enumerate(self.buckets):
            for p in b:
                if id(p) in self._of:
                    raise ValueError("a parameter may be in one bucket only")
                self._of[id(p)] = bi
        self._pending = [{id(p) for p in b} for b in self.buckets]
        self._issued = [False] * len(self.buckets)
        self._work = []             # (handle or None, staging or None) per issued collective
        self._stage = {}            # bucket -> flat staging buffer of its small loners
        self._comm, self._comm_tried, self._side = None, False, None
        self.always = False         # run the collectives in a group of one as well (self-test)
        self.last_collectives = 0
        self.last_overlapped = 0    # collectives started from a hook, i.e. before the backward pass had ended
        self.bytes_per_step = 0
        self._armed = False
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for b in self.buckets for p in b]

    def close(self):
        for h in self._handles:
            h.remove()
        self._handles = []
        if self._comm is not None:
            self._comm.close()
            self._comm = None

    # ---- per step ---------------------------------------------------------------------------------------------------
    def arm(self):
        """Before backward(): a new step begins."""
        self._pending = [{id(p) for p in b} for b in self.buckets]
        self._issued = [False] * len(self.buckets)
        self._work = []
        self.last_collectives = self.last_overlapped = self.bytes_per_step = 0
        self._armed = True

    def _world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def _hook(self, p):
        if not self._armed:
            return
        bi = self._of[id(p)]
        self._pending[bi].discard(id(p))
        # launch every bucket that is complete AND whose predecessors have been issued: the same order on every rank
        for k in range(len(self.buckets)):
            if self._issued[k]:
                continue
            if self._pending[k]:
                break
            self._launch(k, from_hook=True)

    def _tensors(self, bi):
        """What bucket bi reduces: (tensors reduced where they are, small loners to stage)."""
        grads = [p.grad for p in self.buckets[bi] if p.grad is not None]
        in_place, loose = [], []
        for flat, gs in FlatGradAllReduce._spans(grads):
            if flat is not None:
                in_place.append(flat)
            else:
                for g in gs:
                    if g.is_contiguous() and g.dtype == torch.float32 and g.numel() * 4 >= self.IN_PLACE_BYTES:
                        in_place.append(g.view(-1))
                    else:
                        loose.append(g)
        return in_place, loose

    def _launch(self, bi, from_hook):
        self._issued[bi] = True
        in_place, loose = self._tensors(bi)
        if not in_place and not loose:
            return
        world = self._world()
        if world == 1 and not self.always:
            return
        stage = None
        if loose:
            n = sum(g.numel() for g in loose)
            stage = self._stage.get(bi)
            if stage is None or stage.numel() != n or stage.device != loose[0].device:
                stage = self._stage[bi] = torch.empty(n, dtype=torch.float32, device=loose[0].device)
            views, o = [], 0
            for g in loose:
                views.append(stage[o:o + g.numel()].view_as(g))
                o += g.numel()
            torch._foreach_copy_(views, loose)
            self._work.append((None, (loose, views)))
        tensors = in_place + ([stage] if stage is not None else [])
        backend = dist.get_backend(self.group) if dist.is_initialized() else ""
        native_avg = self.op == "avg" and backend == "nccl"
        rop = dist.ReduceOp.AVG if native_avg else dist.ReduceOp.SUM
        comm = self._cabi(tensors[0].device, backend)
        if comm is not None:
            cur = torch.cuda.current_stream(tensors[0].device)
            if self._side is None:
                self._side = torch.cuda.Stream(tensors[0].device)
            self._side.wait_stream(cur)                      # the gradients are complete on the compute stream
            with torch.cuda.stream(self._side):
                comm.all_reduce(tensors, "avg" if native_avg else "sum")
            self._work.append(("side", None))
        else:
            for t in tensors:
                self._work.append((dist.all_reduce(t, op=rop, group=self.group, async_op=True), None))
        if self.op == "avg" and not native_avg:
            self._work.append((None, ("scale", tensors, 1.0 / world)))
        self.last_collectives += len(tensors)
        self.last_overlapped += len(tensors) if from_hook else 0
        self.bytes_per_step += 4 * sum(t.numel() for t in tensors)

    def _cabi(self, device, backend):
        if self.transport != "cabi" or backend != "nccl" or device.type != "cuda":
            return None
        if not self._comm_tried:
            self._comm_tried = True
            comm = CabiComm(device, self.group)
            self._comm = comm if comm.handle else None
        return self._comm

    def wait(self):
        """After backward(): launch what the hooks could not, join every collective (the current stream waits for the
        communication stream; over gloo the host does), copy staged gradients back."""
        for k in range(len(self.buckets)):
            if not self._issued[k]:
                self._launch(k, from_hook=False)
        later = []
        for handle, extra in self._work:
            if handle == "side":
                torch.cuda.current_stream(self._side.device).wait_stream(self._side)
            elif handle is not None:
                handle.wait()
            if extra is not None:
                later.append(extra)
        for extra in later:
            if extra[0] == "scale":
                for t in extra[1]:
                    t.mul_(extra[2])
        for extra in later:
            if extra[0] != "scale":
                loose, views = extra
                torch._foreach_copy_(loose, views)
        self._work = []
        self._armed = False

    def exchange_ranks(self):
        if self._comm is not None:
            return self._comm.ranks(), "ncclCommCount (RCCL communicator behind the C ABI)"
        if not dist.is_initialized():
            return 1, "no process group"
        return dist.get_world_size(self.group), f"torch.distributed.get_world_size (backend {dist.get_backend(self.group)})"

    def transport_used(self):
        return "cabi:vmlmf_flat_allreduce_group(rccl), side stream" if self._comm is not None else \
            f"torch.distributed:{dist.get_backend(self.group) if dist.is_initialized() else 'none'}, async_op"


def lm_buckets(model):
    """Buckets of the LM network (vmlmf_lm.py:366-440) in the order its backward pass completes them: the vocabulary
    projection first (its gradients exist before the recurrent layers' backward starts), then the recurrent layers from
    the top one down, the embedding table last (its gradient needs layer 0's dx)."""
    out = [list(model.fc.parameters())]
    out += [list(r.parameters()) for r in reversed(list(model.rnns))]
    out.append(list(model.embed.parameters()))
    seen = {id(p) for b in out for p in b}
    rest = [p for p in model.parameters() if id(p) not in seen]
    return out + ([rest] if rest else [])


class LmDataParallel:
    """The reference's LM training step (lm_test.py:196-207) on one rank of a data-parallel job:

        states = model.detach(states)                 rank-local rows: the carried (h, c) never leave the rank
        scores, states = model(x_local, states)
        loss = nll_loss(scores, y_local)              = (1/T) Σ over the LOCAL tokens  (lm_test.py:147-153)
        loss.backward()                               buckets all-reduce (SUM) as they complete
        clip_grad_norm_(REDUCED gradients, max_norm); param -= lr * grad        (lm_test.py:204-207)

    With the batch columns split contiguously over the ranks (shard()), Σ over ranks of the local losses and gradients are
    the single-process loss and gradients of the global minibatch, so every rank clips by the same norm and takes the same
    update: the replicas stay identical without ever exchanging parameters.
    loss_fn / update_fn default to the package's fused kernels (HIP tensors); the CPU tests pass stock formulations."""

    def __init__(self, model, lr, max_norm, group=None, transport="torch", loss_fn=None, update_fn=None, buckets=None):
        from .functional import nll_loss
        from .optim import clip_sgd_step
        self.model, self.lr, self.max_norm, self.group = model, lr, max_norm, group
        # the package's own loss: taken inside Model.loss (projection + loss with the scores' gradient formed in place)
        self._fused_loss = loss_fn is None and hasattr(model, "loss")
        self.loss_fn = loss_fn or nll_loss
        self.update_fn = update_fn or clip_sgd_step
        self.reducer = BucketedGradAllReduce(buckets or lm_buckets(model), op="sum", group=group, transport=transport)
        broadcast_parameters(model, group=group)

    def _rank_world(self):
        if not dist.is_initialized():
            return 0, 1
        return dist.get_rank(self.group), dist.get_world_size(self.group)

    def shard(self, x):
        """This rank's contiguous block of batch columns of a (T, B) token tensor."""
        rank, world = self._rank_world()
        return shard_batch(x, rank, world, dim=1)

    def forward_backward(self, x, y, states):
        self.model.zero_grad(set_to_none=True)
        states = self.model.detach(states)
        if self._fused_loss and x.is_cuda:
            from .functional import unit_gradient
            loss, states = self.model.loss(x, y, states)
            root = unit_gradient(x.device)
        else:
            scores, states = self.model(x, states)
            loss, root = self.loss_fn(scores, y), None
        self.reducer.arm()
        loss.backward(root)
        self.reducer.wait()
        return loss.detach(), states

    def step(self, x, y, states):
        """x, y: this rank's (T, B_local) tokens.  Returns (local loss, norm of the reduced gradients, new states)."""
        loss, states = self.forward_backward(x, y, states)
        norm = self.update_fn(self.model.parameters(), self.lr, self.max_norm)
        return loss, norm, states

    def global_loss(self, loss):
        """Σ over ranks of the local losses = the reference's loss on the global minibatch."""
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return loss
        t = loss.detach().clone().reshape(1)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t[0]


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
