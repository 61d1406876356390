"""Data-parallel gradient exchange for the VMLMF hot path: ONE flat fp32 buffer, ONE collective per step.

The reference has no distributed code.  Batch rows are independent through the
whole forward/backward, so the only exchange is the sum over ranks of the parameter gradients.  The
payload is tiny (HAR Net: 30 951 floats = 121 KiB), i.e. latency-bound on xGMI: bucketing would only add
launches, so every gradient goes through a single all-reduce on the compute stream
(torch.distributed backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests).

Reduction op must reproduce single-process semantics:
  HAR  loss = mean CE over the local batch  (train.py:63)      -> AVG over ranks
  LM   loss = mean token NLL * B_local      (lm_test.py:147-153) -> SUM over ranks
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class FlatGradAllReduce:
    """Owns a flat buffer covering the gradients of `params` (only those that can receive one)."""

    def __init__(self, params, op="avg", group=None):
        assert op in ("avg", "sum")
        self.op = op
        self.group = group
        self.params = [p for p in params if p.requires_grad]
        self.flat = None
        self.active = None

    def _layout(self):
        # parameters that never receive a gradient (e.g. Net.cell, the reference's unused duplicate,
        # vmlmf.py:349-350) are left out; every rank sees the same set because the model is replicated
        self.active = [p for p in self.params if p.grad is not None]
        n = sum(p.numel() for p in self.active)
        dev = self.active[0].device if self.active else torch.device("cpu")
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.views, o = [], 0
        for p in self.active:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()

    def numel(self):
        return 0 if self.flat is None else self.flat.numel()

    def reduce(self):
        """Call after backward().  In-place on the parameters' .grad tensors."""
        if self.flat is None:
            self._layout()
        if not self.active:
            return
        grads = [p.grad for p in self.active]
        torch._foreach_copy_(self.views, grads)
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            if self.op == "avg":
                self.flat.mul_(1.0 / world)
        torch._foreach_copy_(grads, self.views)


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
