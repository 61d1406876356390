"""CPU, 2 processes over gloo: the flat-buffer gradient all-reduce reproduces single-process gradients
(AVG for the HAR loss, SUM for the LM loss), tolerates parameters without a gradient
(Net.cell, vmlmf.py:349-350), and batch sharding is a contiguous split."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Toy(torch.nn.Module):
    """Row-independent model: per-row loss, parameter gradients sum over rows (like the RNN)."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.a = torch.nn.Parameter(torch.randn(6, 4, generator=g))
        self.b = torch.nn.Parameter(torch.randn(4, generator=g))
        self.unused = torch.nn.Parameter(torch.randn(3, generator=g))   # never receives a gradient

    def forward(self, x):
        return torch.tanh(x @ self.a + self.b)


class _FlatGradAffine(torch.autograd.Function):
    """x @ a + b whose backward hands autograd VIEWS of one flat buffer, like functional.VmlmfSeqFn does."""

    @staticmethod
    def forward(ctx, x, a, b):
        ctx.save_for_backward(x, a)
        return x @ a + b

    @staticmethod
    def backward(ctx, dy):
        x, a = ctx.saved_tensors
        flat = torch.empty(a.numel() + dy.shape[1])
        da, db = flat[:a.numel()].view_as(a), flat[a.numel():]
        da.copy_(x.t() @ dy)
        db.copy_(dy.sum(0))
        return None, da, db


class TiledToy(Toy):
    def forward(self, x):
        return torch.tanh(_FlatGradAffine.apply(x, self.a, self.b))


def _worker(rank, world, port, op, out, tiled=False):
    sys.path.insert(0, ROOT)
    from vmlmf_amd.dp import FlatGradAllReduce, broadcast_parameters, shard_batch
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                 # replicas start different ...
    m = TiledToy() if tiled else Toy()
    with torch.no_grad():
        m.a.add_(rank)
    broadcast_parameters(m)                       # ... and are made identical
    x = torch.randn(10, 6, generator=torch.Generator().manual_seed(1))
    xs = shard_batch(x, rank, world)
    y = m(xs)
    loss = y.mean(dim=1).mean() if op == "avg" else y.mean(dim=1).sum()
    loss.backward()
    red = FlatGradAllReduce(m.parameters(), op=op)
    in_place = [flat is not None for flat, _ in red._spans([p.grad for p in m.parameters() if p.grad is not None])]
    red.reduce()
    if rank == 0:
        torch.save({"a": m.a.grad.clone(), "b": m.b.grad.clone(), "n": red.numel(), "a0": m.a.detach(),
                    "in_place": in_place, "staged": red.flat is not None, "collectives": red.last_collectives,
                    "ranks": red.rccl_ranks()}, out)
    dist.destroy_process_group()


def _run(op, tmp_path, tiled=False):
    out = str(tmp_path / f"dp_{op}_{int(tiled)}.pt")
    port = 29500 + (os.getpid() % 2000) + (7 if tiled else 0)
    mp.spawn(_worker, args=(2, port, op, out, tiled), nprocs=2, join=True)
    got = torch.load(out)
    assert got["collectives"] == 1 and got["ranks"] == 2          # ONE all-reduce per step, over both ranks
    if tiled:      # both gradients live in one allocation (as a VMLMF layer's and its classifier's do): reduced where they are
        assert got["in_place"] == [True] and not got["staged"]
    else:          # gradients alone in their allocations share the staging buffer (one collective)
        assert not any(got["in_place"]) and got["staged"]
    m = Toy()
    x = torch.randn(10, 6, generator=torch.Generator().manual_seed(1))
    y = m(x)
    loss = y.mean(dim=1).mean() if op == "avg" else y.mean(dim=1).sum()
    loss.backward()
    assert torch.equal(got["a0"], m.a.detach())                    # rank 0's parameters won the broadcast
    assert got["n"] == 6 * 4 + 4                                   # the unused parameter is not in the buffer
    assert torch.allclose(got["a"], m.a.grad, atol=1e-6)
    assert torch.allclose(got["b"], m.b.grad, atol=1e-6)


def test_avg_matches_global_batch_mean_loss(tmp_path):
    _run("avg", tmp_path)


def test_sum_matches_global_batch_sum_loss(tmp_path):
    _run("sum", tmp_path)


def test_gradients_sharing_a_flat_allocation_are_reduced_in_place(tmp_path):
    _run("avg", tmp_path, tiled=True)
    _run("sum", tmp_path, tiled=True)


def test_shard_batch_is_a_contiguous_partition():
    sys.path.insert(0, ROOT)
    from vmlmf_amd.dp import shard_batch
    x = torch.arange(512 * 3).view(512, 3)
    for world in (1, 2, 4, 8):
        parts = [shard_batch(x, r, world) for r in range(world)]
        assert all(p.shape[0] == 512 // world for p in parts)
        assert torch.equal(torch.cat(parts), x)
    parts = [shard_batch(x[:10], r, 4) for r in range(4)]           # ragged: 3,3,3,1
    assert [p.shape[0] for p in parts] == [3, 3, 3, 1]
