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


def _worker(rank, world, port, op, out, tiled=False, transport="torch"):
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
    red = FlatGradAllReduce(m.parameters(), op=op, transport=transport)
    in_place = [flat is not None for flat, _ in red._spans([p.grad for p in m.parameters() if p.grad is not None])]
    red.reduce()
    if rank == 0:
        torch.save({"a": m.a.grad.clone(), "b": m.b.grad.clone(), "n": red.numel(), "a0": m.a.detach(),
                    "in_place": in_place, "staged": red.flat is not None, "collectives": red.last_collectives,
                    "ranks": red.rccl_ranks(), "transport": red.transport_used()}, out)
    dist.destroy_process_group()


def _run(op, tmp_path, tiled=False, transport="torch"):
    out = str(tmp_path / f"dp_{op}_{int(tiled)}.pt")
    port = 29500 + (os.getpid() % 2000) + (7 if tiled else 0) + (13 if transport != "torch" else 0)
    mp.spawn(_worker, args=(2, port, op, out, tiled, transport), nprocs=2, join=True)
    got = torch.load(out)
    assert got["transport"].startswith("torch.distributed:gloo")   # (CPU tensors: "p2p" / "cabi" fall back to the process group, on every rank alike)
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


def test_the_peer_to_peer_transport_falls_back_on_cpu_tensors(tmp_path):
    """FlatGradAllReduce(transport="p2p") (ABI 13: hipIpc staging areas) with CPU gradients: the exchange runs on the process group,
    decided the same way on every rank - same results as the default transport."""
    _run("avg", tmp_path, tiled=True, transport="p2p")


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


# ---- the LM network's data-parallel step (BASELINE configs[4]; lm_test.py:196-207) -------------------
def _lm_model_on_cpu(d):
    """vmlmf_amd.Model with the golden initial parameters; its forward (HIP only in the product) is replaced by the
    oracle's literal restatement of Model.forward so that the DATA-PARALLEL logic around it runs on CPU."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vmlmf_oracle as O
    from vmlmf_amd import Model
    V, H, L, B, T, rw, ru = (int(v) for v in d["meta"])
    model = Model(V, H, L, 0.0, 0.1, w_rank=rw, u_ranks=[ru], lstm_type="vmlmf")
    model.load_state_dict({k[len("init/"):]: torch.tensor(d[k]) for k in d.files if k.startswith("init/")})
    model.forward = lambda x, states: O.literal_lm_forward(dict(model.named_parameters()), x, states, L)
    return model, (V, H, L, B, T)


def _stock_clip_sgd(params, lr, max_norm):
    """lm_test.py:204-207 as written (the product's fused clip_sgd_step is HIP only)."""
    params = [p for p in params if p.grad is not None]
    with torch.no_grad():
        norm = torch.nn.utils.clip_grad_norm_(params, max_norm)
        for p in params:
            p -= lr * p.grad
    return norm


def _lm_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import numpy as np
    from vmlmf_amd.dp import LmDataParallel
    if world > 1:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    d = np.load(os.path.join(ROOT, "tests", "golden", "lm_model_v3.npz"))
    model, (V, H, L, B, T) = _lm_model_on_cpu(d)
    if rank != 0:                       # replicas start different: the constructor's broadcast must make them rank 0's
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.5)
    dp = LmDataParallel(model, lr=1.0, max_norm=0.25, update_fn=_stock_clip_sgd)
    dp.reducer.IN_PLACE_BYTES = 2048    # at this size fc.w / embed.w (3840 B) are reduced where they are, the rest is staged
    states = model.state_init(B // world)
    rec = {"losses": [], "norms": [], "grads": [], "collectives": [], "overlapped": [], "bytes": []}
    for i in range(2):
        x, y = dp.shard(torch.tensor(d[f"x{i}"])), dp.shard(torch.tensor(d[f"y{i}"]))
        assert x.shape == (T, B // world)
        loss, states = dp.forward_backward(x, y, states)
        rec["grads"].append({k: p.grad.clone() for k, p in model.named_parameters()})
        rec["norms"].append(float(dp.update_fn(model.parameters(), dp.lr, dp.max_norm)))
        rec["losses"].append(float(dp.global_loss(loss)))
        rec["collectives"].append(dp.reducer.last_collectives)
        rec["overlapped"].append(dp.reducer.last_overlapped)
        rec["bytes"].append(dp.reducer.bytes_per_step)
    rec["final"] = {k: v.detach().clone() for k, v in model.state_dict().items()}
    rec["hT"] = torch.stack([s[0].detach() for s in states])
    rec["ranks"] = dp.reducer.exchange_ranks()
    torch.save(rec, f"{out}.{rank}")
    if world > 1:
        dist.destroy_process_group()


def test_lm_network_data_parallel_step_matches_the_reference_run(tmp_path):
    """Two ranks, two columns of the golden minibatch each (lm_model_v3: the reference's own two-minibatch run of
    lm_test.py:196-209 at B = 4): SUM-reduced gradients, the clip norm taken AFTER the reduce, the updated parameters and
    the rank-local carried states reproduce the single-process run; the exchange is bucketed (vocabulary projection first,
    from a hook inside the backward pass, embedding last) and identical replicas never exchange parameters."""
    import numpy as np
    d = np.load(os.path.join(ROOT, "tests", "golden", "lm_model_v3.npz"))
    out = str(tmp_path / "lm_dp")
    port = 29500 + (os.getpid() % 2000) + 23
    mp.spawn(_lm_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    _lm_worker(0, 1, 0, out + ".single")            # the same code in a group of one, whole minibatch
    single = torch.load(out + ".single.0")

    def close(a, b, what, tol=2e-5):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        assert np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-6), what

    for i in range(2):
        for k in r0["grads"][i]:
            assert torch.equal(r0["grads"][i][k], r1["grads"][i][k]), k          # both ranks hold the same reduced gradient
            close(r0["grads"][i][k], d[f"G{i}/{k}"], f"G{i}/{k} vs the reference")
            close(r0["grads"][i][k], single["grads"][i][k], f"G{i}/{k} vs one process")
        assert r0["norms"][i] == r1["norms"][i]
        assert abs(r0["norms"][i] - float(d[f"norm{i}"][0])) < 1e-5 * r0["norms"][i]        # clipped by the GLOBAL norm
        assert abs(r0["losses"][i] - float(d[f"loss{i}"][0])) < 1e-5 * abs(r0["losses"][i])  # Σ local losses = global loss
        # buckets: fc | rnns.1 | rnns.0 | embed; fc.w and embed.w in place, fc.b and each layer's tensors staged per bucket
        assert r0["collectives"][i] == 5 and r0["overlapped"][i] >= 4, (r0["collectives"], r0["overlapped"])
        assert r0["bytes"][i] == 4 * sum(v.numel() for v in r0["grads"][i].values())
    for k, v in r0["final"].items():
        assert torch.equal(v, r1["final"][k]), k                                   # replicas stay identical
        close(v, d[f"final/{k}"], f"final/{k}", tol=1e-4)
    close(torch.cat([r0["hT"], r1["hT"]], dim=1), d["hT"], "carried states stay with their rows", tol=1e-4)
    assert r0["ranks"][0] == 2 and "gloo" in r0["ranks"][1] and single["ranks"][0] == 1
    assert single["collectives"] == [0, 0]


def test_bucketed_exchange_handles_a_parameter_without_gradient_and_avg(tmp_path):
    """A bucket whose parameter gets no gradient this step is launched by wait() (same order on every rank), AVG over gloo
    is SUM + scale; results equal FlatGradAllReduce's."""
    out = str(tmp_path / "bk")
    port = 29500 + (os.getpid() % 2000) + 31
    mp.spawn(_bucket_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    m = Toy()
    x = torch.randn(10, 6, generator=torch.Generator().manual_seed(1))
    m(x).mean(dim=1).mean().backward()
    assert torch.allclose(got["a"], m.a.grad, atol=1e-6) and torch.allclose(got["b"], m.b.grad, atol=1e-6)
    assert got["unused_grad"] is None and got["collectives"] == 2 and got["overlapped"] == 1


def _bucket_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    from vmlmf_amd.dp import BucketedGradAllReduce, broadcast_parameters, shard_batch
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = Toy()
    broadcast_parameters(m)
    red = BucketedGradAllReduce([[m.b], [m.unused], [m.a]], op="avg")       # backward order: b, (never), a
    x = torch.randn(10, 6, generator=torch.Generator().manual_seed(1))
    red.arm()
    m(shard_batch(x, rank, world)).mean(dim=1).mean().backward()
    red.wait()
    if rank == 0:
        torch.save({"a": m.a.grad.clone(), "b": m.b.grad.clone(), "unused_grad": m.unused.grad,
                    "collectives": red.last_collectives, "overlapped": red.last_overlapped}, out)
    dist.destroy_process_group()
