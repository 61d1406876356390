"""GPU: the data-parallel path on the REAL kernels (BASELINE configs[3]).

* one GPU stands in for eight: the gradients of 8 contiguous shards of 64 rows through Net, averaged, equal the
  B = 512 gradients of the same Net and the fp64 literal oracle's at B = 512, T = 128, H = 180;
* the C-ABI RCCL entry points (vmlmf_comm_*, vmlmf_flat_allreduce_group) run on hardware in a group of one.
"""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest
import torch

import vmlmf_oracle as O
from hip_util import assert_grad, assert_out

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV = "cuda"


def _net_and_data():
    import bench
    from vmlmf_amd import MyLSTM, MyVMLMFCell, Net
    torch.manual_seed(0)
    net = Net(bench.I, layer_sizes=[bench.H], w_rank=bench.RW, u_rank=[bench.RU], model=MyLSTM, cell=MyVMLMFCell)
    P = bench.numpy_params(3)
    with torch.no_grad():
        for k, v in P.items():
            getattr(net.rnn.rnncells[0], k).copy_(torch.tensor(v))
    x, tgt = bench.synthetic_batch(0, 512, 512)          # the global minibatch of the strong-scaling mode
    return net.to(DEV), P, x, tgt


def _grads(net, x, tgt):
    import vmlmf_amd
    net.zero_grad(set_to_none=True)
    loss = vmlmf_amd.cross_entropy(net(x), tgt)
    loss.backward()
    return float(loss), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}


def test_eight_shards_of_64_averaged_equal_the_b512_step_and_the_oracle():
    net, P, x_np, tgt_np = _net_and_data()
    x, tgt = torch.tensor(x_np, device=DEV), torch.tensor(tgt_np, device=DEV)
    from vmlmf_amd.dp import shard_batch
    loss_full, g_full = _grads(net, x, tgt)
    # what 8 ranks compute: each the mean-CE gradient of its contiguous 64 rows; AVG over ranks (dp.FlatGradAllReduce op="avg")
    acc, losses = None, []
    for r in range(8):
        xs, ts = shard_batch(x, r, 8), shard_batch(tgt, r, 8)
        assert xs.shape[0] == 64 and xs.data_ptr() == x[64 * r:].data_ptr()
        l, g = _grads(net, xs, ts)
        losses.append(l)
        acc = g if acc is None else {k: acc[k] + g[k] for k in g}
    g_avg = {k: v / 8 for k, v in acc.items()}
    assert set(g_avg) == set(g_full) and len(g_full) == 10     # 8 cell tensors + lin.weight + lin.bias; Net.cell gets none
    assert abs(np.mean(losses) - loss_full) <= 1e-6 * abs(loss_full)
    # stated tolerance, shard-vs-full on the same kernels: summation order differs (8 partial sums), nothing else
    for k in g_full:
        assert_grad(g_avg[k].cpu().numpy(), g_full[k].cpu().numpy(), f"shards-vs-full.{k}", rel=2e-5)

    # fp64 literal oracle at the full configs[3] size (B 512, T 128, I 9, H 180): the tolerance of every other parity test
    Pt = O.to_torch(P, dtype=torch.float64, requires_grad=True)
    lw = net.lin.weight.detach().cpu().double().requires_grad_(True)
    lb = net.lin.bias.detach().cpu().double().requires_grad_(True)
    loss_ref, logits_ref = O.literal_train_step_har(Pt, lw, lb, torch.tensor(x_np).double(), torch.tensor(tgt_np))
    loss_ref.backward()
    assert abs(loss_full - float(loss_ref)) <= 1e-5 * abs(float(loss_ref))
    with torch.no_grad():
        assert_out(net(x).cpu().numpy(), logits_ref.detach().numpy(), "logits B=512")
    ref = {f"rnn.rnncells.0.{k}": v.grad.numpy() for k, v in Pt.items()}
    ref["lin.weight"], ref["lin.bias"] = lw.grad.numpy(), lb.grad.numpy()
    for k in g_full:
        assert_grad(g_full[k].cpu().numpy(), ref[k], f"full-vs-oracle.{k}")
        assert_grad(g_avg[k].cpu().numpy(), ref[k], f"shards-vs-oracle.{k}")


def test_cabi_rccl_allreduce_in_a_group_of_one():
    """vmlmf_comm_unique_id -> vmlmf_comm_init -> vmlmf_flat_allreduce(_group) -> vmlmf_comm_destroy on the device: SUM and
    AVG over one rank leave the buffers unchanged; the call is asynchronous on torch's current stream."""
    import torch.distributed as dist
    from vmlmf_amd import _lib
    from vmlmf_amd.dp import CabiComm, FlatGradAllReduce
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dev = torch.device("cuda", torch.cuda.current_device())
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        comm = CabiComm(dev)
        a = torch.randn(30951, device=dev)
        b = torch.randn(3258, device=dev)
        a0, b0 = a.clone(), b.clone()
        comm.all_reduce([a, b], "avg")
        comm.all_reduce([a], "sum")
        torch.cuda.synchronize()
        assert torch.equal(a, a0) and torch.equal(b, b0)
        lib = _lib.lib()
        assert lib.vmlmf_flat_allreduce(ctypes.c_void_p(a.data_ptr()), a.numel(), 9, comm.handle, None) == _lib.E_BADARG
        comm.close()
        # the reducer picks the C-ABI transport on request and reports it
        p = torch.nn.Parameter(torch.randn(8, 4, device=dev))
        q = torch.nn.Parameter(torch.randn(5, device=dev))
        p.grad, q.grad = torch.randn_like(p), torch.randn_like(q)
        gp, gq = p.grad.clone(), q.grad.clone()
        red = FlatGradAllReduce([p, q], op="avg", transport="cabi")
        red.always = True
        red.reduce()
        torch.cuda.synchronize()
        assert red.transport_used().startswith("cabi")
        assert torch.equal(p.grad, gp) and torch.equal(q.grad, gq)
    finally:
        dist.destroy_process_group()
