"""Worker of tests/test_gpu_rehearsal.py::test_two_processes_exchange_over_the_peer_to_peer_path: one of two ranks on ONE GPU.
The handles travel over gloo, the buffers over the hipIpc-mapped staging areas (vmlmf_p2p_*, ABI 13); every exchange is compared
bit for bit with the same buffers reduced over gloo.  Prints "P2P-OK <n exchanges>" on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

from vmlmf_amd.dp import FlatGradAllReduce, P2PExchange


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(100 + rank)
    ex = P2PExchange(dev, 40000)
    assert ex.handle, ex.error
    done = 0
    for n, op in ((30951, "avg"), (7, "sum"), (1024, "sum"), (30951, "avg"), (30951, "avg"), (4, "avg"), (39999, "sum")):
        g = torch.randn(n, generator=gen)
        ref = g.clone()
        dist.all_reduce(ref, op=dist.ReduceOp.SUM)
        if op == "avg":
            ref.mul_(1.0 / world)
        buf = g.to(dev)
        ex.all_reduce([buf], op)
        torch.cuda.synchronize()
        assert torch.equal(buf.cpu(), ref), (n, op, float((buf.cpu() - ref).abs().max()))
        done += 1
    # a run of back-to-back exchanges of random sizes without a host synchronisation in between (the parities alternate, a fast rank
    # runs ahead of a slow one): checked at the end against gloo
    soak = int(os.environ.get("VMLMF_P2P_SOAK", "200"))
    sizes = torch.randint(1, 40000, (soak,), generator=torch.Generator().manual_seed(7)).tolist()   # (the same on both ranks)
    bufs, refs = [], []
    for n in sizes:
        g = torch.randn(n, generator=gen)
        bufs.append(g.to(dev))
        refs.append(g)
    for b in bufs:
        ex.all_reduce([b], "sum")
    torch.cuda.synchronize()
    for b, r in zip(bufs, refs):
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        assert torch.equal(b.cpu(), r), b.numel()
    done += 1
    ex.close()
    # through the gradient reducer of the data-parallel path: the parameters' gradients tile one flat allocation
    flat = torch.randn(5000, generator=gen).to(dev)
    w = torch.nn.Parameter(torch.zeros(40, 100, device=dev))
    b = torch.nn.Parameter(torch.zeros(1000, device=dev))
    w.grad, b.grad = flat[:4000].view(40, 100), flat[4000:]
    ref = flat.cpu().clone()
    dist.all_reduce(ref, op=dist.ReduceOp.SUM)
    ref.mul_(1.0 / world)
    red = FlatGradAllReduce([w, b], op="avg", transport="p2p")
    red.reduce()
    torch.cuda.synchronize()
    assert red.transport_used().startswith("p2p"), red.transport_used()
    assert torch.equal(flat.cpu(), ref)
    assert red.exchange_ranks()[0] == world
    print("P2P-OK", done + 1, flush=True)   # 7 single exchanges + the run + the reducer
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
