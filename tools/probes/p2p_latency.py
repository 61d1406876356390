"""Latency of vmlmf_p2p_allreduce in a group of ONE rank (the two launches + the local part of the protocol; no peer, no xGMI):
the floor of the one-shot exchange for DESIGN.md section 6's predicted table.  30 951 floats = the HAR network's gradients."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
from vmlmf_amd.dp import P2PExchange
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29731")
dist.init_process_group("gloo", rank=0, world_size=1)
torch.cuda.set_device(0)
ex = P2PExchange(torch.device("cuda:0"), 40000)
assert ex.handle, ex.error
buf = torch.randn(30951, device="cuda")
for _ in range(20): ex.all_reduce([buf], "avg")
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ex.all_reduce([buf], "avg")
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(10): ex.all_reduce([buf], "avg")
for _ in range(5): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): g.replay()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 500
print({"p2p_allreduce_us_world1_graph_replay": round(dt * 1e6, 2), "floats": 30951, "finite": bool(torch.isfinite(buf).all())})
