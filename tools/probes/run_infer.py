"""Inference (torch.no_grad) of Net at the headline shape: eager and hipGraph replay, ms per batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vmlmf_amd import MyLSTM, MyVMLMFCell, Net
torch.manual_seed(0)
net = Net(9, layer_sizes=[180], w_rank=16, u_rank=[16], model=MyLSTM, cell=MyVMLMFCell).cuda().eval()
x = torch.randn(64, 128, 9, device="cuda")
with torch.no_grad():
    for _ in range(5): out = net(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): out = net(x)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 200 * 1e3
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): out = net(x)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): out = net(x)
    for _ in range(10): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): g.replay()
    torch.cuda.synchronize()
print(f"inference B=64 T=128: eager {eager:.4f} ms, hipGraph {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms per batch")
