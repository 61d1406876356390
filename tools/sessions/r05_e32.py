"""two config E group layers at 32 and 256 rows: eager launches against a hipGraph replay of the same forward + backward"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from vmlmf_amd import MyVMLSTMGroup
torch.manual_seed(0)
layers = [MyVMLSTMGroup(650, 650, w_rank=32, u_ranks=[32, 32]).cuda() for _ in range(2)]
for l in layers:
    for p in l.parameters():
        torch.nn.init.uniform_(p, -0.05, 0.05)
for B in (32, 256):
    xe = 0.05 * torch.randn(35, B, 650, device="cuda")
    st = [(torch.zeros(B, 650, device="cuda"), torch.zeros(B, 650, device="cuda")) for _ in layers]
    def fbe():
        for l in layers:
            l.zero_grad(set_to_none=True)
        h = xe
        for l, s in zip(layers, st):
            h, _ = l(h, s)
        h.sum().backward()
    def timeit(fn, n=30):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    e = timeit(fbe)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): fbe()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fbe()
    r = timeit(g.replay)
    print(f"B={B}: eager {e:.4f} ms, hipGraph replay {r:.4f} ms", flush=True)
