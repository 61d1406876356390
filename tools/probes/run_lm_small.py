"""Small LM layers on the persistent path: MyVMLSTM vs MyVMLSTMGroup at the same size (hipGraph replay, fwd+bwd)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vmlmf_amd import MyVMLSTM, MyVMLSTMGroup

def run(tag, layer, B, T, H):
    for p in layer.parameters(): torch.nn.init.uniform_(p, -0.05, 0.05)
    x = 0.05 * torch.randn(T, B, H, device="cuda")
    st = (torch.zeros(B, H, device="cuda"), torch.zeros(B, H, device="cuda"))
    def step():
        layer.zero_grad(set_to_none=True)
        y, _ = layer(x, st)
        y.sum().backward()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): step()
    for _ in range(10): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): g.replay()
    torch.cuda.synchronize()
    print(f"{tag:44s} {(time.perf_counter() - t0) / 100 * 1e3:.4f} ms")

for H in (200, 256, 400, 512):
    run(f"V3 H={H} rank 32 B=64 T=35", MyVMLSTM(H, H, w_rank=16, u_ranks=32).cuda(), 64, 35, H)
    run(f"V4 H={H} ranks [16,16] B=64 T=35", MyVMLSTMGroup(H, H, w_rank=16, u_ranks=[16, 16]).cuda(), 64, 35, H)
