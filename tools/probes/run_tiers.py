import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from vmlmf_amd import MyLSTM, MyVMLMFCell, MyVMLMFCellg2
def run(cell, B, T, I, H, rw, ru):
    torch.manual_seed(0)
    rnn = MyLSTM(I, hidden_layer_sizes=[H], batch_first=True, w_rank=rw, u_ranks=ru, cell=cell).cuda()
    x = torch.randn(B, T, I, device="cuda")
    def step():
        rnn.zero_grad(set_to_none=True)
        y, _ = rnn(x); y[:, -1].sum().backward()
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
    print(f"{cell.__name__:14s} H={H} r={ru} {(time.perf_counter()-t0)/100*1e3:.4f} ms", flush=True)
for H in (256, 320, 384, 448, 512):
    run(MyVMLMFCell, 64, 128, 9, H, 16, [16])
for H in (320, 384):
    run(MyVMLMFCell, 64, 128, 9, H, 8, [8])
run(MyVMLMFCellg2, 64, 128, 9, 360, 16, [16, 16])
