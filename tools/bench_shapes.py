"""Time one VMLMF layer (fwd+bwd, hipGraph replay) over a grid of shapes: finds pathological instantiations.
    python tools/bench_shapes.py            # prints one line per shape: ms per step and us per timestep"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vmlmf_amd import MyLSTM, MyVMLMFCell, MyVMLMFCellg2


def run(cell, B, T, I, H, rw, ru):
    torch.manual_seed(0)
    rnn = MyLSTM(I, hidden_layer_sizes=[H], batch_first=True, w_rank=rw, u_ranks=ru, cell=cell).cuda()
    x = torch.randn(B, T, I, device="cuda")

    def step():
        rnn.zero_grad(set_to_none=True)
        y, _ = rnn(x)
        y[:, -1].sum().backward()
    for _ in range(3):
        step()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 30 * 1e3


if __name__ == "__main__":
    for (cell, B, T, I, H, rw, ru) in [
            (MyVMLMFCell, 64, 128, 9, 64, 16, [16]), (MyVMLMFCell, 64, 128, 9, 128, 16, [16]),
            (MyVMLMFCell, 64, 128, 9, 180, 16, [16]), (MyVMLMFCell, 64, 128, 9, 256, 16, [16]),
            (MyVMLMFCell, 64, 128, 9, 256, 24, [24]), (MyVMLMFCell, 64, 128, 9, 256, 32, [32]),
            (MyVMLMFCell, 64, 128, 9, 320, 16, [16]), (MyVMLMFCell, 64, 128, 9, 384, 24, [24]),
            (MyVMLMFCell, 64, 128, 9, 512, 16, [16]), (MyVMLMFCell, 64, 128, 9, 512, 32, [32]),
            (MyVMLMFCellg2, 64, 128, 9, 180, 16, [16, 16]), (MyVMLMFCellg2, 64, 128, 9, 512, 16, [16, 16]),
            (MyVMLMFCell, 64, 128, 9, 640, 16, [16]), (MyVMLMFCell, 64, 128, 9, 256, 16, [40])]:
        ms = run(cell, B, T, I, H, rw, ru)
        print(f"{cell.__name__:14s} B={B} T={T} I={I} H={H} rw={rw} ru={ru}: {ms:8.4f} ms/step  {ms * 1e3 / (2 * T):6.3f} us per dependent step",
              flush=True)
