"""One PTB-medium group layer (BASELINE config E shape: H 650, ranks 32 / [32,32], B 256, T 35) forward + backward on
the step-wise path; prints ms per iteration (eager and hipGraph replay).  `--v3` times the plain LM layer instead.
Used under rocprofv3 for the per-kernel split and with VMLMF_SKINNY=0/1 for A/B runs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from vmlmf_amd import MyVMLSTM, MyVMLSTMGroup
torch.manual_seed(0)
H, B, T = 650, 256, 35
if "--batch" in sys.argv:
    B = int(sys.argv[sys.argv.index("--batch") + 1])
if "--v3" in sys.argv:
    l = MyVMLSTM(H, H, w_rank=32, u_ranks=32).cuda()
else:
    l = MyVMLSTMGroup(H, H, w_rank=32, u_ranks=[32, 32]).cuda()
for p in l.parameters(): torch.nn.init.uniform_(p, -0.05, 0.05)
x = 0.05 * torch.randn(T, B, H, device="cuda")
st = (torch.zeros(B, H, device="cuda"), torch.zeros(B, H, device="cuda"))


def step():
    if "--infer" in sys.argv:   # forward only, no tapes (torch.no_grad): the recurrent kernel without its tape stores
        with torch.no_grad():
            l(x, st)
        return
    l.zero_grad(set_to_none=True)
    y, _ = l(x, st)
    y.sum().backward()


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / 10
if "--nograph" in sys.argv:
    print(f"eager {eager * 1e3:.3f} ms per layer iteration")
    sys.exit(0)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print(f"eager {eager * 1e3:.3f} ms, hipGraph {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per layer iteration "
      f"(VMLMF_SKINNY={os.environ.get('VMLMF_SKINNY', '1')})")
