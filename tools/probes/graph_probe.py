import os, sys, time, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from vmlmf_amd import MyLSTM, MyVMLMFCell, Net
torch.manual_seed(0)
net = Net(9, layer_sizes=[180], w_rank=16, u_rank=[16], model=MyLSTM, cell=MyVMLMFCell).cuda()
x = torch.randn(64, 128, 9, device="cuda"); tgt = torch.randint(0, 6, (64,), device="cuda")
def fwd_bwd():
    net.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(net(x), tgt)
    loss.backward()
    return loss
for _ in range(5): fwd_bwd()
torch.cuda.synchronize(); print("eager ok", flush=True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): fwd_bwd()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize(); print("side-stream warmup ok", flush=True)
g = torch.cuda.CUDAGraph()
net.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    l = fwd_bwd()
print("captured", flush=True)
for _ in range(10): g.replay()
torch.cuda.synchronize(); print("replay ok", l.item(), flush=True)
t0 = time.perf_counter()
for _ in range(200): g.replay()
torch.cuda.synchronize()
print("graph ms/step", (time.perf_counter() - t0) / 200 * 1e3)
