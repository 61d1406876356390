"""How long does the HOST need to enqueue one training step (no sync)?  If that is >= the GPU time per step the
bench is host-bound."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from vmlmf_amd import MyLSTM, MyVMLMFCell, Net
torch.manual_seed(0)
net = Net(9, layer_sizes=[180], w_rank=16, u_rank=[16], model=MyLSTM, cell=MyVMLMFCell).cuda()
x = torch.randn(64, 128, 9, device="cuda"); tgt = torch.randint(0, 6, (64,), device="cuda")
def step():
    net.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(net(x), tgt)
    loss.backward()
for _ in range(20): step()
torch.cuda.synchronize()
for n in (50, 200):
    t0 = time.perf_counter()
    for _ in range(n): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"n={n}: enqueue {1e6*(t1-t0)/n:.1f} us/step, total {1e6*(t2-t0)/n:.1f} us/step")
# rnn only
rnn = net.rnn
def step2():
    rnn.zero_grad(set_to_none=True)
    y, _ = rnn(x)
    y.sum().backward()
for _ in range(10): step2()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step2()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"rnn only: enqueue {1e6*(t1-t0)/200:.1f} us/step, total {1e6*(t2-t0)/200:.1f} us/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(100): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
