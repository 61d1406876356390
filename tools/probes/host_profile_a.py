"""cProfile of the eager training step at config A (Net, one layer of 180, rank 16, B 64, T 128, fused criterion): host time per step."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import vmlmf_amd
from vmlmf_amd import MyLSTM, MyVMLMFCell, Net

torch.manual_seed(0)
net = Net(9, layer_sizes=[180], w_rank=16, u_rank=[16], model=MyLSTM, cell=MyVMLMFCell).cuda()
x = torch.randn(64, 128, 9, device="cuda")
t = torch.randint(0, 18, (64,), device="cuda")
one = vmlmf_amd.unit_gradient("cuda")


def step():
    net.zero_grad(set_to_none=True)
    loss = vmlmf_amd.cross_entropy(net(x), t)
    loss.backward(one)


for _ in range(50):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    step()
torch.cuda.synchronize()
print("eager ms per step", (time.perf_counter() - t0) / 300 * 1e3)
t0 = time.perf_counter()
for _ in range(300):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host ms per step (no sync)", (t1 - t0) / 300 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
