import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vmlmf_amd import MyVMLSTMGroup
torch.manual_seed(0)
H, B, T = 650, 256, 35
l = MyVMLSTMGroup(H, H, w_rank=32, u_ranks=[32, 32]).cuda()
for p in l.parameters(): torch.nn.init.uniform_(p, -0.05, 0.05)
x = 0.05 * torch.randn(T, B, H, device="cuda")
st = (torch.zeros(B, H, device="cuda"), torch.zeros(B, H, device="cuda"))
for _ in range(5):
    l.zero_grad(set_to_none=True)
    y, _ = l(x, st)
    y.sum().backward()
torch.cuda.synchronize()
