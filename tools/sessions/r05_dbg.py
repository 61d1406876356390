import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vmlmf_amd
from vmlmf_amd import Net, MyLSTM, MyVMLMFCell, _lib
torch.manual_seed(0)
net = Net(9, layer_sizes=[180], w_rank=16, u_rank=[16], model=MyLSTM, cell=MyVMLMFCell).cuda()
x = torch.randn(64, 128, 9, device="cuda")
outs = {}
for d in (1, 0):
    _lib.tune("direct", d)
    with torch.no_grad():
        y, hid = net.rnn.run_layers(x)
    outs[d] = (y.clone(), hid[-1].clone())
dy = (outs[1][0] - outs[0][0]).abs()
print("max |y1 - y0|", float(dy.max()), "at t", int(dy.amax(dim=(0, 2)).argmax()), "unit", int(dy.amax(dim=(0, 1)).argmax()))
print("per-unit max diff (first 24):", [round(float(v), 6) for v in dy.amax(dim=(0, 1))[:24]])
print("t=0 diff max", float(dy[:, 0].max()), "t=1", float(dy[:, 1].max()))
