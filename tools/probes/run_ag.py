"""The A-group shape (UCI: one MyVMLMFCellg2 layer of 180, w_rank 16, u_ranks [16, 16], B 64, T 128, I 9) -- a few fwd+bwd
iterations for rocprofv3 --kernel-trace --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from vmlmf_amd import MyLSTM, MyVMLMFCellg2
torch.manual_seed(0)
rnn = MyLSTM(9, hidden_layer_sizes=[180], batch_first=True, w_rank=16, u_ranks=[16, 16], cell=MyVMLMFCellg2).cuda()
x = torch.randn(64, 128, 9, device="cuda")
for _ in range(30):
    rnn.zero_grad(set_to_none=True)
    y, _ = rnn(x)
    y[:, -1].sum().backward()
torch.cuda.synchronize()
