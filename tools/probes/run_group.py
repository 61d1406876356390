"""UCI shape with the group cell (MyVMLMFCellg2, ranks [16,16]): a few eager fwd+bwd iterations for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vmlmf_amd import MyLSTM, MyVMLMFCellg2
torch.manual_seed(0)
rnn = MyLSTM(9, hidden_layer_sizes=[180], batch_first=True, w_rank=16, u_ranks=[16, 16], cell=MyVMLMFCellg2).cuda()
x = torch.randn(64, 128, 9, device="cuda")
for _ in range(20):
    rnn.zero_grad(set_to_none=True)
    y, _ = rnn(x)
    y[:, -1].sum().backward()
torch.cuda.synchronize()
