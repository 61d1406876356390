"""Config C (fp32): two VMLMF layers of 256, ranks 24, B 128, T 24, I 77 -- a few fwd+bwd iterations for rocprofv3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from vmlmf_amd import MyLSTM, MyVMLMFCell
torch.manual_seed(0)
rnn = MyLSTM(77, hidden_layer_sizes=[256, 256], batch_first=True, w_rank=24, u_ranks=[24], cell=MyVMLMFCell).cuda()
x = torch.randn(128, 24, 77, device="cuda")
if "--bf16" in sys.argv:      # configs[2] as BASELINE.json words it: bf16 MFMA (row-block kernels), fp32 state
    import vmlmf_amd
    vmlmf_amd.set_compute_dtype(rnn, "bf16")
for _ in range(20):
    rnn.zero_grad(set_to_none=True)
    y, _ = rnn(x)
    y[:, -1].sum().backward()
torch.cuda.synchronize()
