"""Probe: config C through the wavefront launches a few times, print the time per iteration and the error word of the
progress hand-over (0 = no bounded spin gave up)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("VMLMF_STACK", "1")
os.environ["VMLMF_PYBIND"] = "ctypes"      # the probe reads the ctypes binding's workspace (the C++ binding keeps its own)
import torch
from vmlmf_amd import MyLSTM, MyVMLMFCell, functional as F
torch.manual_seed(0)
L, B, T, H = 2, 128, 24, 256
rnn = MyLSTM(77, hidden_layer_sizes=[H] * L, batch_first=True, w_rank=24, u_ranks=[24], cell=MyVMLMFCell).cuda()
x = torch.randn(B, T, 77, device="cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for i in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rnn.zero_grad(set_to_none=True)
    y, _ = rnn(x)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    y[:, -1].sum().backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ws = list(F._WORKSPACE.values())[0]
    err = ws[(L - 1) * B * 32 * 4:(L - 1) * B * 32 * 4 + 4].view(torch.int32).item()
    print(f"iter {i}: fwd {1e3 * (t1 - t0):.2f} ms  bwd {1e3 * (t2 - t1):.2f} ms  err_word {err}  finite {bool(torch.isfinite(y).all())}", flush=True)
