import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from vmlmf_amd import MyLSTM, MyVMLMFCell
torch.manual_seed(0)
rnn = MyLSTM(77, hidden_layer_sizes=[256, 256], batch_first=True, w_rank=24, u_ranks=[24], cell=MyVMLMFCell).cuda()
x = torch.randn(128, 24, 77, device="cuda")
def step():
    rnn.zero_grad(set_to_none=True)
    y, _ = rnn(x)
    y[:, -1].sum().backward()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): step()
res=[]
for rep in range(5):
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(200): g.replay()
    torch.cuda.synchronize(); res.append((time.perf_counter()-t0)/200*1e3)
print("WMIN", os.environ.get("VMLMF_WMIN","32"), "WCHUNKS", os.environ.get("VMLMF_WCHUNKS","64"), " ".join(f"{r:.4f}" for r in res))
