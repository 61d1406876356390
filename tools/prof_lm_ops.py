import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from torch.profiler import profile, ProfilerActivity
from vmlmf_amd import Model, nll_loss, optim
T,B,H,V=35,256,650,10000
torch.manual_seed(0)
m=Model(V,H,2,0.0,0.05,w_rank=32,u_ranks=[32],lstm_type="vmlmf").cuda()
x=torch.randint(0,V,(T,B),device="cuda"); y=torch.randint(0,V,(T,B),device="cuda")
st=m.state_init(B)
def step():
    m.zero_grad(set_to_none=True)
    sc,_=m(x,[(h.detach(),c.detach()) for h,c in st])
    l=nll_loss(sc,y); l.backward()
    optim.clip_sgd_step(m.parameters(), lr=1e-3, max_norm=5.0)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
rows=[e for e in prof.key_averages() if e.device_time_total>50]
rows.sort(key=lambda e:-e.device_time_total)
for e in rows[:25]: print(f"{e.device_time_total:9.0f} us  n={e.count:3d}  {e.key[:90]}")
