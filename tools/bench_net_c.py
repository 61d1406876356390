"""Whole training step of the OPP-shaped classifier (BASELINE configs[2] as a Net: two VMLMF layers of 256, rank 24, B 128,
T 24, I 77, Linear(256, 18), fused criterion, fused Adam): eager and replayed from one hipGraph (vmlmf_amd.GraphedTrainStep),
wavefront launches against the chained per-layer kernels (VMLMF_STACK=0)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vmlmf_amd
from vmlmf_amd import MyLSTM, MyVMLMFCell, Net

for mode in ("0", "auto"):
    os.environ["VMLMF_STACK"] = mode
    torch.manual_seed(0)
    net = Net(77, layer_sizes=[256, 256], w_rank=24, u_rank=[24], model=MyLSTM, cell=MyVMLMFCell).cuda()
    x = torch.randn(128, 24, 77, device="cuda")
    t = torch.randint(0, 18, (128,), device="cuda")
    crit = vmlmf_amd.CrossEntropyLoss()
    opt = vmlmf_amd.optim.Adam(net.parameters(), lr=1e-3)
    one = vmlmf_amd.unit_gradient("cuda")

    def eager():
        opt.zero_grad(set_to_none=True)
        crit(net(x), t).backward(one)
        opt.step()

    for _ in range(20):
        eager()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        eager()
    torch.cuda.synchronize(); e = (time.perf_counter() - t0) / 200 * 1e3
    step = vmlmf_amd.GraphedTrainStep(net, crit, opt, x, t)
    for _ in range(20):
        step(x, t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300):
        step(x, t)
    torch.cuda.synchronize(); g = (time.perf_counter() - t0) / 300 * 1e3
    print(json.dumps({"config": "C as Net: 2x256 r24 B128 T24 I77 + Linear(256,18) + CE + Adam", "VMLMF_STACK": mode,
                      "train_step_ms_eager": round(e, 4), "train_step_ms_hipgraph": round(g, 4)}), flush=True)
