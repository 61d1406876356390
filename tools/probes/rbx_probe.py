"""Two PTB layers (H 650, ranks 32 / [32, 32], T 35) forward + backward at the rows a GPU of an 8-GPU node holds: chained per-layer
launches (VMLMF_STACK=0) against the clustered one-launch form (csrc/vmlmf_rbx.hip), eager and replayed from a hipGraph."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from vmlmf_amd import functional as F, _lib
from vmlmf_amd.lm import MyVMLSTM, MyVMLSTMGroup

plain = "--plain" in sys.argv
T, H = 35, 650
torch.manual_seed(0)
NL = 1 if "--one" in sys.argv else 2
layers = [(MyVMLSTM(H, H, w_rank=32, u_ranks=32) if plain else MyVMLSTMGroup(H, H, w_rank=32, u_ranks=[32, 32], g=2)).cuda() for _ in range(NL)]
for l in layers:
    for p in l.parameters():
        torch.nn.init.uniform_(p, -0.05, 0.05)
variant = layers[0].variant
ur = [32] if plain else [32, 32]
g = 1 if plain else 2


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def graphed(fn, n=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    torch.cuda.synchronize()
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        gr.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for B in [int(a) for a in sys.argv[1:] if a.isdigit()] or [32, 64, 128]:
    x = (0.05 * torch.randn(T, B, H, device="cuda")).requires_grad_(True)
    st = [(torch.zeros(B, H, device="cuda"), torch.zeros(B, H, device="cuda")) for _ in layers]
    h0 = torch.stack([s[0] for s in st])
    c0 = torch.stack([s[1] for s in st])

    def chained():
        for l in layers:
            l.zero_grad(set_to_none=True)
        x.grad = None
        h = x
        for l, s in zip(layers, st):
            h, _ = l(h, s)
        h.sum().backward()

    def stacked():
        for l in layers:
            l.zero_grad(set_to_none=True)
        x.grad = None
        out = F.vmlmf_stack(variant, x, [l.kernel_params() for l in layers], 32, ur, g=g, time_major=True, h0=h0, c0=c0)
        assert out is not None
        out[0].sum().backward()

    rec = {"B": B, "plain": plain}
    if "--stacked-only" in sys.argv:      # (for rocprofv3 --kernel-trace --stats: 23 eager steps of the one-launch form, nothing else)
        rec["stacked_eager_ms"] = round(timed(stacked), 4)
    elif "--chained-only" in sys.argv:
        rec["chained_eager_ms"] = round(timed(chained), 4)
    else:
        rec["chained_eager_ms"] = round(timed(chained), 4)
        rec["stacked_eager_ms"] = round(timed(stacked), 4)
        rec["chained_graph_ms"] = round(graphed(chained), 4)
        rec["stacked_graph_ms"] = round(graphed(stacked), 4)
    rec["status"] = _lib.lib().vmlmf_check_status()
    print(json.dumps(rec), flush=True)
