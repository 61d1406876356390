"""Two PTB layers with the sequence cut into time chunks and the layers' chunks on two streams: layer 2 works on chunk k while layer 1
works on chunk k + 1 (the clustered recurrent kernels are latency-bound and leave room for a second workgroup per CU).
    python tools/experiments/chunk_pipeline_probe.py [--v3] [--chunks N]
Prints ms per forward + backward of both layers: chained as the LM network runs them, and pipelined."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vmlmf_amd import MyVMLSTM, MyVMLSTMGroup
torch.manual_seed(0)
H, B, T = 650, 256, 35
NCH = int(sys.argv[sys.argv.index("--chunks") + 1]) if "--chunks" in sys.argv else 2
mk = (lambda: MyVMLSTM(H, H, w_rank=32, u_ranks=32)) if "--v3" in sys.argv else (lambda: MyVMLSTMGroup(H, H, w_rank=32, u_ranks=[32, 32]))
L = [mk().cuda() for _ in range(2)]
for l in L:
    for p in l.parameters(): torch.nn.init.uniform_(p, -0.05, 0.05)
x = 0.05 * torch.randn(T, B, H, device="cuda")
dy = torch.randn(T, B, H, device="cuda")
st0 = [(torch.zeros(B, H, device="cuda"), torch.zeros(B, H, device="cuda")) for _ in L]
bounds = [round(i * T / NCH) for i in range(NCH + 1)]
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def chained():
    for l in L: l.zero_grad(set_to_none=True)
    h = x
    for l, st in zip(L, st0): h, _ = l(h, st)
    torch.autograd.backward(h, dy)
    return h


def pipelined():
    for l in L: l.zero_grad(set_to_none=True)
    cur = torch.cuda.current_stream()
    sA.wait_stream(cur); sB.wait_stream(cur)
    keep, outs = [], []
    s1, s2 = st0[0], st0[1]
    for k in range(NCH):
        xs = x[bounds[k]:bounds[k + 1]]
        with torch.cuda.stream(sA):
            y1, s1 = L[0](xs, s1)
        ev = torch.cuda.Event(); ev.record(sA)
        with torch.cuda.stream(sB):
            sB.wait_event(ev)
            y2, s2 = L[1](y1, s2)
        keep += [y1, s1, s2]; outs.append(y2)
    with torch.cuda.stream(sB):
        torch.autograd.backward(outs, [dy[bounds[k]:bounds[k + 1]] for k in range(NCH)])
    cur.wait_stream(sA); cur.wait_stream(sB)
    return torch.cat(outs), keep


def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

ya = chained(); ga = [p.grad.clone() for l in L for p in l.parameters()]
yb, _ = pipelined(); torch.cuda.synchronize(); gb = [p.grad.clone() for l in L for p in l.parameters()]
print("max |y diff| %.3g, max rel grad diff %.3g" % (float((ya - yb).abs().max()), max(float((a - b).abs().max() / (a.abs().max() + 1e-12)) for a, b in zip(ga, gb))))
print("chained %.3f ms, pipelined (%d chunks) %.3f ms per forward + backward of two layers" % (t(chained), NCH, t(pipelined)))
