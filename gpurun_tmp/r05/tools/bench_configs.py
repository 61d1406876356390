"""Time the RNN stack (fwd+bwd, hipGraph replay and eager) at every BASELINE.json config on one MI355X, beside the
oracle's CPU port on the host (bounded sample).  Prints one JSON line per config.  Not the graded bench (bench.py)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
import torch
import vmlmf_oracle as O
from vmlmf_amd import MyLSTM, MyLSTMCell, MyVMLMFCell, MyVMLMFCellg2, MyVMLMFgCellg2, MyVMLSTM, MyVMLSTMGroup

DEV = "cuda"


def time_fn(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def graph_of(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


ONLY = os.environ.get("BENCH_ONLY", "")   # substring filter on the config names (A/B runs of single configs)


def har(name, cellcls, B, T, I, layers, rw, ru, variant, cpu=True):
    if ONLY and ONLY not in name:
        return
    if os.environ.get("BENCH_NOCPU"):
        cpu = False
    torch.manual_seed(0)
    rnn = MyLSTM(I, hidden_layer_sizes=layers, batch_first=True, w_rank=rw, u_ranks=ru, cell=cellcls).to(DEV)
    x = torch.randn(B, T, I, device=DEV)

    def step():
        rnn.zero_grad(set_to_none=True)
        y, _ = rnn(x)
        y[:, -1].sum().backward()

    eager = time_fn(step, 50)
    g = graph_of(step)
    graph = time_fn(g.replay, 100)
    out = {"config": name, "B": B, "T": T, "ms_eager": round(eager * 1e3, 4), "ms_hipgraph": round(graph * 1e3, 4),
           "timesteps_per_s": round(T / graph, 1)}
    if cpu:
        torch.set_num_threads(1)
        Ps = []
        in_size = I
        for H in layers:
            Ps.append(O.to_torch(O.make_params(variant, in_size, H, rw, ru if variant in (O.V2, O.V6) else ru[0]), requires_grad=True))
            in_size = H
        xc = x.cpu()
        def cpu_step():
            h = xc
            for P in Ps:
                h, _, _ = O.literal_sequence(variant, P, h, time_major=False)
            h[:, -1].sum().backward()
        cpu_step()
        t0 = time.perf_counter(); cpu_step(); dt = time.perf_counter() - t0
        out["cpu_port_s_per_step_1thread"] = round(dt, 4)
        out["speedup_vs_cpu"] = round(dt / graph, 1)
    print(json.dumps(out), flush=True)


def lm(name, cls, variant, B, T, H, rw, ru, nlayers, cpu_B=None):
    if ONLY and ONLY not in name:
        return
    torch.manual_seed(0)
    layers = [cls(H, H, w_rank=rw, u_ranks=ru).to(DEV) for _ in range(nlayers)]
    for l in layers:
        for p in l.parameters():
            torch.nn.init.uniform_(p, -0.05, 0.05)
    x = 0.05 * torch.randn(T, B, H, device=DEV)
    states = [(torch.zeros(B, H, device=DEV), torch.zeros(B, H, device=DEV)) for _ in layers]

    def step():
        for l in layers:
            l.zero_grad(set_to_none=True)
        h = x
        for l, st in zip(layers, states):
            h, _ = l(h, st)
        h.sum().backward()

    eager = time_fn(step, 10)
    g = graph_of(step)
    graph = time_fn(g.replay, 20)
    out = {"config": name, "B": B, "T": T, "layers": nlayers, "ms_eager": round(eager * 1e3, 3),
           "ms_hipgraph": round(graph * 1e3, 3), "timesteps_per_s": round(T / graph, 1)}
    if cpu_B:
        torch.set_num_threads(1)
        P = O.to_torch(O.make_params(variant, H, H, rw, ru, scale=0.05), requires_grad=True)
        xc = 0.05 * torch.randn(T, cpu_B, H)
        t0 = time.perf_counter()
        y, _, _ = O.literal_sequence(variant, P, xc, torch.zeros(cpu_B, H), torch.zeros(cpu_B, H), v4_scratch_rows=cpu_B)
        y.sum().backward()
        dt = time.perf_counter() - t0
        out["cpu_port_s_per_layer_step_1thread"] = round(dt, 4)
        out["cpu_batch"] = cpu_B
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), flush=True)
    har("A/B: UCI V1 H=180 r=16 B=64 T=128", MyVMLMFCell, 64, 128, 9, [180], 16, [16], O.V1)
    har("A-group: UCI V2 H=180 r=[16,16] B=64", MyVMLMFCellg2, 64, 128, 9, [180], 16, [16, 16], O.V2)
    har("C(fp32): OPP V1 2x256 r=24 B=128 T=24", MyVMLMFCell, 128, 24, 77, [256, 256], 24, [24], O.V1)
    har("D@1GPU: UCI V1 B=512", MyVMLMFCell, 512, 128, 9, [180], 16, [16], O.V1, cpu=False)
    har("demo.sh: OPP V1 H=180 w8 u6 B=81 T=24", MyVMLMFCell, 81, 24, 77, [180], 8, [6], O.V1)
    har("demo.sh: OPP V2 H=180 w8 u[2,4] B=81", MyVMLMFCellg2, 81, 24, 77, [180], 8, [2, 4], O.V2)
    if "--compare" in sys.argv:   # the reference's compression-vs-speed comparison: same shape, cells without vm
        har("A-lmf: UCI MyLSTMCell low-rank H=180 r=16 B=64", MyLSTMCell, 64, 128, 9, [180], 16, [16], O.V5)
        har("A-group-novm: UCI MyVMLMFgCellg2 H=180 r=[16,16] B=64", MyVMLMFgCellg2, 64, 128, 9, [180], 16, [16, 16], O.V6)
        sys.exit(0)
    lm("E: PTB V4 group H=650 r=32/[32,32] B=256 T=35 (2 layers)", MyVMLSTMGroup, O.V4, 256, 35, 650, 32, [32, 32], 2, cpu_B=40)
    lm("E-shape: PTB V3 H=650 r=32 B=256 T=35 (2 layers)", MyVMLSTM, O.V3, 256, 35, 650, 32, 32, 2, cpu_B=64)
