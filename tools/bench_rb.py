"""A/B of the recurrent kernel families on one MI355X: the one-row-per-CU VALU kernels (or the step-wise path for large
layers) against the row-block MFMA kernels (vmlmf_tune("rb", 0 / 1)), forward + backward of the RNN stack, hipGraph
replay, plus the library's own event timing of the two recurrent kernels.  One JSON line per (shape, batch, family)."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from vmlmf_amd import MyLSTM, MyVMLMFCell, MyVMLMFCellg2, MyVMLSTM, MyVMLSTMGroup, _lib

DEV = "cuda"


def time_fn(fn, iters):
    for _ in range(3):
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


def kernel_us(fn, n=5):
    lib = _lib.lib()
    lib.vmlmf_profile_enable(0xff)
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    usec = (ctypes.c_float * _lib.NKERNELS)()
    cnt = (ctypes.c_int32 * _lib.NKERNELS)()
    lib.vmlmf_profile_read(usec, cnt, 1)
    lib.vmlmf_profile_enable(0)
    return {lib.vmlmf_kernel_name(k).decode().replace("_kernel", ""): round(usec[k] / n, 1) for k in range(8) if cnt[k]}


def run(name, make, step_of, rb, iters):
    _lib.tune("rb", rb)
    torch.manual_seed(0)
    mod, x = make()
    step = step_of(mod, x)
    ks = kernel_us(step)
    try:
        g = graph_of(step)
        ms = time_fn(g.replay, iters) * 1e3
    except Exception as e:  # noqa: BLE001
        ms = time_fn(step, iters) * 1e3
        name += " (eager: %s)" % type(e).__name__
    print(json.dumps({"shape": name, "family": "row-block MFMA" if rb else "VALU row-per-CU / step-wise", "ms_fwd_bwd": round(ms, 4),
                      "kernels_us_per_step": ks}), flush=True)


def har(B, T=128, I=9, H=180, rw=16, ru=(16,), layers=1, cell=MyVMLMFCell):
    def make():
        rnn = MyLSTM(I, hidden_layer_sizes=[H] * layers, batch_first=True, w_rank=rw, u_ranks=list(ru), cell=cell).to(DEV)
        return rnn, torch.randn(B, T, I, device=DEV)

    def step_of(rnn, x):
        def step():
            rnn.zero_grad(set_to_none=True)
            y, _ = rnn(x)
            y[:, -1].sum().backward()
        return step
    return make, step_of


def lm(B, group, T=35, H=650, layers=2):
    def make():
        ls = [(MyVMLSTMGroup(H, H, w_rank=32, u_ranks=[32, 32]) if group else MyVMLSTM(H, H, w_rank=32, u_ranks=32)).to(DEV)
              for _ in range(layers)]
        for l in ls:
            for p in l.parameters():
                torch.nn.init.uniform_(p, -0.05, 0.05)
        return torch.nn.ModuleList(ls), 0.05 * torch.randn(T, B, H, device=DEV)

    def step_of(ls, x):
        st = [(torch.zeros(B, H, device=DEV), torch.zeros(B, H, device=DEV)) for _ in ls]

        def step():
            ls.zero_grad(set_to_none=True)
            h = x
            for l, s in zip(ls, st):
                h, _ = l(h, s)
            h.sum().backward()
        return step
    return make, step_of


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "uci"):
        for B in (64, 256, 512, 1024, 2048):
            for rb in (0, 1):
                run(f"UCI V1 H=180 r=16 T=128 B={B}", *har(B), rb, 30)
    if which == "uci_rows":     # live rows per row-block workgroup: 16 / 8 / 4
        for B in (512, 1024, 2048, 4096):
            run(f"UCI V1 H=180 r=16 T=128 B={B}", *har(B), 0, 20)
            for rows in (16, 8, 4):
                _lib.tune("rb_rows", rows)
                run(f"UCI V1 H=180 r=16 T=128 B={B}, {rows} live rows per workgroup", *har(B), 1, 20)
            _lib.tune("rb_rows", 0)
    if which in ("all", "c"):
        for rb in (0, 1):
            run("C: OPP V1 2x256 r=24 B=128 T=24", *har(128, 24, 77, 256, 24, (24,), 2), rb, 50)
        for rb in (0, 1):
            run("C-shape at B=1024", *har(1024, 24, 77, 256, 24, (24,), 2), rb, 30)
    if which in ("all", "e"):
        run("E: PTB V4 group H=650 [32,32] B=256 T=35 x2 layers", *lm(256, True), 0, 10)
        for S in (4, 8, 16):
            _lib.tune("rb_cluster", S)
            run(f"E: PTB V4 group H=650 [32,32] B=256 T=35 x2 layers, cluster of {S}", *lm(256, True), 1, 10)
        run("E-shape: PTB V3 H=650 r=32 B=256 T=35 x2 layers", *lm(256, False), 0, 10)
        for S in (4, 8, 16):
            _lib.tune("rb_cluster", S)
            run(f"E-shape: PTB V3 H=650 r=32 B=256 T=35 x2 layers, cluster of {S}", *lm(256, False), 1, 10)
        _lib.tune("rb_cluster", 0)
    if which == "e_rows":  # configs[4]: fewer live rows per workgroup with a smaller cluster (the same 256 workgroups)
        for S, rows in ((16, 16), (8, 8), (4, 4), (8, 16)):
            _lib.tune("rb_cluster", S)
            _lib.tune("rb_rows", rows)
            run(f"E: PTB V4 group B=256 T=35 x2 layers, cluster of {S}, {rows} live rows", *lm(256, True), 1, 10)
            run(f"E-shape: PTB V3 B=256 T=35 x2 layers, cluster of {S}, {rows} live rows", *lm(256, False), 1, 10)
        _lib.tune("rb_cluster", 0)
        _lib.tune("rb_rows", 0)
    if which == "e32":     # configs[4] per GPU on an 8-GPU node: 32 rows
        for S in (4, 8, 16):
            _lib.tune("rb_cluster", S)
            run(f"E/8 GPUs: PTB V4 group B=32 T=35 x2 layers, cluster of {S}", *lm(32, True), 1, 10)
        run("E/8 GPUs: PTB V4 group B=32 T=35 x2 layers", *lm(32, True), 0, 10)
        _lib.tune("rb_cluster", 0)
    _lib.tune("rb", -1)
