"""Which library GEMM form serves the LM head best (R = 8960 rows, H = 650, V = 10000, fp32)?  Forward scores = h W^T (+ b),
backward dh = dz W, dW = dz^T h.  Times every layout variant torch can hand to rocBLAS / hipBLASLt."""
import sys, time, torch
R, H, V = 8960, 650, 10000
dev = "cuda"
torch.manual_seed(0)
h = torch.randn(R, H, device=dev); W = torch.randn(V, H, device=dev) * 0.05; b = torch.randn(V, device=dev)
dz = torch.randn(R, V, device=dev)


def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


for lib in ("default", "hipblaslt", "cublas"):
    if lib != "default":
        try: torch.backends.cuda.preferred_blas_library(lib)
        except Exception as e: print(lib, "unavailable", e); continue
    out = torch.empty(R, V, device=dev); Wt = W.t().contiguous(); outT = torch.empty(V, R, device=dev); hT = h.t().contiguous()
    dh = torch.empty(R, H, device=dev); dW = torch.empty(V, H, device=dev); dWt = torch.empty(H, V, device=dev); dzT = dz.t().contiguous()
    res = {
        "fwd addmm(b, h, W.t()) [NT+bias]": t(lambda: torch.addmm(b, h, W.t(), out=out)),
        "fwd mm(h, W.t()) [NT]": t(lambda: torch.mm(h, W.t(), out=out)),
        "fwd mm(h, Wt) [NN, W^T copy outside]": t(lambda: torch.mm(h, Wt, out=out)),
        "fwd W.t().contiguous()": t(lambda: W.t().contiguous()),
        "fwd mm(W, hT) -> scores^T [NN]": t(lambda: torch.mm(W, hT, out=outT)),
        "fwd mm(W, h.t()) -> scores^T [NT]": t(lambda: torch.mm(W, h.t(), out=outT)),
        "bwd dh = mm(dz, W) [NN]": t(lambda: torch.mm(dz, W, out=dh)),
        "bwd dW = mm(dz.t(), h) [TN]": t(lambda: torch.mm(dz.t(), h, out=dW)),
        "bwd dW^T = mm(h.t(), dz) [TN]": t(lambda: torch.mm(h.t(), dz, out=dWt)),
        "bwd dW = mm(dzT, h) [NN, dz^T given]": t(lambda: torch.mm(dzT, h, out=dW)),
        "bwd dh = mm(dzT.t(), W) [TN, dz^T given]": t(lambda: torch.mm(dzT.t(), W, out=dh)),
        "db = dz.sum(0)": t(lambda: dz.sum(0)),
    }
    for k, v in res.items():
        fl = 2 * R * H * V / (v * 1e-3) / 1e12 if "mm" in k else 0
        print(f"{lib:10s} {k:45s} {v:8.3f} ms  {fl:6.1f} TFLOP/s", flush=True)
