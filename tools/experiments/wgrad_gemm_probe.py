"""What would a library GEMM do with the weight-gradient products of a PTB layer (R = B T = 8960 rows)?
G1 = dpre^T [qx | Q]  (4 NT x R)(R x 96),  G2 = h^T dQ (650 x R)(R x 64),  G3 = x^T dqx (650 x R)(R x 32); plus the element sums.
Compared against wgrad_mfma_kernel + reduce_cg (146 + 32 us per layer, profiles/r04_config_e_layer_kernel_stats.csv)."""
import time, torch
R = 8960
dev = "cuda"
torch.manual_seed(0)


def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6


for C in (2688, 2600):
    dpre = torch.randn(R, C, device=dev); qxq = torch.randn(R, 96, device=dev); qxq64 = torch.randn(R, 64, device=dev)
    h = torch.randn(R, 650, device=dev); dq = torch.randn(R, 64, device=dev); dqx = torch.randn(R, 32, device=dev)
    for lib in ("default", "hipblaslt", "cublas"):
        if lib != "default":
            try: torch.backends.cuda.preferred_blas_library(lib)
            except Exception as e: print(lib, "unavailable", e); continue
        o1 = torch.empty(C, 96, device=dev); o1t = torch.empty(96, C, device=dev); o164 = torch.empty(C, 64, device=dev)
        o2 = torch.empty(650, 64, device=dev); o3 = torch.empty(650, 32, device=dev)
        res = {
            "G1 mm(dpre.t(), qxq) 96": (t(lambda: torch.mm(dpre.t(), qxq, out=o1)), 2 * R * C * 96),
            "G1^T mm(qxq.t(), dpre) 96": (t(lambda: torch.mm(qxq.t(), dpre, out=o1t)), 2 * R * C * 96),
            "G1 mm(dpre.t(), qxq64) 64": (t(lambda: torch.mm(dpre.t(), qxq64, out=o164)), 2 * R * C * 64),
            "G2 mm(h.t(), dq)": (t(lambda: torch.mm(h.t(), dq, out=o2)), 2 * R * 650 * 64),
            "G3 mm(h.t(), dqx)": (t(lambda: torch.mm(h.t(), dqx, out=o3)), 2 * R * 650 * 32),
        }
        for k, (us, fl) in res.items():
            print(f"C={C} {lib:10s} {k:32s} {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s", flush=True)
    hh = torch.randn(R, C // 4, device=dev)
    print("C=%d element sums  dpre.sum(0): %.1f us   (dpre.view(R,-1,4) * h[...,None]).sum(0): %.1f us" % (
        C, t(lambda: dpre.sum(0)), t(lambda: (dpre.view(R, -1, 4) * hh[..., None]).sum(0))), flush=True)
