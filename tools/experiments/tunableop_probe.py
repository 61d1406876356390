import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vmlmf_amd import functional as F
R, H, V = 8960, 650, 10000
dev = torch.device("cuda:0")
h = torch.randn(R, H, device=dev); w = torch.randn(V, H, device=dev) * 0.05; dz = torch.randn(R, V, device=dev)
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("before: fwd %.3f dh %.3f dw %.3f" % (t(lambda: torch.mm(h, w.t())), t(lambda: torch.mm(dz, w)), t(lambda: torch.mm(dz.t(), h))))
forms = F.head_forms(R, H, V, dev)
print({k: (v[0], v[1], v[3]) for k, v in forms.items()})
print("enabled", torch.cuda.tunable.is_enabled(), "tuning", torch.cuda.tunable.tuning_is_enabled())
for r in torch.cuda.tunable.get_results(): print(r)
print("after:  fwd %.3f dh %.3f dw %.3f dw2 %.3f" % (t(lambda: torch.mm(h, w.t())), t(lambda: torch.mm(dz, w)), t(lambda: torch.mm(dz.t(), h)), t(lambda: torch.mm(h.t(), dz))))
