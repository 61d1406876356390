import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vmlmf_amd import MyVMLSTM, MyVMLSTMGroup, _lib
torch.manual_seed(0)
H, T = 650, 35
for name, mk, B in (("group B256", lambda: MyVMLSTMGroup(H, H, w_rank=32, u_ranks=[32, 32]), 256), ("plain B256", lambda: MyVMLSTM(H, H, w_rank=32, u_ranks=32), 256), ("group B32", lambda: MyVMLSTMGroup(H, H, w_rank=32, u_ranks=[32, 32]), 32)):
    L = [mk().cuda() for _ in range(2)]
    for l in L:
        for p in l.parameters(): torch.nn.init.uniform_(p, -0.05, 0.05)
    x = 0.05 * torch.randn(T, B, H, device="cuda"); dy = torch.randn(T, B, H, device="cuda")
    st = [(torch.zeros(B, H, device="cuda"), torch.zeros(B, H, device="cuda")) for _ in L]
    ref = None; bad = 0
    for it in range(300):
        for l in L: l.zero_grad(set_to_none=True)
        h = x
        for l, s in zip(L, st): h, _ = l(h, s)
        torch.autograd.backward(h, dy)
        cur = [h.detach().clone()] + [p.grad.clone() for l in L for p in l.parameters()]
        if ref is None: ref = cur
        else: bad += int(any(not torch.equal(a, b) for a, b in zip(ref, cur)))
    torch.cuda.synchronize()
    print(name, "300 iterations, differing from the first:", bad, "status", _lib.lib().vmlmf_check_status(), "finite", all(bool(torch.isfinite(t).all()) for t in ref))
