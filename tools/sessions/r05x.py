"""config E group layer, forward + backward: cluster sizes (vmlmf_tune rb_cluster) and rows per block (rb_rows) at 32 and 256 rows"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from vmlmf_amd import MyVMLSTMGroup, _lib
torch.manual_seed(0)
H, T = 650, 35
for B in (32, 256):
    l = MyVMLSTMGroup(H, H, w_rank=32, u_ranks=[32, 32]).cuda()
    for p in l.parameters(): torch.nn.init.uniform_(p, -0.05, 0.05)
    x = 0.05 * torch.randn(T, B, H, device="cuda")
    st = (torch.zeros(B, H, device="cuda"), torch.zeros(B, H, device="cuda"))
    def step():
        l.zero_grad(set_to_none=True)
        y, _ = l(x, st)
        y.sum().backward()
    for S in (0, 2, 4, 8, 16):
        for rows in (0, 8, 4):
            _lib.tune("rb_cluster", S); _lib.tune("rb_rows", rows)
            try:
                for _ in range(3): step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10): step()
                torch.cuda.synchronize()
                q = _lib.query(_lib.make_desc(_lib.V4_LM_GROUP, B, T, H, H, 32, [32, 32], g=2, time_major=True))
                print(f"B={B} rb_cluster={S} rb_rows={rows}: {(time.perf_counter()-t0)/10*1e3:.3f} ms  (rows/wg {q.rows_per_wg}, workgroups {q.workgroups})", flush=True)
            except Exception as e:
                print(f"B={B} rb_cluster={S} rb_rows={rows}: {type(e).__name__} {str(e)[:80]}", flush=True)
    _lib.tune("rb_cluster", 0); _lib.tune("rb_rows", 0)
