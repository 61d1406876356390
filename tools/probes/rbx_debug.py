"""Debug: stacked (rbx) forward against the chained per-layer forward, error per layer and time step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import vmlmf_oracle as O
from hip_util import ORDER
from vmlmf_amd import functional as F, vmlmf_sequence
variant = O.V3 if "--plain" in sys.argv else O.V4
rw, ru, g = (32, [32, 32], 2) if variant == O.V4 else (32, [32], 1)
H, L, B, T = 650, 2, 32, 5
Ps = [O.make_params(variant, H, H, rw, ru if g == 2 else ru[0], seed=41 + l, scale=0.05) for l in range(L)]
r = np.random.Generator(np.random.PCG64(3))
x = torch.tensor((0.5 * r.standard_normal((T, B, H))).astype(np.float32), device="cuda")
h0 = torch.tensor((0.3 * r.standard_normal((L, B, H))).astype(np.float32), device="cuda")
c0 = torch.tensor((0.3 * r.standard_normal((L, B, H))).astype(np.float32), device="cuda")
names = ORDER[variant]
params = [[torch.tensor(np.asarray(P[k]), device="cuda") for k in names] for P in Ps]
with torch.no_grad():
    for rep in range(3):
        out = F.vmlmf_stack(variant, x, params, rw, ru, g=g, time_major=True, h0=h0, c0=c0)
        cur, hs, cs = x, [], []
        ys = []
        for l in range(L):
            cur, hT, cT = vmlmf_sequence(variant, cur, h0[l], c0[l], params[l], rw, ru, g=g, time_major=True)
            hs.append(hT); cs.append(cT); ys.append(cur)
        torch.cuda.synchronize()
        y, hs2, cs2 = out
        print("rep", rep, "y err per t:", [float((y[t] - ys[-1][t]).abs().max()) for t in range(T)])
        print("   hT err per layer:", [float((hs2[l] - hs[l]).abs().max()) for l in range(L)], "cT:", [float((cs2[l] - cs[l]).abs().max()) for l in range(L)])
