import os, sys, numpy as np
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."); sys.path[:0] = [R, os.path.join(R, "oracle"), os.path.join(R, "tests")]; sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from hip_util import *
from conftest import load_golden
d = load_golden("cfgA_v1_uci")
_, B, T, I, H, rw, ru = (int(v) for v in d["meta"])
P = O.make_params(O.V1, I, H, rw, ru, seed=int(d["seeds"][0]))
x, _ = O.synthetic_batch(B, T, I, seed=int(d["seeds"][1]))
dy = np.random.Generator(np.random.PCG64(int(d["seeds"][2]))).standard_normal((B, T, H)).astype(np.float32)
for it in range(3):
    got = run_hip(O.V1, P, x, None, None, dy, None, None)
    e = np.abs(got["dx"] - d["dx"])
    bad = np.argwhere(e > 1e-3)
    print("it", it, "dx err", e.max(), "scale", np.abs(d["dx"]).max(), "nbad", len(bad), "first", bad[:4].tolist(), "t of bad", sorted(set(bad[:,1].tolist()))[:20])
    for k, v in d["G"].items():
        print("   G", k, np.abs(got["G"][k] - v).max() / max(np.abs(v).max(), 1e-6))
