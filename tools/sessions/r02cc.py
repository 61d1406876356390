import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."); sys.path[:0] = [R, os.path.join(R, "oracle"), os.path.join(R, "tests")]
import numpy as np, vmlmf_oracle as O
from hip_util import run_hip, run_literal, compare_all
variant, B, T, I, H, rw, ru = O.V2, 32, 12, 9, 180, 16, [16, 16]
rng = np.random.Generator(np.random.PCG64(5))
P = O.make_params(variant, I, H, rw, ru, seed=7)
x = rng.standard_normal((B, T, I)).astype(np.float32)
dy = rng.standard_normal((B, T, H)).astype(np.float32)
print("start", flush=True)
got = run_hip(variant, P, x, None, None, dy, None, None)
print("ran", flush=True)
compare_all(got, run_literal(variant, P, x, None, None, dy, None, None), "v2")
print("ok")
