"""GPU parity of wgrad_ring_kernel (vmlmf_wgrad_ring.hip): the weight-gradient products of large layers with their operands
streamed through an LDS ring.  Forced on through vmlmf_tune("wring", 1) for shapes of every layout it takes (plain, group,
flat group, cells without vm; clustered and one-row-per-workgroup recurrences), with row counts that are not multiples of
the 16-row stage, with and without initial states, and compared with the fp64 literal oracle at the tolerances of the other
kernels; then against wgrad_mfma_kernel on the same inputs (different chunking: same sums up to fp32 rounding)."""
import numpy as np
import pytest

import vmlmf_oracle as O
from hip_util import run_hip, run_literal, compare_all
from vmlmf_amd import _lib

pytestmark = pytest.mark.gpu

CASES = [
    # variant, B, T, I, H, rw, ru, with_state
    (O.V3, 24, 5, 650, 650, 32, [32], True),        # PTB plain layer: 11 column tiles, ranks 32 + 32
    (O.V4, 21, 3, 650, 650, 32, [32, 32], True),    # PTB group layer (flat layout): 63 rows = 3 full stages + 15 rows
    (O.V4, 40, 9, 650, 650, 32, [32, 32], False),   # the same without initial states (t = 0 rows masked), 360 rows
    (O.V1, 18, 7, 20, 600, 8, [8], True),           # narrow input (20 of 600 units have an x), rank 8
    (O.V3, 16, 4, 400, 400, 12, [20], False),       # rank 20 -> 24 columns: a B tile that is not full
    (O.V2, 10, 6, 40, 300, 16, [16, 16], True),     # group cell: a tile's vector is its group's
    (O.V5, 9, 5, 30, 260, 8, [12], True),           # cell without vm
    (O.V1, 33, 3, 77, 256, 24, [24], False),        # OPP layer: one-row-per-workgroup recurrence, ONE 256-column... four tiles
    (O.V3, 70, 40, 300, 300, 16, [16], True),       # 2800 rows: several chunks per product
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "v%d_B%d_T%d_I%d_H%d_r%d_%s" % (c[:6] + ("x".join(map(str, c[6])),)))
def test_ring_weight_gradients_vs_oracle(case):
    variant, B, T, I, H, rw, ru, with_state = case
    rng = np.random.Generator(np.random.PCG64(23 * B + H + T))
    P = O.make_params(variant, I, H, rw, ru[0] if len(ru) == 1 else ru, seed=H + B, scale=0.05 if H > 300 else 0.1)
    x = (0.5 * rng.standard_normal((T, B, I))).astype(np.float32)
    h0 = (0.3 * rng.standard_normal((B, H))).astype(np.float32) if with_state else None
    c0 = (0.3 * rng.standard_normal((B, H))).astype(np.float32) if with_state else None
    dy = rng.standard_normal((T, B, H)).astype(np.float32)
    dhT = rng.standard_normal((B, H)).astype(np.float32)
    ref = run_literal(variant, P, x, h0, c0, dy, dhT, None, time_major=True)
    try:
        _lib.tune("wring", 0)
        old = run_hip(variant, P, x, h0, c0, dy, dhT, None, time_major=True)
        _lib.tune("wring", 1)
        got = run_hip(variant, P, x, h0, c0, dy, dhT, None, time_major=True)
    finally:
        _lib.tune("wring", -1)
    compare_all(got, ref, "wring")
    # and against the per-wave kernel: the same products, other chunk boundaries
    for k in got:
        if k.startswith("d") and got[k] is not None and old.get(k) is not None:
            a, b = np.asarray(got[k], np.float64), np.asarray(old[k], np.float64)
            scale = max(1e-6, float(np.abs(b).max()))
            assert float(np.abs(a - b).max()) <= 2e-4 * scale, (k, float(np.abs(a - b).max()), scale)
