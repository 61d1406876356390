"""GPU parity: the HIP path (through the C ABI) against (a) the golden vectors captured from the imported
reference and (b) the oracle on the same seeded inputs.  Tolerances: tests/hip_util.py."""
import numpy as np
import pytest
import torch

import vmlmf_oracle as O
from conftest import load_golden
from hip_util import run_hip, run_literal, compare_all, assert_out, assert_grad

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["cell_v1", "cell_v1_b1", "cell_v1_ieqh", "cell_v2", "cell_v3", "cell_v4", "cell_v5",
                                  "cell_v6"])
def test_bare_cell_vs_reference_golden(name):
    d = load_golden(name)
    variant = int(d["meta"][0])
    tm = variant in (O.V3, O.V4)
    x = d["x"][None] if tm else d["x"][:, None]
    got = run_hip(variant, d["P"], x, d["h0"], d["c0"], None, d["dh"], d["dc"], time_major=tm)
    ref = {"hT": d["h1"], "cT": d["c1"], "dx": d["dx"][None] if tm else d["dx"][:, None],
           "dh0": d["dh0"], "dc0": d["dc0"], "G": d["G"]}
    compare_all(got, ref, name)


@pytest.mark.parametrize("name", ["seq_v1", "seq_v1_wide", "seq_v2", "seq_v2_demo", "seq_v1_demo", "seq_v5",
                                  "seq_v5_wide", "seq_v6", "seq_v6_demo"])
def test_har_sequence_vs_reference_golden(name):
    d = load_golden(name)
    variant = int(d["meta"][0])
    got = run_hip(variant, d["P"], d["x"], None, None, d["dy"], d["dhT"], None, time_major=False)
    compare_all(got, {"y": d["y"], "hT": d["hT"], "dx": d["dx"], "G": d["G"]}, name)


@pytest.mark.parametrize("name", ["seq_v3", "seq_v4"])
def test_lm_sequence_vs_reference_golden(name):
    d = load_golden(name)
    variant = int(d["meta"][0])
    got = run_hip(variant, d["P"], d["x"], d["h0"], d["c0"], d["dy"], d["dhT"], d["dcT"], time_major=True)
    compare_all(got, {k: d[k] for k in ("y", "hT", "cT", "dx", "dh0", "dc0", "G")}, name)


def test_config_a_full_size_vs_reference_golden():
    """BASELINE config A/B: B=64 T=128 I=9 H=180 rank 16 (the bench workload)."""
    d = load_golden("cfgA_v1_uci")
    _, B, T, I, H, rw, ru = (int(v) for v in d["meta"])
    P = O.make_params(O.V1, I, H, rw, ru, seed=int(d["seeds"][0]))
    x, _ = O.synthetic_batch(B, T, I, seed=int(d["seeds"][1]))
    dy = np.random.Generator(np.random.PCG64(int(d["seeds"][2]))).standard_normal((B, T, H)).astype(np.float32)
    got = run_hip(O.V1, P, x, None, None, dy, None, None)
    assert_out(got["y"][:, ::16], d["y_s"], "y")
    assert_out(got["hT"], d["hT"], "hT")
    assert_grad(got["dx"], d["dx"], "dx")
    for k, v in d["G"].items():
        assert_grad(got["G"][k], v, "G." + k)


def test_config_a_group_full_size_vs_reference_golden():
    d = load_golden("cfgA_v2_uci")
    meta = [int(v) for v in d["meta"]]
    _, B, T, I, H, rw = meta[:6]
    ru = meta[6:]
    P = O.make_params(O.V2, I, H, rw, ru, seed=int(d["seeds"][0]))
    x, _ = O.synthetic_batch(B, T, I, seed=int(d["seeds"][1]))
    dy = np.random.Generator(np.random.PCG64(int(d["seeds"][2]))).standard_normal((B, T, H)).astype(np.float32)
    got = run_hip(O.V2, P, x, None, None, dy, None, None)
    assert_out(got["y"][:, ::16], d["y_s"], "y")
    assert_out(got["hT"], d["hT"], "hT")
    assert_grad(got["dx"], d["dx"], "dx")
    for k, v in d["G"].items():
        assert_grad(got["G"][k], v, "G." + k)


@pytest.mark.parametrize("name", ["cfgA_v5_uci", "cfgA_v6_uci"])
def test_config_a_comparison_cells_full_size_vs_reference_golden(name):
    """UCI-HAR shape through the two cells without vm: MyLSTMCell in low-rank mode
    (rank 16) and MyVMLMFgCellg2 (ranks [16,16])."""
    d = load_golden(name)
    meta = [int(v) for v in d["meta"]]
    variant, B, T, I, H, rw = meta[:6]
    ru = meta[6:]
    P = O.make_params(variant, I, H, rw, ru, seed=int(d["seeds"][0]))
    x, _ = O.synthetic_batch(B, T, I, seed=int(d["seeds"][1]))
    dy = np.random.Generator(np.random.PCG64(int(d["seeds"][2]))).standard_normal((B, T, H)).astype(np.float32)
    got = run_hip(variant, P, x, None, None, dy, None, None)
    assert_out(got["y"][:, ::16], d["y_s"], "y")
    assert_out(got["hT"], d["hT"], "hT")
    assert_grad(got["dx"], d["dx"], "dx")
    for k, v in d["G"].items():
        assert_grad(got["G"][k], v, "G." + k)


def test_comparison_cell_with_more_inputs_than_units_vs_reference_golden():
    """The reference's cells without vm accept input_size > hidden_size (no vm_x to pad); such a layer runs the step-wise
    path, whose x side is indexed by input rather than by unit slot."""
    d = load_golden("cell_v5_iwide")
    got = run_hip(O.V5, d["P"], d["x"][:, None], d["h0"], d["c0"], None, d["dh"], d["dc"])
    ref = {"hT": d["h1"], "cT": d["c1"], "dx": d["dx"][:, None], "dh0": d["dh0"], "dc0": d["dc0"], "G": d["G"]}
    compare_all(got, ref, "cell_v5_iwide")


CASES = [
    # variant, B, T, I, H, rw, ru, time_major, with_state
    (O.V1, 3, 5, 4, 16, 2, [3], False, False),
    (O.V1, 7, 9, 16, 64, 8, [8], False, True),       # exactly one full wave
    (O.V1, 5, 4, 9, 65, 5, [11], True, True),        # one unit into the second wave, odd ranks
    (O.V1, 2, 3, 30, 200, 16, [24], False, False),   # rank 24 (pass of 16 + half pass)
    (O.V1, 3, 4, 12, 130, 32, [32], False, True),    # rank 32
    (O.V1, 300, 3, 6, 40, 4, [4], False, False),     # more rows than CUs: workgroups queue up
    (O.V1, 513, 2, 6, 40, 4, [4], False, True),      # odd batch, two workgroups per CU
    (O.V2, 4, 5, 6, 20, 3, [2, 5], False, False),
    (O.V2, 3, 4, 10, 136, 8, [16, 8], False, True),  # two waves per group, second one ragged
    (O.V2, 260, 2, 5, 24, 3, [4, 4], False, False),  # group cell, B > 256
    (O.V3, 6, 5, 24, 24, 4, [6], True, True),
    (O.V4, 9, 4, 20, 20, 3, [4, 2], True, True),     # batch != 40 (the reference cannot run this)
    (O.V4, 40, 3, 72, 72, 8, [16, 16], True, True),
    (O.V4, 63, 31, 10, 10, 9, [11, 4], False, False),  # a flat layer small enough for the x-fold (its backward was refused since round 5's pruning: found by tools/fuzz_parity.py, round 6)
    (O.V4, 51, 35, 8, 8, 14, [5, 12], False, True),
    # 257..512 thread slots: the 8-wave instantiations of the persistent kernels
    (O.V1, 4, 5, 20, 300, 8, [16], False, True),     # 5 compute waves
    (O.V1, 3, 3, 9, 500, 16, [32], False, False),    # 8 compute waves, rank 32
    (O.V1, 260, 2, 9, 300, 8, [8], False, False),    # ... with more rows than CUs
    (O.V2, 3, 4, 12, 280, 8, [8, 16], False, True),  # two groups of three waves (384 slots), rank 8 + 16
    (O.V3, 5, 4, 330, 330, 8, [24], True, True),     # LM layer, 6 waves
    (O.V4, 6, 3, 264, 264, 6, [8, 8], True, True),   # flat layout on 2 x 3 waves
    # shapes that do not fit the register-resident kernels -> step-wise path (vmlmf_generic.hip)
    (O.V1, 5, 4, 12, 40, 6, [40], False, True),      # rank 40 > 32
    (O.V1, 3, 3, 20, 600, 8, [8], False, False),     # 640 thread slots > 512
    (O.V2, 4, 3, 10, 48, 4, [24, 20], False, True),  # group ranks pad to 24 + 24 = 48
    (O.V3, 6, 4, 70, 70, 9, [33], True, True),       # rank 33 -> padded 40
    (O.V4, 7, 3, 44, 44, 5, [20, 36], True, True),   # flat layout, padded 24 + 40 = 64
    (O.V1, 512, 16, 20, 330, 32, [40], False, False),  # 8192 rows, K = 1536: the 16 x 32 skinny tiles of dqx = dpre VxT
    # the cells without vm: per-gate factor tensors (V5), both sides chunked (f,i,n,o) (V6)
    (O.V5, 5, 6, 9, 70, 5, [7], False, True),
    (O.V5, 3, 4, 40, 200, 24, [32], False, False),   # wide input: x-projection kernel, no x-fold
    (O.V5, 4, 3, 30, 40, 6, [40], False, True),      # rank 40 -> step-wise path
    (O.V5, 6, 5, 12, 64, 4, [8], True, True),        # time-major
    (O.V6, 4, 5, 10, 136, 8, [16, 8], False, True),
    (O.V6, 300, 2, 5, 24, 3, [4, 4], False, False),
    (O.V6, 3, 3, 20, 48, 4, [24, 20], False, True),  # step-wise path
    (O.V5, 5, 4, 77, 40, 8, [6], False, True),       # more inputs than units (valid without vm): step-wise path
    (O.V6, 4, 3, 100, 24, 5, [4, 4], True, False),   # ... group cell, time-major, three input tiles
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"v{c[0]}_B{c[1]}_T{c[2]}_I{c[3]}_H{c[4]}_r{c[5]}_{'x'.join(map(str, c[6]))}")
def test_seeded_shapes_vs_oracle(case):
    variant, B, T, I, H, rw, ru, tm, with_state = case
    rng = np.random.Generator(np.random.PCG64(1000 + B + 7 * T + 13 * H))
    P = O.make_params(variant, I, H, rw, ru if variant in (O.V2, O.V4, O.V6) else ru[0], seed=H + rw)
    shp = (T, B, I) if tm else (B, T, I)
    x = rng.standard_normal(shp).astype(np.float32)
    h0 = (0.4 * rng.standard_normal((B, H))).astype(np.float32) if with_state else None
    c0 = (0.4 * rng.standard_normal((B, H))).astype(np.float32) if with_state else None
    dy = rng.standard_normal(shp[:2] + (H,)).astype(np.float32)
    dhT = rng.standard_normal((B, H)).astype(np.float32)
    dcT = rng.standard_normal((B, H)).astype(np.float32)
    got = run_hip(variant, P, x, h0, c0, dy, dhT, dcT, time_major=tm)
    ref = run_literal(variant, P, x, h0, c0, dy, dhT, dcT, time_major=tm)
    compare_all(got, ref, "case")


def _regen_lm_inputs(seed, B, T, H, xscale):
    """Same draw order as oracle/make_golden.py:case_lm_seq (the large fixtures store only the seed)."""
    r = np.random.Generator(np.random.PCG64(seed))
    x = (xscale * r.standard_normal((T, B, H))).astype(np.float32)
    h0 = (0.3 * r.standard_normal((B, H))).astype(np.float32)
    c0 = (0.3 * r.standard_normal((B, H))).astype(np.float32)
    dy = r.standard_normal((T, B, H)).astype(np.float32)
    dhT = r.standard_normal((B, H)).astype(np.float32)
    dcT = r.standard_normal((B, H)).astype(np.float32)
    return x, h0, c0, dy, dhT, dcT


@pytest.mark.parametrize("name", ["cfgE_v4_b40", "cfgE_v3_b64"])
def test_config_e_shape_vs_reference_golden(name):
    """BASELINE config E shape: PTB layer H=650, ranks 32 / [32,32], T=35 (B=40 is the only batch the
    reference's group layer can run).  Runs the step-wise path."""
    d = load_golden(name)
    meta = [int(v) for v in d["meta"]]
    variant, B, T, _, H, rw = meta[:6]
    ru = meta[6:]
    seed, (scale, xscale) = int(d["seed"][0]), (float(d["scale"][0]), float(d["scale"][1]))
    P = O.make_params(variant, H, H, rw, ru if variant == O.V4 else ru[0], seed=seed + 1, scale=scale)
    x, h0, c0, dy, dhT, dcT = _regen_lm_inputs(seed, B, T, H, xscale)
    got = run_hip(variant, P, x, h0, c0, dy, dhT, dcT, time_major=True)
    assert_out(got["y"][::4, ::4], d["y_s"], "y")
    assert_out(got["hT"], d["hT"], "hT")
    assert_out(got["cT"], d["cT"], "cT")
    assert_grad(got["dx"][::4, ::4], d["dx_s"], "dx")
    assert_grad(got["dh0"], d["dh0"], "dh0")
    assert_grad(got["dc0"], d["dc0"], "dc0")
    for k, v in d["G"].items():
        assert_grad(got["G"][k], v, "G." + k)


def test_inference_mode_matches_training_forward():
    P = O.make_params(O.V1, 9, 180, 16, 16, seed=3)
    x, _ = O.synthetic_batch(16, 12, 9, seed=5)
    a = run_hip(O.V1, P, x, None, None, np.ones((16, 12, 180), np.float32))
    with torch.no_grad():
        b = run_hip(O.V1, P, x)
    assert np.array_equal(a["y"], b["y"])          # same kernels, bitwise


def test_backward_is_deterministic():
    P = O.make_params(O.V1, 9, 180, 16, 16, seed=3)
    x, _ = O.synthetic_batch(64, 32, 9, seed=5)
    dy = np.random.Generator(np.random.PCG64(9)).standard_normal((64, 32, 180)).astype(np.float32)
    a = run_hip(O.V1, P, x, None, None, dy)
    b = run_hip(O.V1, P, x, None, None, dy)
    for k in a["G"]:
        assert np.array_equal(a["G"][k], b["G"][k]), k
    assert np.array_equal(a["dx"], b["dx"])


def test_linearity_of_backward_in_upstream_gradient():
    """Size-independent property at the full bench size: backward is linear in dy."""
    P = O.make_params(O.V1, 9, 180, 16, 16, seed=3)
    x, _ = O.synthetic_batch(64, 128, 9, seed=1234)
    rng = np.random.Generator(np.random.PCG64(77))
    d1 = rng.standard_normal((64, 128, 180)).astype(np.float32)
    d2 = rng.standard_normal((64, 128, 180)).astype(np.float32)
    g1 = run_hip(O.V1, P, x, None, None, d1)
    g2 = run_hip(O.V1, P, x, None, None, d2)
    g3 = run_hip(O.V1, P, x, None, None, d1 + 2 * d2)
    for k in g1["G"]:
        assert_grad(g1["G"][k] + 2 * g2["G"][k], g3["G"][k], "lin." + k, rel=2e-4)
    assert_grad(g1["dx"] + 2 * g2["dx"], g3["dx"], "lin.dx", rel=2e-4)


def test_dy_is_never_read_past_its_end():
    """Pad thread slots (H = 180 on 192 slots) used to index dy with their slot number: for the last batch row of
    the last timestep that is up to 12 floats past the tensor.  B*T*H*4 = 45 * 2 MiB here, so the allocation
    has no slack behind it and the old kernel faulted."""
    from vmlmf_amd import MyLSTM, MyVMLMFCell
    torch.manual_seed(0)
    rnn = MyLSTM(9, hidden_layer_sizes=[180], batch_first=True, w_rank=16, u_ranks=[16], cell=MyVMLMFCell).cuda()
    x = torch.randn(1024, 128, 9, device="cuda")
    y, _ = rnn(x)
    assert y.numel() * 4 % (2 << 20) == 0
    y[:, -1].sum().backward()
    torch.cuda.synchronize()
    assert all(torch.isfinite(p.grad).all() for p in rnn.parameters())


@pytest.mark.parametrize("env", [{"VMLMF_SKINNY": "0"}, {"VMLMF_FUSE_GATES": "0"}, {"VMLMF_DQ_SPLIT": "0"},
                                 {"VMLMF_FUSE_GATES": "3"}, {"VMLMF_XWAVE": "0"}, {"VMLMF_XEXP": "0"}],
                         ids=lambda e: "_".join(f"{k[6:]}{v}" for k, v in e.items()))
def test_measurement_switches_compute_the_same_thing(env):
    """The A/B switches of the native code (read once when the library loads) select alternative kernels for the same
    arithmetic: each is run in a fresh interpreter on a step-wise-path shape and on the headline cell, against the oracle."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import sys; sys.path[:0] = [%r, %r, %r]\n"
        "import numpy as np, vmlmf_oracle as O\n"
        "from hip_util import run_hip, run_literal, compare_all\n"
        "for variant, B, T, I, H, rw, ru, tm in [(O.V4, 7, 3, 44, 44, 5, [20, 36], True), (O.V1, 5, 4, 12, 40, 6, [40], False),\n"
        "                                          (O.V1, 6, 5, 9, 70, 4, [8], False)]:\n"
        "    rng = np.random.Generator(np.random.PCG64(5))\n"
        "    P = O.make_params(variant, I, H, rw, ru if variant == O.V4 else ru[0], seed=7)\n"
        "    shp = (T, B, I) if tm else (B, T, I)\n"
        "    x = rng.standard_normal(shp).astype(np.float32)\n"
        "    h0 = (0.4 * rng.standard_normal((B, H))).astype(np.float32)\n"
        "    c0 = (0.4 * rng.standard_normal((B, H))).astype(np.float32)\n"
        "    dy = rng.standard_normal(shp[:2] + (H,)).astype(np.float32)\n"
        "    dhT = rng.standard_normal((B, H)).astype(np.float32)\n"
        "    dcT = rng.standard_normal((B, H)).astype(np.float32)\n"
        "    compare_all(run_hip(variant, P, x, h0, c0, dy, dhT, dcT, time_major=tm),\n"
        "                run_literal(variant, P, x, h0, c0, dy, dhT, dcT, time_major=tm), 'switch')\n"
        "print('ok')\n") % (os.path.dirname(here), os.path.join(os.path.dirname(here), "oracle"), here)
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]
