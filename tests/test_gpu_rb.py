"""GPU parity of the row-block MFMA recurrent kernels (vmlmf_rb.hip): forced on through vmlmf_tune("rb", 1) and compared
with the fp64 literal oracle and the reference's golden vectors at the same tolerances as the VALU kernels
(tests/hip_util.py); the cluster form (a 16-row block's units split over S workgroups) runs on the H = 650 PTB shapes."""
import numpy as np
import pytest
import torch

import vmlmf_oracle as O
from conftest import load_golden
from hip_util import run_hip, run_literal, compare_all, assert_out, assert_grad, ranks_of
from vmlmf_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def force_row_block_kernels():
    _lib.tune("rb", 1)
    yield
    _lib.tune("rb", -1)
    _lib.tune("rb_cluster", 0)


def uses_rb(variant, B, T, I, H, rw, ru, tm=False):
    g = 2 if variant in (O.V2, O.V4, O.V6) else 1
    s = _lib.query(_lib.make_desc(variant, B, T, I, H, rw, ru, g=g, time_major=tm))
    return s.rows_per_wg in (4, 8, 16), s.workgroups   # the VALU kernels own one row per workgroup


RB_CASES = [
    # variant, B, T, I, H, rw, ru, time_major, with_state
    (O.V1, 3, 5, 4, 16, 2, [3], False, False),        # one tile, rank 3 -> 2 contraction steps
    (O.V1, 7, 9, 16, 64, 8, [8], False, True),        # 4 tiles, one per wave
    (O.V1, 5, 4, 9, 65, 5, [11], True, True),         # ragged last tile, time-major
    (O.V1, 17, 6, 9, 180, 16, [16], False, True),     # UCI layer, two row blocks (the second one ragged)
    (O.V1, 2, 3, 30, 200, 16, [24], False, False),    # rank 24: 6 contraction steps, 2 M-tiles
    (O.V1, 3, 4, 12, 130, 32, [32], False, True),     # rank 32
    (O.V1, 33, 3, 77, 256, 24, [24], False, False),   # OPP layer (config C), 4 tiles per wave
    (O.V1, 300, 3, 6, 40, 4, [4], False, False),      # 19 row blocks
    (O.V2, 4, 5, 6, 20, 3, [2, 5], False, False),     # group cell, ranks pad to 8 + 8
    (O.V2, 3, 4, 10, 136, 8, [16, 8], False, True),   # ranks 16 + 8 -> 24: no instantiation, falls back (still must be right)
    (O.V2, 20, 6, 9, 180, 16, [16, 16], False, True), # UCI group cell
    (O.V3, 6, 5, 24, 24, 4, [6], True, True),
    (O.V4, 9, 4, 20, 20, 3, [4, 2], True, True),      # flat layout, batch != 40
    (O.V4, 40, 3, 72, 72, 8, [16, 16], True, True),
    (O.V5, 5, 6, 9, 70, 5, [7], False, True),         # cells without vm
    (O.V6, 4, 5, 10, 100, 8, [8, 8], False, True),
    # clusters: layers beyond the register-resident kernels
    (O.V1, 18, 3, 20, 600, 8, [8], False, True),      # 640 thread slots: 38 tiles over 4 workgroups
    (O.V3, 5, 4, 650, 650, 32, [32], True, True),     # PTB plain layer
    (O.V4, 21, 3, 650, 650, 32, [32, 32], True, True),  # PTB group layer (config E), two row blocks
]
EXPECT_RB = {i for i in range(len(RB_CASES))} - {9}


@pytest.mark.parametrize("idx", range(len(RB_CASES)), ids=lambda i: "v%d_B%d_T%d_I%d_H%d_r%d_%s" % (
    RB_CASES[i][:6] + ("x".join(map(str, RB_CASES[i][6])),)))
def test_row_block_kernels_vs_oracle(idx):
    variant, B, T, I, H, rw, ru, tm, with_state = RB_CASES[idx]
    on, wgs = uses_rb(variant, B, T, I, H, rw, ru, tm)
    assert on == (idx in EXPECT_RB), f"row-block selection changed: {on} ({wgs} workgroups)"
    rng = np.random.Generator(np.random.PCG64(2000 + B + 7 * T + 13 * H))
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
    compare_all(got, ref, "rb")


def test_config_a_full_size_vs_reference_golden_on_row_blocks():
    d = load_golden("cfgA_v1_uci")
    _, B, T, I, H, rw, ru = (int(v) for v in d["meta"])
    assert uses_rb(O.V1, B, T, I, H, rw, [ru])[0]
    P = O.make_params(O.V1, I, H, rw, ru, seed=int(d["seeds"][0]))
    x, _ = O.synthetic_batch(B, T, I, seed=int(d["seeds"][1]))
    dy = np.random.Generator(np.random.PCG64(int(d["seeds"][2]))).standard_normal((B, T, H)).astype(np.float32)
    got = run_hip(O.V1, P, x, None, None, dy, None, None)
    assert_out(got["y"][:, ::16], d["y_s"], "y")
    assert_out(got["hT"], d["hT"], "hT")
    assert_grad(got["dx"], d["dx"], "dx")
    for k, v in d["G"].items():
        assert_grad(got["G"][k], v, "G." + k)


def _regen_lm_inputs(seed, B, T, H, xscale):
    r = np.random.Generator(np.random.PCG64(seed))
    x = (xscale * r.standard_normal((T, B, H))).astype(np.float32)
    h0 = (0.3 * r.standard_normal((B, H))).astype(np.float32)
    c0 = (0.3 * r.standard_normal((B, H))).astype(np.float32)
    dy = r.standard_normal((T, B, H)).astype(np.float32)
    dhT = r.standard_normal((B, H)).astype(np.float32)
    dcT = r.standard_normal((B, H)).astype(np.float32)
    return x, h0, c0, dy, dhT, dcT


@pytest.mark.parametrize("name", ["cfgE_v4_b40", "cfgE_v3_b64"])
def test_config_e_shape_vs_reference_golden_on_clusters(name):
    """BASELINE config E shape (H = 650, ranks 32 / [32,32], T = 35) against the imported reference's vectors, on the
    clustered row-block kernels instead of the step-wise path."""
    d = load_golden(name)
    meta = [int(v) for v in d["meta"]]
    variant, B, T, _, H, rw = meta[:6]
    ru = meta[6:]
    on, wgs = uses_rb(variant, B, T, H, H, rw, ru, True)
    assert on and wgs > (B + 15) // 16, "expected a cluster of workgroups per row block"
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


def test_config_e_at_batch_256_vs_oracle():
    """configs[4] at its own batch size (the reference's group layer only runs B = 40): the HIP path against the fp64
    restatement with v4_scratch_rows = 256, T shortened to keep the CPU side in seconds."""
    variant, B, T, H, rw, ru = O.V4, 256, 6, 650, 32, [32, 32]
    assert uses_rb(variant, B, T, H, H, rw, ru, True)[0]
    P = O.make_params(variant, H, H, rw, ru, seed=11, scale=0.05)
    x, h0, c0, dy, dhT, dcT = _regen_lm_inputs(5, B, T, H, 0.05)
    got = run_hip(variant, P, x, h0, c0, dy, dhT, dcT, time_major=True)
    ref = run_literal(variant, P, x, h0, c0, dy, dhT, dcT, time_major=True)
    compare_all(got, ref, "E.B256")


def test_config_e_two_layers_at_its_own_size_vs_oracle():
    """configs[4] exactly as bench.py times it: TWO chained MyVMLSTMGroup layers (vmlmf_lm.py:53-174 under the layer loop of
    vmlmf_lm.py:437-439), B = 256, T = 35, H = 650, ranks 32 / [32, 32], states carried in and gradients into both layers'
    final states - against the literal fp64 restatement with v4_scratch_rows = 256 (the reference itself only executes
    B = 40).  About a minute of host time: the oracle replays 2 x 35 cell steps at batch 256 in fp64."""
    from vmlmf_amd import vmlmf_sequence
    from hip_util import ORDER
    variant, B, T, H, rw, ru, L = O.V4, 256, 35, 650, 32, [32, 32], 2
    assert uses_rb(variant, B, T, H, H, rw, ru, True)[0]
    Ps = [O.make_params(variant, H, H, rw, ru, seed=21 + l, scale=0.05) for l in range(L)]
    r = np.random.Generator(np.random.PCG64(35))
    x = (0.05 * r.standard_normal((T, B, H))).astype(np.float32)
    h0 = (0.3 * r.standard_normal((L, B, H))).astype(np.float32)
    c0 = (0.3 * r.standard_normal((L, B, H))).astype(np.float32)
    dy = r.standard_normal((T, B, H)).astype(np.float32)
    dhT = r.standard_normal((L, B, H)).astype(np.float32)
    dcT = r.standard_normal((L, B, H)).astype(np.float32)
    names = ORDER[variant]
    # ---- HIP: the two layers chained through autograd, as Model.forward does
    dev = "cuda"
    params = [[torch.tensor(np.asarray(P[k]), device=dev).requires_grad_(True) for k in names] for P in Ps]
    xg = torch.tensor(x, device=dev).requires_grad_(True)
    h0g = torch.tensor(h0, device=dev).requires_grad_(True)
    c0g = torch.tensor(c0, device=dev).requires_grad_(True)
    cur, loss = xg, 0.0
    outs = []
    for l in range(L):
        cur, hT, cT = vmlmf_sequence(variant, cur, h0g[l], c0g[l], params[l], rw, ru, g=2, time_major=True)
        outs.append((hT, cT))
        loss = loss + (hT * torch.tensor(dhT[l], device=dev)).sum() + (cT * torch.tensor(dcT[l], device=dev)).sum()
    loss = loss + (cur * torch.tensor(dy, device=dev)).sum()
    loss.backward()
    torch.cuda.synchronize()
    # ---- oracle, fp64
    Pt = [O.to_torch(P, dtype=torch.float64, requires_grad=True) for P in Ps]
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    h0t = torch.tensor(h0, dtype=torch.float64, requires_grad=True)
    c0t = torch.tensor(c0, dtype=torch.float64, requires_grad=True)
    curr, lossr = xt, 0.0
    outr = []
    for l in range(L):
        curr, hT, cT = O.literal_sequence(variant, Pt[l], curr, h0t[l], c0t[l], time_major=True, v4_scratch_rows=B)
        outr.append((hT, cT))
        lossr = lossr + (hT * torch.tensor(dhT[l], dtype=torch.float64)).sum() + (cT * torch.tensor(dcT[l], dtype=torch.float64)).sum()
    lossr = lossr + (curr * torch.tensor(dy, dtype=torch.float64)).sum()
    lossr.backward()
    assert_out(cur.detach().cpu().numpy(), curr.detach().numpy(), "y")
    for l in range(L):
        assert_out(outs[l][0].detach().cpu().numpy(), outr[l][0].detach().numpy(), f"hT[{l}]")
        assert_out(outs[l][1].detach().cpu().numpy(), outr[l][1].detach().numpy(), f"cT[{l}]")
    assert_grad(xg.grad.cpu().numpy(), xt.grad.numpy(), "dx")
    assert_grad(h0g.grad.cpu().numpy(), h0t.grad.numpy(), "dh0")
    assert_grad(c0g.grad.cpu().numpy(), c0t.grad.numpy(), "dc0")
    for l in range(L):
        for k, p_ in zip(names, params[l]):
            assert_grad(p_.grad.cpu().numpy(), Pt[l][k].grad.numpy(), f"layer {l} {k}")


def test_row_block_and_valu_kernels_agree_at_large_batch():
    """B = 1024, T = 16 at the UCI layer: both kernel families against each other (they sum in different orders)."""
    P = O.make_params(O.V1, 9, 180, 16, 16, seed=3)
    rng = np.random.Generator(np.random.PCG64(8))
    x = rng.standard_normal((1024, 16, 9)).astype(np.float32)
    dy = rng.standard_normal((1024, 16, 180)).astype(np.float32)
    a = run_hip(O.V1, P, x, None, None, dy)
    _lib.tune("rb", 0)
    b = run_hip(O.V1, P, x, None, None, dy)
    assert_out(a["y"], b["y"], "y")
    assert_grad(a["dx"], b["dx"], "dx", rel=2e-5)
    for k in a["G"]:
        assert_grad(a["G"][k], b["G"][k], "G." + k, rel=5e-5)


def test_cluster_exchange_is_bit_stable_under_uneven_load():
    """The cluster hand-offs (write-through partials + epoch words, agent scope) must not depend on timing or placement:
    the PTB group layer runs forward + backward 12 times while a second stream streams copies through HBM at random
    intervals; every repetition must reproduce the first one bit for bit (sums are taken in member order, so any stale or
    torn read shows up as a different bit pattern).  Idle, evenly loaded chips hide such failures."""
    from vmlmf_amd import MyVMLSTMGroup
    torch.manual_seed(0)
    H, B, T = 650, 48, 12
    layer = MyVMLSTMGroup(H, H, w_rank=32, u_ranks=[32, 32]).cuda()
    for p in layer.parameters():
        torch.nn.init.uniform_(p, -0.05, 0.05)
    assert uses_rb(O.V4, B, T, H, H, 32, [32, 32], True)[1] > 16
    x = (0.05 * torch.randn(T, B, H, device="cuda")).requires_grad_(True)
    st = (torch.zeros(B, H, device="cuda"), torch.zeros(B, H, device="cuda"))
    dy = torch.randn(T, B, H, device="cuda")
    hog_stream = torch.cuda.Stream()
    a = torch.empty(64 << 20, device="cuda", dtype=torch.float32)
    b = torch.empty_like(a)
    g = torch.Generator().manual_seed(1)
    first = None
    for it in range(12):
        n_hog = int(torch.randint(0, 4, (1,), generator=g))
        with torch.cuda.stream(hog_stream):
            for _ in range(n_hog):
                b.copy_(a)
                a.mul_(1.0001)
        layer.zero_grad(set_to_none=True)
        x.grad = None
        y, (hT, cT) = layer(x, st)
        (y * dy).sum().backward()
        torch.cuda.synchronize()
        got = [y.detach().clone(), hT.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in layer.parameters()]
        assert all(torch.isfinite(t).all() for t in got)
        if first is None:
            first = got
        else:
            for k, (u, v) in enumerate(zip(first, got)):
                assert torch.equal(u, v), f"repetition {it}: tensor {k} differs from the first run (max {float((u - v).abs().max()):.3e})"


@pytest.mark.parametrize("case", [(O.V4, 40, 4, 650, 650, 32, [32, 32]), (O.V3, 24, 5, 650, 650, 32, [32]), (O.V1, 18, 3, 20, 600, 8, [8]),
                                  (O.V3, 16, 3, 400, 400, 12, [20])],
                         ids=lambda c: "v%d_B%d_T%d_I%d_H%d_r%d" % c[:6])
def test_clusters_of_sixteen_with_full_tiles_vs_oracle(case):
    """Clusters of sixteen workgroups with all 16 rows of a tile live, on layers with narrow inputs (I < H), x-ranks that are not a
    multiple of 16, both cluster layouts (two groups flat, one group), given initial states: against the oracle."""
    variant, B, T, I, H, rw, ru = case
    tm = variant in (O.V3, O.V4)
    rng = np.random.Generator(np.random.PCG64(17 * B + H))
    P = O.make_params(variant, I, H, rw, ru if variant == O.V4 else ru[0], seed=H + B, scale=0.05 if tm else 0.1)
    shp = (T, B, I) if tm else (B, T, I)
    x = (0.5 * rng.standard_normal(shp)).astype(np.float32)
    h0 = (0.3 * rng.standard_normal((B, H))).astype(np.float32)
    c0 = (0.3 * rng.standard_normal((B, H))).astype(np.float32)
    dy = rng.standard_normal(shp[:2] + (H,)).astype(np.float32)
    dhT = rng.standard_normal((B, H)).astype(np.float32)
    ref = run_literal(variant, P, x, h0, c0, dy, dhT, None, time_major=tm)
    _lib.tune("rb_cluster", 16)
    _lib.tune("rb_rows", 16)
    try:
        got = run_hip(variant, P, x, h0, c0, dy, dhT, None, time_major=tm)
        compare_all(got, ref, "rb.cluster16.rows16")
    finally:
        _lib.tune("rb_cluster", 0)
        _lib.tune("rb_rows", 0)


@pytest.mark.parametrize("case", [(O.V3, 24, 5, 650, 650, 32, [32]), (O.V1, 18, 3, 20, 600, 8, [8]), (O.V3, 16, 4, 400, 400, 12, [20]),
                                  (O.V1, 33, 2, 300, 520, 16, [24])],
                         ids=lambda c: "v%d_B%d_T%d_I%d_H%d_r%d" % c[:6])
def test_clustered_layers_weight_gradients_vs_oracle(case):
    """One-group clustered layers: every gradient against the oracle with 16 and with fewer live rows per workgroup, ragged last
    tiles, a rank that is not a multiple of 16, zero and given initial states."""
    variant, B, T, I, H, rw, ru = case
    tm = variant in (O.V3, O.V4)
    rng = np.random.Generator(np.random.PCG64(19 * B + H))
    P = O.make_params(variant, I, H, rw, ru[0], seed=H + B, scale=0.05 if tm else 0.1)
    shp = (T, B, I) if tm else (B, T, I)
    x = (0.5 * rng.standard_normal(shp)).astype(np.float32)
    h0 = (0.3 * rng.standard_normal((B, H))).astype(np.float32) if B % 2 == 0 else None
    c0 = (0.3 * rng.standard_normal((B, H))).astype(np.float32) if B % 2 == 0 else None
    dy = rng.standard_normal(shp[:2] + (H,)).astype(np.float32)
    dhT = rng.standard_normal((B, H)).astype(np.float32)
    ref = run_literal(variant, P, x, h0, c0, dy, dhT, None, time_major=tm)
    _lib.tune("rb_cluster", 16)
    try:
        for rows in (16, 0):
            _lib.tune("rb_rows", rows)
            got = run_hip(variant, P, x, h0, c0, dy, dhT, None, time_major=tm)
            compare_all(got, ref, f"rb.cluster16.rows{rows}")
    finally:
        _lib.tune("rb_cluster", 0)
        _lib.tune("rb_rows", 0)
