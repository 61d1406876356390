"""CPU: pin the oracle (both restatements) to the golden vectors captured from the imported reference.

Tolerances: the literal restatement replays the reference's ATen ops in fp32, so it must agree to fp32
round-off (observed <= 2e-6 abs); the unified fp64 restatement is compared at 2e-5 abs / 1e-4 rel, the
band SURVEY.md section 8c derives from the fp32 noise floor of the reference itself.
"""
import numpy as np
import pytest
import torch

import vmlmf_oracle as O
from conftest import load_golden

torch.set_num_threads(4)


def ru_of(meta):
    ru = [int(v) for v in meta[6:]]
    return ru


def close(a, b, atol, rtol, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    lim = atol + rtol * np.abs(b)
    assert np.all(err <= lim), f"{what}: max err {err.max():.3e} (|ref| max {np.abs(b).max():.3e})"


def grad_close(Ga, Gb, what, rel=1e-4):
    for k in Gb:
        a, b = np.asarray(Ga[k], np.float64).reshape(-1), np.asarray(Gb[k], np.float64).reshape(-1)
        scale = max(np.abs(b).max(), 1e-6)
        assert np.abs(a - b).max() <= rel * scale + 1e-6, f"{what}.{k}: {np.abs(a - b).max():.3e} vs scale {scale:.3e}"


CELLS = ["cell_v1", "cell_v1_b1", "cell_v1_ieqh", "cell_v2", "cell_v3", "cell_v4", "cell_v5", "cell_v5_iwide",
         "cell_v6"]


@pytest.mark.parametrize("name", CELLS)
def test_literal_cell_matches_reference(name):
    d = load_golden(name)
    variant = int(d["meta"][0])
    P = O.to_torch(d["P"], requires_grad=True)
    x = torch.tensor(d["x"], requires_grad=True)
    h = torch.tensor(d["h0"], requires_grad=True)
    c = torch.tensor(d["c0"], requires_grad=True)
    hn, cn = O.literal_step(variant, P, x, h, c)
    ((hn * torch.tensor(d["dh"])).sum() + (cn * torch.tensor(d["dc"])).sum()).backward()
    close(hn.detach(), d["h1"], 2e-6, 1e-5, "h1")
    close(cn.detach(), d["c1"], 2e-6, 1e-5, "c1")
    close(x.grad, d["dx"], 5e-6, 1e-4, "dx")
    close(h.grad, d["dh0"], 5e-6, 1e-4, "dh0")
    close(c.grad, d["dc0"], 5e-6, 1e-4, "dc0")
    grad_close({k: v.grad.numpy() for k, v in P.items()}, d["G"], name)


@pytest.mark.parametrize("name", CELLS)
def test_unified_cell_matches_reference(name):
    d = load_golden(name)
    variant = int(d["meta"][0])
    y, hT, cT, dx, dh0, dc0, G = O.unified_run(variant, d["P"], d["x"][None], d["h0"], d["c0"],
                                               d["dh"][None], np.zeros_like(d["dh"]), d["dc"])
    close(hT, d["h1"], 2e-5, 1e-4, "h1")
    close(cT, d["c1"], 2e-5, 1e-4, "c1")
    close(dx[0], d["dx"], 2e-5, 1e-4, "dx")
    close(dh0, d["dh0"], 2e-5, 1e-4, "dh0")
    close(dc0, d["dc0"], 2e-5, 1e-4, "dc0")
    grad_close(G, d["G"], name)


HAR_SEQS = ["seq_v1", "seq_v1_wide", "seq_v2", "seq_v2_demo", "seq_v1_demo", "seq_v5", "seq_v5_wide", "seq_v6",
            "seq_v6_demo"]


@pytest.mark.parametrize("name", HAR_SEQS)
def test_har_sequence_both_restatements(name):
    d = load_golden(name)
    variant = int(d["meta"][0])
    # literal, fp32, autograd
    P = O.to_torch(d["P"], requires_grad=True)
    x = torch.tensor(d["x"], requires_grad=True)
    y, hT, _ = O.literal_sequence(variant, P, x, time_major=False)
    ((y * torch.tensor(d["dy"])).sum() + (hT * torch.tensor(d["dhT"])).sum()).backward()
    close(y.detach(), d["y"], 2e-6, 1e-5, "y literal")
    close(x.grad, d["dx"], 1e-5, 1e-4, "dx literal")
    grad_close({k: v.grad.numpy() for k, v in P.items()}, d["G"], name + " literal")
    # unified, fp64, analytic
    B, T, H = d["y"].shape
    z = np.zeros((B, H))
    yu, hTu, cTu, dxu, _, _, G = O.unified_run(variant, d["P"], d["x"].transpose(1, 0, 2), z, z,
                                               d["dy"].transpose(1, 0, 2), d["dhT"], z)
    close(yu.transpose(1, 0, 2), d["y"], 2e-5, 1e-4, "y unified")
    close(hTu, d["hT"], 2e-5, 1e-4, "hT unified")
    close(dxu.transpose(1, 0, 2), d["dx"], 2e-5, 1e-4, "dx unified")
    grad_close(G, d["G"], name + " unified")


@pytest.mark.parametrize("name", ["seq_v3", "seq_v4"])
def test_lm_sequence_both_restatements(name):
    d = load_golden(name)
    variant = int(d["meta"][0])
    P = O.to_torch(d["P"], requires_grad=True)
    x = torch.tensor(d["x"], requires_grad=True)
    h0 = torch.tensor(d["h0"], requires_grad=True)
    c0 = torch.tensor(d["c0"], requires_grad=True)
    y, hT, cT = O.literal_sequence(variant, P, x, h0, c0)
    ((y * torch.tensor(d["dy"])).sum() + (hT * torch.tensor(d["dhT"])).sum()
     + (cT * torch.tensor(d["dcT"])).sum()).backward()
    close(y.detach(), d["y"], 2e-6, 1e-5, "y literal")
    close(cT.detach(), d["cT"], 2e-6, 1e-5, "cT literal")
    close(h0.grad, d["dh0"], 1e-5, 1e-4, "dh0 literal")
    grad_close({k: v.grad.numpy() for k, v in P.items()}, d["G"], name + " literal")
    yu, hTu, cTu, dxu, dh0u, dc0u, G = O.unified_run(variant, d["P"], d["x"], d["h0"], d["c0"],
                                                     d["dy"], d["dhT"], d["dcT"])
    close(yu, d["y"], 2e-5, 1e-4, "y unified")
    close(cTu, d["cT"], 2e-5, 1e-4, "cT unified")
    close(dxu, d["dx"], 2e-5, 1e-4, "dx unified")
    close(dh0u, d["dh0"], 2e-5, 1e-4, "dh0 unified")
    close(dc0u, d["dc0"], 2e-5, 1e-4, "dc0 unified")
    grad_close(G, d["G"], name + " unified")


def test_v4_other_batch_raises_like_reference():
    """vmlmf_lm.py:112-113 hard-codes 40 scratch rows: the literal restatement keeps that behaviour."""
    P = O.to_torch(O.make_params(O.V4, 12, 12, 3, [2, 3]))
    with pytest.raises(RuntimeError):
        O.literal_step(O.V4, P, torch.randn(8, 12), torch.zeros(8, 12), torch.zeros(8, 12))
    # the kernels' specification has no such limit; the oracle is run with matching scratch rows for B != 40
    O.literal_step(O.V4, P, torch.randn(8, 12), torch.zeros(8, 12), torch.zeros(8, 12), v4_scratch_rows=8)


def test_v1_input_wider_than_hidden_raises_like_reference():
    """vmlmf.py:94: vm_x is None when I > H and the slice write fails."""
    P = O.to_torch(O.make_params(O.V1, 10, 8, 3, 3))
    with pytest.raises(Exception):
        O.literal_step(O.V1, P, torch.randn(2, 10), torch.zeros(2, 8), torch.zeros(2, 8))


def test_lm_state_carry_two_minibatches():
    d = load_golden("lm_v3_carry")
    P = O.to_torch(d["P"], requires_grad=True)
    B, H = int(d["meta"][1]), int(d["meta"][4])
    h, c = torch.zeros(B, H), torch.zeros(B, H)
    for i in range(2):
        for p in P.values():
            p.grad = None
        y, h, c = O.literal_sequence(O.V3, P, torch.tensor(d[f"x{i}"]), h.detach(), c.detach())
        loss = torch.mean(y * torch.tensor(d[f"w{i}"])) * B
        loss.backward()
        assert abs(loss.item() - float(d[f"loss{i}"][0])) < 1e-6
        close(h.detach(), d[f"hT{i}"], 2e-6, 1e-5, "hT")
        close(c.detach(), d[f"cT{i}"], 2e-6, 1e-5, "cT")
        grad_close({k: v.grad.numpy() for k, v in P.items()}, d[f"G{i}"], f"carry{i}")


def test_config_a_unified_fp64_vs_reference():
    """Full BASELINE shape (B=64 T=128 I=9 H=180 r=16): unified fp64 vs the reference's fp32 outputs."""
    d = load_golden("cfgA_v1_uci")
    _, B, T, I, H, rw, ru = (int(v) for v in d["meta"])
    P = O.make_params(O.V1, I, H, rw, ru, seed=int(d["seeds"][0]))
    x, _ = O.synthetic_batch(B, T, I, seed=int(d["seeds"][1]))
    assert np.array_equal(x[:2, :4], d["x_head"])
    dy = np.random.Generator(np.random.PCG64(int(d["seeds"][2]))).standard_normal((B, T, H)).astype(np.float32)
    z = np.zeros((B, H))
    y, hT, cT, dx, _, _, G = O.unified_run(O.V1, P, x.transpose(1, 0, 2), z, z, dy.transpose(1, 0, 2), z, z)
    close(y.transpose(1, 0, 2)[:, ::16], d["y_s"], 2e-5, 1e-4, "y")
    close(hT, d["hT"], 2e-5, 1e-4, "hT")
    close(dx.transpose(1, 0, 2), d["dx"], 5e-5, 1e-4, "dx")
    grad_close(G, d["G"], "cfgA")


def test_two_layers_of_different_sizes_literal_fp64_vs_reference():
    """MyLSTM(hidden_layer_sizes=[128, 256]) (vmlmf.py:283-314): the literal restatement chained layer by layer, fp64, against the
    imported reference's fp32 vectors."""
    d = load_golden("seq_v1_h128_h256")
    _, B, T, I, H0, H1, rw, ru = (int(v) for v in d["meta"])
    P0 = O.to_torch(O.make_params(O.V1, I, H0, rw, ru, seed=int(d["seeds"][0])), dtype=torch.float64, requires_grad=True)
    P1 = O.to_torch(O.make_params(O.V1, H0, H1, rw, ru, seed=int(d["seeds"][1])), dtype=torch.float64, requires_grad=True)
    x_np, _ = O.synthetic_batch(B, T, I, seed=int(d["seeds"][2]), classes=18)
    dy = np.random.Generator(np.random.PCG64(int(d["seeds"][3]))).standard_normal((B, T, H1)).astype(np.float32)
    x = torch.tensor(x_np, dtype=torch.float64, requires_grad=True)
    y0, h0, _ = O.literal_sequence(O.V1, P0, x, None, None, time_major=False)
    y1, h1, _ = O.literal_sequence(O.V1, P1, y0, None, None, time_major=False)
    (y1 * torch.tensor(dy, dtype=torch.float64)).sum().backward()
    close(y1.detach()[:, ::4], d["y_s"], 2e-5, 1e-4, "y")
    close(torch.cat([h0, h1], -1).detach(), d["hT"], 2e-5, 1e-4, "hT")
    close(x.grad[::4], d["dx_s"], 5e-5, 1e-4, "dx")
    grad_close({k: v.grad.numpy() for k, v in P0.items()}, d["G0"], "layer 0")
    grad_close({k: v.grad.numpy() for k, v in P1.items()}, d["G1"], "layer 1")


@pytest.mark.parametrize("name", ["cfgA_v5_uci", "cfgA_v6_uci"])
def test_config_a_comparison_cells_unified_fp64_vs_reference(name):
    """The two cells without vm (plain low-rank LSTM, group ablation) at the UCI-HAR shape."""
    d = load_golden(name)
    variant, B, T, I, H, rw = (int(v) for v in d["meta"][:6])
    ru = ru_of(d["meta"])
    P = O.make_params(variant, I, H, rw, ru, seed=int(d["seeds"][0]))
    x, _ = O.synthetic_batch(B, T, I, seed=int(d["seeds"][1]))
    dy = np.random.Generator(np.random.PCG64(int(d["seeds"][2]))).standard_normal((B, T, H)).astype(np.float32)
    z = np.zeros((B, H))
    y, hT, cT, dx, _, _, G = O.unified_run(variant, P, x.transpose(1, 0, 2), z, z, dy.transpose(1, 0, 2), z, z)
    close(y.transpose(1, 0, 2)[:, ::16], d["y_s"], 2e-5, 1e-4, "y")
    close(hT, d["hT"], 2e-5, 1e-4, "hT")
    close(dx.transpose(1, 0, 2), d["dx"], 5e-5, 1e-4, "dx")
    grad_close(G, d["G"], name)


def test_net_adam_three_steps_literal():
    """train.py:58-65 counterpart: Net forward + CE + Adam, 3 steps, loss trajectory and logits."""
    d = load_golden("cfgA_net_adam3")
    _, B, T, I, H, rw, ru = (int(v) for v in d["meta"])
    P = O.to_torch(O.make_params(O.V1, I, H, rw, ru, seed=int(d["seeds"][0])), requires_grad=True)
    x, tgt = O.synthetic_batch(B, T, I, seed=int(d["seeds"][1]))
    lw = torch.tensor(d["lin_w"], requires_grad=True)
    lb = torch.tensor(d["lin_b"], requires_grad=True)
    opt = torch.optim.Adam(list(P.values()) + [lw, lb], lr=0.002)
    for step in range(3):
        opt.zero_grad()
        loss, logits = O.literal_train_step_har(P, lw, lb, torch.tensor(x), torch.tensor(tgt))
        loss.backward()
        opt.step()
        assert abs(loss.item() - float(d["losses"][step])) < 2e-5
        close(logits.detach(), d["logits"][step], 2e-5, 1e-4, f"logits[{step}]")
    assert int(d["unused_cell_has_grad"][0]) == 0   # Net.cell duplicate never receives a gradient
    for k, v in d["final"].items():
        close(P[k.split(".")[-1]].detach(), v, 2e-5, 1e-3, "param " + k)


def test_nll_loss_restatements_vs_reference():
    """lm_test.py:140-153 at the PTB vocabulary width: literal (torch) and stable (numpy fp64) restatements."""
    d = load_golden("nll_v10000")
    T, B, V, seed = (int(v) for v in d["meta"])
    r = np.random.Generator(np.random.PCG64(seed))
    z = (2.0 * r.standard_normal((T * B, V))).astype(np.float32)
    y = r.integers(0, V, size=(T, B))
    assert np.array_equal(y, d["y"])
    zt = torch.tensor(z, requires_grad=True)
    loss = O.nll_loss_literal(zt, torch.tensor(y))
    (float(d["upstream"][0]) * loss).backward()
    assert abs(loss.item() - float(d["loss"][0])) < 1e-5
    close(zt.grad.numpy()[:, ::97], d["g_s"], 1e-9, 1e-5, "literal grad sample")
    l2, g2 = O.nll_loss_stable(z, y)
    assert abs(l2 - float(d["loss"][0])) < 1e-4 * abs(l2)
    close(float(d["upstream"][0]) * g2[:, ::97], d["g_s"], 1e-8, 1e-4, "stable grad sample")
    close(float(d["upstream"][0]) * g2[np.arange(T * B), y.reshape(-1)], d["g_target"], 1e-8, 1e-4, "stable grad at targets")


def test_lm_network_literal_two_minibatches():
    """Model (lstm_type "vmlmf") + nll_loss + clip/SGD over two minibatches with carried state (lm_test.py:196-209)."""
    d = load_golden("lm_model_v3")
    V, H, L, B, T, rw, ru = (int(v) for v in d["meta"])
    sd = {k: torch.tensor(v, requires_grad=True) for k, v in d["init"].items()}
    states = [(torch.zeros(B, H), torch.zeros(B, H)) for _ in range(L)]
    for i in range(2):
        for p in sd.values():
            p.grad = None
        states = [(h.detach(), c.detach()) for h, c in states]
        scores, states = O.literal_lm_forward(sd, torch.tensor(d[f"x{i}"]), states, L)
        loss = O.nll_loss_literal(scores, torch.tensor(d[f"y{i}"]))
        loss.backward()
        close(scores.detach(), d[f"scores{i}"], 2e-6, 1e-5, f"scores{i}")
        assert abs(loss.item() - float(d[f"loss{i}"][0])) < 1e-5
        grad_close({k: v.grad.numpy() for k, v in sd.items()}, d[f"G{i}"], f"lm G{i}")
        with torch.no_grad():
            norm = torch.nn.utils.clip_grad_norm_(list(sd.values()), 0.25)
            for p in sd.values():
                p -= 1.0 * p.grad
        assert abs(float(norm) - float(d[f"norm{i}"][0])) < 1e-4 * float(norm)
    for k, v in d["final"].items():
        close(sd[k].detach(), v, 2e-6, 1e-4, "final " + k)
