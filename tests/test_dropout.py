"""Dropout of the LM network (V/src/models/vmlmf_lm.py:402,434-439) without mask tensors (C ABI 11, csrc/vmlmf_dropout.h).

nn.Dropout's ALGORITHM is "zero with probability p, scale the rest by 1/(1-p)"; which elements is the generator's draw.  The
parity chain: (1) the oracle's numpy Philox4x32-10 is pinned on Random123's known-answer vectors (CPU test); (2) the factors the
library applies equal that restatement bit for bit - as a tensor from vmlmf_dropout_factors, identity columns and the row-block
kernels' mapped columns; (3) every place that applies them - the stand-alone launch, the embedding gather and its scatter-add
backward, the row-block layer kernels (forward copy, backward mask on dy), the whole Model - equals the oracle's forward /
backward with THOSE factors multiplied in (literal restatement, fp64), at the tolerances of tests/hip_util.py; (4) the draws
have the statistics dropout needs (rate, independence across sites / steps / replays of a captured graph)."""
import numpy as np
import pytest
import torch

import vmlmf_oracle as O

SEED = 0x1234_5678_9ABC_DEF


# ---------------------------------------------------------------- CPU: the restatement itself
def test_philox_restatement_vs_random123_known_answers():
    for ctr, key, want in O.PHILOX_KAT:
        got = O.philox4x32_10(np.array(ctr, dtype=np.uint32), np.array(key, dtype=np.uint32))
        assert tuple(int(v) for v in got) == want
    # vectorised over leading axes = element by element
    r = np.random.Generator(np.random.PCG64(1))
    ctr = r.integers(0, 2 ** 32, size=(5, 3, 4), dtype=np.uint64).astype(np.uint32)
    key = r.integers(0, 2 ** 32, size=(2,), dtype=np.uint64).astype(np.uint32)
    all_ = O.philox4x32_10(ctr, key)
    for i in range(5):
        for j in range(3):
            assert np.array_equal(all_[i, j], O.philox4x32_10(ctr[i, j], key))


def test_oracle_factors_are_zero_or_the_scale_at_the_asked_rate():
    for p in (0.1, 0.5, 0.9):
        f = O.dropout_factors(SEED, 3, 1, 512, 256, p)
        vals = np.unique(f)
        assert len(vals) == 2 and vals[0] == 0.0 and vals[1] == np.float32(1) / (np.float32(1) - np.float32(p))
        n = f.size
        assert abs((f == 0).mean() - p) < 5 * np.sqrt(p * (1 - p) / n)
    assert np.array_equal(O.dropout_factors(SEED, 3, 1, 8, 32, 0.0), np.ones((8, 32), np.float32))
    # sites, offsets and seeds are different streams
    a = O.dropout_factors(SEED, 3, 1, 64, 64, 0.5)
    for other in (O.dropout_factors(SEED, 3, 2, 64, 64, 0.5), O.dropout_factors(SEED, 4, 1, 64, 64, 0.5),
                  O.dropout_factors(SEED + 1, 3, 1, 64, 64, 0.5), O.dropout_factors(SEED, 3 + 2 ** 32, 1, 64, 64, 0.5)):
        assert 0.4 < (a != other).mean() < 0.6


# ---------------------------------------------------------------- GPU
gpu = pytest.mark.gpu


def _state(seed=SEED, offset=0):
    return torch.tensor([seed, offset], dtype=torch.int64, device="cuda")


def _rb_desc(variant, B, T, H, rw, ru, g):
    from vmlmf_amd import _lib
    return _lib.make_desc(variant, B, T, H, H, rw, ru, g=g, time_major=True, training=True)


@gpu
@pytest.mark.parametrize("case", [(96, 64, 0.5, 0, 0), (1000, 650, 0.5, 7, 2), (33, 130, 0.25, 2 ** 32 + 5, 1), (5, 7, 0.9, 1, 3),
                                  (2, 1, 0.5, 0, 0), (300, 1024, 0.1, 123456789012, 31)], ids=str)
def test_factors_from_the_library_equal_the_restatement(case):
    from vmlmf_amd.functional import dropout_factors
    R, H, p, off, site = case
    got = dropout_factors(R, H, p, _state(SEED, off), site).cpu().numpy()
    want = O.dropout_factors(SEED, off, site, R, H, p)
    assert np.array_equal(got, want)


@gpu
def test_factors_of_a_row_block_layer_use_its_thread_slots_as_columns():
    """Inside rb_fwd_kernel / rb_bwd_kernel a lane's four units are one generator call: the column of the counter is the unit's
    thread slot, group g's units starting at slot 64 W g."""
    from vmlmf_amd import _lib
    from vmlmf_amd.functional import dropout_factors
    for variant, g, H, ru in ((O.V4, 2, 650, [32, 32]), (O.V3, 1, 650, [32])):
        B, T = 21, 3
        desc = _rb_desc(variant, B, T, H, 32, ru, g)
        assert _lib.lib().vmlmf_dropout_fused(desc) == 1
        Hg = H // g
        W = (Hg + 63) // 64
        got = dropout_factors(T * B, H, 0.5, _state(SEED, 9), 2, layer_desc=desc).cpu().numpy()
        assert np.array_equal(got, O.dropout_factors(SEED, 9, 2, T * B, H, 0.5, Hg=Hg, gstride=64 * W))
    # a layer on the one-row-per-workgroup kernels: not fused, identity columns
    desc = _lib.make_desc(O.V3, 8, 4, 32, 32, 8, [8], g=1, time_major=True, training=True)
    assert _lib.lib().vmlmf_dropout_fused(desc) == 0


@gpu
def test_snapshot_and_advance():
    from vmlmf_amd.functional import dropout_advance
    st = _state(SEED, 2 ** 32 - 1)
    s1 = dropout_advance(st)
    s2 = dropout_advance(st)
    assert s1.tolist() == [SEED, 2 ** 32 - 1] and s2.tolist() == [SEED, 2 ** 32] and st.tolist() == [SEED, 2 ** 32 + 1]


@gpu
@pytest.mark.parametrize("shape", [(35, 16, 650), (7, 3, 33), (4, 1, 2)], ids=str)
def test_stand_alone_dropout_forward_and_backward(shape):
    from vmlmf_amd.functional import dropout
    r = np.random.Generator(np.random.PCG64(3))
    x = r.standard_normal(shape).astype(np.float32)
    dy = r.standard_normal(shape).astype(np.float32)
    snap = _state(SEED, 4)
    xt = torch.tensor(x, device="cuda", requires_grad=True)
    y = dropout(xt, 0.5, snap, 1)
    y.backward(torch.tensor(dy, device="cuda"))
    F = O.dropout_factors(SEED, 4, 1, shape[0] * shape[1], shape[2], 0.5).reshape(shape)
    assert np.array_equal(y.detach().cpu().numpy(), x * F)
    assert np.array_equal(xt.grad.cpu().numpy(), dy * F)
    assert dropout(xt, 0.0, snap, 1) is xt


@gpu
@pytest.mark.parametrize("case", [(35, 8, 1000, 64), (11, 3, 50, 652), (5, 2, 9, 30), (35, 16, 300, 650), (3, 2, 7, 33), (4, 1, 5, 1100)], ids=str)
def test_embedding_with_dropout_vs_restatement(case):
    """vmlmf_lm.py:434-435 in one launch per direction: out = w[tokens] * F; dw = scatter-add of (dy * F) in position order.
    (Widths that are not a multiple of four - the PTB network's 650 - take scalar accesses with the same one call per four columns;
    H = 1100: beyond the embedding kernels' 1024 columns, the gather and the dropout are two launches - same values.)"""
    from vmlmf_amd.functional import embedding_dropout
    T, B, V, H = case
    r = np.random.Generator(np.random.PCG64(5))
    w = r.standard_normal((V, H)).astype(np.float32)
    tok = r.integers(0, V, size=(T, B))
    tok[0, 0] = tok[-1, -1] = 3    # a row hit more than once
    dy = r.standard_normal((T, B, H)).astype(np.float32)
    snap = _state(SEED, 1)
    wt = torch.tensor(w, device="cuda", requires_grad=True)
    out = embedding_dropout(wt, torch.tensor(tok, device="cuda"), 0.5, snap, 0)
    out.backward(torch.tensor(dy, device="cuda"))
    F = O.dropout_factors(SEED, 1, 0, T * B, H, 0.5).reshape(T, B, H)
    assert np.array_equal(out.detach().cpu().numpy(), w[tok] * F)
    want = np.zeros((V, H), np.float64)
    np.add.at(want, tok.reshape(-1), (dy * F).reshape(-1, H).astype(np.float64))
    got = wt.grad.cpu().numpy()
    assert np.abs(got - want).max() <= 1e-5 * max(np.abs(want).max(), 1.0)
    assert np.array_equal(got[np.setdiff1d(np.arange(V), tok.reshape(-1))], np.zeros((V - len(np.unique(tok)), H), np.float32))


@gpu
@pytest.mark.parametrize("case", [(O.V4, 2, 21, 3, 650, [32, 32], 0.5), (O.V3, 1, 5, 4, 650, [32], 0.5), (O.V4, 2, 40, 6, 650, [32, 32], 0.25)],
                         ids=lambda c: "v%d_B%d_T%d_p%g" % (c[0], c[2], c[3], c[6]))
def test_layer_with_dropout_inside_its_launches_vs_oracle(case):
    """The PTB layers (config E's kernels): y_dropped = y * F out of the forward launch, dy * F inside the backward launch - against
    the literal fp64 restatement with the same factors multiplied in; the final states are the undropped ones."""
    from hip_util import ORDER, assert_out, assert_grad
    from vmlmf_amd import vmlmf_sequence, _lib
    from vmlmf_amd.functional import dropout_factors
    variant, g, B, T, H, ru, p = case
    rw = 32
    desc = _rb_desc(variant, B, T, H, rw, ru, g)
    assert _lib.lib().vmlmf_dropout_fused(desc) == 1
    P = O.make_params(variant, H, H, rw, ru if g == 2 else ru[0], seed=31, scale=0.05)
    r = np.random.Generator(np.random.PCG64(8))
    x = (0.5 * r.standard_normal((T, B, H))).astype(np.float32)
    h0 = (0.3 * r.standard_normal((B, H))).astype(np.float32)
    c0 = (0.3 * r.standard_normal((B, H))).astype(np.float32)
    dy, dhT, dcT = (r.standard_normal(s).astype(np.float32) for s in ((T, B, H), (B, H), (B, H)))
    snap = _state(SEED, 11)
    names = ORDER[variant]
    params = [torch.tensor(np.asarray(P[k]), device="cuda").requires_grad_(True) for k in names]
    xg, h0g, c0g = (torch.tensor(a, device="cuda").requires_grad_(True) for a in (x, h0, c0))
    yd, hT, cT = vmlmf_sequence(variant, xg, h0g, c0g, params, rw, ru, g=g, time_major=True, drop=(p, snap, 2))
    ((yd * torch.tensor(dy, device="cuda")).sum() + (hT * torch.tensor(dhT, device="cuda")).sum()
     + (cT * torch.tensor(dcT, device="cuda")).sum()).backward()
    F = dropout_factors(T * B, H, p, snap, 2, layer_desc=desc).cpu().numpy().reshape(T, B, H)
    assert abs((F == 0).mean() - p) < 0.02
    # oracle
    Pt = O.to_torch(P, dtype=torch.float64, requires_grad=True)
    xt, h0t, c0t = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (x, h0, c0))
    y, hTr, cTr = O.literal_sequence(variant, Pt, xt, h0t, c0t, time_major=True, v4_scratch_rows=B)
    ydr = y * torch.tensor(F, dtype=torch.float64)
    ((ydr * torch.tensor(dy, dtype=torch.float64)).sum() + (hTr * torch.tensor(dhT, dtype=torch.float64)).sum()
     + (cTr * torch.tensor(dcT, dtype=torch.float64)).sum()).backward()
    got = yd.detach().cpu().numpy()
    assert_out(got, ydr.detach().numpy(), "y_dropped")
    assert_out(hT.detach().cpu().numpy(), hTr.detach().numpy(), "hT")
    assert_out(cT.detach().cpu().numpy(), cTr.detach().numpy(), "cT")
    assert_grad(xg.grad.cpu().numpy(), xt.grad.numpy(), "dx")
    assert_grad(h0g.grad.cpu().numpy(), h0t.grad.numpy(), "dh0")
    assert_grad(c0g.grad.cpu().numpy(), c0t.grad.numpy(), "dc0")
    for k, p_ in zip(names, params):
        assert_grad(p_.grad.cpu().numpy(), Pt[k].grad.numpy(), k)


@gpu
def test_layer_outside_the_fused_envelope_takes_the_stand_alone_launch():
    from hip_util import ORDER, assert_out, assert_grad
    from vmlmf_amd import vmlmf_sequence
    variant, B, T, H, rw, ru = O.V3, 6, 5, 24, 4, [6]
    P = O.make_params(variant, H, H, rw, ru[0], seed=2)
    r = np.random.Generator(np.random.PCG64(9))
    x = r.standard_normal((T, B, H)).astype(np.float32)
    dy = r.standard_normal((T, B, H)).astype(np.float32)
    snap = _state(SEED, 0)
    names = ORDER[variant]
    params = [torch.tensor(np.asarray(P[k]), device="cuda").requires_grad_(True) for k in names]
    xg = torch.tensor(x, device="cuda").requires_grad_(True)
    yd, hT, cT = vmlmf_sequence(variant, xg, None, None, params, rw, ru, g=1, time_major=True, drop=(0.5, snap, 1))
    (yd * torch.tensor(dy, device="cuda")).sum().backward()
    F = O.dropout_factors(SEED, 0, 1, T * B, H, 0.5).reshape(T, B, H)
    Pt = O.to_torch(P, dtype=torch.float64, requires_grad=True)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    y, _, _ = O.literal_sequence(variant, Pt, xt, None, None, time_major=True)
    ((y * torch.tensor(F, dtype=torch.float64)) * torch.tensor(dy, dtype=torch.float64)).sum().backward()
    assert_out(yd.detach().cpu().numpy(), (y * torch.tensor(F, dtype=torch.float64)).detach().numpy(), "y_dropped")
    assert_grad(xg.grad.cpu().numpy(), xt.grad.numpy(), "dx")
    for k, p_ in zip(names, params):
        assert_grad(p_.grad.cpu().numpy(), Pt[k].grad.numpy(), k)


def _site_factors(model, snap, T, B):
    """The factors of every dropout site of `model` for `snap`, as the kernels of each site apply them."""
    from vmlmf_amd import _lib
    from vmlmf_amd.functional import dropout_factors
    H, p = model.hidden_size, model.dropout.p
    out = [dropout_factors(T * B, H, p, snap, 0).reshape(T, B, H)]
    for i, rnn in enumerate(model.rnns):
        g = getattr(rnn, "g", 1)
        ur = list(rnn.u_ranks) if isinstance(rnn.u_ranks, (list, tuple)) else [rnn.u_ranks]
        desc = _lib.make_desc(rnn.variant, B, T, H, H, rnn.w_rank, ur, g=g, time_major=True, training=True)
        fused = _lib.lib().vmlmf_dropout_fused(desc) == 1
        out.append(dropout_factors(T * B, H, p, snap, i + 1, layer_desc=desc if fused else None).reshape(T, B, H))
    return out


@gpu
@pytest.mark.parametrize("H,B,T", [(32, 6, 5), (650, 20, 4)], ids=["valu_layers", "row_block_layers"])
def test_model_training_step_with_dropout_vs_oracle(H, B, T):
    """Model.forward + nll_loss + backward in training mode at p = 0.5 (lm_test.py:196-203) against literal_lm_forward with the
    factors of the three sites multiplied in: scores, loss, every gradient; then eval mode = no dropout."""
    from hip_util import assert_out, assert_grad
    from vmlmf_amd.lm import Model
    V, L, p = 120, 2, 0.5
    torch.manual_seed(5)
    m = Model(V, H, L, p, 0.08, w_rank=8 if H < 100 else 32, u_ranks=[8 if H < 100 else 32], lstm_type="vmlmf").cuda()
    st = m.dropout_state(seed=SEED)
    st[1] = 41
    r = np.random.Generator(np.random.PCG64(4))
    tok = torch.tensor(r.integers(0, V, size=(T, B)), device="cuda")
    tgt = torch.tensor(r.integers(0, V, size=(T * B,)), device="cuda")
    states = [(torch.tensor((0.2 * r.standard_normal((B, H))).astype(np.float32), device="cuda"),
               torch.tensor((0.2 * r.standard_normal((B, H))).astype(np.float32), device="cuda")) for _ in range(L)]
    snap = st.clone()
    m.train()
    scores, new_states = m(tok, [tuple(s) for s in states])
    assert st.tolist() == [SEED, 42]
    loss = torch.nn.functional.cross_entropy(scores, tgt)
    loss.backward()
    F = [f.cpu().double() for f in _site_factors(m, snap, T, B)]
    sd = {k: v.detach().cpu().double().requires_grad_(True) for k, v in m.state_dict().items()}
    ref_scores, ref_states = O.literal_lm_forward(sd, tok.cpu(), [(h.cpu().double(), c.cpu().double()) for h, c in states], L, factors=F)
    ref_loss = torch.nn.functional.cross_entropy(ref_scores, tgt.cpu())
    ref_loss.backward()
    assert_out(scores.detach().cpu().numpy(), ref_scores.detach().numpy(), "scores")
    assert abs(loss.item() - ref_loss.item()) <= 1e-5 * max(1.0, abs(ref_loss.item()))
    for i in range(L):
        assert_out(new_states[i][0].detach().cpu().numpy(), ref_states[i][0].detach().numpy(), f"hT[{i}]")
    for k, v in m.named_parameters():
        assert_grad(v.grad.cpu().numpy(), sd[k].grad.numpy(), k)
    # the three sites together zero about half of each activation - and eval mode none
    assert all(abs((f == 0).double().mean().item() - p) < 0.05 for f in F)
    m.eval()
    with torch.no_grad():
        ev, _ = m(tok, [tuple(s) for s in states])
        ref_ev, _ = O.literal_lm_forward({k: v.detach() for k, v in sd.items()}, tok.cpu(), [(h.cpu().double(), c.cpu().double()) for h, c in states], L)
    assert_out(ev.cpu().numpy(), ref_ev.numpy(), "eval scores")
    assert st.tolist() == [SEED, 42]


@gpu
def test_the_stock_launches_remain_selectable_and_the_group_layers_are_covered():
    from vmlmf_amd.lm import Model, MyVMLSTMGroup
    torch.manual_seed(1)
    m = Model(60, 40, 2, 0.5, 0.1, w_rank=8, u_ranks=[8], lstm_type="vmlmf").cuda().train()
    m.rnns = torch.nn.ModuleList([MyVMLSTMGroup(40, 40, w_rank=8, u_ranks=[4, 4]) for _ in range(2)]).cuda()
    m.reset_parameters()
    tok = torch.randint(0, 60, (5, 4), device="cuda")
    st0 = m.state_init(4)
    a, _ = m(tok, st0)
    a.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    m.stock_dropout = True
    before = m.dropout_state().clone()
    b, _ = m(tok, m.state_init(4))
    assert torch.equal(m.dropout_state(), before) and b.shape == a.shape


@gpu
def test_rate_and_independence_at_the_lm_shape():
    """8960 positions x 650 units (configs[4]): drop rate within five sigma at p = 0.1 / 0.5 / 0.9, rows and columns both mixed,
    sites / consecutive forwards independent of one another."""
    from vmlmf_amd.functional import dropout_factors, dropout_advance
    R, H = 8960, 650
    st = _state(SEED, 0)
    masks = []
    for p in (0.1, 0.5, 0.9):
        snap = dropout_advance(st)
        f = dropout_factors(R, H, p, snap, 1)
        z = (f == 0)
        n = R * H
        assert abs(z.double().mean().item() - p) < 5 * (p * (1 - p) / n) ** 0.5
        rows, cols = z.double().mean(1), z.double().mean(0)
        assert (rows - p).abs().max().item() < 6 * (p * (1 - p) / H) ** 0.5
        assert (cols - p).abs().max().item() < 6 * (p * (1 - p) / R) ** 0.5
        masks.append(z)
    snap = dropout_advance(st)
    a, b = dropout_factors(R, H, 0.5, snap, 1) == 0, dropout_factors(R, H, 0.5, snap, 2) == 0
    c = dropout_factors(R, H, 0.5, dropout_advance(st), 1) == 0
    for u, v in ((a, b), (a, c)):
        both = (u & v).double().mean().item()
        assert abs(both - 0.25) < 5 * (0.25 * 0.75 / (R * H)) ** 0.5
    # neighbouring elements (the four words of one call; consecutive calls) are uncorrelated
    for shift, dim in ((1, 1), (4, 1), (1, 0)):
        u = a.narrow(dim, 0, a.shape[dim] - shift)
        v = a.narrow(dim, shift, a.shape[dim] - shift)
        assert abs((u & v).double().mean().item() - 0.25) < 5 * (0.25 * 0.75 / u.numel()) ** 0.5


@gpu
def test_a_captured_training_step_draws_fresh_factors_on_every_replay():
    from vmlmf_amd.lm import Model
    torch.manual_seed(2)
    m = Model(80, 650, 2, 0.5, 0.05, w_rank=32, u_ranks=[32], lstm_type="vmlmf").cuda().train()
    st = m.dropout_state(seed=SEED)
    tok = torch.randint(0, 80, (4, 20), device="cuda")
    tgt = torch.randint(0, 80, (4, 20), device="cuda")
    states = m.state_init(20)

    def step():
        for p_ in m.parameters():
            p_.grad = None
        loss, _ = m.loss(tok, tgt, [tuple(s) for s in states])
        loss.backward()
        return loss

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    off0 = st[1].item()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss = step()
    losses = []
    for _ in range(4):
        g.replay()
        losses.append(loss.item())
    assert st[1].item() == off0 + 4
    assert len(set(losses)) == 4, losses
    # ... and each equals the eager step at the same offset
    st[1] = off0 + 1
    assert abs(step().item() - losses[1]) <= 1e-6 * abs(losses[1])


@gpu
def test_config_e_layer_at_full_size_drop_equals_the_undropped_layer_times_the_factors_bit_for_bit():
    """BASELINE configs[4]'s own size (B 256, T 35, H 650, ranks 32 / [32, 32]) - too large for the fp64 oracle in a test, so the
    size-independent property: the fused layer's y_dropped IS (the same layer's y) * factors, and its gradients ARE the
    undropped layer's gradients for the upstream gradient dy * factors - the same kernels with one multiplication moved, every
    bit equal; the factors at that size have the asked rate."""
    from hip_util import ORDER
    from vmlmf_amd import vmlmf_sequence, _lib
    from vmlmf_amd.functional import dropout_factors
    variant, g, B, T, H, rw, ru, p = O.V4, 2, 256, 35, 650, 32, [32, 32], 0.5
    desc = _rb_desc(variant, B, T, H, rw, ru, g)
    assert _lib.lib().vmlmf_dropout_fused(desc) == 1
    P = O.make_params(variant, H, H, rw, ru, seed=77, scale=0.05)
    torch.manual_seed(3)
    x = 0.5 * torch.randn(T, B, H, device="cuda")
    h0, c0 = 0.3 * torch.randn(B, H, device="cuda"), 0.3 * torch.randn(B, H, device="cuda")
    dy, dhT, dcT = torch.randn(T, B, H, device="cuda"), torch.randn(B, H, device="cuda"), torch.randn(B, H, device="cuda")
    snap = _state(SEED, 2024)
    F = dropout_factors(T * B, H, p, snap, 1, layer_desc=desc).reshape(T, B, H)
    assert abs((F == 0).double().mean().item() - p) < 5 * (0.25 / F.numel()) ** 0.5

    def run(drop, upstream):
        params = [torch.tensor(np.asarray(P[k]), device="cuda").requires_grad_(True) for k in ORDER[variant]]
        xg, h0g, c0g = x.clone().requires_grad_(True), h0.clone().requires_grad_(True), c0.clone().requires_grad_(True)
        y, hT, cT = vmlmf_sequence(variant, xg, h0g, c0g, params, rw, ru, g=g, time_major=True, drop=drop)
        ((y * upstream).sum() + (hT * dhT).sum() + (cT * dcT).sum()).backward()
        return y.detach(), hT.detach(), cT.detach(), [xg.grad, h0g.grad, c0g.grad] + [q.grad for q in params]

    yd, hTd, cTd, gd = run((p, snap, 1), dy)
    y, hT, cT, gu = run(None, dy * F)
    assert torch.equal(yd, y * F) and torch.equal(hTd, hT) and torch.equal(cTd, cT)
    for a, b in zip(gd, gu):
        assert torch.equal(a, b)


@gpu
@pytest.mark.parametrize("L,B,T,H,rw,ru,p", [(2, 6, 5, 32, 8, 8, 0.5), (3, 9, 7, 100, 16, 16, 0.25), (2, 128, 6, 256, 24, 24, 0.5), (2, 5, 1, 64, 8, 16, 0.5)],
                         ids=lambda v: str(v))
def test_dropout_between_the_layers_of_a_wavefront_stack_vs_oracle(L, B, T, H, rw, ru, p):
    """Round 6 (verdict r5 item 7): nn.Dropout(p) behind every layer of a stack the WAVEFRONT launches run (vmlmf_lm.py:437-439 with hidden
    sizes up to 256): the loader wave of a layer forms the step's factors, its storer writes the dropped copy the layer above reads, the
    backward multiplies dy by the same factors - against the fp64 oracle with the factors (identity columns: one-group layers)
    multiplied in; outputs, carried states, dx, dh0 / dc0 and every parameter gradient."""
    from hip_util import ORDER, assert_out, assert_grad
    from vmlmf_amd import functional as F, _lib
    variant = O.V3
    cfg = (variant, 1, rw, (ru,), True, _lib.DT_F32)
    assert F._stack_plan(cfg, L, B, T, H, H, True) is not None and F.stack_takes_dropout(cfg, L, B, T, H, H, True)
    Ps = [O.make_params(variant, H, H, rw, ru, seed=71 + l, scale=0.1) for l in range(L)]
    r = np.random.Generator(np.random.PCG64(12))
    x = (0.5 * r.standard_normal((T, B, H))).astype(np.float32)
    h0 = (0.3 * r.standard_normal((L, B, H))).astype(np.float32)
    c0 = (0.3 * r.standard_normal((L, B, H))).astype(np.float32)
    dy = r.standard_normal((T, B, H)).astype(np.float32)
    dhT = r.standard_normal((L, B, H)).astype(np.float32)
    dcT = r.standard_normal((L, B, H)).astype(np.float32)
    snap = _state(SEED, 23)
    names = ORDER[variant]
    params = [[torch.tensor(np.asarray(P[k]), device="cuda").requires_grad_(True) for k in names] for P in Ps]
    xg, h0g, c0g = (torch.tensor(a, device="cuda").requires_grad_(True) for a in (x, h0, c0))
    out = F.vmlmf_stack(variant, xg, params, rw, [ru], g=1, time_major=True, h0=h0g, c0=c0g, drops=[(p, snap, l + 1) for l in range(L)])
    assert out is not None
    yd, hs, cs = out
    loss = (yd * torch.tensor(dy, device="cuda")).sum()
    for l in range(L):
        loss = loss + (hs[l] * torch.tensor(dhT[l], device="cuda")).sum() + (cs[l] * torch.tensor(dcT[l], device="cuda")).sum()
    loss.backward()
    Fs = [O.dropout_factors(SEED, 23, l + 1, T * B, H, p).reshape(T, B, H) for l in range(L)]
    assert all(abs((f == 0).mean() - p) < 0.06 for f in Fs)
    f64 = torch.float64
    Pt = [O.to_torch(P, dtype=f64, requires_grad=True) for P in Ps]
    xt, h0t, c0t = (torch.tensor(a, dtype=f64, requires_grad=True) for a in (x, h0, c0))
    cur, lossr, hr, cr = xt, 0.0, [], []
    for l in range(L):
        cur, hT, cT = O.literal_sequence(variant, Pt[l], cur, h0t[l], c0t[l], time_major=True)
        cur = cur * torch.tensor(Fs[l], dtype=f64)
        hr.append(hT), cr.append(cT)
        lossr = lossr + (hT * torch.tensor(dhT[l], dtype=f64)).sum() + (cT * torch.tensor(dcT[l], dtype=f64)).sum()
    (lossr + (cur * torch.tensor(dy, dtype=f64)).sum()).backward()
    assert_out(yd.detach().cpu().numpy(), cur.detach().numpy(), "wf.drop.y_dropped")
    for l in range(L):
        assert_out(hs[l].detach().cpu().numpy(), hr[l].detach().numpy(), f"wf.drop.hT[{l}]")
        assert_out(cs[l].detach().cpu().numpy(), cr[l].detach().numpy(), f"wf.drop.cT[{l}]")
    assert_grad(xg.grad.cpu().numpy(), xt.grad.numpy(), "wf.drop.dx")
    assert_grad(h0g.grad.cpu().numpy(), h0t.grad.numpy(), "wf.drop.dh0")
    assert_grad(c0g.grad.cpu().numpy(), c0t.grad.numpy(), "wf.drop.dc0")
    for l in range(L):
        for k, p_ in zip(names, params[l]):
            assert_grad(p_.grad.cpu().numpy(), Pt[l][k].grad.numpy(), f"wf.drop.layer{l}.{k}")
