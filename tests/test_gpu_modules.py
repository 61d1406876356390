"""GPU: the nn.Module surface (what the reference's harness calls) against golden vectors from the reference:
Net + Adam loop (train.py:58-65), two-layer MyLSTM at BASELINE config C's shape, LM state carry
(lm_test.py:196-203), bare cell calls."""
import os

import numpy as np
import pytest
import torch

import vmlmf_oracle as O
from conftest import load_golden
from hip_util import assert_out, assert_grad
from vmlmf_amd import MyLSTM, MyVMLMFCell, MyVMLMFCellg2, MyVMLSTM, MyVMLSTMGroup, Net

pytestmark = pytest.mark.gpu
DEV = "cuda"


def load_cell(cell, P, group_sep=None):
    with torch.no_grad():
        for k, v in P.items():
            if hasattr(cell, "layers") and k in cell.layers:
                cell.layers[k].copy_(torch.tensor(v))
            elif "." in k:                                   # u_h.0 -> ParameterList
                name, idx = k.split(".")
                getattr(cell, name)[int(idx)].copy_(torch.tensor(v))
            else:
                getattr(cell, k).copy_(torch.tensor(v))


def test_net_adam_three_steps_vs_reference():
    d = load_golden("cfgA_net_adam3")
    _, B, T, I, H, rw, ru = (int(v) for v in d["meta"])
    torch.manual_seed(0)                                      # same creation order => same lin init
    net = Net(I, layer_sizes=[H], w_rank=rw, u_rank=[ru], model=MyLSTM, cell=MyVMLMFCell)
    assert np.array_equal(net.lin.weight.detach().numpy(), d["lin_w"])
    load_cell(net.rnn.rnncells[0], O.make_params(O.V1, I, H, rw, ru, seed=int(d["seeds"][0])))
    net = net.to(DEV)
    x, tgt = O.synthetic_batch(B, T, I, seed=int(d["seeds"][1]))
    x, tgt = torch.tensor(x, device=DEV), torch.tensor(tgt, device=DEV)
    opt = torch.optim.Adam(net.parameters(), lr=0.002)
    for step in range(3):
        net.zero_grad()
        out = net(x)
        loss = torch.nn.functional.cross_entropy(out, tgt.long())
        loss.backward()
        opt.step()
        assert abs(loss.item() - float(d["losses"][step])) < 2e-5, (step, loss.item(), d["losses"][step])
        assert_out(out.detach().cpu().numpy(), d["logits"][step], f"logits[{step}]", atol=2e-5)
    assert all(p.grad is None for p in net.cell.parameters())          # the unused duplicate stays untouched
    sd = net.state_dict()
    for k, v in d["final"].items():
        assert_out(sd[k].cpu().numpy(), v, "final." + k, atol=2e-5, rtol=1e-3)


def test_config_c_two_layer_mylstm_vs_reference():
    """BASELINE config C shape (fp32): Opportunity, 2 layers H=256 rank 24, B=128 T=24 I=77."""
    d = load_golden("cfgC_v1_opp2")
    _, B, T, I, H, rw, ru = (int(v) for v in d["meta"])
    rnn = MyLSTM(I, hidden_layer_sizes=[H, H], batch_first=True, w_rank=rw, u_ranks=[ru], cell=MyVMLMFCell)
    load_cell(rnn.rnncells[0], O.make_params(O.V1, I, H, rw, ru, seed=int(d["seeds"][0])))
    load_cell(rnn.rnncells[1], O.make_params(O.V1, H, H, rw, ru, seed=int(d["seeds"][1])))
    rnn = rnn.to(DEV)
    x_np, _ = O.synthetic_batch(B, T, I, seed=int(d["seeds"][2]), classes=18)
    dy = np.random.Generator(np.random.PCG64(int(d["seeds"][3]))).standard_normal((B, T, H)).astype(np.float32)
    x = torch.tensor(x_np, device=DEV, requires_grad=True)
    y, hcat = rnn(x)
    (y * torch.tensor(dy, device=DEV)).sum().backward()
    assert_out(y.detach().cpu().numpy()[:, ::6], d["y_s"], "y")
    assert_out(hcat.detach().cpu().numpy(), d["hT"], "hT")
    assert_grad(x.grad.cpu().numpy()[::4], d["dx_s"], "dx")
    for li, G in ((0, d["G0"]), (1, d["G1"])):
        for k, v in G.items():
            assert_grad(getattr(rnn.rnncells[li], k).grad.cpu().numpy(), v, f"layer{li}.{k}")


def test_two_layers_of_different_sizes_vs_reference():
    """MyLSTM(hidden_layer_sizes=[128, 256]) on the default launch policy (one wavefront launch per direction since round 6)
    against the imported reference's vectors."""
    d = load_golden("seq_v1_h128_h256")
    _, B, T, I, H0, H1, rw, ru = (int(v) for v in d["meta"])
    rnn = MyLSTM(I, hidden_layer_sizes=[H0, H1], batch_first=True, w_rank=rw, u_ranks=[ru], cell=MyVMLMFCell)
    load_cell(rnn.rnncells[0], O.make_params(O.V1, I, H0, rw, ru, seed=int(d["seeds"][0])))
    load_cell(rnn.rnncells[1], O.make_params(O.V1, H0, H1, rw, ru, seed=int(d["seeds"][1])))
    rnn = rnn.to(DEV)
    x_np, _ = O.synthetic_batch(B, T, I, seed=int(d["seeds"][2]), classes=18)
    dy = np.random.Generator(np.random.PCG64(int(d["seeds"][3]))).standard_normal((B, T, H1)).astype(np.float32)
    x = torch.tensor(x_np, device=DEV, requires_grad=True)
    y, hcat = rnn(x)
    (y * torch.tensor(dy, device=DEV)).sum().backward()
    assert_out(y.detach().cpu().numpy()[:, ::4], d["y_s"], "y")
    assert_out(hcat.detach().cpu().numpy(), d["hT"], "hT")
    assert_grad(x.grad.cpu().numpy()[::4], d["dx_s"], "dx")
    for li, G in ((0, d["G0"]), (1, d["G1"])):
        for k, v in G.items():
            assert_grad(getattr(rnn.rnncells[li], k).grad.cpu().numpy(), v, f"layer{li}.{k}")


def test_lm_state_carry_two_minibatches_vs_reference():
    d = load_golden("lm_v3_carry")
    _, B, T, H, _, rw, ru = (int(v) for v in d["meta"])
    layer = MyVMLSTM(H, H, w_rank=rw, u_ranks=ru)
    load_cell(layer, d["P"])
    layer = layer.to(DEV)
    states = (torch.zeros(B, H, device=DEV), torch.zeros(B, H, device=DEV))
    for i in range(2):
        layer.zero_grad()
        states = (states[0].detach(), states[1].detach())
        y, states = layer(torch.tensor(d[f"x{i}"], device=DEV), states)
        loss = torch.mean(y * torch.tensor(d[f"w{i}"], device=DEV)) * B
        loss.backward()
        assert abs(loss.item() - float(d[f"loss{i}"][0])) < 1e-5
        assert_out(states[0].detach().cpu().numpy(), d[f"hT{i}"], "hT")
        assert_out(states[1].detach().cpu().numpy(), d[f"cT{i}"], "cT")
        for k, v in d[f"G{i}"].items():
            assert_grad(getattr(layer, k).grad.cpu().numpy(), v, f"carry{i}.{k}")


@pytest.mark.parametrize("name,cls", [("cell_v1", MyVMLMFCell), ("cell_v2", MyVMLMFCellg2),
                                      ("cell_v3", MyVMLSTM), ("cell_v4", MyVMLSTMGroup)])
def test_bare_cell_calls_vs_reference(name, cls):
    d = load_golden(name)
    meta = [int(v) for v in d["meta"]]
    variant, B, _, I, H, rw = meta[:6]
    ru = meta[6:]
    if variant in (O.V1, O.V3):
        cell = cls(I, H, w_rank=rw, u_ranks=ru[0])
    else:
        cell = cls(I, H, w_rank=rw, u_ranks=ru)
    load_cell(cell, d["P"])
    cell = cell.to(DEV)
    x = torch.tensor(d["x"], device=DEV)
    h = torch.tensor(d["h0"], device=DEV)
    c = torch.tensor(d["c0"], device=DEV)
    with torch.no_grad():
        hn, cn = cell(x, (h, c)) if variant in (O.V1, O.V2) else cell.lstm_step(x, h, c)
    assert_out(hn.cpu().numpy(), d["h1"], "h1")
    assert_out(cn.cpu().numpy(), d["c1"], "c1")


def test_unsupported_shape_raises_and_never_falls_back():
    """w_rank > 32 is outside this round's kernels: explicit error, no fallback."""
    from vmlmf_amd import _lib
    layer = MyVMLSTM(64, 64, w_rank=40, u_ranks=8).to(DEV)
    # (VmlmfError from the ctypes binding, a plain RuntimeError with the same text from the C++ one)
    with pytest.raises(RuntimeError, match=r"vmlmf_hip error %d: padded w_rank > 32" % _lib.E_UNSUPPORTED):
        layer(torch.zeros(3, 4, 64, device=DEV), (torch.zeros(4, 64, device=DEV), torch.zeros(4, 64, device=DEV)))


@pytest.mark.parametrize("B,H,C,strided", [(64, 180, 18, False), (64, 180, 18, True), (1, 7, 1, False),
                                           (513, 650, 32, True), (40, 33, 6, False)])
def test_head_linear_matches_library_linear(B, H, C, strided):
    """Classifier head kernels (Net.lin, V/src/models/vmlmf.py:345,353-355) against F.linear in fp64."""
    from vmlmf_amd.functional import HeadLinearFn
    g = torch.Generator().manual_seed(B * 1000 + H + C)
    seq = torch.randn(B, 3, H, generator=g).cuda()
    h = (seq[:, -1] if strided else seq[:, -1].contiguous()).requires_grad_(True)
    W = (0.1 * torch.randn(C, H, generator=g)).cuda().requires_grad_(True)
    b = torch.randn(C, generator=g).cuda().requires_grad_(True)
    dl = torch.randn(B, C, generator=g).cuda()
    out = HeadLinearFn.apply(h, W, b)
    gh, gW, gb = torch.autograd.grad(out, (h, W, b), dl)
    h64, W64, b64 = (t.detach().double().requires_grad_(True) for t in (h, W, b))
    ref = torch.nn.functional.linear(h64, W64, b64)
    rh, rW, rb = torch.autograd.grad(ref, (h64, W64, b64), dl.double())
    for got, want in ((out, ref), (gh, rh), (gW, rW), (gb, rb)):
        scale = max(1.0, float(want.abs().max()))
        assert float((got.double() - want).abs().max()) <= 2e-5 * scale
    out2 = HeadLinearFn.apply(h, W, b)
    gW2, = torch.autograd.grad(out2, (W,), dl)
    assert torch.equal(out, out2) and torch.equal(gW, gW2)     # fixed summation order


def test_head_rejects_too_many_classes_in_backward_only_through_dispatch():
    from vmlmf_amd.functional import head_linear
    h = torch.randn(4, 16).cuda().requires_grad_(True)
    W = torch.randn(40, 16).cuda().requires_grad_(True)
    out = head_linear(h, W, None)           # > 32 classes: stock library linear
    assert out.shape == (4, 40)
    out.sum().backward()
    assert h.grad is not None


@pytest.mark.parametrize("B,C,ignored", [(64, 18, 0), (64, 18, 5), (1, 2, 0), (300, 7, 17), (513, 100, 0)])
def test_fused_cross_entropy_matches_library(B, C, ignored):
    """vmlmf_amd.cross_entropy against torch.nn.functional.cross_entropy (the reference's criterion) in fp64."""
    import vmlmf_amd
    g = torch.Generator().manual_seed(B + C)
    z = (3 * torch.randn(B, C, generator=g)).cuda().requires_grad_(True)
    t = torch.randint(0, C, (B,), generator=g)
    if ignored:
        t[torch.randperm(B, generator=g)[:min(ignored, B - 1)]] = -100
    t = t.cuda()
    loss = vmlmf_amd.cross_entropy(z, t)
    (3.0 * loss).backward()
    z64 = z.detach().double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(z64, t)
    (3.0 * ref).backward()
    assert abs(float(loss) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    assert float((z.grad.double() - z64.grad).abs().max()) <= 1e-7 + 2e-6 * float(z64.grad.abs().max())
    loss2 = vmlmf_amd.CrossEntropyLoss()(z.detach(), t)
    assert torch.equal(loss.detach(), loss2)


def test_cross_entropy_dispatch_leaves_other_cases_to_the_library():
    import vmlmf_amd
    z = torch.randn(4, 3, 5).cuda()            # (N, C, d) form: not the classifier case
    t = torch.randint(0, 3, (4, 5)).cuda()
    assert torch.allclose(vmlmf_amd.cross_entropy(z, t), torch.nn.functional.cross_entropy(z, t))


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_fused_adam_matches_torch_adam(wd):
    """vmlmf_amd.optim.Adam against torch.optim.Adam (the reference's optimizer, train.py:47) over 6 steps,
    including a parameter that never gets a gradient (Net.cell) and one that gets it late."""
    import vmlmf_amd
    g = torch.Generator().manual_seed(11)
    shapes = [(9, 16), (720, 16), (1, 180), (720,), (18, 180), (5,)]
    mine = [torch.nn.Parameter(torch.randn(*s, generator=g).cuda()) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    o1 = vmlmf_amd.optim.Adam(mine, lr=2e-3, weight_decay=wd)
    o2 = torch.optim.Adam(ref, lr=2e-3, weight_decay=wd)
    for it in range(6):
        for k, (a, b) in enumerate(zip(mine, ref)):
            if k == 5 or (k == 4 and it < 2):       # never / late
                a.grad = b.grad = None
                continue
            gr = torch.randn(*a.shape, generator=g).cuda()
            a.grad, b.grad = gr.clone(), gr.clone()
        o1.step()
        o2.step()
    for a, b in zip(mine, ref):
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max()))
    assert float(o1.state[mine[0]]["step"]) == 6.0
    sd = o1.state_dict()                        # same structure as the stock optimizer's
    assert set(sd["param_groups"][0]) == set(o2.state_dict()["param_groups"][0]) - {
        "amsgrad", "maximize", "foreach", "capturable", "differentiable", "fused", "decoupled_weight_decay"}


def test_clip_sgd_step_matches_the_lm_loop():
    """clip_grad_norm_ + `param -= lr * grad` (lm_test.py:203-209), clipping active and inactive."""
    import vmlmf_amd
    g = torch.Generator().manual_seed(3)
    # (shapes: 16-byte accesses with and without a tail of n % 4 elements; one tensor of more than 2^20 elements - the norm's
    # partial sums then run on 1024-thread workgroups)
    for max_norm, scale, shapes in ((0.25, 1.0, [(650, 32), (2600,), (1, 650)]), (1e3, 1.0, [(650, 32), (2600,), (1, 650)]),
                                    (0.25, 1.0, [(7, 3), (1100, 1000), (5,)]), (1e4, 1.0, [(7, 3), (1100, 1000), (5,)])):
        mine = [torch.nn.Parameter(torch.randn(*s, generator=g).cuda()) for s in shapes]
        ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
        for a, b in zip(mine, ref):
            gr = scale * torch.randn(*a.shape, generator=g).cuda()
            a.grad, b.grad = gr.clone(), gr.clone()
        norm = vmlmf_amd.optim.clip_sgd_step(mine, lr=0.5, max_norm=max_norm)
        with torch.no_grad():
            rnorm = torch.nn.utils.clip_grad_norm_(ref, max_norm)
            for p in ref:
                p -= 0.5 * p.grad
        assert abs(float(norm) - float(rnorm)) <= 1e-5 * float(rnorm)
        for a, b in zip(mine, ref):
            assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max()))
            assert float((a.grad - b.grad).abs().max()) <= 2e-6 * max(1.0, float(b.grad.abs().max()))


def test_graphed_train_step_matches_the_eager_loop():
    """GraphedTrainStep (forward + fused CE + backward + fused Adam in one hipGraph) against the same loop eager."""
    import copy
    import vmlmf_amd
    torch.manual_seed(1)
    a = Net(9, layer_sizes=[40], w_rank=8, u_rank=[8], model=MyLSTM, cell=MyVMLMFCell).cuda()
    b = copy.deepcopy(a)
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(16, 12, 9, generator=g).cuda() for _ in range(4)]
    ts = [torch.randint(0, 18, (16,), generator=g).cuda() for _ in range(4)]
    step = vmlmf_amd.GraphedTrainStep(a, vmlmf_amd.cross_entropy, vmlmf_amd.optim.Adam(a.parameters(), lr=2e-3),
                                      xs[0], ts[0], warmup=2)
    opt = vmlmf_amd.optim.Adam(b.parameters(), lr=2e-3)
    # constructing the graphed step must leave model and optimizer untouched (its warm-up runs on a snapshot): the
    # eager twin does NO extra steps, the first call is step 1 of the loop (train.py:58-65)
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(pa, pb)
    xs.append(torch.randn(5, 12, 9, generator=g).cuda())      # the last, shorter DataLoader batch: eager fallback
    ts.append(torch.randint(0, 18, (5,), generator=g).cuda())
    for it, (x, t) in enumerate(zip(xs, ts)):
        if it == 2:                         # a scheduler changes the learning rate: the replay must follow it
            for o in (step.optimizer, opt):
                o.param_groups[0]["lr"] = 5e-4
        la = step(x, t).clone()
        b.zero_grad(set_to_none=True)
        lb = vmlmf_amd.cross_entropy(b(x), t)
        lb.backward()
        opt.step()
        assert abs(float(la) - float(lb)) <= 1e-6 * max(1.0, abs(float(lb)))
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-7)
    assert float(step.optimizer.state[next(a.rnn.parameters())]["step"]) == float(len(xs))


def test_fused_adam_resumes_from_a_checkpoint_like_torch_adam():
    """save -> load_state_dict -> step: moments and step counts must survive, also when the checkpoint was
    written by torch.optim.Adam itself."""
    import vmlmf_amd
    g = torch.Generator().manual_seed(2)
    shapes = [(9, 16), (180,), (18, 180)]
    mine = [torch.nn.Parameter(torch.randn(*s, generator=g).cuda()) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    o1, o2 = vmlmf_amd.optim.Adam(mine, lr=2e-3), torch.optim.Adam(ref, lr=2e-3)

    def feed():
        for a, b in zip(mine, ref):
            gr = torch.randn(*a.shape, generator=g).cuda()
            a.grad, b.grad = gr.clone(), gr.clone()

    for _ in range(3):
        feed()
        o1.step(), o2.step()
    sd_mine, sd_ref = o1.state_dict(), o2.state_dict()
    fresh = vmlmf_amd.optim.Adam(mine, lr=2e-3)             # resume on a fresh optimizer
    fresh.load_state_dict(sd_mine)
    o1.load_state_dict(sd_ref)                              # and on one that has stepped, from the stock optimizer's file
    mine2 = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    third = vmlmf_amd.optim.Adam(mine2, lr=2e-3)
    third.load_state_dict(sd_ref)
    for _ in range(2):
        feed()
        for a, c in zip(mine, mine2):
            c.grad = a.grad.clone()
        fresh.step(), o2.step(), third.step()
    for a, b, c in zip(mine, ref, mine2):
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max()))
        assert float((c - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max()))
    assert float(fresh.state[mine[0]]["step"]) == 5.0 and float(third.state[mine2[0]]["step"]) == 5.0


@pytest.mark.parametrize("name", ["seq_v5_wide", "seq_v6_demo"])
def test_comparison_cells_through_mylstm_vs_reference(name):
    """MyLSTM(cell=MyLSTMCell, low-rank) and MyLSTM(cell=MyVMLMFgCellg2) are one sequence
    pipeline per layer on the same kernels; gradients land on the reference's own parameter tensors."""
    from vmlmf_amd import MyLSTMCell, MyVMLMFgCellg2
    d = load_golden(name)
    meta = [int(v) for v in d["meta"]]
    variant, B, T, I, H, rw = meta[:6]
    ru = meta[6:]
    cellcls = MyLSTMCell if variant == O.V5 else MyVMLMFgCellg2
    rnn = MyLSTM(I, hidden_layer_sizes=[H], batch_first=True, w_rank=rw, u_ranks=ru, cell=cellcls)
    load_cell(rnn.rnncells[0], d["P"])
    rnn = rnn.to(DEV)
    x = torch.tensor(d["x"], device=DEV, requires_grad=True)
    y, hcat = rnn(x)
    ((y * torch.tensor(d["dy"], device=DEV)).sum() + (hcat * torch.tensor(d["dhT"], device=DEV)).sum()).backward()
    assert_out(y.detach().cpu().numpy(), d["y"], "y")
    assert_out(hcat.detach().cpu().numpy(), d["hT"], "hT")
    assert_grad(x.grad.cpu().numpy(), d["dx"], "dx")
    named = dict(rnn.rnncells[0].named_parameters())
    for k, v in d["G"].items():
        p = named[k] if k in named else named["layers." + k]
        assert_grad(p.grad.cpu().numpy(), v, "G." + k)


@pytest.mark.parametrize("name", ["cell_v5", "cell_v6"])
def test_comparison_cells_single_step_call(name):
    """cell(x, (h, c)) -> (h', c'), the signature MyLSTM's loop in the reference uses (vmlmf.py:306)."""
    from vmlmf_amd import MyLSTMCell, MyVMLMFgCellg2
    d = load_golden(name)
    meta = [int(v) for v in d["meta"]]
    variant, B, _, I, H, rw = meta[:6]
    ru = meta[6:]
    cell = MyLSTMCell(I, H, w_rank=rw, u_ranks=ru[0]) if variant == O.V5 else MyVMLMFgCellg2(I, H, w_rank=rw, u_ranks=ru)
    load_cell(cell, d["P"])
    cell = cell.to(DEV)
    h1, c1 = cell(torch.tensor(d["x"], device=DEV), (torch.tensor(d["h0"], device=DEV), torch.tensor(d["c0"], device=DEV)))
    assert_out(h1.detach().cpu().numpy(), d["h1"], "h1")
    assert_out(c1.detach().cpu().numpy(), d["c1"], "c1")


def test_vanilla_baseline_cell_keeps_stock_ops_on_gpu():
    """MyLSTMCell without ranks (dense gate matrices) is not on the HIP path: stock GEMMs, reference loop."""
    from vmlmf_amd import MyLSTMCell
    torch.manual_seed(3)
    rnn = MyLSTM(9, hidden_layer_sizes=[32], cell=MyLSTMCell).to(DEV)
    y, h = rnn(torch.randn(4, 6, 9, device=DEV))
    assert y.shape == (4, 6, 32) and torch.isfinite(y).all()


def test_nll_loss_vs_reference_at_ptb_vocabulary():
    """lm_test.py:140-153 on the fused kernels: loss and gradient against values captured from the reference."""
    from vmlmf_amd import nll_loss
    d = load_golden("nll_v10000")
    T, B, V, seed = (int(v) for v in d["meta"])
    r = np.random.Generator(np.random.PCG64(seed))
    z = torch.tensor((2.0 * r.standard_normal((T * B, V))).astype(np.float32), device=DEV, requires_grad=True)
    y = torch.tensor(r.integers(0, V, size=(T, B)), device=DEV)
    loss = nll_loss(z, y)
    (float(d["upstream"][0]) * loss).backward()
    assert abs(loss.item() - float(d["loss"][0])) < 1e-4 * abs(float(d["loss"][0]))
    g = z.grad.cpu().numpy()
    assert_grad(g[:, ::97], d["g_s"], "dscores sample")
    assert_grad(g[np.arange(T * B), d["y"].reshape(-1)], d["g_target"], "dscores at targets")


@pytest.mark.parametrize("R,B,V", [(6, 3, 1003), (8, 4, 20000), (35, 5, 16384), (4, 2, 7)])
def test_nll_loss_shapes_vs_oracle(R, B, V):
    """Rows that are not a multiple of four wide (scalar path), wider than the register-resident limit (two-pass
    path), exactly at it, and tiny; scores large enough that the reference's plain exp would overflow."""
    from vmlmf_amd import nll_loss
    r = np.random.Generator(np.random.PCG64(R + V))
    z = (30.0 * r.standard_normal((R, V))).astype(np.float32)      # row maxima around 100: exp overflows in fp32
    y = r.integers(0, V, size=(R // B, B))
    want, gwant = O.nll_loss_stable(z, y)
    zt = torch.tensor(z, device=DEV, requires_grad=True)
    loss = nll_loss(zt, torch.tensor(y, device=DEV))
    loss.backward()
    assert np.isfinite(loss.item()) and abs(loss.item() - want) < 1e-4 * abs(want)
    assert_grad(zt.grad.cpu().numpy(), gwant, "dscores")


def test_lm_network_two_minibatches_vs_reference():
    """The LM loop of lm_test.py:196-209 on this package: Model(lstm_type="vmlmf") -> nll_loss -> backward ->
    clip_sgd_step, two minibatches with detached state carry, against the reference's own run."""
    from vmlmf_amd import Model, nll_loss, optim
    d = load_golden("lm_model_v3")
    V, H, L, B, T, rw, ru = (int(v) for v in d["meta"])
    model = Model(V, H, L, 0.0, 0.1, w_rank=rw, u_ranks=[ru], lstm_type="vmlmf")
    model.load_state_dict({k: torch.tensor(v) for k, v in d["init"].items()})
    model = model.to(DEV)
    states = model.state_init(B)
    for i in range(2):
        model.zero_grad()
        states = model.detach(states)
        scores, states = model(torch.tensor(d[f"x{i}"], device=DEV), states)
        loss = nll_loss(scores, torch.tensor(d[f"y{i}"], device=DEV))
        loss.backward()
        assert_out(scores.detach().cpu().numpy(), d[f"scores{i}"], f"scores{i}")
        assert abs(loss.item() - float(d[f"loss{i}"][0])) < 1e-4 * float(d[f"loss{i}"][0])
        for k, p in model.named_parameters():
            assert_grad(p.grad.cpu().numpy(), d[f"G{i}"][k], f"G{i}.{k}")
        norm = optim.clip_sgd_step(model.parameters(), lr=1.0, max_norm=0.25)
        assert abs(float(norm) - float(d[f"norm{i}"][0])) < 1e-4 * float(d[f"norm{i}"][0])
    sd = model.state_dict()
    for k, v in d["final"].items():
        assert_out(sd[k].cpu().numpy(), v, "final." + k, atol=2e-5, rtol=1e-3)
    assert_out(torch.stack([s[0].detach() for s in states]).cpu().numpy(), d["hT"], "hT")


def test_cross_entropy_unit_gradient_shortcut_equals_the_general_backward():
    """loss.backward(unit_gradient(dev)) returns the gradient the forward kernel wrote; loss.backward() and an explicit
    other scale take the backward kernel.  All three must agree (the first two bit for bit)."""
    import vmlmf_amd
    torch.manual_seed(5)
    z0 = torch.randn(64, 18, device=DEV)
    t = torch.randint(0, 18, (64,), device=DEV)
    t[3] = -100
    grads = []
    for mode in ("unit", "implicit", "scaled"):
        z = z0.clone().requires_grad_(True)
        loss = vmlmf_amd.cross_entropy(z, t)
        if mode == "unit":
            loss.backward(vmlmf_amd.unit_gradient(DEV))
        elif mode == "implicit":
            loss.backward()
        else:
            loss.backward(torch.tensor(2.5, device=DEV))
        grads.append(z.grad.clone())
    assert torch.equal(grads[0], grads[1])
    assert torch.allclose(grads[2], 2.5 * grads[0], rtol=1e-6, atol=0)
    zr = z0.clone().requires_grad_(True)
    torch.nn.functional.cross_entropy(zr, t).backward()
    assert torch.allclose(grads[0], zr.grad, rtol=1e-5, atol=1e-7)
    assert vmlmf_amd.unit_gradient(DEV) is vmlmf_amd.unit_gradient("cuda") and float(vmlmf_amd.unit_gradient(DEV)) == 1.0


def test_unit_gradient_shortcut_survives_a_retained_graph():
    """Leaf logits + retain_graph: the second backward must add the same gradient again, not a doubled one."""
    import vmlmf_amd
    torch.manual_seed(6)
    z = torch.randn(16, 7, device=DEV, requires_grad=True)
    t = torch.randint(0, 7, (16,), device=DEV)
    loss = vmlmf_amd.cross_entropy(z, t)
    loss.backward(vmlmf_amd.unit_gradient(DEV), retain_graph=True)
    g1 = z.grad.clone()
    loss.backward(vmlmf_amd.unit_gradient(DEV), retain_graph=True)
    assert torch.allclose(z.grad, 2 * g1, rtol=1e-6, atol=0)
    # in-place work on the leaf's gradient must not reach the buffer the graph keeps
    z.grad.zero_()
    torch.nn.utils.clip_grad_norm_([z], 1e-3)
    loss.backward(vmlmf_amd.unit_gradient(DEV))
    assert torch.allclose(z.grad, g1, rtol=1e-6, atol=0)


def test_out_of_range_targets_poison_the_loss_without_reading_out_of_bounds():
    """A class index outside [0, C) that is not ignore_index: PyTorch asserts on the device; the fused criteria return NaN."""
    import vmlmf_amd
    z = torch.randn(8, 5, device=DEV)
    t = torch.tensor([0, 1, 2, 3, 4, 0, 1, 2], device=DEV)
    assert torch.isfinite(vmlmf_amd.cross_entropy(z, t))
    for bad in (5, -3, 1 << 40):
        t2 = t.clone()
        t2[4] = bad
        assert torch.isnan(vmlmf_amd.cross_entropy(z, t2))
    s = torch.randn(6, 1000, device=DEV)
    y = torch.randint(0, 1000, (3, 2), device=DEV)
    assert torch.isfinite(vmlmf_amd.nll_loss(s, y))
    y[1, 1] = 1000
    assert torch.isnan(vmlmf_amd.nll_loss(s, y))


def test_both_bindings_give_the_same_results():
    """The package reaches the C ABI through the C++ TORCH_LIBRARY binding when it is built (csrc/torch_binding.cpp) and
    through ctypes otherwise (VMLMF_PYBIND=ctypes): same kernels, so the same bits - checked in a fresh interpreter for
    the ctypes side, on a training step of Net with the fused criterion."""
    import os
    import subprocess
    import sys
    from vmlmf_amd import functional as F
    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import sys; sys.path[:0] = [%r]\n"
        "import torch, vmlmf_amd\n"
        "from vmlmf_amd import Net, MyLSTM, MyVMLMFCell, functional as F\n"
        "torch.manual_seed(3)\n"
        "net = Net(9, layer_sizes=[24, 40], w_rank=8, u_rank=[8], model=MyLSTM, cell=MyVMLMFCell).cuda()\n"
        "x = torch.randn(6, 7, 9, device='cuda'); t = torch.randint(0, 18, (6,), device='cuda')\n"
        "loss = vmlmf_amd.cross_entropy(net(x), t); loss.backward(vmlmf_amd.unit_gradient('cuda'))\n"
        "with torch.no_grad(): y = net(x)\n"
        "print('binding', 'cpp' if F.torch_ops() is not None else 'ctypes')\n"
        "print('vals', repr(float(loss)), repr(float(y.double().sum())), repr(sum(float(p.grad.double().sum()) for p in net.parameters() if p.grad is not None)))\n"
    ) % (os.path.dirname(here),)
    outs = {}
    for mode in ("cpp", "ctypes"):
        env = dict(os.environ)
        if mode == "ctypes":
            env["VMLMF_PYBIND"] = "ctypes"
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = dict(l.split(" ", 1) for l in r.stdout.strip().splitlines() if " " in l)
        outs[mode] = lines
    assert outs["ctypes"]["binding"] == "ctypes"
    if F.torch_ops() is not None:
        assert outs["cpp"]["binding"] == "cpp"
        assert outs["cpp"]["vals"] == outs["ctypes"]["vals"]


def test_kept_parameter_images_follow_every_visible_parameter_change():
    """vmlmf_amd.cache_packed_parameters: forward reuses the packed images while (data_ptr, _version) of the parameters are
    unchanged and re-packs after an in-place update, an optimizer step (stock and fused), a load_state_dict; results are
    bit-identical to the uncached module throughout, in eager mode and inside a captured training step."""
    import copy
    import vmlmf_amd
    torch.manual_seed(2)
    a = Net(9, layer_sizes=[40], w_rank=8, u_rank=[8], model=MyLSTM, cell=MyVMLMFCell).cuda()
    b = copy.deepcopy(a)
    assert vmlmf_amd.cache_packed_parameters(a, True) == 2          # rnn layer + the reference's spare `cell`
    cache = a.rnn.rnncells[0]._pack_cache
    g = torch.Generator().manual_seed(4)
    x = torch.randn(8, 10, 9, generator=g).cuda()
    t = torch.randint(0, 18, (8,), generator=g).cuda()
    oa, ob = vmlmf_amd.optim.Adam(a.parameters(), lr=1e-2), vmlmf_amd.optim.Adam(b.parameters(), lr=1e-2)

    def step(net, opt=None):
        net.zero_grad(set_to_none=True)
        loss = vmlmf_amd.cross_entropy(net(x), t)
        loss.backward()
        if opt is not None:
            opt.step()
        return loss.detach().clone(), [p.grad.clone() for p in net.rnn.parameters()]

    def same():
        la, ga = step(a)
        lb, gb = step(b)
        assert torch.equal(la, lb) and all(torch.equal(u, v) for u, v in zip(ga, gb))

    same()
    same()
    assert (cache.fills, cache.hits) == (1, 1)
    step(a, oa), step(b, ob)                   # fused Adam writes through raw pointers: versions must move
    same()
    assert cache.fills == 2
    with torch.no_grad():
        for net in (a, b):
            net.rnn.rnncells[0].u_h.mul_(1.5)  # the reference's manual update style (lm_test.py:205-207)
    same()
    assert cache.fills == 3
    sd = {k: v * 0.5 for k, v in b.state_dict().items()}
    a.load_state_dict(sd), b.load_state_dict(sd)
    same()
    assert cache.fills == 4
    with torch.no_grad():                      # inference keeps the images as well
        ya, yb = a(x), b(x)
    assert torch.equal(ya, yb) and cache.fills == 4
    # a captured training step packs inside the graph (the cache is never filled during capture), and replays move the versions
    sa = vmlmf_amd.GraphedTrainStep(a, vmlmf_amd.cross_entropy, oa, x, t, warmup=1)
    sb = vmlmf_amd.GraphedTrainStep(b, vmlmf_amd.cross_entropy, ob, x, t, warmup=1)
    for _ in range(3):
        assert torch.equal(sa(x, t), sb(x, t))
    same()


@pytest.mark.parametrize("layers,B,T", [([180], 64, 16), ([24, 40], 7, 5), ([600], 5, 3)])
def test_classifier_riding_on_the_last_layer_equals_the_separate_head(layers, B, T):
    """Net.forward hands Net.lin to the last layer's call (vmlmf_seq_*_ex: logits from the recurrence's epilogue, d(hT) in the
    backward's prologue, dW / db from the final gradient kernel; stand-alone head kernels inside the call for the row-block /
    step-wise layers): same logits, loss and gradients as the layer followed by vmlmf_amd.head_linear."""
    import copy
    import vmlmf_amd
    from vmlmf_amd.functional import head_linear
    torch.manual_seed(7)
    a = Net(9, layer_sizes=layers, w_rank=8, u_rank=[8], model=MyLSTM, cell=MyVMLMFCell).cuda()
    b = copy.deepcopy(a)
    x = torch.randn(B, T, 9, device=DEV)
    t = torch.randint(0, 18, (B,), device=DEV)
    la = vmlmf_amd.cross_entropy(a(x), t)
    la.backward()
    _, hid = b.rnn.run_layers(x)
    lb = vmlmf_amd.cross_entropy(head_linear(hid[-1], b.lin.weight, b.lin.bias), t)
    lb.backward()
    assert abs(float(la) - float(lb)) <= 1e-6 * abs(float(lb))
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if pb.grad is None:
            assert pa.grad is None, k
            continue
        assert_grad(pa.grad.cpu().numpy(), pb.grad.cpu().numpy(), k, rel=2e-5)
    with torch.no_grad():
        assert_out(a(x).cpu().numpy(), head_linear(b.rnn.run_layers(x)[1][-1], b.lin.weight, b.lin.bias).cpu().numpy(), "logits")


def test_linear_nll_without_the_score_tensor_equals_projection_plus_loss():
    """vmlmf_amd.linear_nll: Linear (vmlmf_lm.py:355-358) + nll_loss (lm_test.py:140-153) in row chunks whose scores never form
    a (T*B, V) tensor - same loss, same gradients (the backward recomputes the chunk's scores) as the projection followed by
    the loss; by default only taken when no gradient is needed."""
    import vmlmf_amd
    torch.manual_seed(4)
    T, B, H, V = 7, 9, 40, 301
    h = torch.randn(T, B, H, device="cuda", requires_grad=True)
    w = (0.2 * torch.randn(V, H, device="cuda")).requires_grad_(True)
    b = (0.1 * torch.randn(V, device="cuda")).requires_grad_(True)
    y = torch.randint(0, V, (T, B), device="cuda")
    ref = vmlmf_amd.nll_loss(torch.addmm(b, h.reshape(-1, H), w.t()), y)
    ref.backward()
    want = [t.grad.clone() for t in (h, w, b)]
    for chunk in (5, 16, 64):
        h.grad = w.grad = b.grad = None
        got = vmlmf_amd.linear_nll(h, w, b, y, chunk_rows=chunk, fused=True)
        got.backward()
        assert abs(float(got) - float(ref)) <= 1e-5 * abs(float(ref))
        for t, g in zip((h, w, b), want):
            assert float((t.grad - g).abs().max()) <= 1e-5 * float(g.abs().max()) + 1e-7
    with torch.no_grad():
        assert abs(float(vmlmf_amd.linear_nll(h, w, b, y)) - float(ref)) <= 1e-5 * abs(float(ref))      # the chunked form
    assert abs(float(vmlmf_amd.linear_nll(h, w, b, y)) - float(ref)) <= 1e-6 * abs(float(ref))          # needs a gradient: unfused


# ---- the LM head for training and the embedding table's gradient (verdict r3 item 5) --------------
def test_lm_network_with_the_fused_head_and_embedding_gradient_vs_reference():
    """The LM loop of lm_test.py:196-209 through Model.loss: projection + loss with the scores' gradient formed in place
    (vmlmf_nll_forward_grad), the two backward GEMMs reading it there, the embedding table's gradient by vmlmf_embed_backward -
    two minibatches with carried states against the reference's own run, once with the package's unit gradient and once with a
    plain loss.backward() (a foreign d(loss) tensor: the scaled path)."""
    import vmlmf_amd
    from vmlmf_amd import Model, optim
    d = load_golden("lm_model_v3")
    V, H, L, B, T, rw, ru = (int(v) for v in d["meta"])
    for plain_backward in (False, True):
        model = Model(V, H, L, 0.0, 0.1, w_rank=rw, u_ranks=[ru], lstm_type="vmlmf")
        model.load_state_dict({k: torch.tensor(v) for k, v in d["init"].items()})
        model = model.to(DEV)
        states = model.state_init(B)
        for i in range(2):
            model.zero_grad()
            states = model.detach(states)
            loss, states = model.loss(torch.tensor(d[f"x{i}"], device=DEV), torch.tensor(d[f"y{i}"], device=DEV), states)
            if plain_backward:
                loss.backward()
            else:
                loss.backward(vmlmf_amd.unit_gradient(DEV))
            assert abs(loss.item() - float(d[f"loss{i}"][0])) < 1e-4 * float(d[f"loss{i}"][0])
            for k, p in model.named_parameters():
                assert_grad(p.grad.cpu().numpy(), d[f"G{i}"][k], f"G{i}.{k}")
            norm = optim.clip_sgd_step(model.parameters(), lr=1.0, max_norm=0.25)
            assert abs(float(norm) - float(d[f"norm{i}"][0])) < 1e-4 * float(d[f"norm{i}"][0])
        sd = model.state_dict()
        for k, v in d["final"].items():
            assert_out(sd[k].cpu().numpy(), v, "final." + k, atol=2e-5, rtol=1e-3)
        assert_out(torch.stack([s[0].detach() for s in states]).cpu().numpy(), d["hT"], "hT")


def test_nll_forward_grad_at_the_vocabulary_width_vs_reference():
    """vmlmf_nll_forward_grad on the reference's own loss fixture (R = 70 rows of 10 000 scores): loss, the in-place gradient
    (sampled columns and the target columns, upstream gradient 3) and the bias gradient = its column sums; the bias is added
    by the kernel (the scores arrive without it)."""
    from vmlmf_amd.functional import lm_head_loss
    d = load_golden("nll_v10000")
    T, B, V, seed = (int(v) for v in d["meta"])
    r = np.random.Generator(np.random.PCG64(seed))
    z = (2.0 * r.standard_normal((T * B, V))).astype(np.float32)
    y = d["y"]
    # scores = h W^T + b with H = V and W = I would be a 400 MB weight: drive the kernel through the C ABI instead
    import ctypes
    from vmlmf_amd import _lib
    lib = _lib.lib()
    rb = np.random.Generator(np.random.PCG64(5))
    bias = rb.standard_normal(V).astype(np.float32)
    zt = torch.tensor(z - bias[None, :], device=DEV)                 # what the projection would hand over: scores without the bias
    bt = torch.tensor(bias, device=DEV)
    yt = torch.tensor(y.reshape(-1), device=DEV)
    R = T * B
    stats = torch.empty(1 + R, device=DEV)
    db = torch.empty(V, device=DEV)
    scratch = torch.empty(lib.vmlmf_nll_grad_scratch_floats(R, V), device=DEV)
    _lib.check(lib.vmlmf_nll_forward_grad(R, V, zt.data_ptr(), bt.data_ptr(), yt.data_ptr(), ctypes.c_float(B / R), stats.data_ptr(),
                                          stats.data_ptr() + 4, db.data_ptr(), scratch.data_ptr(), _lib.raw_stream(torch.device(DEV, 0))))
    torch.cuda.synchronize()
    up = float(d["upstream"][0])
    g = zt.cpu().numpy().astype(np.float64) * up
    # z - bias + bias differs from z by one rounding: the fixture's own tolerance for the stable form (1e-4 relative)
    assert abs(stats[0].item() - float(d["loss"][0])) < 1e-4 * float(d["loss"][0])
    assert_grad(g[:, ::97], d["g_s"], "dscores sample", rel=2e-4)
    assert_grad(g[np.arange(R), y.reshape(-1)], d["g_target"], "dscores at the targets", rel=2e-4)
    assert_grad(db.cpu().numpy(), zt.cpu().numpy().astype(np.float64).sum(0), "dbias = column sums")
    want, gwant = O.nll_loss_stable(z, y)
    assert_grad(g / up, gwant, "dscores vs the stable oracle", rel=2e-4)
    # and through autograd on a small projection: values and gradients against stock ops in fp64
    torch.manual_seed(3)
    Hs, Vs, Ts, Bs = 24, 64, 5, 6
    h = torch.randn(Ts, Bs, Hs, device=DEV, requires_grad=True)
    w = (0.3 * torch.randn(Vs, Hs, device=DEV)).requires_grad_(True)
    b = torch.randn(Vs, device=DEV, requires_grad=True)
    yy = torch.randint(0, Vs, (Ts, Bs), device=DEV)
    loss = lm_head_loss(h, w, b, yy)
    (2.5 * loss).backward()
    hd, wd, bd = (t.detach().double().cpu().requires_grad_(True) for t in (h, w, b))
    ref = -torch.log_softmax(hd.reshape(-1, Hs) @ wd.t() + bd, 1)[torch.arange(Ts * Bs), yy.cpu().reshape(-1)].mean() * Bs
    (2.5 * ref).backward()
    assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item())
    for got, want_, name in ((h.grad, hd.grad, "dh"), (w.grad, wd.grad, "dW"), (b.grad, bd.grad, "db")):
        assert_grad(got.cpu().numpy(), want_.numpy(), "head." + name)


def test_embedding_gradient_is_a_position_ordered_sum_without_atomics():
    """vmlmf_embed_backward: every table row the sum of the dy rows of its positions in ascending order - equal to an fp64
    scatter-add within rounding, identical bits on every run, zeros where no token points (the call fills the whole matrix),
    correct with heavy repetition (one token at a third of the positions) and for widths that are not a multiple of 64."""
    from vmlmf_amd.functional import embedding
    rng = np.random.Generator(np.random.PCG64(11))
    for (V, H, T, B) in ((100, 650, 35, 64), (10000, 650, 35, 16), (37, 16, 5, 4), (300, 129, 9, 33)):
        tok = rng.integers(0, V, size=(T, B))
        tok[rng.random((T, B)) < 0.33] = V // 2          # one very frequent token
        tok[0, 0] = V - 1
        w = torch.tensor(rng.standard_normal((V, H)).astype(np.float32), device=DEV, requires_grad=True)
        dy = torch.tensor(rng.standard_normal((T, B, H)).astype(np.float32), device=DEV)
        tt = torch.tensor(tok, device=DEV)
        grads = []
        for _ in range(2):
            w.grad = None
            x = embedding(w, tt)
            assert torch.equal(x, w.detach()[tt])
            x.backward(dy)
            grads.append(w.grad.clone())
        assert torch.equal(grads[0], grads[1])
        ref = np.zeros((V, H))
        np.add.at(ref, tok.reshape(-1), dy.cpu().numpy().reshape(-1, H).astype(np.float64))
        assert_grad(grads[0].cpu().numpy(), ref, f"embed.dW V{V} H{H}", rel=1e-5)
        unused = np.setdiff1d(np.arange(V), tok.reshape(-1))
        assert unused.size == 0 or not grads[0][torch.tensor(unused, device=DEV)].any()


def test_transpose_is_exact_for_ragged_shapes():
    """vmlmf_transpose (the LM head's dW^T -> dW): bit-equal to torch's strided copy, shapes that are not multiples of the 64 x 64
    tile, the PTB head's own (650, 10000), single rows and columns; the same buffer twice is refused."""
    from vmlmf_amd import _lib
    from vmlmf_amd.functional import transposed
    g = torch.Generator(device=DEV).manual_seed(5)
    for rows, cols in ((650, 10000), (1, 7), (7, 1), (64, 64), (65, 63), (129, 1000), (3, 300)):
        m = torch.randn(rows, cols, device=DEV, generator=g)
        t = transposed(m)
        assert t.shape == (cols, rows) and t.is_contiguous() and torch.equal(t, m.t().contiguous()), (rows, cols)
    m = torch.randn(8, 8, device=DEV)
    assert _lib.lib().vmlmf_transpose(8, 8, m.data_ptr(), m.data_ptr(), None) == _lib.E_BADARG


def test_a_packed_image_made_for_something_else_is_refused():
    """Verdict r3 (hygiene): the geometry header in front of a kept parameter image was written and never checked.  The
    signature now lives on the host (address -> what the image was packed for): a *_packed call with an image of another
    descriptor, an address vmlmf_pack_params never filled, or an image from before a vmlmf_tune() call returns VMLMF_E_BADARG
    and launches nothing; the matching image still works, for any batch / length."""
    import ctypes
    from vmlmf_amd import _lib
    from vmlmf_amd.functional import _params_struct, _ptr
    lib = _lib.lib()
    dev = torch.device("cuda", 0)
    stream = _lib.raw_stream(dev)

    def layer(I, H, rw, ru, B=8, T=5):
        P = O.make_params(O.V1, I, H, rw, ru, seed=H)
        names = ["dia_x", "dia_h", "u_x", "v_x", "b_x", "b_h", "u_h", "v_h"]
        params = [torch.tensor(np.asarray(P[k]), device=DEV) for k in names]
        desc = _lib.make_desc(_lib.V1_CELL, B, T, I, H, rw, [ru], training=False)
        return desc, params, _params_struct(params, 1, _lib.V1_CELL)

    def pack(desc, ps):
        n = ctypes.c_size_t()
        _lib.check(lib.vmlmf_pack_bytes(ctypes.byref(desc), ctypes.byref(n)))
        img = torch.empty(n.value, device=DEV, dtype=torch.uint8)
        _lib.check(lib.vmlmf_pack_params(ctypes.byref(desc), ctypes.byref(ps), _ptr(img), stream))
        return img

    def forward(desc, ps, img, B, T, I, H):
        d2 = _lib.make_desc(_lib.V1_CELL, B, T, I, H, desc.w_rank, [desc.u_ranks[0]], training=False)
        sz = _lib.query(d2)
        x = torch.randn(B, T, I, device=DEV)
        y, hT, cT = torch.empty(B, T, H, device=DEV), torch.empty(B, H, device=DEV), torch.empty(B, H, device=DEV)
        ws = torch.empty(sz.workspace_bytes, device=DEV, dtype=torch.uint8)
        return lib.vmlmf_seq_forward_packed(ctypes.byref(d2), ctypes.byref(ps), _ptr(x), None, None, _ptr(y), _ptr(hT), _ptr(cT), None,
                                            _ptr(ws), sz.workspace_bytes, stream, _ptr(img)), y

    dA, pA, sA = layer(9, 64, 8, 8)
    dB, pB, sB = layer(9, 100, 8, 16)
    imgA, imgB = pack(dA, sA), pack(dB, sB)
    rc, y1 = forward(dA, sA, imgA, 8, 5, 9, 64)
    assert rc == 0
    rc, y2 = forward(dA, sA, imgA, 3, 11, 9, 64)                  # batch and length may differ
    assert rc == 0 and torch.isfinite(y2).all()
    rc, _ = forward(dA, sA, imgB, 8, 5, 9, 64)                    # an image packed for the other layer
    assert rc == _lib.E_BADARG and "another descriptor" in lib.vmlmf_last_error().decode()
    stray = torch.zeros(imgA.numel(), device=DEV, dtype=torch.uint8)
    rc, _ = forward(dA, sA, stray, 8, 5, 9, 64)                   # memory vmlmf_pack_params never filled
    # (refused either way: as memory no image was ever packed into - or, when the allocator hands the test an address where an earlier
    #  test's image lived, as an image packed for another descriptor: the library's registry is keyed by address)
    assert rc == _lib.E_BADARG and any(m in lib.vmlmf_last_error().decode() for m in ("not an image", "another descriptor"))
    _lib.tune("rec3", 6)                                          # any vmlmf_tune() call moves the generation
    rc, _ = forward(dA, sA, imgA, 8, 5, 9, 64)
    assert rc == _lib.E_BADARG
    imgA2 = pack(dA, sA)
    rc, y3 = forward(dA, sA, imgA2, 8, 5, 9, 64)
    assert rc == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("layers,B,T,C,ignored", [([180], 64, 16, 18, 0), ([180], 64, 128, 18, 5), ([24], 1, 3, 18, 0), ([64], 300, 4, 18, 17),
                                                  ([24, 40], 7, 5, 18, 2), ([600], 5, 3, 18, 0)])
def test_criterion_riding_on_the_forward_launch_equals_the_two_lines(layers, B, T, C, ignored):
    """Net.loss(x, target) = criterion(net(x), target) of train.py:61-63 as one call (C ABI 10, vmlmf_ce): on the VALU recurrent
    kernels the row's log-sum-exp, loss term and d(loss)/d(logits) come out of the forward recurrence's epilogue, the mean is the
    last workgroup's fixed-order sum.  Same loss (to the summation order of the mean), same logits bit for bit, same gradients as
    the package's criterion on the logits - with ignored rows, a single row, more rows than CUs, and on the families whose head is a
    launch of its own (two-layer wavefront stack, H = 600 step-wise / clustered layer: there the criterion is a launch too)."""
    import copy
    import vmlmf_amd
    torch.manual_seed(11)
    a = Net(9, layer_sizes=layers, w_rank=8, u_rank=[8], model=MyLSTM, cell=MyVMLMFCell).cuda()
    b = copy.deepcopy(a)
    x = torch.randn(B, T, 9, device=DEV)
    t = torch.randint(0, C, (B,), device=DEV)
    if ignored:
        t[torch.randperm(B, device=DEV)[:min(ignored, B - 1)]] = -100
    for rep in range(2):                                  # (twice: the ticket word must be back at zero)
        a.zero_grad(set_to_none=True)
        la, za = a.loss(x, t, return_logits=True)
        la.backward(vmlmf_amd.unit_gradient(DEV))
    zb = b(x)
    lb = vmlmf_amd.cross_entropy(zb, t)
    lb.backward()
    assert torch.equal(za.reshape(zb.shape), zb)
    assert abs(float(la) - float(lb)) <= 2e-6 * max(1.0, abs(float(lb))), (float(la), float(lb))
    ref = torch.nn.functional.cross_entropy(zb.detach().double().cpu(), t.cpu())
    assert abs(float(la) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if pb.grad is None:
            assert pa.grad is None, k
            continue
        assert_grad(pa.grad.cpu().numpy(), pb.grad.cpu().numpy(), k, rel=2e-5)


def test_criterion_riding_scales_with_a_foreign_loss_gradient_and_runs_without_grad():
    """d(loss) that is not the package's constant one scales the stored gradient; a logits gradient from elsewhere adds to it;
    under torch.no_grad() the call returns the loss of the inference kernels; an out-of-range target poisons the loss."""
    import copy
    import vmlmf_amd
    torch.manual_seed(12)
    a = Net(9, layer_sizes=[64], w_rank=8, u_rank=[8], model=MyLSTM, cell=MyVMLMFCell).cuda()
    b = copy.deepcopy(a)
    x = torch.randn(10, 6, 9, device=DEV)
    t = torch.randint(0, 18, (10,), device=DEV)
    la, za = a.loss(x, t, return_logits=True)
    (3.0 * la + za.square().sum()).backward()
    zb = b(x)
    (3.0 * torch.nn.functional.cross_entropy(zb, t) + zb.square().sum()).backward()
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if pb.grad is not None:
            assert_grad(pa.grad.cpu().numpy(), pb.grad.cpu().numpy(), k, rel=2e-5)
    with torch.no_grad():
        l0 = a.loss(x, t)
    assert abs(float(l0) - float(la)) <= 1e-6
    t[3] = 18
    assert torch.isnan(a.loss(x, t))


def test_criterion_riding_tiers_of_the_fixed_point_mean():
    """The mean of the riding criterion is an integer sum (vmlmf_ce, ce_epilogue): terms below 2048 at 2^-29 (64 rows), terms of 2048
    and more at 2^-10 through the second ticket word, beyond 2^36 / B the loss is +Inf, a NaN term makes it NaN - and the two
    ticket words are back at zero afterwards (the next call is right again)."""
    import vmlmf_amd
    torch.manual_seed(13)
    net = Net(9, layer_sizes=[64], w_rank=8, u_rank=[8], model=MyLSTM, cell=MyVMLMFCell).cuda()
    x = torch.randn(12, 5, 9, device=DEV)
    t = torch.randint(1, 18, (12,), device=DEV)          # never class 0
    with torch.no_grad():
        base = float(net.loss(x, t))
        for bias0, kind in ((5000.0, "coarse"), (1.0e12, "inf"), (float("nan"), "nan"), (0.1, "fine")):
            net.lin.bias[0] = bias0                        # a logit of class 0 that far above the others: every row's term ~ bias0
            got, logits = net.loss(x, t, return_logits=True)
            ref = torch.nn.functional.cross_entropy(logits.double().cpu(), t.cpu())
            if kind == "coarse":
                assert abs(float(got) - float(ref)) <= 2e-6 * float(ref), (float(got), float(ref))
            elif kind == "inf":
                assert torch.isinf(got) and float(got) > 0
            elif kind == "nan":
                assert torch.isnan(got)
            else:
                assert abs(float(got) - base) <= 1e-6, (float(got), base)
    from vmlmf_amd.functional import ce_ticket
    assert int(ce_ticket(x.device).abs().sum()) == 0


@pytest.mark.parametrize("variant,I,H,rw,ru,B,T,tm", [(O.V1, 9, 64, 8, [8], 6, 5, False), (O.V3, 650, 650, 32, [32], 16, 3, True)])
def test_plain_c_abi_calls_behind_a_dirty_stack(variant, I, H, rw, ru, B, T, tm, tmp_path):
    """vmlmf_seq_forward / vmlmf_seq_backward go through the *_packed wrappers, whose stack `vmlmf_extra` left the
    fields added later (ce, drop) unset.  Called behind a function that filled the stack with 0xA5 (tests/shims/dirty_stack.c, gcc),
    on a register-resident layer and on a row-block layer (the family that applies dropout inside its launches), the plain entry
    points must return 0 and the oracle's results."""
    import ctypes
    import subprocess
    from hip_util import ORDER, assert_grad, assert_out, run_literal
    from vmlmf_amd import _lib
    from vmlmf_amd.functional import _params_struct, _ptr
    so = str(tmp_path / "dirty_stack.so")
    subprocess.run(["gcc", "-O1", "-shared", "-fPIC", os.path.join(os.path.dirname(__file__), "shims", "dirty_stack.c"), "-o", so], check=True)
    shim = ctypes.CDLL(so)
    lib = _lib.lib()
    rng = np.random.Generator(np.random.PCG64(11))
    P = O.make_params(variant, I, H, rw, ru[0], seed=2)
    if variant == O.V3:
        P = {k: (np.asarray(v) * 0.3).astype(np.float32) for k, v in P.items()}
    shape = (T, B, I) if tm else (B, T, I)
    x = (rng.standard_normal(shape) * 0.5).astype(np.float32)
    dy = rng.standard_normal(shape[:2] + (H,)).astype(np.float32)
    ref = run_literal(variant, P, x, None, None, dy, time_major=tm)
    params = [torch.tensor(np.asarray(P[k]), device=DEV) for k in ORDER[variant]]
    grads = [torch.empty_like(p) for p in params]
    desc = _lib.make_desc(variant, B, T, I, H, rw, ru, time_major=tm, training=True)
    sz = _lib.query(desc)
    xt, dyt = torch.tensor(x, device=DEV), torch.tensor(dy, device=DEV)
    y, hT, cT = torch.empty(shape[:2] + (H,), device=DEV), torch.empty(B, H, device=DEV), torch.empty(B, H, device=DEV)
    dx = torch.empty_like(xt)
    ws = torch.empty(sz.workspace_bytes, device=DEV, dtype=torch.uint8)
    reserve = torch.empty(sz.reserve_bytes, device=DEV, dtype=torch.uint8)
    ps, gs = _params_struct(params, 1, variant), _params_struct(grads, 1, variant)
    stream = _lib.raw_stream(torch.device("cuda", 0))
    vp = ctypes.c_void_p
    shim.dirty_forward.argtypes = [vp] * 11 + [ctypes.c_size_t, vp]
    shim.dirty_backward.argtypes = [vp] * 16 + [ctypes.c_size_t, vp]
    fn = lambda f: ctypes.cast(f, vp)
    rc = shim.dirty_forward(fn(lib.vmlmf_seq_forward), ctypes.addressof(desc), ctypes.addressof(ps), _ptr(xt), None, None, _ptr(y), _ptr(hT),
                            _ptr(cT), _ptr(reserve), _ptr(ws), sz.workspace_bytes, stream)
    assert rc == 0, lib.vmlmf_last_error().decode()
    rc = shim.dirty_backward(fn(lib.vmlmf_seq_backward), ctypes.addressof(desc), ctypes.addressof(ps), _ptr(xt), None, None, _ptr(y),
                             _ptr(reserve), _ptr(dyt), None, None, _ptr(dx), None, None, ctypes.addressof(gs), _ptr(ws),
                             sz.workspace_bytes, stream)
    assert rc == 0, lib.vmlmf_last_error().decode()
    torch.cuda.synchronize()
    assert_out(y.cpu().numpy(), ref["y"], "dirty.y")
    assert_out(hT.cpu().numpy(), ref["hT"], "dirty.hT")
    assert_grad(dx.cpu().numpy(), ref["dx"], "dirty.dx")
    for k, g_ in zip(ORDER[variant], grads):
        assert_grad(g_.cpu().numpy(), ref["G"][k], f"dirty.{k}")


def _cfg_a_net(d):
    _, B, T, I, H, rw, ru = (int(v) for v in d["meta"])
    torch.manual_seed(0)
    net = Net(I, layer_sizes=[H], w_rank=rw, u_rank=[ru], model=MyLSTM, cell=MyVMLMFCell)
    load_cell(net.rnn.rnncells[0], O.make_params(O.V1, I, H, rw, ru, seed=int(d["seeds"][0])))
    x, tgt = O.synthetic_batch(B, T, I, seed=int(d["seeds"][1]))
    return net.to(DEV), torch.tensor(x, device=DEV), torch.tensor(tgt, device=DEV).long()


@pytest.mark.parametrize("graphed", [False, True])
def test_the_graded_step_itself_vs_reference_golden(graphed):
    """the bench's own step - Net.loss (criterion in the forward launch) -> backward(unit_gradient) ->
    finish2_kernel -> vmlmf_amd.optim.Adam, eager and replayed from GraphedTrainStep's hipGraph - against the reference's own three
    Adam steps at config A (cfgA_net_adam3: V/src/train_test/train.py:47,58-65 run on the imported reference): the three losses,
    the three logit matrices and the final parameters, at the tolerances of test_net_adam_three_steps_vs_reference."""
    import vmlmf_amd
    d = load_golden("cfgA_net_adam3")
    net, x, tgt = _cfg_a_net(d)
    opt = vmlmf_amd.optim.Adam(net.parameters(), lr=0.002)
    if graphed:
        step = vmlmf_amd.GraphedTrainStep(net, vmlmf_amd.cross_entropy, opt, x, tgt, warmup=2)
        assert step._fused_loss
    for it in range(3):
        if graphed:
            loss = float(step(x, tgt))        # (the replay has applied the update already: the logits are checked on the eager leg)
        else:
            opt.zero_grad(set_to_none=True)
            l, logits = net.loss(x, tgt, return_logits=True)
            l.backward(vmlmf_amd.unit_gradient(l.device))
            opt.step()
            loss = float(l)
            assert_out(logits.detach().cpu().numpy(), d["logits"][it], f"logits[{it}]", atol=2e-5)
        assert abs(loss - float(d["losses"][it])) < 2e-5, (graphed, it, loss, d["losses"][it])
    assert all(p.grad is None for p in net.cell.parameters())
    sd = net.state_dict()
    for k, v in d["final"].items():
        assert_out(sd[k].cpu().numpy(), v, "final." + k, atol=2e-5, rtol=1e-3)


@pytest.mark.parametrize("variant,cls,ru,kw", [(O.V1, MyVMLMFCell, [6], {}), (O.V2, MyVMLMFCellg2, [4, 6], {"g": 2})])
def test_mylstm_time_major_module_vs_oracle(variant, cls, ru, kw):
    """MyLSTM(batch_first=False) (V/src/models/vmlmf.py:255,273-276: x is (T, B, I), y (T, B, H)) through the MODULE for V1 and V2,
    two layers, against literal_sequence(time_major=True) chained layer by layer in fp64."""
    from hip_util import ORDER
    B, T, I, H, rw = 5, 7, 12, 24, 5
    rng = np.random.Generator(np.random.PCG64(21))
    rnn = MyLSTM(I, hidden_layer_sizes=[H, H], batch_first=False, w_rank=rw, u_ranks=ru, cell=cls, **kw)
    Ps = [O.make_params(variant, I if l == 0 else H, H, rw, ru if len(ru) > 1 else ru[0], seed=30 + l) for l in range(2)]
    for l in range(2):
        load_cell(rnn.rnncells[l], Ps[l])
    rnn = rnn.to(DEV)
    x = rng.standard_normal((T, B, I)).astype(np.float32)
    dy = rng.standard_normal((T, B, H)).astype(np.float32)
    dh = rng.standard_normal((B, 2 * H)).astype(np.float32)
    xg = torch.tensor(x, device=DEV, requires_grad=True)
    y, hcat = rnn(xg)
    assert y.shape == (T, B, H) and hcat.shape == (B, 2 * H)
    ((y * torch.tensor(dy, device=DEV)).sum() + (hcat * torch.tensor(dh, device=DEV)).sum()).backward()
    Pt = [O.to_torch(P, dtype=torch.float64, requires_grad=True) for P in Ps]
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    cur, hs = xt, []
    for l in range(2):
        cur, h, _ = O.literal_sequence(variant, Pt[l], cur, None, None, time_major=True)
        hs.append(h)
    ((cur * torch.tensor(dy, dtype=torch.float64)).sum() + (torch.cat(hs, -1) * torch.tensor(dh, dtype=torch.float64)).sum()).backward()
    assert_out(y.detach().cpu().numpy(), cur.detach().numpy(), "tm.y")
    assert_out(hcat.detach().cpu().numpy(), torch.cat(hs, -1).detach().numpy(), "tm.hcat")
    assert_grad(xg.grad.cpu().numpy(), xt.grad.numpy(), "tm.dx")
    for l in range(2):
        cell = rnn.rnncells[l]
        for k in ORDER[variant]:
            got = (cell.layers[k] if hasattr(cell, "layers") else getattr(cell, k)).grad
            assert_grad(got.cpu().numpy(), Pt[l][k].grad.numpy(), f"tm.layer{l}.{k}")
