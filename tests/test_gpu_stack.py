"""GPU: stacked layers in one wavefront launch per direction (C ABI vmlmf_stack_*, vmlmf_wave.inc) against the chained
per-layer calls of the same library (VMLMF_STACK=0) and against the fp64 oracle.  The reference's layer loop:
V/src/models/vmlmf.py:300-314."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(model, x, mode, head_grad=None):
    os.environ["VMLMF_STACK"] = mode
    try:
        for p in model.parameters():
            p.grad = None
        xx = x.clone().requires_grad_(True)
        y, hid = model(xx)
        loss = (y * head_grad[0]).sum() + (hid * head_grad[1]).sum()
        loss.backward()
        torch.cuda.synchronize()
        return (y.detach().clone(), hid.detach().clone(), xx.grad.clone(),
                {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    finally:
        os.environ.pop("VMLMF_STACK", None)


CASES = [
    # L, B, T, I, H, rank, cell
    (2, 8, 5, 12, 40, 8, "vm"),
    (2, 5, 7, 40, 40, 16, "vm"),
    (3, 16, 9, 20, 100, 16, "vm"),
    (2, 128, 24, 77, 256, 24, "vm"),      # BASELINE configs[2] (fp32)
    (2, 19, 11, 9, 180, 16, "vm"),
    (4, 3, 6, 30, 64, 24, "vm"),
    (2, 7, 8, 33, 130, 32, "vm"),
    (1, 9, 6, 77, 180, 8, "vm"),
    (2, 6, 5, 24, 72, 16, "lmf"),         # MyLSTMCell in low-rank mode (variant 5)
    (2, 300, 4, 10, 64, 8, "vm"),         # more workgroups than CUs
    (2, 6, 7, 20, 64, (8, 16), "vm"),     # padded w_rank != padded u_rank: both sides at the wider one
    (3, 9, 5, 33, 130, (24, 5), "vm"),
    (2, 128, 24, 77, 256, (16, 24), "vm"),
    (1, 81, 24, 77, 180, (8, [2, 4]), "g2"),      # the reference's vmlmf_group2 demo (script/demo.sh): two groups, ranks 2 and 4
    (2, 7, 6, 12, 64, (8, [4, 6]), "g2"),
    (3, 5, 9, 40, 128, (16, [8, 9]), "g2"),
    (2, 9, 5, 30, 200, (12, [8, 8]), "g2"),       # two waves per group
    (2, 6, 7, 20, 96, (8, [5, 3]), "g2novm"),     # the group cell without the vector multiplication (variant 6)
    (3, 4, 1, 12, 40, 8, "vm"),           # one and two time steps: the pipelines of the x-team and of the hand-over barely start
    (2, 6, 2, 30, 64, 16, "vm"),
    (2, 520, 3, 16, 128, 16, "vm"),
]


@pytest.mark.parametrize("L,B,T,I,H,r,kind", CASES)
def test_stack_matches_chained_layers(L, B, T, I, H, r, kind):
    import vmlmf_amd
    from vmlmf_amd import functional as F
    torch.manual_seed(1234 + L * 7 + B)
    rw, ru = r if isinstance(r, tuple) else (r, r)
    cell = {"vm": vmlmf_amd.MyVMLMFCell, "lmf": vmlmf_amd.MyLSTMCell, "g2": vmlmf_amd.MyVMLMFCellg2,
            "g2novm": vmlmf_amd.MyVMLMFgCellg2}[kind]
    model = vmlmf_amd.MyLSTM(I, hidden_layer_sizes=[H] * L, batch_first=True, w_rank=rw, u_ranks=ru, cell=cell).cuda()
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.5)       # livelier gates than the reference's init
    x = torch.randn(B, T, I, device="cuda")
    gy = torch.randn(B, T, H, device="cuda")
    gh = torch.randn(B, L * H, device="cuda")
    ref = _run(model, x, "0", (gy, gh))
    got = _run(model, x, "1", (gy, gh))
    # the stack really ran on the wavefront launch
    cfg = model.rnncells[0].kernel_cfg()
    ur = tuple(ru) if isinstance(ru, list) else (ru,)
    assert F._stack_plan((cfg["variant"], cfg["g"], rw, ur, False, 0), L, B, T, I, H, True) is not None
    for a, b, what in ((got[0], ref[0], "y"), (got[1], ref[1], "hidden"), (got[2], ref[2], "dx")):
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, scale), what
    for name, gref in ref[3].items():
        scale = float(gref.abs().max()) + 1e-12
        err = float((got[3][name] - gref).abs().max())
        assert err <= 1e-4 * scale + 1e-6, (name, err, scale)


UNEQUAL = [
    # B, T, I, hidden_layer_sizes, rank, cell      (a VMLMF cell needs input_size <= hidden_size, vmlmf.py:94: sizes cannot shrink)
    (8, 5, 12, [40, 64], 8, "vm"),
    (128, 24, 77, [128, 256], 24, "vm"),           # BASELINE configs[2]'s shape with a narrower bottom layer
    (5, 7, 20, [64, 100, 180], 16, "vm"),          # three sizes, the middle one no multiple of 64
    (9, 6, 24, [72, 130], 16, "lmf"),
    (33, 9, 30, [64, 64, 192], 32, "vm"),
    (6, 1, 10, [24, 200], 8, "vm"),
    (300, 3, 16, [128, 129], 16, "vm"),
]


@pytest.mark.parametrize("B,T,I,Hs,r,kind", UNEQUAL)
def test_stack_of_layers_of_different_hidden_sizes(B, T, I, Hs, r, kind):
    """MyLSTM builds any hidden_layer_sizes (vmlmf.py:283-292: layer i reads hidden_layer_sizes[i - 1]); round 6: such a stack rides
    the wavefront launches too, every layer on the widest layer's thread-slot geometry - against the chained per-layer calls."""
    import vmlmf_amd
    from vmlmf_amd import functional as F
    torch.manual_seed(77 + B + len(Hs))
    L = len(Hs)
    cell = {"vm": vmlmf_amd.MyVMLMFCell, "lmf": vmlmf_amd.MyLSTMCell}[kind]
    model = vmlmf_amd.MyLSTM(I, hidden_layer_sizes=Hs, batch_first=True, w_rank=r, u_ranks=r, cell=cell).cuda()
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.5)
    x = torch.randn(B, T, I, device="cuda")
    gy = torch.randn(B, T, Hs[-1], device="cuda")
    gh = torch.randn(B, sum(Hs), device="cuda")
    ref = _run(model, x, "0", (gy, gh))
    got = _run(model, x, "1", (gy, gh))
    cfg = model.rnncells[0].kernel_cfg()
    assert F._stack_plan((cfg["variant"], cfg["g"], r, (r,), False, 0), L, B, T, I, tuple(Hs), True) is not None
    for a, b, what in ((got[0], ref[0], "y"), (got[1], ref[1], "hidden"), (got[2], ref[2], "dx")):
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, scale), what
    assert set(got[3]) == set(ref[3])
    for name, gref in ref[3].items():
        scale = float(gref.abs().max()) + 1e-12
        err = float((got[3][name] - gref).abs().max())
        assert err <= 1e-4 * scale + 1e-6, (name, err, scale)


def test_unequal_stack_with_the_classifier_against_the_fp64_oracle():
    """Net(layer_sizes=[128, 256]) (vmlmf.py:330-355: Linear(256, 18) on the top layer's last step) through the default launch
    policy against the literal fp64 restatement chained layer by layer: logits, dx, every gradient of both layers and the head."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.join(os.path.dirname(here), "oracle"), here]
    import vmlmf_oracle as O
    import vmlmf_amd
    from hip_util import ORDER, assert_grad, assert_out
    B, T, I, Hs, r = 64, 16, 77, [128, 256], 24
    torch.manual_seed(9)
    net = vmlmf_amd.Net(I, layer_sizes=Hs, w_rank=r, u_rank=[r], model=vmlmf_amd.MyLSTM, cell=vmlmf_amd.MyVMLMFCell).cuda()
    with torch.no_grad():
        net.lin.weight.mul_(10.0)
    rng = np.random.Generator(np.random.PCG64(5))
    x = rng.standard_normal((B, T, I)).astype(np.float32)
    tgt = rng.integers(0, 18, B)
    xg = torch.tensor(x, device="cuda").requires_grad_(True)
    logits = net(xg)
    vmlmf_amd.cross_entropy(logits, torch.tensor(tgt, device="cuda")).backward()
    torch.cuda.synchronize()
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    cur, Pts = xt, []
    for cell in net.rnn.rnncells:
        named = dict(cell.named_parameters())
        Pt = O.to_torch({k: named[k].detach().cpu().numpy() for k in ORDER[O.V1]}, dtype=torch.float64, requires_grad=True)
        Pts.append((named, Pt))
        cur, _, _ = O.literal_sequence(O.V1, Pt, cur, None, None, time_major=False)
    W = net.lin.weight.detach().cpu().double().requires_grad_(True)
    b = net.lin.bias.detach().cpu().double().requires_grad_(True)
    lr = cur[:, -1] @ W.t() + b
    torch.nn.functional.cross_entropy(lr, torch.tensor(tgt)).backward()
    assert_out(logits.detach().cpu().numpy(), lr.detach().numpy(), "logits")
    assert_grad(xg.grad.cpu().numpy(), xt.grad.numpy(), "dx")
    assert_grad(net.lin.weight.grad.cpu().numpy(), W.grad.numpy(), "lin.weight")
    assert_grad(net.lin.bias.grad.cpu().numpy(), b.grad.numpy(), "lin.bias")
    for l, (named, Pt) in enumerate(Pts):
        for k in ORDER[O.V1]:
            assert_grad(named[k].grad.cpu().numpy(), Pt[k].grad.numpy(), "layer %d %s" % (l, k))


def test_stack_inference_matches_training_forward():
    import vmlmf_amd
    torch.manual_seed(5)
    model = vmlmf_amd.MyLSTM(20, hidden_layer_sizes=[96, 96], batch_first=True, w_rank=16, u_ranks=16,
                             cell=vmlmf_amd.MyVMLMFCell).cuda()
    x = torch.randn(12, 10, 20, device="cuda")
    os.environ["VMLMF_STACK"] = "1"
    try:
        y1, h1 = model(x)
        with torch.no_grad():
            y2, h2 = model(x)
        torch.cuda.synchronize()
    finally:
        os.environ.pop("VMLMF_STACK", None)
    assert torch.equal(y1.detach(), y2) and torch.equal(h1.detach(), h2)


def test_stack_repeats_bit_identically_under_load():
    """The layer hand-over (progress words) must not depend on timing: the same launch repeated while a second stream keeps
    the chip busy gives the same bits."""
    import vmlmf_amd
    torch.manual_seed(9)
    model = vmlmf_amd.MyLSTM(77, hidden_layer_sizes=[256, 256], batch_first=True, w_rank=24, u_ranks=24,
                             cell=vmlmf_amd.MyVMLMFCell).cuda()
    x = torch.randn(128, 24, 77, device="cuda")
    os.environ["VMLMF_STACK"] = "1"
    try:
        with torch.no_grad():
            y0, h0 = model(x)
            side = torch.cuda.Stream()
            a = torch.randn(4096, 4096, device="cuda")
            for i in range(6):
                with torch.cuda.stream(side):
                    for _ in range(3):
                        a = (a @ a).clamp_(-1, 1)
                y, h = model(x)
                assert torch.equal(y, y0) and torch.equal(h, h0), i
        torch.cuda.synchronize()
    finally:
        os.environ.pop("VMLMF_STACK", None)


def test_stack_backward_twice_over_one_graph():
    """retain_graph: the second backward finds the progress words of the backward launch cleared again."""
    import vmlmf_amd
    torch.manual_seed(3)
    model = vmlmf_amd.MyLSTM(12, hidden_layer_sizes=[64, 64, 64], batch_first=True, w_rank=16, u_ranks=16,
                             cell=vmlmf_amd.MyVMLMFCell).cuda()
    x = torch.randn(9, 13, 12, device="cuda", requires_grad=True)
    os.environ["VMLMF_STACK"] = "1"
    try:
        y, hid = model(x)
        loss = y.square().sum() + hid.sum()
        g1 = torch.autograd.grad(loss, [x] + list(model.parameters()), retain_graph=True)
        g2 = torch.autograd.grad(loss, [x] + list(model.parameters()))
        torch.cuda.synchronize()
    finally:
        os.environ.pop("VMLMF_STACK", None)
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)


def test_stack_through_both_bindings():
    """vmlmf::stack of the C++ binding and VmlmfStackFn over ctypes call the same C ABI: same bits (fresh interpreters)."""
    import subprocess
    import sys
    from vmlmf_amd import functional as F
    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import sys; sys.path[:0] = [%r]\n"
        "import torch, vmlmf_amd\n"
        "from vmlmf_amd import MyLSTM, MyVMLMFCell, functional as F\n"
        "torch.manual_seed(3)\n"
        "m = MyLSTM(20, hidden_layer_sizes=[72, 72, 72], batch_first=True, w_rank=16, u_ranks=16, cell=MyVMLMFCell).cuda()\n"
        "x = torch.randn(6, 7, 20, device='cuda', requires_grad=True)\n"
        "y, h = m(x); (y.square().sum() + h.sum()).backward()\n"
        "print('binding', 'cpp' if F.torch_ops() is not None else 'ctypes')\n"
        "print('vals', repr(float(y.double().sum())), repr(float(x.grad.double().sum())), repr(sum(float(p.grad.double().sum()) for p in m.parameters())))\n"
    ) % (os.path.dirname(here),)
    outs = {}
    for mode in ("cpp", "ctypes"):
        env = dict(os.environ)
        env["VMLMF_STACK"] = "1"
        if mode == "ctypes":
            env["VMLMF_PYBIND"] = "ctypes"
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = dict(l.split(" ", 1) for l in r.stdout.strip().splitlines() if " " in l)
    assert outs["ctypes"]["binding"] == "ctypes"
    if F.torch_ops() is not None:
        assert outs["cpp"]["binding"] == "cpp"
        assert outs["cpp"]["vals"] == outs["ctypes"]["vals"]


def test_stack_time_major_lm_layers():
    """Stacks of LM layers (MyVMLSTM, time-major, vmlmf_lm.py:437-439) through vmlmf_stack against the chained layer calls."""
    import vmlmf_amd
    from vmlmf_amd import _lib, functional as F
    torch.manual_seed(21)
    T, B, H, r, L = 11, 10, 96, 16, 3
    layers = [vmlmf_amd.MyVMLSTM(H, H, w_rank=r, u_ranks=r).cuda() for _ in range(L)]
    for l in layers:
        for p in l.parameters():
            torch.nn.init.uniform_(p, -0.3, 0.3)
    x = (0.5 * torch.randn(T, B, H, device="cuda")).requires_grad_(True)
    gy = torch.randn(T, B, H, device="cuda")
    zeros = (torch.zeros(B, H, device="cuda"), torch.zeros(B, H, device="cuda"))
    h = x
    for l in layers:
        h, _ = l(h, zeros)
    (h * gy).sum().backward()
    ref_y, ref_dx = h.detach().clone(), x.grad.clone()
    ref_g = [p.grad.clone() for l in layers for p in l.parameters()]
    x.grad = None
    for l in layers:
        l.zero_grad(set_to_none=True)
    os.environ["VMLMF_STACK"] = "1"
    try:
        out = F.vmlmf_stack(_lib.V3_LM, x, [l.kernel_params() for l in layers], r, [r], g=1, time_major=True)
    finally:
        os.environ.pop("VMLMF_STACK", None)
    assert out is not None
    y = out[0]
    (y * gy).sum().backward()
    torch.cuda.synchronize()
    assert float((y.detach() - ref_y).abs().max()) <= 2e-5 * max(1.0, float(ref_y.abs().max()))
    assert float((x.grad - ref_dx).abs().max()) <= 1e-4 * float(ref_dx.abs().max()) + 1e-6
    for got, ref in zip([p.grad for l in layers for p in l.parameters()], ref_g):
        assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-6


@pytest.mark.parametrize("L,B,T,I,H,r", [(2, 128, 24, 77, 256, 24), (2, 9, 7, 12, 40, 8), (3, 5, 4, 30, 100, 16),
                                         (2, 11, 6, 20, 180, (8, [4, 6]))])
def test_net_classifier_rides_on_the_stack(L, B, T, I, H, r):
    """Net (MyLSTM + Linear(H, 18) on the last time step, vmlmf.py:330-355) with the wavefront launches: the logits come out of
    the forward launch, the classifier's backward is part of the stack's backward - against the chained per-layer form."""
    import vmlmf_amd
    from vmlmf_amd import Net, MyLSTM, MyVMLMFCell
    torch.manual_seed(100 + L + B)
    if isinstance(r, tuple):    # group cell: (w_rank, [u_rank of shift 0, of shift 1])
        net = Net(I, layer_sizes=[H] * L, w_rank=r[0], u_rank=r[1], model=MyLSTM, cell=vmlmf_amd.MyVMLMFCellg2).cuda()
    else:
        net = Net(I, layer_sizes=[H] * L, w_rank=r, u_rank=[r], model=MyLSTM, cell=MyVMLMFCell).cuda()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(1.5)
        net.lin.weight.mul_(20.0)
    x = torch.randn(B, T, I, device="cuda")
    tgt = torch.randint(0, 18, (B,), device="cuda")
    res = {}
    for mode in ("0", "1"):
        os.environ["VMLMF_STACK"] = mode
        try:
            net.zero_grad(set_to_none=True)
            xx = x.clone().requires_grad_(True)
            logits = net(xx)
            loss = vmlmf_amd.cross_entropy(logits, tgt)
            loss.backward()
            torch.cuda.synchronize()
            res[mode] = (logits.detach().clone(), xx.grad.clone(),
                         {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None})
        finally:
            os.environ.pop("VMLMF_STACK", None)
    a, b = res["1"], res["0"]
    assert float((a[0] - b[0]).abs().max()) <= 2e-5 * max(1.0, float(b[0].abs().max()))
    assert float((a[1] - b[1]).abs().max()) <= 1e-4 * float(b[1].abs().max()) + 1e-7
    assert set(a[2]) == set(b[2])
    for name, gref in b[2].items():
        assert float((a[2][name] - gref).abs().max()) <= 1e-4 * float(gref.abs().max()) + 1e-7, name


def test_stack_random_shapes_sweep():
    """Thirty seeded random stacks (layers, batch, length, input width, hidden size, rank) through the wavefront launches
    against the chained per-layer kernels: outputs, input gradient and every parameter gradient."""
    import random
    import vmlmf_amd
    rng = random.Random(20261002)
    for trial in range(30):
        L = rng.choice([1, 2, 2, 3, 4])
        r = rng.choice([8, 16, 24, 32])
        hmax = 192 if r == 32 else 256
        H = rng.randint(max(r, 12), hmax)
        I = rng.randint(1, H)
        B = rng.choice([1, 2, 3, 7, 8, 9, 17, 33, 70])
        T = rng.randint(1, 20)
        ru = rng.randint(max(1, r - 7), r)          # true ranks anywhere inside the padded width ...
        rw = rng.randint(max(1, r - 7), r)
        if trial % 3 == 2:                          # ... and every third stack with a narrower x or h side
            if trial % 2:
                rw = rng.randint(1, r)
            else:
                ru = rng.randint(1, r)
        torch.manual_seed(trial)
        model = vmlmf_amd.MyLSTM(I, hidden_layer_sizes=[H] * L, batch_first=True, w_rank=rw, u_ranks=ru,
                                 cell=vmlmf_amd.MyVMLMFCell).cuda()
        with torch.no_grad():
            for p in model.parameters():
                p.mul_(1.3)
        x = torch.randn(B, T, I, device="cuda")
        gy = torch.randn(B, T, H, device="cuda")
        gh = torch.randn(B, L * H, device="cuda")
        ref = _run(model, x, "0", (gy, gh))
        got = _run(model, x, "1", (gy, gh))
        what = f"trial {trial}: L={L} B={B} T={T} I={I} H={H} rw={rw} ru={ru}"
        for a, b in ((got[0], ref[0]), (got[1], ref[1])):
            assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max())), what
        assert float((got[2] - ref[2]).abs().max()) <= 1e-4 * float(ref[2].abs().max()) + 1e-6, what
        for name, gref in ref[3].items():
            assert float((got[3][name] - gref).abs().max()) <= 1e-4 * float(gref.abs().max()) + 1e-6, (what, name)


def test_stack_random_group_cells_sweep():
    """Twenty seeded random stacks of the group cells (MyVMLMFCellg2 / MyVMLMFgCellg2): wavefront against chained kernels."""
    import random
    import vmlmf_amd
    rng = random.Random(777)
    for trial in range(20):
        L = rng.choice([1, 2, 2, 3])
        H = 2 * rng.randint(6, 128)
        rw = rng.randint(1, 16)
        ru = [rng.randint(1, 8), rng.randint(1, 8)] if H > 128 else [rng.randint(1, 16), rng.randint(1, 8)]
        I = rng.randint(1, H)
        B = rng.choice([1, 3, 8, 9, 20, 41])
        T = rng.randint(1, 14)
        cell = vmlmf_amd.MyVMLMFCellg2 if trial % 3 else vmlmf_amd.MyVMLMFgCellg2
        torch.manual_seed(trial)
        model = vmlmf_amd.MyLSTM(I, hidden_layer_sizes=[H] * L, batch_first=True, w_rank=rw, u_ranks=ru, cell=cell).cuda()
        with torch.no_grad():
            for p in model.parameters():
                p.mul_(1.3)
        x = torch.randn(B, T, I, device="cuda")
        gy = torch.randn(B, T, H, device="cuda")
        gh = torch.randn(B, L * H, device="cuda")
        ref = _run(model, x, "0", (gy, gh))
        got = _run(model, x, "1", (gy, gh))
        what = f"trial {trial}: {cell.__name__} L={L} B={B} T={T} I={I} H={H} rw={rw} ru={ru}"
        for a, b in ((got[0], ref[0]), (got[1], ref[1])):
            assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max())), what
        assert float((got[2] - ref[2]).abs().max()) <= 1e-4 * float(ref[2].abs().max()) + 1e-6, what
        for name, gref in ref[3].items():
            assert float((got[3][name] - gref).abs().max()) <= 1e-4 * float(gref.abs().max()) + 1e-6, (what, name)


def test_stack_backward_switch_chains_the_per_layer_kernels():
    """VMLMF_WF_BWD=0 (read when the library loads): the stack's backward as the chained per-layer kernels behind the same
    vmlmf_stack_backward call - same gradients as the wavefront backward, in a fresh interpreter each."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import sys; sys.path[:0] = [%r]\n"
        "import torch, vmlmf_amd\n"
        "torch.manual_seed(11)\n"
        "m = vmlmf_amd.MyLSTM(33, hidden_layer_sizes=[130, 130], batch_first=True, w_rank=24, u_ranks=24, cell=vmlmf_amd.MyVMLMFCell).cuda()\n"
        "x = torch.randn(10, 9, 33, device='cuda', requires_grad=True)\n"
        "y, h = m(x); (y.square().sum() + h.sum()).backward()\n"
        "print('vals', ' '.join(repr(float(v)) for v in [y.double().sum(), x.grad.double().abs().sum()] + [p.grad.double().abs().sum() for p in m.parameters()]))\n"
    ) % (os.path.dirname(here),)
    vals = {}
    for sw in ("1", "0"):
        env = dict(os.environ)
        env["VMLMF_STACK"] = "1"
        env["VMLMF_WF_BWD"] = sw
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("vals ")][-1]
        vals[sw] = [float(v) for v in line.split()[1:]]
    for a, b in zip(vals["1"], vals["0"]):
        assert abs(a - b) <= 1e-4 * abs(b) + 1e-6


def test_one_finishing_launch_against_the_two_it_replaces(tmp_path):
    """finish_units_stack_kernel (round 6: a workgroup per hidden unit sums that unit's partial sums once - every accumulator's blocks in
    reduce_cg_stack_kernel's order - and finishes its gradient entries with finish_body) against reduce_cg_stack_kernel + finish_stack_kernel
    (VMLMF_FINISH_UNITS=0, read when the library loads): the block sums run in the same order and the finishing arithmetic is the same
    function - every gradient, the classifier's too, bit for bit.  Three stacks: config C's shape through Net, layers of different sizes, the per-gate layout of MyLSTMCell."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import sys; sys.path[:0] = [%r]\n"
        "import torch, vmlmf_amd\n"
        "out = []\n"
        "torch.manual_seed(5)\n"
        "net = vmlmf_amd.Net(77, layer_sizes=[256, 256], w_rank=24, u_rank=[24], model=vmlmf_amd.MyLSTM, cell=vmlmf_amd.MyVMLMFCell).cuda()\n"
        "x = torch.randn(128, 24, 77, device='cuda'); t = torch.randint(0, 18, (128,), device='cuda')\n"
        "vmlmf_amd.cross_entropy(net(x), t).backward(); out += [p.grad.cpu() for p in net.parameters() if p.grad is not None]\n"
        "m = vmlmf_amd.MyLSTM(20, hidden_layer_sizes=[64, 100, 180], batch_first=True, w_rank=16, u_ranks=16, cell=vmlmf_amd.MyVMLMFCell).cuda()\n"
        "x = torch.randn(6, 7, 20, device='cuda', requires_grad=True)\n"
        "y, h = m(x); (y.square().sum() + h.sum()).backward(); out += [p.grad.cpu() for p in m.parameters()] + [x.grad.cpu()]\n"
        "m = vmlmf_amd.MyLSTM(24, hidden_layer_sizes=[72, 72], batch_first=True, w_rank=16, u_ranks=16, cell=vmlmf_amd.MyLSTMCell).cuda()\n"
        "x = torch.randn(8, 5, 24, device='cuda', requires_grad=True)\n"
        "y, h = m(x); (y.square().sum() + h.sum()).backward(); out += [p.grad.cpu() for p in m.parameters() if p.grad is not None] + [x.grad.cpu()]\n"
        "torch.save(out, sys.argv[1])\n"
    ) % (os.path.dirname(here),)
    got = {}
    for sw in ("1", "0"):
        env = dict(os.environ)
        env["VMLMF_STACK"] = "1"
        env["VMLMF_FINISH_UNITS"] = sw
        path = str(tmp_path / ("grads%s.pt" % sw))
        r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got[sw] = torch.load(path)
    assert len(got["1"]) == len(got["0"]) and len(got["1"]) > 30
    for i, (a, b) in enumerate(zip(got["1"], got["0"])):
        assert a.shape == b.shape and torch.isfinite(a).all()
        assert torch.equal(a, b), i


@pytest.mark.parametrize("variant_name,time_major", [("V1", False), ("V3", True), ("V5", False), ("V2", False), ("V6", False)])
def test_stack_with_initial_states_against_the_fp64_oracle(variant_name, time_major):
    """The stack entry points with everything the per-layer calls take - initial states of every layer, gradients into
    every layer's final states, gradients of the initial states - against the literal fp64 restatement of the reference
    chained layer by layer (vmlmf.py:300-314 / vmlmf_lm.py:437-439)."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.join(os.path.dirname(here), "oracle"), here]
    import vmlmf_oracle as O
    from hip_util import ORDER, assert_grad, assert_out
    from vmlmf_amd import functional as F
    variant = getattr(O, variant_name)
    L, B, T, I, H, rw, ru = 3, 6, 7, (48 if variant_name == "V3" else 20), 48, 13, 10
    grouped = variant_name in ("V2", "V6")   # the group cells (vmlmf_group.py:85-155 / 158-251): two groups, ranks per shift
    if grouped:
        ru = [6, 5]
    rng = np.random.Generator(np.random.PCG64(77))
    Ps = [O.make_params(variant, I if l == 0 else H, H, rw, ru, seed=3 + l) for l in range(L)]
    shp = (T, B, I) if time_major else (B, T, I)
    x = rng.standard_normal(shp).astype(np.float32)
    h0 = (0.5 * rng.standard_normal((L, B, H))).astype(np.float32)
    c0 = (0.5 * rng.standard_normal((L, B, H))).astype(np.float32)
    dy = rng.standard_normal(shp[:2] + (H,)).astype(np.float32)
    dhT = rng.standard_normal((L, B, H)).astype(np.float32)
    dcT = rng.standard_normal((L, B, H)).astype(np.float32)
    # ---- oracle, fp64
    Pt = [O.to_torch(P, dtype=torch.float64, requires_grad=True) for P in Ps]
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    h0t = torch.tensor(h0, dtype=torch.float64, requires_grad=True)
    c0t = torch.tensor(c0, dtype=torch.float64, requires_grad=True)
    cur, loss = xt, 0.0
    hTs, cTs = [], []
    for l in range(L):
        cur, hT, cT = O.literal_sequence(variant, Pt[l], cur, h0t[l], c0t[l], time_major=time_major)
        hTs.append(hT), cTs.append(cT)
        loss = loss + (hT * torch.tensor(dhT[l], dtype=torch.float64)).sum() + (cT * torch.tensor(dcT[l], dtype=torch.float64)).sum()
    loss = loss + (cur * torch.tensor(dy, dtype=torch.float64)).sum()
    loss.backward()
    # ---- HIP, through vmlmf_stack
    names = ORDER[variant]
    params = [[torch.tensor(np.asarray(P[k]), dtype=torch.float32, device="cuda").requires_grad_(True) for k in names] for P in Ps]
    xg = torch.tensor(x, device="cuda").requires_grad_(True)
    h0g = torch.tensor(h0, device="cuda").requires_grad_(True)
    c0g = torch.tensor(c0, device="cuda").requires_grad_(True)
    os.environ["VMLMF_STACK"] = "1"
    try:
        out = F.vmlmf_stack(variant, xg, params, rw, ru if grouped else [ru], g=2 if grouped else 1, time_major=time_major, h0=h0g, c0=c0g)
    finally:
        os.environ.pop("VMLMF_STACK", None)
    assert out is not None
    y, hs, cs = out
    lossg = (y * torch.tensor(dy, device="cuda")).sum()
    for l in range(L):
        lossg = lossg + (hs[l] * torch.tensor(dhT[l], device="cuda")).sum() + (cs[l] * torch.tensor(dcT[l], device="cuda")).sum()
    lossg.backward()
    torch.cuda.synchronize()
    assert_out(y.detach().cpu().numpy(), cur.detach().numpy(), "y")
    for l in range(L):
        assert_out(hs[l].detach().cpu().numpy(), hTs[l].detach().numpy(), f"hT[{l}]")
        assert_out(cs[l].detach().cpu().numpy(), cTs[l].detach().numpy(), f"cT[{l}]")
    assert_grad(xg.grad.cpu().numpy(), xt.grad.numpy(), "dx")
    assert_grad(h0g.grad.cpu().numpy(), h0t.grad.numpy(), "dh0")
    assert_grad(c0g.grad.cpu().numpy(), c0t.grad.numpy(), "dc0")
    for l in range(L):
        for k, p in zip(names, params[l]):
            assert_grad(p.grad.cpu().numpy(), Pt[l][k].grad.numpy(), f"layer {l} {k}")


@pytest.mark.parametrize("cell_name,variant_name", [("MyVMLMFCellg2", "V2"), ("MyVMLMFCell", "V1")])
def test_reference_demo_shapes_through_mylstm_on_the_default_path_against_the_fp64_oracle(cell_name, variant_name):
    """script/demo.sh's two models (1 x 180, w_rank 8, u_ranks [2, 4] for the group cell / 6 for the plain one, batch 81, 24 steps,
    77 inputs) through MyLSTM with the default launch policy (VMLMF_STACK unset = auto: a single layer with a wide input takes
    the wavefront launch) against the literal fp64 restatement of vmlmf_group.py:85-155 / vmlmf.py:78-125 under the loop of
    vmlmf.py:300-314: outputs, input gradient, every parameter gradient."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.join(os.path.dirname(here), "oracle"), here]
    import vmlmf_oracle as O
    import vmlmf_amd
    from hip_util import ORDER, assert_grad, assert_out
    from vmlmf_amd import functional as F
    variant = getattr(O, variant_name)
    B, T, I, H, rw = 81, 24, 77, 180, 8
    ru = [2, 4] if variant_name == "V2" else 6
    assert os.environ.get("VMLMF_STACK") is None
    torch.manual_seed(3)
    model = vmlmf_amd.MyLSTM(I, hidden_layer_sizes=[H], batch_first=True, w_rank=rw, u_ranks=ru, cell=getattr(vmlmf_amd, cell_name)).cuda()
    cell = model.rnncells[0]
    named = dict(cell.named_parameters())
    P = {}
    for k in ORDER[variant]:
        key = ("layers." + k) if ("layers." + k) in named else k
        P[k] = named[key].detach().cpu().numpy()
    rng = np.random.Generator(np.random.PCG64(81))
    x = rng.standard_normal((B, T, I)).astype(np.float32)
    dy = rng.standard_normal((B, T, H)).astype(np.float32)
    cfg = cell.kernel_cfg()
    ur = tuple(ru) if isinstance(ru, list) else (ru,)
    assert F._stack_plan((cfg["variant"], cfg["g"], rw, ur, False, 0), 1, B, T, I, H, True) is not None, "auto must pick the wavefront launch here"
    xg = torch.tensor(x, device="cuda").requires_grad_(True)
    y, hid = model(xg)
    (y * torch.tensor(dy, device="cuda")).sum().backward()
    torch.cuda.synchronize()
    Pt = O.to_torch(P, dtype=torch.float64, requires_grad=True)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    yr, hT, cT = O.literal_sequence(variant, Pt, xt, None, None, time_major=False)
    (yr * torch.tensor(dy, dtype=torch.float64)).sum().backward()
    assert_out(y.detach().cpu().numpy(), yr.detach().numpy(), "y")
    assert_out(hid.detach().cpu().numpy(), hT.detach().numpy(), "hidden")
    assert_grad(xg.grad.cpu().numpy(), xt.grad.numpy(), "dx")
    for k in ORDER[variant]:
        key = ("layers." + k) if ("layers." + k) in named else k
        assert_grad(named[key].grad.cpu().numpy(), Pt[k].grad.numpy(), k)


def test_lm_model_layer_loop_runs_as_one_wavefront_launch_with_carried_states():
    """Model.forward (vmlmf_lm.py:433-441) with covered MyVMLSTM layers and no active dropout: the layer loop goes through
    vmlmf_stack with the carried (h, c) of every layer as initial states - same scores, same new states, same gradients as
    the layer-by-layer loop (VMLMF_STACK=0), over two consecutive minibatches (truncated BPTT: states detached in between)."""
    import vmlmf_amd
    from vmlmf_amd import functional as F
    V, H, L, B, T, rw, ru = 60, 96, 2, 12, 9, 12, 10
    res = {}
    for mode in ("auto", "0"):
        os.environ["VMLMF_STACK"] = mode
        try:
            torch.manual_seed(11)
            m = vmlmf_amd.Model(V, H, L, 0.0, 0.1, w_rank=rw, u_ranks=[ru], lstm_type="vmlmf").cuda()
            if mode == "auto":
                cfg = m.rnns[0]
                assert F._stack_plan((cfg.variant, 1, rw, (ru,), True, 0), L, B, T, H, H, True) is not None
            g = torch.Generator().manual_seed(3)
            states = m.state_init(B)
            outs = []
            for step in range(2):
                xs = torch.randint(0, V, (T, B), generator=g).cuda()
                ys = torch.randint(0, V, (T, B), generator=g).cuda()
                states = m.detach(states)
                m.zero_grad(set_to_none=True)
                scores, states = m(xs, states)
                loss = vmlmf_amd.nll_loss(scores, ys)
                loss.backward()
                outs.append((scores.detach().clone(), [(h.detach().clone(), c.detach().clone()) for h, c in states],
                             {n: p.grad.clone() for n, p in m.named_parameters()}))
            torch.cuda.synchronize()
            res[mode] = outs
        finally:
            os.environ.pop("VMLMF_STACK", None)
    for a, b in zip(res["auto"], res["0"]):
        assert float((a[0] - b[0]).abs().max()) <= 2e-5 * max(1.0, float(b[0].abs().max()))
        for (ha, ca), (hb, cb) in zip(a[1], b[1]):
            assert float((ha - hb).abs().max()) <= 2e-5 and float((ca - cb).abs().max()) <= 2e-5
        for n, gref in b[2].items():
            assert float((a[2][n] - gref).abs().max()) <= 1e-4 * float(gref.abs().max()) + 1e-6, n
