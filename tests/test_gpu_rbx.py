"""GPU: clustered layers stacked in one launch per direction (csrc/vmlmf_rbx.hip; the layer loop V/src/models/vmlmf_lm.py:437-439 over
the PTB layers :53-174 / :178-280 at the batch sizes a GPU of an 8-GPU node holds) against the literal fp64 restatement, against the
chained per-layer launches, with dropout between the layers, and for run-to-run bit stability."""
import ctypes

import numpy as np
import pytest
import torch

import vmlmf_oracle as O
from hip_util import ORDER, assert_grad, assert_out
from vmlmf_amd import _lib, vmlmf_sequence
from vmlmf_amd import functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _cfg(variant):
    return (32, [32, 32], 2) if variant == O.V4 else (32, [32], 1)


def _plan_ok(variant, L, B, T, H):
    rw, ru, g = _cfg(variant)
    cfg = (variant, g, rw, tuple(ru), True, _lib.DT_F32)
    return F._stack_plan(cfg, L, B, T, H, H, True) is not None and F.stack_takes_dropout(cfg, L, B, T, H, H, True)


def _inputs(seed, L, B, T, H):
    r = np.random.Generator(np.random.PCG64(seed))
    x = (0.5 * r.standard_normal((T, B, H))).astype(np.float32)
    h0 = (0.3 * r.standard_normal((L, B, H))).astype(np.float32)
    c0 = (0.3 * r.standard_normal((L, B, H))).astype(np.float32)
    dy = r.standard_normal((T, B, H)).astype(np.float32)
    dhT = r.standard_normal((L, B, H)).astype(np.float32)
    dcT = r.standard_normal((L, B, H)).astype(np.float32)
    return x, h0, c0, dy, dhT, dcT


def _run_stack(variant, Ps, x, h0, c0, dy, dhT, dcT, drops=None, need_dx=True):
    rw, ru, g = _cfg(variant)
    L = len(Ps)
    names = ORDER[variant]
    params = [[torch.tensor(np.asarray(P[k]), device=DEV).requires_grad_(True) for k in names] for P in Ps]
    xg = torch.tensor(x, device=DEV).requires_grad_(need_dx)
    h0g = torch.tensor(h0, device=DEV).requires_grad_(True)
    c0g = torch.tensor(c0, device=DEV).requires_grad_(True)
    out = F.vmlmf_stack(variant, xg, params, rw, ru, g=g, time_major=True, h0=h0g, c0=c0g, drops=drops)
    assert out is not None, "the clustered form must cover this stack"
    y, hs, cs = out
    loss = (y * torch.tensor(dy, device=DEV)).sum()
    for l in range(L):
        loss = loss + (hs[l] * torch.tensor(dhT[l], device=DEV)).sum() + (cs[l] * torch.tensor(dcT[l], device=DEV)).sum()
    loss.backward()
    torch.cuda.synchronize()
    res = {"y": y.detach().cpu().numpy(), "hT": [h.detach().cpu().numpy() for h in hs], "cT": [c_.detach().cpu().numpy() for c_ in cs],
           "dh0": h0g.grad.cpu().numpy(), "dc0": c0g.grad.cpu().numpy(),
           "G": [{k: p.grad.cpu().numpy() for k, p in zip(names, params[l])} for l in range(L)]}
    if need_dx:
        res["dx"] = xg.grad.cpu().numpy()
    return res


def _run_oracle(variant, Ps, x, h0, c0, dy, dhT, dcT, factors=None):
    L, B = len(Ps), x.shape[1]
    f64 = torch.float64
    Pt = [O.to_torch(P, dtype=f64, requires_grad=True) for P in Ps]
    xt = torch.tensor(x, dtype=f64, requires_grad=True)
    h0t = torch.tensor(h0, dtype=f64, requires_grad=True)
    c0t = torch.tensor(c0, dtype=f64, requires_grad=True)
    cur, loss, hs, cs = xt, 0.0, [], []
    for l in range(L):
        cur, hT, cT = O.literal_sequence(variant, Pt[l], cur, h0t[l], c0t[l], time_major=True, v4_scratch_rows=B)
        if factors is not None and factors[l] is not None:
            cur = cur * torch.tensor(factors[l], dtype=f64)
        hs.append(hT), cs.append(cT)
        loss = loss + (hT * torch.tensor(dhT[l], dtype=f64)).sum() + (cT * torch.tensor(dcT[l], dtype=f64)).sum()
    loss = loss + (cur * torch.tensor(dy, dtype=f64)).sum()
    loss.backward()
    return {"y": cur.detach().numpy(), "hT": [h.detach().numpy() for h in hs], "cT": [c_.detach().numpy() for c_ in cs],
            "dx": xt.grad.numpy(), "dh0": h0t.grad.numpy(), "dc0": c0t.grad.numpy(),
            "G": [{k: Pt[l][k].grad.numpy() for k in ORDER[variant]} for l in range(L)]}


def _compare(got, ref, tag):
    assert_out(got["y"], ref["y"], tag + ".y")
    for l in range(len(ref["hT"])):
        assert_out(got["hT"][l], ref["hT"][l], f"{tag}.hT[{l}]")
        assert_out(got["cT"][l], ref["cT"][l], f"{tag}.cT[{l}]")
    if "dx" in got:
        assert_grad(got["dx"], ref["dx"], tag + ".dx")
    assert_grad(got["dh0"], ref["dh0"], tag + ".dh0")
    assert_grad(got["dc0"], ref["dc0"], tag + ".dc0")
    for l in range(len(ref["G"])):
        for k, v in ref["G"][l].items():
            assert_grad(got["G"][l][k], v, f"{tag}.layer{l}.{k}")


@pytest.mark.parametrize("variant,L,B,T", [(O.V4, 2, 32, 5), (O.V4, 2, 64, 4), (O.V4, 2, 128, 3), (O.V3, 2, 32, 6), (O.V3, 2, 100, 3),
                                           (O.V3, 3, 20, 4), (O.V4, 2, 7, 2), (O.V3, 2, 33, 1)],
                         ids=lambda v: str(v))
def test_stacked_clustered_layers_vs_oracle(variant, L, B, T):
    """Outputs, carried states, dx, dh0 / dc0 and every parameter gradient of every layer against the literal fp64 restatement
    (vmlmf_lm.py:437-439 over :97-163 / :222-269 with v4_scratch_rows = B)."""
    H = 650
    assert _plan_ok(variant, L, B, T, H)
    rw, ru, g = _cfg(variant)
    Ps = [O.make_params(variant, H, H, rw, ru if g == 2 else ru[0], seed=41 + l, scale=0.05) for l in range(L)]
    inp = _inputs(3 + B, L, B, T, H)
    got = _run_stack(variant, Ps, *inp)
    ref = _run_oracle(variant, Ps, *inp)
    _compare(got, ref, f"rbx.v{variant}.L{L}.B{B}")


def test_config_e_two_layers_at_the_8_gpu_operating_points_vs_the_chained_launches():
    """configs[4] per GPU (32 / 64 / 128 rows, T = 35): the one-launch form against today's chained per-layer launches (same
    arithmetic per element; the x side is summed on the matrix cores in a different order) and bit-identical run to run."""
    variant, H, T, L = O.V4, 650, 35, 2
    rw, ru, g = _cfg(variant)
    Ps = [O.make_params(variant, H, H, rw, ru, seed=21 + l, scale=0.05) for l in range(L)]
    names = ORDER[variant]
    for B in (32, 64, 128):
        assert _plan_ok(variant, L, B, T, H)
        x, h0, c0, dy, dhT, dcT = _inputs(B, L, B, T, H)
        x *= 0.1
        a = _run_stack(variant, Ps, x, h0, c0, dy, dhT, dcT)
        b = _run_stack(variant, Ps, x, h0, c0, dy, dhT, dcT)
        for k in ("y", "dx", "dh0", "dc0"):
            assert np.array_equal(a[k], b[k]), f"B={B}: {k} differs between two runs"
        for l in range(L):
            for k in names:
                assert np.array_equal(a["G"][l][k], b["G"][l][k]), f"B={B}: layer {l} {k} differs between two runs"
        # chained launches
        params = [[torch.tensor(np.asarray(P[k]), device=DEV).requires_grad_(True) for k in names] for P in Ps]
        xg = torch.tensor(x, device=DEV).requires_grad_(True)
        h0g, c0g = torch.tensor(h0, device=DEV).requires_grad_(True), torch.tensor(c0, device=DEV).requires_grad_(True)
        cur, loss, hs, cs = xg, 0.0, [], []
        for l in range(L):
            cur, hT, cT = vmlmf_sequence(variant, cur, h0g[l], c0g[l], params[l], rw, ru, g=g, time_major=True)
            hs.append(hT), cs.append(cT)
            loss = loss + (hT * torch.tensor(dhT[l], device=DEV)).sum() + (cT * torch.tensor(dcT[l], device=DEV)).sum()
        (loss + (cur * torch.tensor(dy, device=DEV)).sum()).backward()
        ref = {"y": cur.detach().cpu().numpy(), "hT": [h.detach().cpu().numpy() for h in hs], "cT": [c_.detach().cpu().numpy() for c_ in cs],
               "dx": xg.grad.cpu().numpy(), "dh0": h0g.grad.cpu().numpy(), "dc0": c0g.grad.cpu().numpy(),
               "G": [{k: p.grad.cpu().numpy() for k, p in zip(names, params[l])} for l in range(L)]}
        _compare(a, ref, f"rbx.vs_chained.B{B}")


@pytest.mark.parametrize("variant,B,T,p", [(O.V4, 32, 4, 0.5), (O.V3, 48, 3, 0.25)], ids=lambda v: str(v))
def test_dropout_between_the_stacked_layers_vs_oracle(variant, B, T, p):
    """nn.Dropout(p) behind every layer (vmlmf_lm.py:438-439) inside the one launch: the factors are those the single-layer launches
    apply (exported by vmlmf_dropout_factors with the layer's descriptor: thread-slot columns), multiplied into the fp64 oracle."""
    from vmlmf_amd.functional import dropout_factors, dropout_state
    H, L = 650, 2
    rw, ru, g = _cfg(variant)
    Ps = [O.make_params(variant, H, H, rw, ru if g == 2 else ru[0], seed=61 + l, scale=0.05) for l in range(L)]
    inp = _inputs(17, L, B, T, H)
    snap = dropout_state(torch.device("cuda", 0), 1234567)
    snap[1] = 5
    drops = [(p, snap, l + 1) for l in range(L)]
    got = _run_stack(variant, Ps, *inp, drops=drops)
    desc = _lib.make_desc(variant, B, T, H, H, rw, ru, g=g, time_major=True, training=True)
    Fs = [dropout_factors(T * B, H, p, snap, l + 1, layer_desc=desc).cpu().numpy().reshape(T, B, H) for l in range(L)]
    assert all(abs((f == 0).mean() - p) < 0.03 for f in Fs)
    ref = _run_oracle(variant, Ps, *inp, factors=Fs)
    _compare(got, ref, f"rbx.drop.v{variant}")


def test_batches_beyond_co_residency_fall_back_to_the_chained_launches():
    assert not _plan_ok(O.V4, 2, 256, 35, 650)
    assert not _plan_ok(O.V4, 3, 128, 35, 650)
    assert _plan_ok(O.V4, 2, 128, 35, 650)
