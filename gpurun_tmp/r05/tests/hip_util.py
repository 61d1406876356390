"""Helpers for the GPU parity tests: run the HIP path through the package's public entry point
(vmlmf_amd.vmlmf_sequence -> ctypes -> C ABI) on oracle-style parameter dicts."""
import numpy as np
import torch

import vmlmf_oracle as O
from vmlmf_amd import vmlmf_sequence

ORDER = {
    O.V1: ["dia_x", "dia_h", "u_x", "v_x", "b_x", "b_h", "u_h", "v_h"],
    O.V2: ["dia_x", "dia_h", "u_x", "v_x", "bias_x", "bias_h", "u_h_0", "v_h_0", "u_h_1", "v_h_1"],
    O.V3: ["dia_x", "dia_h", "u_x", "w_x", "b_x", "b_h", "u_h", "w_h"],
    O.V4: ["dia_x", "dia_h", "u_x", "w_x", "b_x", "b_h", "u_h.0", "v_h.0", "u_h.1", "v_h.1"],
    O.V5: ["w", "u", "w1", "w2", "w3", "w4", "u1", "u2", "u3", "u4", "bias_i", "bias_f", "bias_o", "bias_c"],
    O.V6: ["u_x", "v_x", "bias_x", "bias_h", "u_h_0", "v_h_0", "u_h_1", "v_h_1"],
}


def ranks_of(variant, P):
    if variant == O.V5:
        return P["w"].shape[1], [P["u"].shape[1]], 1
    rw = P["u_x"].shape[1]
    if variant in (O.V1, O.V3):
        return rw, [P["u_h"].shape[1]], 1
    sep = "." if variant == O.V4 else "_"
    return rw, [P[f"u_h{sep}0"].shape[2], P[f"u_h{sep}1"].shape[2]], 2


def run_hip(variant, P, x, h0=None, c0=None, dy=None, dhT=None, dcT=None, time_major=False, dev="cuda", need_dx=True):
    """Forward (+ backward when any upstream gradient is given).  numpy in, numpy out.  need_dx=False: the input asks for no
    gradient (a first layer: the kernels that form the weight gradients inside the backward launch need that)."""
    names = ORDER[variant]
    params = [torch.tensor(np.asarray(P[k]), dtype=torch.float32, device=dev).requires_grad_(True) for k in names]
    xt = torch.tensor(x, dtype=torch.float32, device=dev).requires_grad_(need_dx)
    h0t = None if h0 is None else torch.tensor(h0, dtype=torch.float32, device=dev).requires_grad_(True)
    c0t = None if c0 is None else torch.tensor(c0, dtype=torch.float32, device=dev).requires_grad_(True)
    rw, ru, g = ranks_of(variant, P)
    y, hT, cT = vmlmf_sequence(variant, xt, h0t, c0t, params, rw, ru, g=g, time_major=time_major)
    out = {"y": y.detach().cpu().numpy(), "hT": hT.detach().cpu().numpy(), "cT": cT.detach().cpu().numpy()}
    if dy is not None or dhT is not None or dcT is not None:
        loss = 0.0
        if dy is not None:
            loss = loss + (y * torch.tensor(dy, device=dev)).sum()
        if dhT is not None:
            loss = loss + (hT * torch.tensor(dhT, device=dev)).sum()
        if dcT is not None:
            loss = loss + (cT * torch.tensor(dcT, device=dev)).sum()
        loss.backward()
        if need_dx:
            out["dx"] = xt.grad.cpu().numpy()
        if h0t is not None:
            out["dh0"] = h0t.grad.cpu().numpy()
            out["dc0"] = c0t.grad.cpu().numpy()
        out["G"] = {k: p.grad.cpu().numpy() for k, p in zip(names, params)}
    torch.cuda.synchronize()
    return out


def run_literal(variant, P, x, h0=None, c0=None, dy=None, dhT=None, dcT=None, time_major=False,
                dtype=torch.float64):
    """The oracle (literal restatement, autograd) on the same inputs, fp64 by default."""
    Pt = O.to_torch(P, dtype=dtype, requires_grad=True)
    xt = torch.tensor(x, dtype=dtype, requires_grad=True)
    B = x.shape[1] if time_major else x.shape[0]
    h0t = None if h0 is None else torch.tensor(h0, dtype=dtype, requires_grad=True)
    c0t = None if c0 is None else torch.tensor(c0, dtype=dtype, requires_grad=True)
    y, hT, cT = O.literal_sequence(variant, Pt, xt, h0t, c0t, time_major=time_major, v4_scratch_rows=B)
    out = {"y": y.detach().numpy(), "hT": hT.detach().numpy(), "cT": cT.detach().numpy()}
    if dy is not None or dhT is not None or dcT is not None:
        loss = 0.0
        if dy is not None:
            loss = loss + (y * torch.tensor(dy, dtype=dtype)).sum()
        if dhT is not None:
            loss = loss + (hT * torch.tensor(dhT, dtype=dtype)).sum()
        if dcT is not None:
            loss = loss + (cT * torch.tensor(dcT, dtype=dtype)).sum()
        loss.backward()
        out["dx"] = xt.grad.numpy()
        if h0t is not None:
            out["dh0"] = h0t.grad.numpy()
            out["dc0"] = c0t.grad.numpy()
        out["G"] = {k: v.grad.numpy() for k, v in Pt.items()}
    return out


# fp32 tolerance of the HIP path (BASELINE.md section 3 / north_star "stated fp32 tolerance"):
#   outputs y, h, c:      |err| <= 1e-5 + 1e-4 |ref|
#   gradients (dx, dh0, dc0, every parameter):  max|err| <= 1e-4 * max|ref| (+1e-6)
ATOL, RTOL, GREL = 1e-5, 1e-4, 1e-4


def assert_out(a, b, what, atol=ATOL, rtol=RTOL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.all(np.isfinite(a)), what + ": non-finite values"
    err = np.abs(a - b)
    bad = err > atol + rtol * np.abs(b)
    assert not bad.any(), f"{what}: max err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)} (|ref|max {np.abs(b).max():.3e}, {bad.sum()} bad)"


def assert_grad(a, b, what, rel=GREL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.all(np.isfinite(a)), what + ": non-finite values"
    scale = max(np.abs(b).max(), 1e-6)
    err = np.abs(a - b).max()
    assert err <= rel * scale + 1e-6, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def compare_all(got, ref, tag, skip=()):
    problems = []
    for k in ("y", "hT", "cT"):
        if k in ref and k not in skip:
            try:
                assert_out(got[k], ref[k], f"{tag}.{k}")
            except AssertionError as e:
                problems.append(str(e))
    for k in ("dx", "dh0", "dc0"):
        if k in ref and k in got:
            try:
                assert_grad(got[k], ref[k], f"{tag}.{k}")
            except AssertionError as e:
                problems.append(str(e))
    if "G" in ref:
        for k in ref["G"]:
            try:
                assert_grad(got["G"][k], ref["G"][k], f"{tag}.G.{k}")
            except AssertionError as e:
                problems.append(str(e))
    assert not problems, "\n".join(problems)
