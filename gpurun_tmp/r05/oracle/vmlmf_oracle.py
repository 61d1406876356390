"""CPU oracle for the VMLMF compressed-LSTM hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  ``vmlmf_amd`` (the shipped package) never does.

Parity status: PINNED.  The reference's own tests hold no value assertions (shape checks only,
``V/src/unittest/unit_test.py:61-93``), so the pin is a set of golden vectors produced *in the build
container* by importing the reference modules themselves (``oracle/make_golden.py`` ->
``tests/golden/*.npz``).  ``tests/test_oracle_golden.py`` checks both restatements below against them.

``V/`` = ``/root/reference/rnn_compression_factorization_vmlmf/``.

Two restatements live here:

* **literal** (``literal_step`` / ``literal_sequence``): the reference's ATen op sequence, op for op,
  for the four cells (the per-timestep 4-iteration diagonal-removal loop with in-place slice writes
  included).  Backward = torch autograd, exactly as in the reference.  This is what ``bench.py`` times as
  ``cpu_baseline`` (kind "port") and what the GPU parity tests compare against.
* **unified** (``canonicalize`` / ``unified_forward`` / ``unified_backward`` / ``uncanonicalize_grads``):
  numpy, any float dtype, analytic backward.  It is the *specification of the HIP kernels*: every variant
  becomes "rank-space reduce -> per-unit expand -> LSTM gates" over one canonical parameter image, and the
  parameter-only diagonal-removal vectors are hoisted out of the time loop.
"""
from __future__ import annotations

import numpy as np
import torch

V1, V2, V3, V4 = 1, 2, 3, 4  # MyVMLMFCell, MyVMLMFCellg2, MyVMLSTM, MyVMLSTMGroup
# the two "no vector-multiplication" comparison cells of the reference:
V5, V6 = 5, 6                 # MyLSTMCell in low-rank mode, MyVMLMFgCellg2 (ablation)
NOVM = (V5, V6)
GATE_NAMES_V5 = ("i", "f", "o", "c")  # w1/u1 -> i, w2/u2 -> f, w3/u3 -> o, w4/u4 -> c~   (vmlmf.py:223-232)

# ----------------------------------------------------------------------------------------------------
# parameter containers
# ----------------------------------------------------------------------------------------------------
# A parameter set is a plain dict  name -> tensor  using the reference's own parameter names:
#   V1: u_x u_h v_x v_h b_x b_h dia_x dia_h                       (V/src/models/vmlmf.py:56-69)
#   V2: dia_x dia_h u_x v_x u_h_{s} v_h_{s} bias_x bias_h          (V/src/models/vmlmf_group.py:61-79)
#   V3: u_x u_h w_x w_h b_x b_h dia_x dia_h                        (V/src/models/vmlmf_lm.py:200-213)
#   V4: u_x w_x u_h.{s} v_h.{s} b_x b_h dia_x dia_h                (V/src/models/vmlmf_lm.py:77-91)
#   V5: w w1..w4 u u1..u4 bias_f bias_i bias_c bias_o              (V/src/models/vmlmf.py:159-186, low-rank mode)
#   V6: u_x v_x u_h_{s} v_h_{s} bias_x bias_h                      (V/src/models/vmlmf_group.py:183-197)


def param_shapes(variant, I, H, rw, ru, g=2):
    """Reference parameter names and shapes for one cell/layer."""
    if variant in (V1, V3):
        r = ru[0] if isinstance(ru, (list, tuple)) else ru
        vx, vh = ("v_x", "v_h") if variant == V1 else ("w_x", "w_h")
        return {"u_x": (I, rw), "u_h": (H, r), vx: (4 * H, rw), vh: (4 * H, r),
                "b_x": (4 * H,), "b_h": (4 * H,), "dia_x": (1, I), "dia_h": (1, H)}
    if variant == V5:
        r = ru[0] if isinstance(ru, (list, tuple)) else ru
        out = {"w": (I, rw)}
        out.update({f"w{k}": (rw, H) for k in range(1, 5)})
        out["u"] = (H, r)
        out.update({f"u{k}": (r, H) for k in range(1, 5)})
        out.update({f"bias_{n}": (1, H) for n in ("f", "i", "c", "o")})
        return out
    Hg = H // g
    if variant in (V2, V6):
        out = {"dia_x": (1, I), "dia_h": (1, H), "u_x": (I, rw), "v_x": (4 * H, rw)} if variant == V2 else \
              {"u_x": (I, rw), "v_x": (4 * H, rw)}
        for s in range(g):
            out[f"u_h_{s}"] = (g, Hg, ru[s])
            out[f"v_h_{s}"] = (g, ru[s], 4 * Hg)
        out["bias_x"] = (1, 4 * H)
        out["bias_h"] = (1, 4 * H)
        return out
    out = {"u_x": (I, rw), "w_x": (4 * H, rw)}
    for s in range(g):
        out[f"u_h.{s}"] = (g, Hg, ru[s])
    for s in range(g):
        out[f"v_h.{s}"] = (g, ru[s], 4 * Hg)
    out.update({"b_x": (4 * H,), "b_h": (4 * H,), "dia_x": (1, I), "dia_h": (1, H)})
    return out


def make_params(variant, I, H, rw, ru, g=2, seed=3, scale=0.1, dtype=np.float32):
    """Seeded numpy-PCG64 parameters (same values on every box)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return {k: (scale * rng.standard_normal(shp)).astype(dtype)
            for k, shp in param_shapes(variant, I, H, rw, ru, g).items()}


def to_torch(P, dtype=torch.float32, requires_grad=False):
    return {k: torch.tensor(np.asarray(v), dtype=dtype).requires_grad_(requires_grad) for k, v in P.items()}


# ----------------------------------------------------------------------------------------------------
# literal restatement (torch, op for op)
# ----------------------------------------------------------------------------------------------------

def _strip_diag(dst, vec, U, V, width, H):
    """The reference's per-step diagonal removal: for each of the four H-wide gate blocks, write
    ``vec * rowsum(U * V[block rows])`` into a slice of ``dst`` (in place, as the reference does).
    V/src/models/vmlmf.py:102-106, vmlmf_lm.py:251-255."""
    for lo in range(0, 4 * H, H):
        dst[:, lo:lo + width] = vec * torch.sum(U * V[lo:lo + width, :], dim=1)


def _rotated_group_product(h, us, vs, g, H, B):
    """Group low-rank h-path: shift s feeds destination group j from source group (j+s) mod g.
    V/src/models/vmlmf_group.py:117-132, vmlmf_lm.py:123-136.  Returns [B, g, 4H/g]."""
    order = list(range(g))
    total = None
    for s in range(g):
        hv = h.view(B, g, H // g)
        if s > 0:
            order = order[1:] + order[0:1]
            hv = hv[:, order, :]
        hv = torch.transpose(hv, 0, 1)
        hv = torch.bmm(hv, us[s])
        hv = torch.bmm(hv, vs[s])
        hv = torch.transpose(hv, 0, 1)
        total = hv if s == 0 else total + hv
    return total


def _lstm_tail(pi, pf, po, pn, c):
    """V/src/models/vmlmf.py:117-124."""
    ig = torch.sigmoid(pi)
    fg = torch.sigmoid(pf)
    og = torch.sigmoid(po)
    ng = torch.tanh(pn)
    c_next = fg * c + ig * ng
    h_next = og * torch.tanh(c_next)
    return h_next, c_next


def literal_step(variant, P, x, h, c, g=2, v4_scratch_rows=None):
    """One cell step, same ATen ops as the reference cell of that variant."""
    B = x.shape[0]
    if variant == V1:  # V/src/models/vmlmf.py:78-125
        H, I = P["dia_h"].shape[1], P["dia_x"].shape[1]
        rx = torch.zeros(B, 4 * H, dtype=x.dtype)
        rh = torch.zeros(B, 4 * H, dtype=x.dtype)
        vm_x = torch.cat([P["dia_x"] * x.squeeze(), torch.zeros([h.shape[0], H - I], dtype=x.dtype)], dim=1)
        vm_h = P["dia_h"] * h.squeeze()
        low_x = torch.matmul(torch.matmul(x, P["u_x"]), P["v_x"].t())
        low_h = torch.matmul(torch.matmul(h, P["u_h"]), P["v_h"].t())
        for lo in range(0, 4 * H, H):
            rx[:, lo:lo + I] = x * torch.sum(P["u_x"] * P["v_x"][lo:lo + I, :], dim=1)
            rh[:, lo:lo + H] = h * torch.sum(P["u_h"] * P["v_h"][lo:lo + H, :], dim=1)
        gx = low_x - rx + P["b_x"]
        gh = low_h - rh + P["b_h"]
        xi, xf, xo, xn = gx.chunk(4, 1)
        hi, hf, ho, hn = gh.chunk(4, 1)
        return _lstm_tail(xi + hi + vm_x + vm_h, xf + hf + vm_x + vm_h,
                          xo + ho + vm_x + vm_h, xn + hn + vm_x + vm_h, c)
    if variant == V2:  # V/src/models/vmlmf_group.py:85-155
        H, I = P["dia_h"].shape[1], P["dia_x"].shape[1]
        r0 = P["u_h_0"].shape[2]
        vm_x = torch.cat([P["dia_x"] * x.squeeze(), torch.zeros([h.shape[0], H - I], dtype=x.dtype)], dim=1)
        vm_h = P["dia_h"] * h.squeeze()
        low_x = torch.matmul(torch.matmul(x, P["u_x"]), P["v_x"].t())
        rx = torch.zeros(B, 4 * H, dtype=x.dtype)
        rh = torch.zeros(B, 4 * H, dtype=x.dtype)
        u0 = P["u_h_0"].view(H, r0)
        v0t = torch.transpose(P["v_h_0"], 1, 2).contiguous()
        Hg = H // g
        for lo in range(0, 4 * H, H):
            rx[:, lo:lo + I] = x * torch.sum(P["u_x"] * P["v_x"][lo:lo + I, :], dim=1)
            glo = int(lo / g)
            gate_v = v0t[:, glo:glo + Hg, :].reshape(-1, r0)
            rh[:, lo:lo + H] = h * torch.sum(u0 * gate_v, dim=1)
        gx = low_x - rx + P["bias_x"]
        xi, xf, xo, xn = gx.chunk(4, 1)
        acc = _rotated_group_product(h, [P[f"u_h_{s}"] for s in range(g)],
                                     [P[f"v_h_{s}"] for s in range(g)], g, H, B)
        f_h, i_h, n_h, o_h = acc.chunk(4, dim=2)
        f_h = f_h.contiguous().view(B, H)
        i_h = i_h.contiguous().view(B, H)
        n_h = n_h.contiguous().view(B, H)
        o_h = o_h.contiguous().view(B, H)
        gh = P["bias_h"] - rh
        hf, hi, hn, ho = gh.chunk(4, 1)
        hf = hf + f_h
        hi = hi + i_h
        hn = hn + n_h
        ho = ho + o_h
        return _lstm_tail(xi + hi + vm_x + vm_h, xf + hf + vm_x + vm_h,
                          xo + ho + vm_x + vm_h, xn + hn + vm_x + vm_h, c)
    if variant == V3:  # V/src/models/vmlmf_lm.py:222-269
        H, I = P["dia_h"].shape[1], P["dia_x"].shape[1]
        rx = torch.zeros(B, 4 * H, dtype=x.dtype)
        rh = torch.zeros(B, 4 * H, dtype=x.dtype)
        vm_x = P["dia_x"] * x.squeeze()
        vm_h = P["dia_h"] * h.squeeze()
        vm_x = torch.cat([vm_x for _ in range(4)], dim=1)
        vm_h = torch.cat([vm_h for _ in range(4)], dim=1)
        low_x = torch.matmul(torch.matmul(x, P["u_x"]), P["w_x"].t())
        low_h = torch.matmul(torch.matmul(h, P["u_h"]), P["w_h"].t())
        for lo in range(0, 4 * H, H):
            rx[:, lo:lo + I] = x * torch.sum(P["u_x"] * P["w_x"][lo:lo + I, :], dim=1)
            rh[:, lo:lo + H] = h * torch.sum(P["u_h"] * P["w_h"][lo:lo + H, :], dim=1)
        gx = vm_x + low_x - rx + P["b_x"]
        gh = vm_h + low_h - rh + P["b_h"]
        xi, xf, xo, xn = gx.chunk(4, 1)
        hi, hf, ho, hn = gh.chunk(4, 1)
        return _lstm_tail(xi + hi, xf + hf, xo + ho, xn + hn, c)
    if variant == V4:  # V/src/models/vmlmf_lm.py:97-163
        H, I = P["dia_h"].shape[1], P["dia_x"].shape[1]
        r0 = P["u_h.0"].shape[2]
        # the reference hard-codes 40 scratch rows (vmlmf_lm.py:112-113); any other batch raises there.
        rows = 40 if v4_scratch_rows is None else v4_scratch_rows
        rx = torch.zeros(rows, 4 * H, dtype=x.dtype)
        rh = torch.zeros(rows, 4 * H, dtype=x.dtype)
        vm_x = P["dia_x"] * x.squeeze()
        vm_h = P["dia_h"] * h.squeeze()
        vm_x = torch.cat([vm_x for _ in range(4)], dim=1)
        vm_h = torch.cat([vm_h for _ in range(4)], dim=1)
        low_x = torch.matmul(torch.matmul(x, P["u_x"]), P["w_x"].t())
        order = list(range(g))
        low_h = None
        for s in range(g):
            top = h.view(-1, g, H // g)
            if s > 0:
                order = order[1:] + order[0:1]
                top = top[:, order, :]
            top = torch.transpose(top, 0, 1)
            top = torch.bmm(top, P[f"u_h.{s}"])
            top = torch.bmm(top, P[f"v_h.{s}"])
            top = torch.transpose(top, 0, 1)
            top = top.contiguous().view(-1, H * 4)
            low_h = top if s == 0 else top + low_h
        re_u = P["u_h.0"].view(H, r0)
        re_v = torch.transpose(P["v_h.0"], 1, 2).contiguous().view(4 * H, r0)
        for lo in range(0, 4 * H, H):
            rx[:, lo:lo + I] = x * torch.sum(P["u_x"] * P["w_x"][lo:lo + I, :], dim=1)
            rh[:, lo:lo + H] = h * torch.sum(re_u * re_v[lo:lo + H, :], dim=1)
        gx = vm_x + low_x - rx + P["b_x"]
        gh = vm_h + low_h - rh + P["b_h"]
        xi, xf, xo, xn = gx.squeeze().chunk(4, 1)
        hi, hf, ho, hn = gh.chunk(4, 1)
        return _lstm_tail(xi + hi, xf + hf, xo + ho, xn + hn, c)
    if variant == V5:  # V/src/models/vmlmf.py:188-236, low-rank branches (198-207, 215-224)
        xw = torch.matmul(x, P["w"])
        w_val1 = torch.matmul(xw, P["w1"])
        w_val2 = torch.matmul(xw, P["w2"])
        w_val3 = torch.matmul(xw, P["w3"])
        w_val4 = torch.matmul(xw, P["w4"])
        hu = torch.matmul(h, P["u"])
        u_val1 = torch.matmul(hu, P["u1"])
        u_val2 = torch.matmul(hu, P["u2"])
        u_val3 = torch.matmul(hu, P["u3"])
        u_val4 = torch.matmul(hu, P["u4"])
        return _lstm_tail(w_val1 + u_val1 + P["bias_i"], w_val2 + u_val2 + P["bias_f"],
                          w_val3 + u_val3 + P["bias_o"], w_val4 + u_val4 + P["bias_c"], c)
    if variant == V6:  # V/src/models/vmlmf_group.py:203-251
        H = P["bias_h"].shape[1] // 4
        gx = torch.matmul(torch.matmul(x, P["u_x"]), P["v_x"].t()) + P["bias_x"]
        xf, xi, xn, xo = gx.chunk(4, 1)
        acc = _rotated_group_product(h, [P[f"u_h_{s}"] for s in range(g)],
                                     [P[f"v_h_{s}"] for s in range(g)], g, H, B)
        f_h, i_h, n_h, o_h = acc.chunk(4, dim=2)
        f_h = f_h.contiguous().view(B, H)
        i_h = i_h.contiguous().view(B, H)
        n_h = n_h.contiguous().view(B, H)
        o_h = o_h.contiguous().view(B, H)
        hf, hi, hn, ho = P["bias_h"].chunk(4, 1)
        return _lstm_tail(xi + (hi + i_h), xf + (hf + f_h), xo + (ho + o_h), xn + (hn + n_h), c)
    raise ValueError(f"unknown variant {variant}")


def hidden_size_of(variant, P):
    if variant == V5:
        return P["u"].shape[0]
    if variant == V6:
        return P["bias_h"].shape[1] // 4
    return P["dia_h"].shape[1]


def literal_sequence(variant, P, x, h0=None, c0=None, g=2, time_major=None, v4_scratch_rows=None):
    """The sequence loop around the cell.

    HAR (V1/V2): ``MyLSTM.forward`` for one layer, batch-first x (B,T,I), zero initial state
    (V/src/models/vmlmf.py:300-314).  LM (V3/V4): ``forward(x, states)``, time-major x (T,B,X)
    (V/src/models/vmlmf_lm.py:272-280).  Returns (y, h_T, c_T) with y in the layout of x.
    """
    if time_major is None:
        time_major = variant in (V3, V4)
    tdim = 0 if time_major else 1
    B = x.shape[1 - tdim]
    H = hidden_size_of(variant, P)
    h = torch.zeros(B, H, dtype=x.dtype) if h0 is None else h0
    c = torch.zeros(B, H, dtype=x.dtype) if c0 is None else c0
    outs = []
    for x_t in torch.unbind(x, tdim):
        h, c = literal_step(variant, P, x_t, h, c, g=g, v4_scratch_rows=v4_scratch_rows)
        outs.append(h)
    return torch.stack(outs, tdim), h, c


# ----------------------------------------------------------------------------------------------------
# unified restatement (numpy) == specification of the HIP kernels
# ----------------------------------------------------------------------------------------------------

GATE_H_CHUNK_V2 = (1, 0, 3, 2)  # canonical gate k=(i,f,o,n) -> h-side chunk (f,i,n,o): vmlmf_group.py:134,149-152


def pad4(n):
    return (n + 3) // 4 * 4


class Canon:
    """Canonical parameter image shared by every variant.

    Units n = 0..H-1, gates k = 0..3 in (i, f, o, n) order, G rank-space vectors Q[j] of width KH.
      ux  (I, KX)        x @ ux = qx
      vx  (H, 4, KX)     pre_x[k][n] = qx . vx[n,k]
      uc  (H, KH)        unit n adds h[n]*uc[n, off_s + r] into Q[dest[n, s]][off_s + r]
      vc  (H, 4, KH)     pre_h[k][n] = Q[qsel[k, n]] . vc[n,k]
      ex  (H, 4)         x[n] * ex[n,k]   (zero for n >= I)
      eh  (H, 4)         h[n] * eh[n,k]
      b   (H, 4)         b_x + b_h
    Rank blocks are padded to multiples of four (zeros), which is what the kernels hold in registers.
    """
    pass


def _group_cols(variant, H, g, k, n):
    """(qsel, column inside the (r_s, 4H/g) matrices of that destination group) for gate k of unit n."""
    Hg = H // g
    if variant in (V2, V6):
        return n // Hg, GATE_H_CHUNK_V2[k] * Hg + (n % Hg)
    return divmod(k * H + n, 4 * Hg)  # V4: flat [B, g*4Hg] then chunk(4): vmlmf_lm.py:135,155


def canonicalize(variant, P, g=2, dtype=np.float64):
    P = {k: np.asarray(v.detach().numpy() if isinstance(v, torch.Tensor) else v, dtype=dtype) for k, v in P.items()}
    C = Canon()
    C.variant = variant
    C.novm = variant in NOVM
    H = hidden_size_of(variant, P)
    ux_name = "w" if variant == V5 else "u_x"
    I, rw = P[ux_name].shape
    C.H, C.I, C.rw = H, I, rw
    C.KX = pad4(rw)
    C.ux = np.zeros((I, C.KX), dtype)
    C.ux[:, :rw] = P[ux_name]
    C.vx = np.zeros((H, 4, C.KX), dtype)
    # x-side chunk that canonical gate k reads: V6 chunks gx as (f,i,n,o) (vmlmf_group.py:211)
    C.xchunk = np.array(GATE_H_CHUNK_V2 if variant == V6 else (0, 1, 2, 3), np.int64)
    if variant == V5:
        for k in range(4):
            C.vx[:, k, :rw] = P[f"w{k + 1}"].T
    else:
        vx_name = "v_x" if variant in (V1, V2, V6) else "w_x"
        C.vx[:, :, :rw] = P[vx_name].reshape(4, H, rw)[C.xchunk].transpose(1, 0, 2)
    if variant in (V1, V3):
        C.g = 1
        vh_name = "v_h" if variant == V1 else "w_h"
        ru = [P["u_h"].shape[1]]
        us = [P["u_h"][None]]                      # (1, H, r)
        vs = [P[vh_name].T[None]]                  # (1, r, 4H): column kH+n
        bx, bh = P["b_x"], P["b_h"]
    elif variant == V5:
        C.g = 1
        ru = [P["u"].shape[1]]
        us = [P["u"][None]]
        vs = [np.concatenate([P[f"u{k + 1}"] for k in range(4)], axis=1)[None]]   # (1, r, 4H): column kH+n
        bx = np.concatenate([P[f"bias_{n}"][0] for n in GATE_NAMES_V5])
        bh = np.zeros_like(bx)
    else:
        C.g = g
        sep = "." if variant == V4 else "_"
        us = [P[f"u_h{sep}{s}"] for s in range(g)]
        vs = [P[f"v_h{sep}{s}"] for s in range(g)]
        ru = [u.shape[2] for u in us]
        bx, bh = (P["bias_x"][0], P["bias_h"][0]) if variant in (V2, V6) else (P["b_x"], P["b_h"])
    G = C.g
    Hg = H // G
    C.ru = ru
    C.off = [0]
    for r in ru:
        C.off.append(C.off[-1] + pad4(r))
    C.KH = C.off[-1]
    C.uc = np.zeros((H, C.KH), dtype)
    C.vc = np.zeros((H, 4, C.KH), dtype)
    C.dest = np.zeros((H, G), np.int64)
    C.qsel = np.zeros((4, H), np.int64)
    C.col = np.zeros((4, H), np.int64)
    C.hchunk = np.zeros(4, np.int64)
    for k in range(4):
        C.hchunk[k] = GATE_H_CHUNK_V2[k] if variant in (V2, V6) else k
    for n in range(H):
        grp, m = divmod(n, Hg)
        for s in range(G):
            j = (grp - s) % G
            C.dest[n, s] = j
            C.uc[n, C.off[s]:C.off[s] + ru[s]] = us[s][j, m, :]
        for k in range(4):
            if variant in (V1, V3, V5):
                q, col = 0, k * H + n
            else:
                q, col = _group_cols(variant, H, G, k, n)
            C.qsel[k, n], C.col[k, n] = q, col
            for s in range(G):
                C.vc[n, k, C.off[s]:C.off[s] + ru[s]] = vs[s][q, :, col]
    r0 = ru[0]
    C.eh = np.zeros((H, 4), dtype)
    C.ex = np.zeros((H, 4), dtype)
    if not C.novm:   # the comparison cells have neither the vector multiplication nor the diagonal removal
        C.eh = P["dia_h"][0][:, None] - np.einsum("nr,nkr->nk", C.uc[:, :r0], C.vc[:, :, :r0])
        C.ex[:I] = P["dia_x"][0][:, None] - np.einsum("mr,mkr->mk", C.ux[:, :rw], C.vx[:I, :, :rw])
    if variant in (V3, V4):
        assert I == H, "vmlmf_lm.py:243 tiles vm_x four times: needs input_size == hidden_size"
    C.b = np.zeros((H, 4), dtype)
    for k in range(4):
        C.b[:, k] = bx[C.xchunk[k] * H:(C.xchunk[k] + 1) * H] + bh[C.hchunk[k] * H:(C.hchunk[k] + 1) * H]
    return C


def _sigmoid(z):
    return 1.0 / (1.0 + np.exp(-z))


def _rank_reduce(C, h):
    """Q[b, j, :] from h (B, H)."""
    B = h.shape[0]
    Q = np.zeros((B, C.g, C.KH), h.dtype)
    for s in range(C.g):
        lo, hi = C.off[s], C.off[s + 1]
        contrib = h[:, :, None] * C.uc[None, :, lo:hi]            # (B, H, blk)
        for j in range(C.g):
            Q[:, j, lo:hi] = contrib[:, C.dest[:, s] == j, :].sum(1)
    return Q


def unified_forward(C, x, h0, c0):
    """x (T,B,I) time-major.  Returns y (T,B,H), hT, cT and the tape the backward needs."""
    T, B, _ = x.shape
    H = C.H
    dt = x.dtype
    y = np.zeros((T, B, H), dt)
    gates = np.zeros((T, B, H, 4), dt)
    cs = np.zeros((T, B, H), dt)
    Qs = np.zeros((T, B, C.g, C.KH), dt)
    h, c = h0.astype(dt), c0.astype(dt)
    xpad = np.zeros((T, B, H), dt)
    xpad[:, :, :min(C.I, H)] = x[:, :, :H]      # I > H only occurs for the cells without the x .* ex term
    for t in range(T):
        qx = x[t] @ C.ux                                           # (B, KX)
        Q = _rank_reduce(C, h)
        Qs[t] = Q
        pre = np.einsum("br,nkr->bnk", qx, C.vx) + xpad[t][:, :, None] * C.ex[None] + C.b[None]
        for k in range(4):
            Qk = Q[:, C.qsel[k], :]                                 # (B, H, KH)
            pre[:, :, k] += np.einsum("bnr,nr->bn", Qk, C.vc[:, k, :])
        pre += h[:, :, None] * C.eh[None]
        i_, f_, o_ = _sigmoid(pre[..., 0]), _sigmoid(pre[..., 1]), _sigmoid(pre[..., 2])
        n_ = np.tanh(pre[..., 3])
        c = f_ * c + i_ * n_
        h = o_ * np.tanh(c)
        y[t], cs[t] = h, c
        gates[t] = np.stack([i_, f_, o_, n_], -1)
    tape = dict(gates=gates, cs=cs, Qs=Qs, x=x, h0=h0.astype(dt), c0=c0.astype(dt), y=y)
    return y, h, c, tape


def unified_backward(C, tape, dy, dhT, dcT):
    """Analytic backward.  Returns dx (T,B,I), dh0, dc0 and canonical gradients."""
    gates, cs, Qs, x, y = tape["gates"], tape["cs"], tape["Qs"], tape["x"], tape["y"]
    T, B, H = y.shape
    dt = y.dtype
    I = C.I
    G = {"ux": np.zeros_like(C.ux), "vx": np.zeros_like(C.vx), "uc": np.zeros_like(C.uc),
         "vc": np.zeros_like(C.vc), "ex": np.zeros_like(C.ex), "eh": np.zeros_like(C.eh),
         "b": np.zeros_like(C.b)}
    dx = np.zeros((T, B, I), dt)
    dh_rec = dhT.astype(dt).copy()
    dc = dcT.astype(dt).copy()
    for t in range(T - 1, -1, -1):
        h_prev = y[t - 1] if t > 0 else tape["h0"]
        c_prev = cs[t - 1] if t > 0 else tape["c0"]
        i_, f_, o_, n_ = (gates[t][..., k] for k in range(4))
        dh = dy[t] + dh_rec
        tc = np.tanh(cs[t])
        dct = dc + dh * o_ * (1 - tc * tc)
        dpre = np.stack([dct * n_ * i_ * (1 - i_), dct * c_prev * f_ * (1 - f_),
                         dh * tc * o_ * (1 - o_), dct * i_ * (1 - n_ * n_)], -1)   # (B,H,4)
        dc = dct * f_
        # rank-space gradient  dQ[b, j, :] = sum_{k,n: qsel[k,n]==j} dpre[b,n,k] * vc[n,k,:]
        dQ = np.zeros((B, C.g, C.KH), dt)
        for k in range(4):
            term = dpre[:, :, k, None] * C.vc[None, :, k, :]
            for j in range(C.g):
                dQ[:, j] += term[:, C.qsel[k] == j, :].sum(1)
        dh_rec = (dpre * C.eh[None]).sum(-1)
        for s in range(C.g):
            lo, hi = C.off[s], C.off[s + 1]
            dQn = dQ[:, C.dest[:, s], lo:hi]                        # (B, H, blk)
            dh_rec += (dQn * C.uc[None, :, lo:hi]).sum(-1)
            G["uc"][:, lo:hi] += np.einsum("bn,bnr->nr", h_prev, dQn)
        qx = x[t] @ C.ux
        dqx = np.einsum("bnk,nkr->br", dpre, C.vx)
        dx[t] = dqx @ C.ux.T
        if not C.novm:
            dx[t] += (dpre[:, :I] * C.ex[None, :I]).sum(-1)
        G["ux"] += x[t].T @ dqx
        G["vx"] += np.einsum("bnk,br->nkr", dpre, qx)
        for k in range(4):
            G["vc"][:, k, :] += np.einsum("bn,bnr->nr", dpre[:, :, k], Qs[t][:, C.qsel[k], :])
        G["eh"] += np.einsum("bnk,bn->nk", dpre, h_prev)
        if not C.novm:
            G["ex"][:I] += np.einsum("bnk,bn->nk", dpre[:, :I], x[t])
        G["b"] += dpre.sum(0)
    return dx, dh_rec, dc, G


def uncanonicalize_grads(C, G, P_like):
    """Scatter canonical gradients back to the reference's parameter layout, folding the gradients of the
    hoisted diagonal-removal vectors ex/eh into dia, U and V."""
    variant, H, I, rw = C.variant, C.H, C.I, C.rw
    r0 = C.ru[0]
    duc, dvc = G["uc"].copy(), G["vc"].copy()
    dux, dvx = G["ux"].copy(), G["vx"].copy()
    db = G["b"]                                                     # (H,4)
    if variant == V5:   # per-gate tensors, (rank, H) each
        out = {"w": dux[:, :rw], "u": duc[:, :r0]}
        for k in range(4):
            out[f"w{k + 1}"] = dvx[:, k, :rw].T
            out[f"u{k + 1}"] = dvc[:, k, :r0].T
            out[f"bias_{GATE_NAMES_V5[k]}"] = db[:, k][None]
        return out
    out = {}
    if not C.novm:
        ddia_h = G["eh"].sum(1)
        duc[:, :r0] -= np.einsum("nk,nkr->nr", G["eh"], C.vc[:, :, :r0])
        dvc[:, :, :r0] -= G["eh"][:, :, None] * C.uc[:, None, :r0]
        ddia_x = G["ex"][:I].sum(1)
        dux[:, :rw] -= np.einsum("mk,mkr->mr", G["ex"][:I], C.vx[:I, :, :rw])
        dvx[:I, :, :rw] -= G["ex"][:I, :, None] * C.ux[:, None, :rw]
        out["dia_x"] = ddia_x[None]
        out["dia_h"] = ddia_h[None]
    vx_name = "v_x" if variant in (V1, V2, V6) else "w_x"
    out["u_x"] = dux[:, :rw]
    dvx_ref = np.zeros((4, H, rw), dvx.dtype)
    dbx = np.zeros(4 * H, db.dtype)
    dbh = np.zeros(4 * H, db.dtype)
    for k in range(4):
        dvx_ref[C.xchunk[k]] = dvx[:, k, :rw]
        dbx[C.xchunk[k] * H:(C.xchunk[k] + 1) * H] = db[:, k]
        dbh[C.hchunk[k] * H:(C.hchunk[k] + 1) * H] = db[:, k]
    out[vx_name] = dvx_ref.reshape(4 * H, rw)
    if variant in (V1, V3):
        vh_name = "v_h" if variant == V1 else "w_h"
        out["u_h"] = duc[:, :r0]
        out[vh_name] = dvc[:, :, :r0].transpose(1, 0, 2).reshape(4 * H, r0)
        out["b_x"], out["b_h"] = dbx, dbh
        return out
    g = C.g
    Hg = H // g
    sep = "." if variant == V4 else "_"
    for s in range(g):
        du = np.zeros((g, Hg, C.ru[s]), duc.dtype)
        dv = np.zeros((g, C.ru[s], 4 * Hg), duc.dtype)
        lo = C.off[s]
        for n in range(H):
            grp, m = divmod(n, Hg)
            du[C.dest[n, s], m, :] = duc[n, lo:lo + C.ru[s]]
            for k in range(4):
                dv[C.qsel[k, n], :, C.col[k, n]] = dvc[n, k, lo:lo + C.ru[s]]
        out[f"u_h{sep}{s}"] = du
        out[f"v_h{sep}{s}"] = dv
    if variant in (V2, V6):
        out["bias_x"], out["bias_h"] = dbx[None], dbh[None]
    else:
        out["b_x"], out["b_h"] = dbx, dbh
    return out


def unified_run(variant, P, x_tm, h0, c0, dy_tm=None, dhT=None, dcT=None, g=2, dtype=np.float64):
    """Convenience: forward (+ backward when dy given) on time-major numpy inputs."""
    C = canonicalize(variant, P, g=g, dtype=dtype)
    x_tm = np.asarray(x_tm, dtype)
    y, hT, cT, tape = unified_forward(C, x_tm, np.asarray(h0, dtype), np.asarray(c0, dtype))
    if dy_tm is None:
        return y, hT, cT
    dx, dh0, dc0, G = unified_backward(C, tape, np.asarray(dy_tm, dtype), np.asarray(dhT, dtype),
                                       np.asarray(dcT, dtype))
    return y, hT, cT, dx, dh0, dc0, uncanonicalize_grads(C, G, P)


# ----------------------------------------------------------------------------------------------------
# harness counterparts (callers of the boundary)
# ----------------------------------------------------------------------------------------------------

def literal_train_step_har(P, lin_w, lin_b, x, target, variant=V1, g=2):
    """``train.py:61-64`` for one batch: Net.forward (vmlmf.py:352-355) -> cross-entropy (mean) -> backward.
    Returns loss, logits and gradients; the optimizer (Adam, train.py:47,65) is applied by the caller."""
    y, _, _ = literal_sequence(variant, P, x, g=g, time_major=False)
    logits = torch.nn.functional.linear(y[:, -1], lin_w, lin_b).squeeze(1)
    loss = torch.nn.functional.cross_entropy(logits, target.long())
    return loss, logits


def synthetic_batch(B, T, I, seed=1234, classes=6, dtype=np.float32):
    """Seeded synthetic HAR batch: x ~ N(0,1) (B,T,I), integer targets."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.standard_normal((B, T, I)).astype(dtype)
    tgt = rng.integers(0, classes, size=(B,)).astype(np.int64)
    return x, tgt


# ----------------------------------------------------------------------------------------------------
# language-model network around the LM layers
# ----------------------------------------------------------------------------------------------------

def nll_loss_literal(scores, y):
    """``nll_loss`` of the LM loop, op for op (V/src/train_test/lm_test.py:140-153)."""
    batch_size = y.size(1)
    expscores = scores.exp()
    probabilities = expscores / expscores.sum(1, keepdim=True)
    answerprobs = probabilities[range(len(y.reshape(-1))), y.reshape(-1)]
    return torch.mean(-torch.log(answerprobs) * batch_size)


def nll_loss_stable(scores, y, dtype=np.float64):
    """The same quantity around the row maximum (numpy): (loss, dloss/dscores).  Specification of the fused kernels."""
    z = np.asarray(scores, dtype)
    t = np.asarray(y).reshape(-1)
    R = z.shape[0]
    batch_size = np.asarray(y).shape[1]
    m = z.max(1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(z - m).sum(1))
    loss = (lse - z[np.arange(R), t]).sum() * batch_size / R
    g = np.exp(z - lse[:, None])
    g[np.arange(R), t] -= 1.0
    return loss, g * batch_size / R


def literal_lm_forward(sd, x, states, layer_num, factors=None):
    """``Model.forward`` with lstm_type "vmlmf" (V/src/models/vmlmf_lm.py:434-440) on a state-dict style
    mapping name -> tensor: embed.w[x] -> MyVMLSTM layers -> addmm(fc.b, ., fc.w^T).  Returns (scores, states).
    factors=None: dropout 0.  Otherwise layer_num + 1 tensors (T, B, H) of nn.Dropout's factors (0, or 1/(1-p) where the element is
    kept): ``x = self.dropout(x)`` behind the embedding (:435) and behind every layer (:439) as a multiplication by the GIVEN
    factors - which elements torch's generator would have dropped is not part of the algorithm; the carried states are the
    layers' undropped final states (:438)."""
    h = sd["embed.w"][x]
    if factors is not None:
        h = h * factors[0]
    out_states = []
    for i in range(layer_num):
        P = {k.split(".", 2)[2]: v for k, v in sd.items() if k.startswith(f"rnns.{i}.")}
        h, hT, cT = literal_sequence(V3, P, h, states[i][0], states[i][1], time_major=True)
        if factors is not None:
            h = h * factors[i + 1]
        out_states.append((hT, cT))
    scores = torch.addmm(sd["fc.b"], h.view(-1, h.size(2)), sd["fc.w"].t())
    return scores, out_states


# ----------------------------------------------------------------------------------------------------
# dropout factors of the HIP path (csrc/vmlmf_dropout.h): Philox4x32-10 restated in numpy
# ----------------------------------------------------------------------------------------------------
# nn.Dropout(p) (V/src/models/vmlmf_lm.py:402,435,439) zeroes an element with probability p and scales the kept ones by 1/(1-p);
# WHICH elements is the generator's business.  The HIP path draws them from Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel
# random numbers: as easy as 1, 2, 3", SC'11; Random123 1.x) - restated here so a test can (1) pin the generator on Random123's
# published known-answer vectors and (2) reproduce the factors a kernel applied, bit for bit.
PHILOX_KAT = [   # Random123 kat_vectors, philox4x32 10 rounds: counter, key, output
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def philox4x32_10(ctr, key):
    """ctr: (..., 4) uint32 counters, key: (2,) or (..., 2) uint32 -> (..., 4) uint32."""
    c = [np.asarray(ctr)[..., i].astype(np.uint64) for i in range(4)]
    key = np.asarray(key)
    k0 = key[..., 0].astype(np.uint64)
    k1 = key[..., 1].astype(np.uint64)
    M0, M1, W0, W1, mask = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85), np.uint64(0xFFFFFFFF)
    sh = np.uint64(32)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> sh) ^ c[1] ^ k0, p1 & mask, (p0 >> sh) ^ c[3] ^ k1, p0 & mask]
        k0, k1 = (k0 + W0) & mask, (k1 + W1) & mask
    return np.stack(c, axis=-1).astype(np.uint32)


def dropout_factors(seed, offset, site, rows, H, p, Hg=None, gstride=0):
    """(rows, H) float32 factors of dropout site `site` for the generator state (seed, offset): element (position, unit n) maps to
    column (n // Hg) * gstride + n % Hg (identity for Hg = None), counter = (position, column >> 2, site, offset low word), key =
    (seed low word, seed high word + offset high word), dropped iff word[column & 3] < round(p 2^32)."""
    n = np.arange(H)
    col = n if Hg is None or gstride == 0 or Hg >= H else (n // Hg) * gstride + n % Hg
    ctr = np.zeros((rows, H, 4), dtype=np.uint32)
    ctr[..., 0] = np.arange(rows, dtype=np.uint32)[:, None]
    ctr[..., 1] = (col >> 2).astype(np.uint32)[None, :]
    ctr[..., 2] = np.uint32(site)
    ctr[..., 3] = np.uint32(offset & 0xFFFFFFFF)
    key = np.array([seed & 0xFFFFFFFF, ((seed >> 32) + (offset >> 32)) & 0xFFFFFFFF], dtype=np.uint32)
    w = philox4x32_10(ctr, key)
    word = np.take_along_axis(w, np.broadcast_to((col & 3)[None, :, None], (rows, H, 1)), axis=2)[..., 0]
    thresh = min(int(float(np.float32(p)) * 4294967296.0 + 0.5), 4294967295)
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return np.where(word < np.uint32(thresh), np.float32(0.0), scale).astype(np.float32)
