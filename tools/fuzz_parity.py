"""Randomised parity sweep of the sequence entry point against the fp64 literal oracle (tests/hip_util.py's run_hip /
run_literal / compare_all at the tolerances of the test suite): shapes, ranks, variants, optional initial states and upstream
gradients drawn at random from a seed.  Prints every case that fails or that the library refuses, and a summary line.
    python tools/fuzz_parity.py [cases] [seed] [seq|stack|rb|big]
big: batches up to 1100 rows, sequences up to 200 steps, H up to 700 (more rows than CUs, the stand-alone weight-gradient
kernels, the clustered layers)
stack: 2 - 4 like layers through vmlmf_stack (the wavefront launches; initial states of every layer at random) against the
chained literal layers; stacks the library does not cover (vmlmf_stack returns None) are counted, not run."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import vmlmf_oracle as O
from hip_util import run_hip, run_literal, compare_all, ORDER, ranks_of, assert_out, assert_grad
from vmlmf_amd import functional as F

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.Generator(np.random.PCG64(SEED))
VARIANTS = [O.V1, O.V2, O.V3, O.V4, O.V5, O.V6]
MODE = sys.argv[3] if len(sys.argv) > 3 else "seq"
if MODE == "rb":                      # the row-block MFMA kernels wherever an instantiation exists (vmlmf_tune "rb"), else the default choice
    from vmlmf_amd import _lib
    _lib.tune("rb", 1)


def pick(lo, hi, small=0.5):
    """mostly small values, sometimes up to hi"""
    if rng.random() < small:
        return int(rng.integers(lo, min(hi, lo + 12) + 1))
    return int(rng.integers(lo, hi + 1))


def draw():
    v = VARIANTS[int(rng.integers(0, len(VARIANTS)))]
    group = v in (O.V2, O.V4, O.V6)
    H = pick(2, 300, 0.3)
    if group and H % 2:
        H += 1
    novm = v in (O.V5, O.V6)
    lm = v in (O.V3, O.V4)            # the LM layers need input_size == hidden_size (vmlmf_lm.py:243)
    if lm:
        H = min(H, 160)
    I = H if lm else pick(2, 150 if novm else min(H, 150), 0.4)   # (I = 1: the reference's own squeeze() breaks the literal oracle)
    rw = pick(1, 32, 0.3)             # the kernels cover padded ranks up to 32 (beyond: VMLMF_E_UNSUPPORTED, by design)
    ru = [pick(1, 32, 0.3), pick(1, 32, 0.3)] if group else pick(1, 32, 0.3)
    B, T = pick(1, 200, 0.4), pick(1, 40, 0.4)
    if MODE == "big":
        B, T = int(rng.integers(64, 1101)), int(rng.integers(16, 201))
        H = int(rng.integers(32, 701)) + (0 if not group else 0)
        if group and H % 2:
            H += 1
        if lm:
            H = min(H, 660)
            I = H
        else:
            I = int(rng.integers(2, (150 if novm else min(H, 150)) + 1))
        while B * T * H > 24_000_000:      # keeps the float64 oracle of a case within seconds
            T = max(8, T // 2)
    if v == O.V4 and B == 1:
        B = 2                         # (B = 1: the reference's squeeze() in vmlmf_lm.py:257 drops the batch dimension and the layer raises)
    return dict(v=v, B=B, T=T, I=I, H=H, rw=rw, ru=ru, states=bool(rng.random() < 0.5), tm=bool(rng.random() < 0.3),
                dy=bool(rng.random() < 0.8), dh=bool(rng.random() < 0.5), dc=bool(rng.random() < 0.4), seed=int(rng.integers(0, 2**31)))


def run(c):
    r = np.random.Generator(np.random.PCG64(c["seed"]))
    P = O.make_params(c["v"], c["I"], c["H"], c["rw"], c["ru"], seed=c["seed"] % 1000)
    shp = (c["T"], c["B"], c["I"]) if c["tm"] else (c["B"], c["T"], c["I"])
    x = r.standard_normal(shp).astype(np.float32)
    h0 = c0 = None
    if c["states"]:
        h0 = (0.5 * r.standard_normal((c["B"], c["H"]))).astype(np.float32)
        c0 = (0.5 * r.standard_normal((c["B"], c["H"]))).astype(np.float32)
    oshp = shp[:2] + (c["H"],)
    dy = r.standard_normal(oshp).astype(np.float32) if c["dy"] else None
    dhT = r.standard_normal((c["B"], c["H"])).astype(np.float32) if c["dh"] else None
    dcT = r.standard_normal((c["B"], c["H"])).astype(np.float32) if c["dc"] else None
    if dy is None and dhT is None and dcT is None:
        dhT = r.standard_normal((c["B"], c["H"])).astype(np.float32)
    got = run_hip(c["v"], P, x, h0, c0, dy, dhT, dcT, time_major=c["tm"])
    ref = run_literal(c["v"], P, x, h0, c0, dy, dhT, dcT, time_major=c["tm"])
    compare_all(got, ref, "fuzz")


class NotCovered(Exception):
    pass


def run_stack(c):
    """L like layers: vmlmf_stack on the GPU, the literal layers chained on the CPU (float64)."""
    r = np.random.Generator(np.random.PCG64(c["seed"]))
    v, L, B, T, I, H = c["v"], c["L"], c["B"], c["T"], c["I"], c["H"]
    Hs = c.get("Hs") or [H] * L          # (round 6: the layers of a stack may differ in hidden size; no initial states then)
    Ps = [O.make_params(v, I if l == 0 else Hs[l - 1], Hs[l], c["rw"], c["ru"], seed=c["seed"] % 1000 + l) for l in range(L)]
    shp = (T, B, I) if c["tm"] else (B, T, I)
    x = r.standard_normal(shp).astype(np.float32)
    dy = r.standard_normal(shp[:2] + (Hs[-1],)).astype(np.float32)
    st = None
    if c["states"] and len(set(Hs)) == 1:
        st = [(0.5 * r.standard_normal((L, B, H))).astype(np.float32) for _ in range(2)]
    names = ORDER[v]
    rw, ru, g = ranks_of(v, Ps[0])
    params = [[torch.tensor(np.asarray(P[k]), device="cuda").requires_grad_(True) for k in names] for P in Ps]
    xt = torch.tensor(x, device="cuda").requires_grad_(True)
    h0 = c0 = None
    if st is not None:
        h0, c0 = (torch.tensor(a, device="cuda").requires_grad_(True) for a in st)
    out = F.vmlmf_stack(v, xt, params, rw, ru, g=g, time_major=c["tm"], h0=h0, c0=c0)
    if out is None:
        raise NotCovered()
    y, hTs, cTs = out[:3]
    dhT = [r.standard_normal((B, Hs[l])).astype(np.float32) for l in range(L)]
    loss = (y * torch.tensor(dy, device="cuda")).sum()
    for l in range(L):
        loss = loss + (hTs[l] * torch.tensor(dhT[l], device="cuda")).sum()
    loss.backward()
    torch.cuda.synchronize()
    # oracle
    Pt = [O.to_torch(P, dtype=torch.float64, requires_grad=True) for P in Ps]
    xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    h0r = c0r = None
    if st is not None:
        h0r, c0r = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in st)
    cur, hr = xr, []
    for l in range(L):
        cur, hT, cT = O.literal_sequence(v, Pt[l], cur, None if h0r is None else h0r[l], None if c0r is None else c0r[l],
                                         time_major=c["tm"], v4_scratch_rows=B)
        hr.append(hT)
    lr = (cur * torch.tensor(dy, dtype=torch.float64)).sum()
    for l in range(L):
        lr = lr + (hr[l] * torch.tensor(dhT[l], dtype=torch.float64)).sum()
    lr.backward()
    assert_out(y.detach().cpu().numpy(), cur.detach().numpy(), "stack.y")
    for l in range(L):
        assert_out(hTs[l].detach().cpu().numpy(), hr[l].detach().numpy(), f"stack.hT[{l}]")
    assert_grad(xt.grad.cpu().numpy(), xr.grad.numpy(), "stack.dx")
    if st is not None:
        assert_grad(h0.grad.cpu().numpy(), h0r.grad.numpy(), "stack.dh0")
        assert_grad(c0.grad.cpu().numpy(), c0r.grad.numpy(), "stack.dc0")
    for l in range(L):
        for k, p in zip(names, params[l]):
            assert_grad(p.grad.cpu().numpy(), Pt[l][k].grad.numpy(), f"stack.G[{l}].{k}")


def draw_stack():
    c = draw()
    while c["v"] == O.V4:             # (the flat V4 layout is not on the wavefront kernels)
        c = draw()
    c["L"] = int(rng.integers(2, 5))
    c["H"] = min(c["H"], 256)
    if c["v"] in (O.V2, O.V6) and c["H"] % 2:
        c["H"] += 1
    if c["v"] == O.V3:
        c["I"] = c["H"]
    elif c["v"] == O.V5 or c["v"] == O.V6:
        pass
    else:
        c["I"] = min(c["I"], c["H"])
    c["B"], c["T"] = min(c["B"], 128), min(c["T"], 30)
    if c["v"] in (O.V1, O.V5) and rng.random() < 0.35:      # growing hidden sizes (a VMLMF cell needs input_size <= hidden_size)
        hs = sorted(int(min(256, max(c["I"] if c["v"] == O.V1 else 2, pick(2, 256, 0.2)))) for _ in range(c["L"]))
        c["Hs"], c["H"] = hs, hs[0]
        if c["v"] == O.V1:
            c["I"] = min(c["I"], hs[0])
    return c


ok = refused = failed = uncovered = 0
t0 = time.time()
for n in range(N):
    c = draw_stack() if MODE == "stack" else draw()
    try:
        if MODE == "stack":
            run_stack(c)
        else:
            run(c)
        ok += 1
    except NotCovered:
        uncovered += 1
    except AssertionError as e:
        failed += 1
        print("FAIL", c, str(e)[:600], flush=True)
    except RuntimeError as e:
        msg = str(e)
        if "error -3" in msg or "error -2" in msg or "unsupported" in msg.lower():
            refused += 1
            print("refused", c, msg[:160], flush=True)
        else:
            failed += 1
            print("ERROR", c, msg[:400], flush=True)
            torch.cuda.synchronize()
    except Exception as e:      # noqa: BLE001
        failed += 1
        print("EXC", c, traceback.format_exc()[-600:], flush=True)
print(f"fuzz {MODE} seed {SEED}: {N} cases, {ok} ok, {refused} refused by the library, {uncovered} not covered by the wavefront "
      f"launches, {failed} FAILED, {time.time() - t0:.0f} s")
sys.exit(1 if failed else 0)
