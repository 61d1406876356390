"""GPU parity of rec4_bwd_kernel (vmlmf_rec4.inc): the backward recurrence that forms the weight gradients inside the rows' own
workgroups (accumulator waves on fp32 MFMA, K = two time steps) - no dpre tape, no weight-gradient launch.  Covers what the
pairing of time steps and the LDS rings can get wrong: odd and even lengths, the shortest sequences (two and three steps),
one / two / three waves of hidden units, padded ranks 8 and 16, narrow and rank-wide inputs, initial states present or not,
gradients arriving through dy, dhT, dcT or only some of them, batches below and above the CU count, repeated launches over
the same buffers, and that the kernel IS the one that ran (no weight-gradient launch in the library's own kernel counters)."""
import ctypes

import numpy as np
import pytest
import torch

import vmlmf_oracle as O
from hip_util import run_hip, run_literal, compare_all

pytestmark = pytest.mark.gpu


def _case(variant, B, T, I, H, rw, ru, seed, states=True):
    rng = np.random.Generator(np.random.PCG64(seed))
    P = O.make_params(variant, I, H, rw, ru[0], seed=seed + 1)
    x = rng.standard_normal((B, T, I)).astype(np.float32)
    h0 = (0.4 * rng.standard_normal((B, H))).astype(np.float32) if states else None
    c0 = (0.4 * rng.standard_normal((B, H))).astype(np.float32) if states else None
    dy = rng.standard_normal((B, T, H)).astype(np.float32)
    dhT = rng.standard_normal((B, H)).astype(np.float32)
    dcT = rng.standard_normal((B, H)).astype(np.float32)
    return P, x, h0, c0, dy, dhT, dcT


def _kernel_counts(fn):
    """Launch counts of the library's internal kernels while fn() runs (vmlmf_profile_*: event pairs on the launch stream)."""
    from vmlmf_amd import _lib
    lib = _lib.lib()
    usec = (ctypes.c_float * _lib.NKERNELS)()
    cnt = (ctypes.c_int32 * _lib.NKERNELS)()
    lib.vmlmf_profile_enable((1 << _lib.NKERNELS) - 1)
    try:
        out = fn()
        torch.cuda.synchronize()
        lib.vmlmf_profile_read(usec, cnt, 1)
    finally:
        lib.vmlmf_profile_enable(0)
    return out, {lib.vmlmf_kernel_name(k).decode(): cnt[k] for k in range(_lib.NKERNELS)}


@pytest.fixture
def inrow():
    from vmlmf_amd import _lib
    _lib.tune("inrow", 1)
    yield
    _lib.tune("inrow", -1)


# (variant, B, T, I, H, w_rank, u_ranks, states)
CASES = [
    (O.V1, 64, 40, 9, 180, 16, [16], True),      # the headline layer, shorter
    (O.V1, 8, 41, 9, 180, 16, [16], False),      # odd length, zero initial states
    (O.V1, 5, 2, 9, 180, 16, [16], True),        # one pair of steps
    (O.V1, 5, 3, 9, 180, 16, [16], True),        # a pair and a single step
    (O.V1, 37, 9, 5, 70, 6, [12], True),         # two waves of units, padded w_rank 8, odd batch
    (O.V1, 16, 6, 8, 64, 8, [8], True),          # one wave, rank 8 on both sides
    (O.V1, 9, 12, 16, 130, 16, [10], False),     # input as wide as the padded rank
    (O.V1, 12, 10, 3, 192, 4, [16], True),       # every thread slot a hidden unit
    (O.V1, 300, 6, 7, 100, 8, [16], True),       # more rows than CUs
    (O.V5, 20, 7, 10, 80, 12, [12], True),       # MyLSTMCell low-rank (no vm vectors)
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "v%d_B%d_T%d_I%d_H%d_r%d_%d_%s" % (c[0], c[1], c[2], c[3], c[4], c[5], c[6][0], "st" if c[7] else "z"))
def test_in_row_weight_gradients_vs_oracle(case, inrow):
    variant, B, T, I, H, rw, ru, states = case
    P, x, h0, c0, dy, dhT, dcT = _case(variant, B, T, I, H, rw, ru, seed=7 * B + T, states=states)
    got, counts = _kernel_counts(lambda: run_hip(variant, P, x, h0, c0, dy, dhT, dcT, need_dx=False))
    assert counts["rec_bwd_kernel"] == 1 and counts["wgrad_mfma_kernel"] == 0 and counts["dqx_dx_kernel"] == 0, counts
    ref = run_literal(variant, P, x, h0, c0, dy, dhT, dcT)
    compare_all(got, ref, "inrow")
    # the forms it replaces, on the same inputs: same tolerances, and close to one another
    from vmlmf_amd import _lib
    _lib.tune("inrow", 0)
    old, counts0 = _kernel_counts(lambda: run_hip(variant, P, x, h0, c0, dy, dhT, dcT, need_dx=False))
    _lib.tune("inrow", 1)
    compare_all(old, ref, "stand-alone / riding")
    assert np.array_equal(old["y"], got["y"])
    for k in ("dh0", "dc0"):
        if k in got:
            assert np.array_equal(old[k], got[k]), k          # the recurrence itself is the same arithmetic


@pytest.mark.parametrize("case", [(O.V1, 300, 6, 7, 100, 8, [16], True), (O.V1, 259, 5, 9, 180, 16, [16], True), (O.V1, 7, 4, 8, 64, 8, [8], False),
                                  (O.V1, 64, 9, 9, 180, 16, [16], True)],
                         ids=lambda c: "B%d_T%d_H%d" % (c[1], c[2], c[4]))
def test_batches_beyond_the_cu_count_vs_oracle(case, inrow):
    """Batches with more rows than the chip has CUs (workgroups queue up), odd batches, a batch below the riding range forced onto
    the in-row form: one backward launch, no weight-gradient launch, every gradient against the oracle."""
    variant, B, T, I, H, rw, ru, states = case
    P, x, h0, c0, dy, dhT, dcT = _case(variant, B, T, I, H, rw, ru, seed=3 * B + T, states=states)
    got, counts = _kernel_counts(lambda: run_hip(variant, P, x, h0, c0, dy, dhT, dcT, need_dx=False))
    assert counts["rec_bwd_kernel"] == 1 and counts["wgrad_mfma_kernel"] == 0, counts
    compare_all(got, run_literal(variant, P, x, h0, c0, dy, dhT, dcT), "inrow.large")


@pytest.mark.parametrize("which", ["dy", "dhT", "dcT", "dy+dcT"])
def test_partial_upstream_gradients(which, inrow):
    """HAR feeds the loss from the last step only (dhT through the classifier); the LM layers from every step."""
    variant, B, T, I, H, rw, ru = O.V1, 24, 11, 9, 180, 16, [16]
    P, x, h0, c0, dy, dhT, dcT = _case(variant, B, T, I, H, rw, ru, seed=99)
    kw = dict(dy=dy if "dy" in which else None, dhT=dhT if "dhT" in which else None, dcT=dcT if "dcT" in which else None)
    got = run_hip(variant, P, x, h0, c0, need_dx=False, **kw)
    compare_all(got, run_literal(variant, P, x, h0, c0, **kw), "inrow." + which)


def test_repeated_launches_over_the_same_buffers_repeat_bit_for_bit(inrow):
    """Fixed-order sums everywhere: the same inputs give the same bits; other data in the same buffers leaves no trace."""
    variant, B, T, I, H, rw, ru = O.V1, 70, 17, 9, 180, 16, [16]
    P, x, h0, c0, dy, dhT, dcT = _case(variant, B, T, I, H, rw, ru, seed=5)
    a = run_hip(variant, P, x, h0, c0, dy, dhT, dcT, need_dx=False)
    P2, x2, h02, c02, dy2, dhT2, dcT2 = _case(variant, B, T, I, H, rw, ru, seed=6)
    run_hip(variant, P2, x2, h02, c02, dy2, dhT2, dcT2, need_dx=False)
    b = run_hip(variant, P, x, h0, c0, dy, dhT, dcT, need_dx=False)
    for k in a["G"]:
        assert np.array_equal(a["G"][k], b["G"][k]), k


def test_automatic_choice_and_the_input_gradient_fallback():
    """Automatic: batches beyond the riding workers' range take the in-row form; a layer whose input needs a gradient never
    does (dx is formed from dpre by dqx_dx_kernel, which needs the tape this form does not write)."""
    variant, I, H, rw, ru = O.V1, 9, 180, 16, [16]
    for B, expect_inrow in ((64, False), (128, True)):
        P, x, h0, c0, dy, dhT, dcT = _case(variant, B, 8, I, H, rw, ru, seed=B)
        got, counts = _kernel_counts(lambda: run_hip(variant, P, x, h0, c0, dy, dhT, dcT, need_dx=False))
        assert (counts["wgrad_mfma_kernel"] == 0) and counts["rec_bwd_kernel"] == 1
        compare_all(got, run_literal(variant, P, x, h0, c0, dy, dhT, dcT), f"auto.B{B}")
    P, x, h0, c0, dy, dhT, dcT = _case(variant, 128, 8, I, H, rw, ru, seed=3)
    got, counts = _kernel_counts(lambda: run_hip(variant, P, x, h0, c0, dy, dhT, dcT, need_dx=True))
    assert counts["wgrad_mfma_kernel"] == 1 and counts["dqx_dx_kernel"] == 1, counts
    compare_all(got, run_literal(variant, P, x, h0, c0, dy, dhT, dcT), "auto.dx")


@pytest.mark.parametrize("rec3", [0, 7])
@pytest.mark.parametrize("case", [(O.V1, 64, 40, 9, 180, 16, [16]), (O.V1, 37, 9, 5, 70, 6, [12]), (O.V1, 300, 6, 7, 100, 8, [16]),
                                  (O.V1, 12, 10, 3, 192, 4, [16])], ids=lambda c: "B%d_T%d_H%d" % (c[1], c[2], c[4]))
def test_both_selections_of_the_third_kernel_form(case, rec3):
    """vmlmf_tune("rec3", 0) (rec_fwd_kernel / rec_bwd_kernel everywhere) and 7 (rec3_fwd_kernel / rec3_bwd_kernel wherever
    they cover the layer) were only ever exercised by the default mask 6.  Both on the standard shapes against the oracle; the two
    forwards sum in slightly different orders: their outputs agree to rounding."""
    from vmlmf_amd import _lib
    variant, B, T, I, H, rw, ru = case
    P, x, h0, c0, dy, dhT, dcT = _case(variant, B, T, I, H, rw, ru, seed=5 * B + T)
    ref = run_literal(variant, P, x, h0, c0, dy, dhT, dcT)
    _lib.tune("rec3", rec3)
    try:
        got = run_hip(variant, P, x, h0, c0, dy, dhT, dcT)
    finally:
        _lib.tune("rec3", 6)
    compare_all(got, ref, f"rec3={rec3}")
    base = run_hip(variant, P, x, h0, c0, dy, dhT, dcT)
    compare_all(base, ref, "rec3=6")
    for k in ("y", "hT", "cT"):
        assert np.abs(got[k] - base[k]).max() <= 2e-6 * max(np.abs(base[k]).max(), 1.0), k


def test_parameter_tensors_that_are_not_16_byte_aligned_take_the_packed_images():
    """Direct mode reads parameter rows as 16-byte loads; a tensor that starts 4 bytes into an allocation (a view into a flat
    buffer) must fall back to pack_kernel's images - same results against the oracle, and a pack launch in the counters."""
    import torch
    from hip_util import ORDER
    from vmlmf_amd import vmlmf_sequence
    variant, B, T, I, H, rw, ru = O.V1, 64, 12, 9, 180, 16, [16]
    P, x, h0, c0, dy, dhT, dcT = _case(variant, B, T, I, H, rw, ru, seed=17)
    names = ORDER[variant]
    total = sum(int(np.asarray(P[k]).size) + 2 for k in names)
    flat = torch.zeros(total + 4, device="cuda")
    params, o = [], 1                       # every tensor starts at an odd float offset
    for k in names:
        a = np.asarray(P[k], np.float32)
        v = flat[o:o + a.size].view(a.shape)
        v.copy_(torch.tensor(a))
        params.append(v.detach().requires_grad_(True))
        o += a.size + (1 if (o + a.size) % 2 == 0 else 2)
    assert any(p.data_ptr() % 16 for p in params)

    def run():
        xt = torch.tensor(x, device="cuda")
        y, hT, cT = vmlmf_sequence(variant, xt, torch.tensor(h0, device="cuda"), torch.tensor(c0, device="cuda"), params, rw, ru, g=1)
        ((y * torch.tensor(dy, device="cuda")).sum() + (hT * torch.tensor(dhT, device="cuda")).sum()).backward()
        return y
    y, counts = _kernel_counts(run)
    assert counts["pack_kernel"] >= 1, counts
    ref = run_literal(variant, P, x, h0, c0, dy, dhT, None)
    from hip_util import assert_out, assert_grad
    assert_out(y.detach().cpu().numpy(), ref["y"], "y")
    for k, p in zip(names, params):
        assert_grad(p.grad.cpu().numpy(), ref["G"][k], k)
