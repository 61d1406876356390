"""GPU parity of the weight-gradient workers that ride on the recurrent backward launch (vmlmf_rec_bwd.inc RIDE,
vmlmf_atb.inc): layers whose input is not wider than the padded w_rank (the x-fold), batch <= 96, fp32, on the persistent
VALU kernels.  The workers read rows that other workgroups write during the same launch, so besides the oracle comparison
the tests look for what such a protocol gets wrong: stale progress words or stale rows when the same buffers are used again
with other data, the last steps of a sequence (published at the end of the launch), very short sequences (shorter than the
lag of the progress words), odd batches, a second backward over the same tape."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import vmlmf_oracle as O
from hip_util import run_hip, run_literal, compare_all, assert_grad, ORDER, ranks_of

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _case(variant, B, T, I, H, rw, ru, seed, states=True):
    rng = np.random.Generator(np.random.PCG64(seed))
    P = O.make_params(variant, I, H, rw, ru if variant in (O.V2, O.V4, O.V6) else ru[0], seed=seed + 1)
    x = rng.standard_normal((B, T, I)).astype(np.float32)
    h0 = (0.4 * rng.standard_normal((B, H))).astype(np.float32) if states else None
    c0 = (0.4 * rng.standard_normal((B, H))).astype(np.float32) if states else None
    dy = rng.standard_normal((B, T, H)).astype(np.float32)
    dhT = rng.standard_normal((B, H)).astype(np.float32)
    dcT = rng.standard_normal((B, H)).astype(np.float32)
    return P, x, h0, c0, dy, dhT, dcT


# (variant, B, T, I, H, w_rank, u_ranks): every one has I <= padded w_rank and B <= 128, i.e. rides
RIDING = [
    (O.V1, 64, 40, 9, 180, 16, [16]),       # the headline layer (its full length: test_gpu_parity, golden vectors)
    (O.V1, 96, 24, 8, 100, 8, [16]),        # largest riding batch
    (O.V1, 128, 10, 8, 100, 8, [16]),       # just above it: the stand-alone kernel
    (O.V1, 37, 9, 5, 70, 6, [12]),          # odd batch: whole-step chunks, row pairs straddle steps
    (O.V1, 3, 1, 4, 40, 4, [8]),            # one step: the only progress word is the final one
    (O.V1, 16, 2, 9, 64, 8, [8]),           # too few chunks to ride (fewer than four): the stand-alone kernel
    (O.V1, 16, 5, 9, 64, 8, [8]),
    (O.V1, 64, 2, 9, 64, 16, [8]),          # rides with fewer steps than the lag of the progress words: only the final word counts
    (O.V1, 96, 3, 7, 100, 8, [16]),         # as many steps as the lag
    (O.V1, 64, 4, 9, 180, 16, [16]),
    (O.V2, 32, 12, 9, 180, 16, [16, 16]),   # group cell: two 32-wide tiles of B columns
    (O.V1, 8, 7, 30, 130, 32, [32]),        # rank 32 on both sides
    (O.V3, 24, 6, 24, 24, 24, [8]),         # PTB cell (input size = hidden size), small enough to fold
    (O.V6, 20, 6, 10, 80, 12, [12, 12]),    # group cell without vm
    (O.V5, 20, 6, 10, 80, 12, [12]),        # MyLSTMCell low-rank (no vm)
]


@pytest.mark.parametrize("case", RIDING, ids=lambda c: "v%d_B%d_T%d_I%d_H%d_r%d_%s" % (c[0], c[1], c[2], c[3], c[4], c[5], "_".join(map(str, c[6]))))
def test_riding_workers_vs_oracle(case):
    variant, B, T, I, H, rw, ru = case
    P, x, h0, c0, dy, dhT, dcT = _case(variant, B, T, I, H, rw, ru, seed=11 * B + T)
    got = run_hip(variant, P, x, h0, c0, dy, dhT, dcT)
    ref = run_literal(variant, P, x, h0, c0, dy, dhT, dcT)
    compare_all(got, ref, "ride")


def test_random_riding_shapes_vs_oracle():
    """Seeded random layers that ride (I <= padded w_rank, B <= 96): hidden sizes across the wave counts, every padded rank
    of the instantiations (8 / 16 / 24 / 32), one and two groups, cells with and without vm, odd batches, with and without
    initial states and upstream gradients of the final states."""
    rng = np.random.Generator(np.random.PCG64(2024))
    for it in range(24):
        variant = [O.V1, O.V2, O.V5, O.V6][it % 4]
        two = variant in (O.V2, O.V6)
        H = int(rng.choice([24, 40, 64, 90, 128, 180, 200, 256]))
        if two and H % 2:
            H += 1
        rw = int(rng.choice([3, 8, 12, 16, 24, 30]))
        I = int(rng.integers(1, ((rw + 7) // 8) * 8 + 1))           # <= padded w_rank
        I = max(2, min(I, H))                                       # (the reference's group cells squeeze() a 1-wide input away)
        ru = [int(rng.choice([4, 8, 13, 16])) for _ in range(2)] if two else [int(rng.choice([4, 8, 16, 21, 32]))]
        B = int(rng.integers(2, 97))
        T = int(rng.integers(1, 30))
        P, x, h0, c0, dy, dhT, dcT = _case(variant, B, T, I, H, rw, ru, seed=300 + it, states=bool(it & 1))
        if it % 3 == 0:
            dhT, dcT = None, None
        got = run_hip(variant, P, x, h0, c0, dy, dhT, dcT)
        ref = run_literal(variant, P, x, h0, c0, dy, dhT, dcT)
        compare_all(got, ref, "it%d v%d B%d T%d I%d H%d rw%d ru%s" % (it, variant, B, T, I, H, rw, ru))


def test_same_buffers_other_data_every_launch():
    """One module, the same shapes (so the caching allocator hands back the same tape and workspace), new inputs and new
    upstream gradients every iteration: a stale progress word or a stale row of an earlier launch would show up in the
    weight gradients, which are compared with the oracle each time."""
    variant, B, T, I, H, rw, ru = O.V1, 64, 48, 9, 180, 16, [16]
    for it in range(6):
        P, x, h0, c0, dy, dhT, dcT = _case(variant, B, T, I, H, rw, ru, seed=100 + it, states=(it % 2 == 0))
        got = run_hip(variant, P, x, h0, c0, dy, dhT, dcT)
        ref = run_literal(variant, P, x, h0, c0, dy, dhT, dcT)
        compare_all(got, ref, "it%d" % it)


def test_second_backward_over_the_same_tape():
    """retain_graph: the progress words go back to zero after a launch (reduce_cg_kernel), so a second backward over the
    same tape waits for its own rows instead of finding the words of the first one."""
    from vmlmf_amd import vmlmf_sequence
    variant, B, T, I, H, rw, ru = O.V1, 48, 20, 9, 120, 16, [16]
    P, x, h0, c0, dy, dhT, dcT = _case(variant, B, T, I, H, rw, ru, seed=5, states=False)
    names = ORDER[variant]
    params = [torch.tensor(np.asarray(P[k]), dtype=torch.float32, device="cuda").requires_grad_(True) for k in names]
    xt = torch.tensor(x, device="cuda")
    r_w, r_u, g = ranks_of(variant, P)
    y, hT, cT = vmlmf_sequence(variant, xt, None, None, params, r_w, r_u, g=g)
    dy1 = torch.tensor(dy, device="cuda")
    dy2 = torch.tensor(dy[::-1].copy(), device="cuda")
    g1 = torch.autograd.grad((y * dy1).sum(), params, retain_graph=True)
    g2 = torch.autograd.grad((y * dy2).sum(), params)
    ref1 = run_literal(variant, P, x, None, None, dy, None, None)
    ref2 = run_literal(variant, P, x, None, None, dy[::-1].copy(), None, None)
    for k, a, b in zip(names, g1, g2):
        assert_grad(a.cpu().numpy(), ref1["G"][k], "first." + k)
        assert_grad(b.cpu().numpy(), ref2["G"][k], "second." + k)


def test_results_repeat_bit_for_bit():
    """Fixed chunk -> worker assignment and a fixed-order sum of the partial blocks: the same inputs give the same bits,
    however the workers were scheduled."""
    variant, B, T, I, H, rw, ru = O.V1, 64, 64, 9, 180, 16, [16]
    P, x, h0, c0, dy, dhT, dcT = _case(variant, B, T, I, H, rw, ru, seed=77)
    first = run_hip(variant, P, x, h0, c0, dy, dhT, dcT)
    for _ in range(4):
        again = run_hip(variant, P, x, h0, c0, dy, dhT, dcT)
        for k in first["G"]:
            assert np.array_equal(first["G"][k], again["G"][k]), k
        assert np.array_equal(first["dx"], again["dx"])


def test_backward_under_load_from_another_stream():
    """A second stream keeps the chip busy with large products while the backward launch with its workers runs: the rows'
    workgroups are dispatched before the workers' (lower block ids), so the workers can be late but never wait for rows that
    have no CU; the gradients are the same bits as in the quiet run."""
    from vmlmf_amd import vmlmf_sequence
    variant, B, T, I, H, rw, ru = O.V1, 64, 96, 9, 180, 16, [16]
    P, x, h0, c0, dy, dhT, dcT = _case(variant, B, T, I, H, rw, ru, seed=21, states=False)
    names = ORDER[variant]
    params = [torch.tensor(np.asarray(P[k]), dtype=torch.float32, device="cuda").requires_grad_(True) for k in names]
    xt = torch.tensor(x, device="cuda")
    dyt = torch.tensor(dy, device="cuda")
    r_w, r_u, g = ranks_of(variant, P)

    def grads():
        y, hT, cT = vmlmf_sequence(variant, xt, None, None, params, r_w, r_u, g=g)
        return [t.clone() for t in torch.autograd.grad((y * dyt).sum(), params)]

    quiet = grads()
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda")
    for i in range(6):
        with torch.cuda.stream(side):
            for _ in range(3):
                a = (a @ a).clamp_(-1, 1)
        busy = grads()
        for k, q, b in zip(names, quiet, busy):
            assert torch.equal(q, b), (i, k)
    torch.cuda.synchronize()
    ref = run_literal(variant, P, x, None, None, dy, None, None)
    for k, q in zip(names, quiet):
        assert_grad(q.cpu().numpy(), ref["G"][k], k)


@pytest.mark.parametrize("env", [{"VMLMF_WRIDE": "0"}, {"VMLMF_WRIDE_K": "5"}, {"VMLMF_WRIDE_RC": "16"}, {"VMLMF_WRIDE_RC": "64"},
                                 {"VMLMF_WRIDE_LAG": "1"}, {"VMLMF_WRIDE_LAG": "6"}, {"VMLMF_WRIDE_MAXB": "8"}],
                         ids=lambda e: "_".join(f"{k[6:]}{v}" for k, v in e.items()))
def test_switches_of_the_riding_workers(env):
    """VMLMF_WRIDE* (read once when the library loads): off, few workers, other chunk sizes, other lags of the progress words,
    a batch limit below the test's batch - in a fresh interpreter each, against the oracle."""
    code = (
        "import sys; sys.path[:0] = [%r, %r, %r]\n"
        "import numpy as np, vmlmf_oracle as O\n"
        "from hip_util import run_hip, run_literal, compare_all\n"
        "for variant, B, T, I, H, rw, ru in [(O.V1, 64, 40, 9, 180, 16, [16]), (O.V1, 10, 3, 12, 40, 12, [40]), (O.V2, 32, 9, 9, 96, 16, [16, 16])]:\n"
        "    rng = np.random.Generator(np.random.PCG64(5))\n"
        "    P = O.make_params(variant, I, H, rw, ru if variant == O.V2 else ru[0], seed=7)\n"
        "    x = rng.standard_normal((B, T, I)).astype(np.float32)\n"
        "    dy = rng.standard_normal((B, T, H)).astype(np.float32)\n"
        "    dhT = rng.standard_normal((B, H)).astype(np.float32)\n"
        "    for it in range(2):\n"
        "        compare_all(run_hip(variant, P, x, None, None, dy, dhT, None), run_literal(variant, P, x, None, None, dy, dhT, None), 'switch')\n"
        "print('ok')\n") % (ROOT, os.path.join(ROOT, "oracle"), HERE)
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]


def test_a_worker_that_gives_up_surfaces_as_an_error_code_not_only_as_nan():
    """Failure path of the riding workers (verdict r2: "in-kernel protocol failures are silent").  With the number of looks a
    worker takes at the rows' progress words cut to one (vmlmf_tune("test_wride_spin", 1): the rows need ~0.4 us a step, so
    the first chunk is never there at the first look) the workers give up: the parameter gradients are NaN - never a
    plausible wrong number - the launch ends, and the library reports VMLMF_E_PROTOCOL: from vmlmf_check_status() once the
    stream is synchronised, and from the next forward / backward call on the device.  Afterwards the same call works again."""
    from vmlmf_amd import _lib
    variant, B, T, I, H, rw, ru = O.V1, 64, 40, 9, 180, 16, [16]
    P, x, h0, c0, dy, dhT, dcT = _case(variant, B, T, I, H, rw, ru, seed=5)
    good = run_hip(variant, P, x, h0, c0, dy, dhT, dcT)
    _lib.check_status()                                  # nothing pending
    _lib.tune("test_wride_spin", 1)
    try:
        bad = run_hip(variant, P, x, h0, c0, dy, dhT, dcT)   # returns normally: launches are asynchronous
        torch.cuda.synchronize()
        assert not np.all(np.isfinite(bad["G"]["v_h"])), "the workers were expected to give up"
        # outputs and the input / state gradients do not come from the workers
        assert np.array_equal(bad["y"], good["y"]) and np.array_equal(bad["dx"], good["dx"])
        with pytest.raises(_lib.VmlmfError) as ei:
            _lib.check_status()
        assert ei.value.code == _lib.E_PROTOCOL and "progress words" in str(ei.value)
        _lib.check_status()                              # reported once, then cleared
        # ... and unsynchronised, the NEXT call on the device is the one that fails
        run_hip(variant, P, x, h0, c0, dy, dhT, dcT)
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError) as ei:     # (_lib.VmlmfError through ctypes, c10::Error through the C++ binding)
            run_hip(variant, P, x, h0, c0, dy, dhT, dcT)
        assert "error -6" in str(ei.value) and "progress words" in str(ei.value)
    finally:
        _lib.tune("test_wride_spin", 0)
        torch.cuda.synchronize()
        try:
            _lib.check_status()
        except _lib.VmlmfError:
            pass
    again = run_hip(variant, P, x, h0, c0, dy, dhT, dcT)
    for k in good["G"]:
        assert np.array_equal(again["G"][k], good["G"][k]), k


def test_riding_can_be_switched_off_and_rearmed_at_run_time():
    """vmlmf_tune("wride", 0): the stand-alone weight-gradient kernel behind the recurrence - what the library switches to by
    itself once a worker has given up under the production bound (a GPU shared with other processes can starve the workers of
    their rows); vmlmf_tune("wride", 1) re-arms the riding form.  Both against the oracle; the riding form bit-identical before
    and after."""
    from vmlmf_amd import _lib
    variant, B, T, I, H, rw, ru = O.V1, 64, 40, 9, 180, 16, [16]
    P, x, h0, c0, dy, dhT, dcT = _case(variant, B, T, I, H, rw, ru, seed=6)
    ref = run_literal(variant, P, x, h0, c0, dy, dhT, dcT)
    riding = run_hip(variant, P, x, h0, c0, dy, dhT, dcT)
    _lib.tune("wride", 0)
    try:
        alone = run_hip(variant, P, x, h0, c0, dy, dhT, dcT)
    finally:
        _lib.tune("wride", 1)
    again = run_hip(variant, P, x, h0, c0, dy, dhT, dcT)
    compare_all(alone, ref, "stand-alone")
    compare_all(riding, ref, "riding")
    for k in riding["G"]:
        assert np.array_equal(again["G"][k], riding["G"][k]), k
    # the two forms sum in different orders: close, and for this shape not bit-identical (which shows the switch did something)
    assert any(not np.array_equal(alone["G"][k], riding["G"][k]) for k in riding["G"])


def test_processes_sharing_the_gpu_get_an_error_code_and_recover():
    """tools/stress_shared_gpu.py: four processes queue forward + backward bursts on ONE GPU.  Their launches overlap, and a
    riding worker can then wait for rows that other processes' workgroups keep off the CUs (seen about once per 10^4 steps with
    four processes, within the first hundred with eight).  Whatever happens in a given run, every process must end with exit code
    0: a give-up is reported as VMLMF_E_PROTOCOL for that step, the process goes on with the stand-alone weight-gradient kernel,
    and no step without an error code differs from the quiet gradients; nothing hangs."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_shared_gpu.py"), "4", "3"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "exit codes [0, 0, 0, 0]" in r.stdout


def _har_net(seed=0):
    from vmlmf_amd import MyLSTM, MyVMLMFCell, Net
    torch.manual_seed(seed)
    net = Net(9, layer_sizes=[180], w_rank=16, u_rank=[16], model=MyLSTM, cell=MyVMLMFCell).cuda()
    # (the headline length: the rows need ~66 us, far longer than the two looks a worker takes under test_wride_spin = 1 - with 40
    #  steps a graph replay finished the whole recurrence inside the first nap and nothing gave up)
    x = torch.randn(64, 128, 9, device="cuda")
    t = torch.randint(0, 18, (64,), device="cuda")
    return net, x, t


def _clear_status():
    from vmlmf_amd import _lib
    torch.cuda.synchronize()
    try:
        _lib.check_status()
    except _lib.VmlmfError:
        pass
    _lib.tune("clear_health", 0)     # (earlier tests wrote NaN gradients that no guarded optimizer step consumed)


def test_nan_gradients_of_a_failed_step_never_reach_the_parameters():
    """nothing between a failing backward and the optimizer looked at the status word, so the NaN gradients
    of step N were applied before step N+1's forward raised.  The optimizers now decide on the device: vmlmf_amd.optim.Adam's
    gate launch skips the whole step (parameters, moments, step counts untouched, skipped_steps() == 1), clip_sgd_step skips
    on a non-finite norm; the steps around the failed one are ordinary Adam steps (same trajectory as torch.optim.Adam with
    that step dropped)."""
    import vmlmf_amd
    from vmlmf_amd import _lib
    net, x, t = _har_net()
    ref, _, _ = _har_net()
    _clear_status()
    opt = vmlmf_amd.optim.Adam(net.parameters(), lr=0.01)
    ropt = torch.optim.Adam(ref.parameters(), lr=0.01)

    def step(m, o, update=True):
        m.zero_grad(set_to_none=True)
        vmlmf_amd.cross_entropy(m(x), t).backward()
        if update:
            o.step()

    step(net, opt), step(ref, ropt)
    held = [p.detach().clone() for p in net.parameters()]
    _lib.tune("test_wride_spin", 1)
    try:
        step(net, opt)                                   # the riding workers give up: NaN parameter gradients
        torch.cuda.synchronize()
        assert not torch.isfinite(net.rnn.rnncells[0].v_h.grad).all(), "the workers were expected to give up"
        for p, h in zip(net.parameters(), held):
            assert torch.equal(p.detach(), h)                # the update was skipped on the device
        assert opt.skipped_steps() == 1
        # (Net.cell, the reference's unused duplicate, never gets a gradient: its counters stay at 0)
        assert all(float(opt.state[p]["step"]) == (1.0 if p.grad is not None else 0.0) for p in net.parameters())
        # the LM loop's update on the same NaN gradients: skipped as well, and the norm says why
        live = [p for p in net.parameters() if p.grad is not None]
        gheld = [p.grad.clone() for p in live]
        norm = vmlmf_amd.optim.clip_sgd_step(live, lr=1.0, max_norm=5.0)
        assert not torch.isfinite(norm)
        for p, h in zip(net.parameters(), held):
            assert torch.equal(p.detach(), h)
        for p, gh in zip(live, gheld):
            assert torch.equal(torch.nan_to_num(p.grad, nan=7.0), torch.nan_to_num(gh, nan=7.0))    # gradients not scaled either
    finally:
        _clear_status()                      # (while the test bound is still set: a give-up under the PRODUCTION bound would
        _lib.tune("test_wride_spin", 0)      #  switch the process to the stand-alone weight-gradient kernel for good)
    step(net, opt), step(ref, ropt)                      # step 2 of the trajectory, as if the failed one had not happened
    for (k, p), q in zip(net.named_parameters(), ref.parameters()):
        assert torch.allclose(p, q, rtol=2e-5, atol=2e-6), k
    assert opt.skipped_steps() == 1


def test_graphed_train_step_captures_again_after_a_failed_replay():
    """Verdict r3 item 6: a launch captured into a hipGraph stays what it was - after a give-up every replay would produce NaN
    again.  GraphedTrainStep reads the status word in front of every replay: the failed step's update is skipped by the
    optimizer's gate (parameters intact), the step is captured again (here: with the production wait bound restored), and
    training goes on; the tune generation alone (kernel selection changed, nothing failed) re-captures as well."""
    import warnings
    import vmlmf_amd
    from vmlmf_amd import _lib
    net, x, t = _har_net(1)
    opt = vmlmf_amd.optim.Adam(net.parameters(), lr=0.01)
    _clear_status()
    _lib.tune("wride", 1)                # (whatever ran before in this process: the riding form is armed)
    gstep = vmlmf_amd.GraphedTrainStep(net, vmlmf_amd.cross_entropy, opt, x, t)
    gstep(x, t)
    torch.cuda.synchronize()
    assert getattr(gstep, "recaptures", 0) == 0 and opt.skipped_steps() == 0
    held = [p.detach().clone() for p in net.parameters()]
    _lib.tune("test_wride_spin", 1)      # (moves the generation: the step is captured again, now with a one-look wait inside)
    try:
        gstep(x, t)                      # capture (executes nothing), replay: the riding workers give up
        torch.cuda.synchronize()
        assert gstep.recaptures == 1
        for p, h in zip(net.parameters(), held):
            assert torch.equal(p.detach(), h)            # the failed step's update never happened
        assert opt.skipped_steps() == 1
    finally:
        # back to the production bound WITHOUT looking at the status word: the failure stays pending for the next call to find
        # (found under the production bound it also switches the process to the stand-alone kernel - what a real failure does)
        _lib.tune("test_wride_spin", 0)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        loss = gstep(x, t)                               # sees E_PROTOCOL of the replay before: captures again, then steps
    torch.cuda.synchronize()
    assert gstep.failed_steps == 1 and gstep.recaptures == 2 and any("captured again" in str(m.message) for m in w)
    assert torch.isfinite(loss) and opt.skipped_steps() == 1
    assert any(not torch.equal(p.detach(), h) for p, h in zip(net.parameters(), held))      # this one did update
    assert all(torch.isfinite(p).all() for p in net.parameters())
    _lib.tune("wride", 1)                                # re-arm (the library switched itself off): the generation moves
    gstep(x, t)
    torch.cuda.synchronize()
    assert gstep.recaptures == 3 and gstep.failed_steps == 1 and all(torch.isfinite(p).all() for p in net.parameters())
    _lib.check_status()


def test_status_word_is_not_lost_when_the_first_call_on_a_device_is_captured():
    """the first library call of a process made inside a stream capture must not disable status reporting for
    good (the word's allocation is skipped for that call only).  Run in a fresh process: capture first, then provoke a give-up
    eagerly and expect the error code."""
    code = (
        "import sys; sys.path[:0] = [%r]\n"
        "import torch, vmlmf_amd\n"
        "from vmlmf_amd import _lib, MyLSTM, MyVMLMFCell, Net\n"
        "torch.manual_seed(0)\n"
        "net = Net(9, layer_sizes=[180], w_rank=16, u_rank=[16], model=MyLSTM, cell=MyVMLMFCell).cuda()\n"
        "x = torch.randn(64, 40, 9, device='cuda'); t = torch.randint(0, 18, (64,), device='cuda')\n"
        "def fb():\n"
        "    net.zero_grad(set_to_none=True)\n"
        "    vmlmf_amd.cross_entropy(net(x), t).backward()\n"
        "z = torch.randn(4, 6, device='cuda', requires_grad=True)      # loads the library's code object; touches no status word\n"
        "vmlmf_amd.cross_entropy(z, torch.zeros(4, dtype=torch.long, device='cuda')).backward()\n"
        "torch.cuda.synchronize()\n"
        "g = torch.cuda.CUDAGraph()\n"
        "with torch.cuda.graph(g):\n"
        "    fb()                           # the first forward / backward entry points of the process: inside a capture\n"
        "g.replay(); torch.cuda.synchronize()\n"
        "assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)\n"
        "_lib.tune('test_wride_spin', 1)\n"
        "fb(); torch.cuda.synchronize()\n"
        "try:\n"
        "    _lib.check_status()\n"
        "    print('no error reported')\n"
        "except _lib.VmlmfError as e:\n"
        "    print('ok' if e.code == _lib.E_PROTOCOL else 'wrong code')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]


def test_a_failed_step_is_skipped_for_every_tensor_list_and_parameter_group():
    """an optimizer step of several launches (more than 48 tensors, two parameter groups) takes ONE verdict.  The
    first form let the first launch read AND clear the health word: the later lists of the failed step saw a clean word and applied
    their NaN gradients.  Here the VMLMF layer's backward fails (the riding workers give up), 60 extra small parameters and a second
    parameter group ride along with NaN-free gradients of their own: nothing may move, one skipped step is counted, and the step
    after it is an ordinary one for every tensor - with the one-launch form (small tensors) and with the tick + update launches."""
    import vmlmf_amd
    from vmlmf_amd import _lib
    for big in (False, True):
        net, x, t = _har_net()
        _clear_status()
        extra = [torch.nn.Parameter(torch.randn(7, device="cuda")) for _ in range(60)]
        other = [torch.nn.Parameter(torch.randn(40000 if big else 33, device="cuda"))]     # (> 32768 elements: the two-launch form)
        opt = vmlmf_amd.optim.Adam([{"params": list(net.parameters()) + extra}, {"params": other, "lr": 0.02}], lr=0.01)

        def step():
            net.zero_grad(set_to_none=True)
            loss = vmlmf_amd.cross_entropy(net(x), t) + sum((e * e).sum() for e in extra) + (other[0] * other[0]).mean()
            for e in extra + other:
                e.grad = None
            loss.backward()
            opt.step()

        step()
        everything = [p for g in opt.param_groups for p in g["params"]]
        held = [p.detach().clone() for p in everything]
        _lib.tune("test_wride_spin", 1)
        try:
            step()
            torch.cuda.synchronize()
            assert not torch.isfinite(net.rnn.rnncells[0].v_h.grad).all(), "the workers were expected to give up"
            assert all(torch.isfinite(e.grad).all() for e in extra + other)
            for p, h in zip(everything, held):
                assert torch.equal(p.detach(), h), "a tensor list behind the first one applied the failed step"
            assert opt.skipped_steps() == 1
            assert all(float(opt.state[p]["step"]) == (1.0 if p.grad is not None else 0.0) for p in everything)
        finally:
            _clear_status()
            _lib.tune("test_wride_spin", 0)
        step()
        torch.cuda.synchronize()
        assert opt.skipped_steps() == 1
        moved = [not torch.equal(p.detach(), h) for p, h in zip(everything, held) if p.grad is not None]
        assert all(moved)
        assert all(float(opt.state[p]["step"]) == (2.0 if p.grad is not None else 0.0) for p in everything)
