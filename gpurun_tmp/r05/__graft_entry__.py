"""Driver entry points: build() compiles every HIP source for gfx950; smoke() runs one small forward+backward
of the hot path on cuda:0 and checks it against the oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build() -> None:
    """hipcc --offload-arch=gfx950 for vmlmf_amd/csrc/*.hip -> vmlmf_amd/lib/libvmlmf_hip.so (in-tree), then
    import the package and check the C ABI exports.  The oracle is pure Python (numpy/torch): nothing to
    compile there, and there is no buildable C/C++ reference (the reference is Python only)."""
    from vmlmf_amd import _lib
    _lib.build()
    _lib.build_torch_binding()      # TORCH_LIBRARY binding over the same C ABI (g++, needs the torch headers)
    import vmlmf_amd  # noqa: F401
    handle = _lib.lib()
    assert handle.vmlmf_abi_version() == _lib.ABI_VERSION
    # the device code just built must be free of the hardware hazards the compiler cannot guard inside the inline-asm stores
    # and DPP chains (tools/check_asm_hazards.py: disassembles every kernel; a flagged store faults or corrupts at run time)
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if os.path.exists(objdump):
        import subprocess
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_hazards.py"), _lib.LIB_PATH],
                           capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            raise RuntimeError("asm hazard check failed:\n" + r.stdout[-3000:] + r.stderr[-1000:])


def smoke() -> None:
    """One tiny training step of the flagship path (Net over MyLSTM[MyVMLMFCell]) on cuda:0, plus a check of
    a VMLMF layer's and of a two-layer wavefront stack's outputs and gradients against the oracle."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import vmlmf_oracle as O
    from hip_util import run_hip, run_literal, compare_all
    from vmlmf_amd import MyLSTM, MyVMLMFCell, Net

    assert torch.cuda.is_available(), "smoke() needs an MI355X"
    rng = np.random.Generator(np.random.PCG64(7))
    for variant, ru in ((O.V1, 16), (O.V2, [8, 8])):
        B, Tn, I, H, rw = 8, 12, 9, 180, 16
        P = O.make_params(variant, I, H, rw, ru, seed=3)
        x = rng.standard_normal((B, Tn, I)).astype(np.float32)
        dy = rng.standard_normal((B, Tn, H)).astype(np.float32)
        got = run_hip(variant, P, x, None, None, dy)
        ref = run_literal(variant, P, x, None, None, dy)
        compare_all(got, ref, f"smoke.v{variant}")
    # a two-layer stack through the wavefront launches (vmlmf_stack_*), layer by layer against the oracle
    from vmlmf_amd import functional as F
    from hip_util import ORDER, assert_grad, assert_out
    Ps = [O.make_params(O.V1, 20 if l == 0 else 64, 64, 16, 16, seed=5 + l) for l in range(2)]
    xs = rng.standard_normal((6, 9, 20)).astype(np.float32)
    dys = rng.standard_normal((6, 9, 64)).astype(np.float32)
    Pt = [O.to_torch(P, dtype=torch.float64, requires_grad=True) for P in Ps]
    xt = torch.tensor(xs, dtype=torch.float64, requires_grad=True)
    cur = xt
    for l in range(2):
        cur, _, _ = O.literal_sequence(O.V1, Pt[l], cur, None, None, time_major=False)
    (cur * torch.tensor(dys, dtype=torch.float64)).sum().backward()
    params = [[torch.tensor(np.asarray(P[k]), dtype=torch.float32, device="cuda:0").requires_grad_(True) for k in ORDER[O.V1]] for P in Ps]
    xg = torch.tensor(xs, device="cuda:0").requires_grad_(True)
    os.environ["VMLMF_STACK"] = "1"
    try:
        out = F.vmlmf_stack(O.V1, xg, params, 16, [16], g=1, time_major=False)
    finally:
        os.environ.pop("VMLMF_STACK", None)
    assert out is not None, "the wavefront kernels must cover this stack"
    (out[0] * torch.tensor(dys, device="cuda:0")).sum().backward()
    assert_out(out[0].detach().cpu().numpy(), cur.detach().numpy(), "smoke.stack.y")
    assert_grad(xg.grad.cpu().numpy(), xt.grad.numpy(), "smoke.stack.dx")
    for l in range(2):
        for k, p in zip(ORDER[O.V1], params[l]):
            assert_grad(p.grad.cpu().numpy(), Pt[l][k].grad.numpy(), f"smoke.stack.layer{l}.{k}")
    torch.manual_seed(0)
    net = Net(9, layer_sizes=[180], w_rank=16, u_rank=[16], model=MyLSTM, cell=MyVMLMFCell).to("cuda:0")
    opt = torch.optim.Adam(net.parameters(), lr=0.002)
    xb = torch.randn(16, 32, 9, device="cuda:0")
    tgt = torch.randint(0, 6, (16,), device="cuda:0")
    losses = []
    for _ in range(3):
        net.zero_grad()
        loss = torch.nn.functional.cross_entropy(net(xb), tgt)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    # the headline step's own form: the criterion riding on the forward launch (Net.loss), the gradients finished by ONE launch
    # behind the backward recurrence, the package's one-launch Adam - against the two-line form on the same state, and the loss
    # against the oracle's literal training step (V/src/train_test/train.py:58-64)
    import vmlmf_amd
    net.zero_grad()
    two_line = torch.nn.functional.cross_entropy(net(xb), tgt)
    two_line.backward()
    g_two = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    net.zero_grad()
    riding = net.loss(xb, tgt)
    riding.backward()
    assert abs(riding.item() - two_line.item()) <= 1e-6 * max(1.0, abs(two_line.item())), (riding.item(), two_line.item())
    assert len(g_two) >= 10
    for k, p in net.named_parameters():
        if k in g_two:
            assert_grad(p.grad.cpu().numpy(), g_two[k].cpu().numpy(), f"smoke.riding_criterion.{k}")
    cell = net.rnn.rnncells[0]
    Pn = {k: getattr(cell, k).detach().cpu().numpy() for k in ORDER[O.V1]}
    ref_loss, _ = O.literal_train_step_har(O.to_torch(Pn, dtype=torch.float64), net.lin.weight.detach().cpu().double(),
                                           net.lin.bias.detach().cpu().double(), xb.cpu().double(), tgt.cpu())
    ref_loss = float(ref_loss)
    assert abs(riding.item() - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (riding.item(), ref_loss)
    popt = vmlmf_amd.optim.Adam(net.parameters(), lr=0.002)
    pl = []
    for _ in range(3):
        popt.zero_grad(set_to_none=True)
        loss = net.loss(xb, tgt)
        loss.backward(vmlmf_amd.unit_gradient(loss.device))
        popt.step()
        pl.append(loss.item())
    assert all(np.isfinite(pl)) and pl[-1] < pl[0], pl
    print("smoke ok: oracle parity (layers and a wavefront stack) + 3 Adam steps, losses", [round(v, 4) for v in losses],
          "+ the three-launch step (Net.loss, finish2, package Adam):", [round(v, 4) for v in pl])


if __name__ == "__main__":
    build()
    if len(sys.argv) > 1 and sys.argv[1] == "smoke":
        smoke()
