"""One GPU shared by N processes, each running forward + backward of its own Net at the headline shape as fast as it can.
The riding weight-gradient workers wait for row workgroups of their own launch; when other processes' launches fill the CUs a
worker can give up its bounded wait.  What must hold in every process: an error code (VMLMF_E_PROTOCOL) for that step - never
a hang, never a wrong finite gradient without an error - and afterwards (the library has switched the process to the stand-alone
weight-gradient kernel) steps that match the quiet run again.
    python tools/stress_shared_gpu.py [N processes] [seconds]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(idx, seconds):
    import torch
    import vmlmf_amd
    from vmlmf_amd import MyLSTM, MyVMLMFCell, Net, _lib
    torch.manual_seed(idx)
    net = Net(9, layer_sizes=[180], w_rank=16, u_rank=[16], model=MyLSTM, cell=MyVMLMFCell).cuda()
    x = torch.randn(64, 128, 9, device="cuda")
    t = torch.randint(0, 18, (64,), device="cuda")
    one = vmlmf_amd.unit_gradient("cuda")

    # an optimizer step behind every backward, as a training loop has it: lr = 0 leaves finite parameters where they are (so
    # the comparison with the quiet gradients below stays valid) but would turn them into NaN if a failing step's NaN gradients
    # ever reached the update (0 * NaN); the device-side gate of vmlmf_amd.optim.Adam has to skip exactly those steps
    _lib.tune("clear_health", 0)
    opt = vmlmf_amd.optim.Adam(net.parameters(), lr=0.0)
    before = [p.detach().clone() for p in net.parameters()]

    def step(update=True):
        net.zero_grad(set_to_none=True)
        vmlmf_amd.cross_entropy(net(x), t).backward(one)
        if update:
            opt.step()

    _lib.tune("wride", 0)          # the reference gradients: stand-alone weight-gradient kernel, nothing waits for anything
    step(update=False)
    torch.cuda.synchronize()
    quiet = [p.grad.clone() for p in net.parameters() if p.grad is not None]
    _lib.tune("wride", 1)
    steps = errors = after = 0
    BURST = int(os.environ.get("STRESS_BURST", "20"))
    t_end = time.time() + seconds
    while time.time() < t_end:
        try:
            for _ in range(BURST):      # queued back to back: the processes' launches overlap on the GPU
                step()
            torch.cuda.synchronize()
            _lib.check_status()
        except (RuntimeError, _lib.VmlmfError) as e:
            errors += 1
            if errors == 1:
                print(f"[{idx}] step {steps}: {str(e)[:140]}", flush=True)
            torch.cuda.synchronize()
            try:
                _lib.check_status()
            except _lib.VmlmfError:
                pass
            steps += BURST
            continue
        steps += BURST
        after += BURST if errors else 0
        for a, b in zip([p.grad for p in net.parameters() if p.grad is not None], quiet):
            if not torch.isfinite(a).all() or (a - b).abs().max() > 5e-5 * (b.abs().max() + 1e-30):
                print(f"[{idx}] step {steps}: a gradient differs from the quiet run WITHOUT an error code", flush=True)
                sys.exit(3)
    torch.cuda.synchronize()
    skipped = opt.skipped_steps()
    for p, b in zip(net.parameters(), before):
        if not torch.equal(p.detach(), b):
            print(f"[{idx}] a parameter changed or went non-finite: NaN gradients reached the optimizer", flush=True)
            sys.exit(4)
    for st in opt.state.values():
        if not (torch.isfinite(st["exp_avg"]).all() and torch.isfinite(st["exp_avg_sq"]).all()):
            print(f"[{idx}] optimizer moments are not finite", flush=True)
            sys.exit(4)
    if (errors > 0) != (skipped > 0):
        print(f"[{idx}] {errors} reported failures but the optimizer's gate skipped {skipped} steps", flush=True)
        sys.exit(5)
    print(f"[{idx}] {steps} steps, {errors} reported VMLMF_E_PROTOCOL, {after} good steps after the first report, "
          f"{skipped} optimizer steps skipped by the device-side gate, parameters intact", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), float(sys.argv[3]))
        sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(i), str(seconds)]) for i in range(n)]
    rcs = [p.wait(timeout=seconds + 240) for p in procs]
    print("exit codes", rcs)
    sys.exit(0 if all(r == 0 for r in rcs) else 1)
