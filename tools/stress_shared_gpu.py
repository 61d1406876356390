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

    def step():
        net.zero_grad(set_to_none=True)
        vmlmf_amd.cross_entropy(net(x), t).backward(one)

    _lib.tune("wride", 0)          # the reference gradients: stand-alone weight-gradient kernel, nothing waits for anything
    step()
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
    print(f"[{idx}] {steps} steps, {errors} reported VMLMF_E_PROTOCOL, {after} good steps after the first report", flush=True)


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
