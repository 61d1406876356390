#!/usr/bin/env python3
"""bench.py - the headline metric of BASELINE.json on MI355X.

Metric: RNN timesteps/sec (fwd+bwd) at B=64 T=128 hid=180 r=16 (UCI-HAR shape, MyVMLMFCell, fp32).
A "step" is one pass of the hot path over one synthetic batch:
    zero_grad -> Net.forward (MyLSTM over T=128 + Linear) -> cross-entropy -> backward
    (+ ONE flat RCCL all-reduce of the gradients when N > 1).  The optimizer is outside the timed region and
reported separately (`adam_ms` stock, `fused_adam_ms` the package's, `train_step_ms` everything in one graph).  `other_configs` (single GPU, outside the metric, `--no-extra` skips it): BASELINE configs[2] in fp32 as the wavefront launches and as the chained per-layer kernels, and configs[4]'s two group layers on one GPU.  Inputs are resident in HBM before the timed region starts.
value = (N ranks x T timesteps per step) / step time: weak scaling, per-GPU batch fixed at 64
(N = 8 is BASELINE config D: global batch 512).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--global-batch G]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`--gpus N` (N > 1) without a torchrun environment starts the N ranks itself: the parent process - before it makes
any GPU call - runs the torchrun line above as a child, relays rank 0's JSON line and exits with the child's code.
`--global-batch G` is the strong-scaling mode (BASELINE configs[3]: G = 512 split contiguously, 512/N rows per GPU);
`value` then counts 64-row batches: (G / 64) x T / step time, the same unit as the weak mode's N x T / step time.

Rank 0 prints ONE JSON line.  Extra objects: "roofline" (dominant kernel, HIP events on the launch stream,
inside the timed region) and, at N = 1, "cpu_baseline" (the oracle's op-for-op PyTorch-CPU port of the
reference cell + time loop, timed on this box's host cores).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU, T, I, H, RW, RU, CLASSES = 64, 128, 9, 180, 16, 16, 6
F32_MATRIX_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 MFMA == fp32 vector peak
# algorithmic flops per sample-timestep, forward
F_X = 2 * I * RW + 8 * H * RW        # input -> hidden  (xproj)
F_H = 10 * H * RU                    # hidden -> hidden (the serial recurrent kernels)
F_FWD = F_X + F_H                    # 52 128
F_STEP = 3 * F_FWD                   # fwd + bwd = 156 384


def numpy_params(seed):
    """Seeded numpy-PCG64 parameters/inputs: identical on the CPU leg and on every GPU box."""
    rng = np.random.Generator(np.random.PCG64(seed))
    shapes = {"u_x": (I, RW), "u_h": (H, RU), "v_x": (4 * H, RW), "v_h": (4 * H, RU),
              "b_x": (4 * H,), "b_h": (4 * H,), "dia_x": (1, I), "dia_h": (1, H)}
    return {k: (0.1 * rng.standard_normal(s)).astype(np.float32) for k, s in shapes.items()}


def synthetic_batch(rank, rows=None, global_rows=None):
    """Weak mode: `rows` rows from the rank's own seed.  Strong mode (global_rows): the rank's contiguous shard of ONE
    global minibatch (seed 1234), so N ranks together compute exactly the single-GPU step on that minibatch."""
    rows = B_PER_GPU if rows is None else rows
    if global_rows is not None:
        rng = np.random.Generator(np.random.PCG64(1234))
        x = rng.standard_normal((global_rows, T, I)).astype(np.float32)
        tgt = rng.integers(0, CLASSES, size=(global_rows,)).astype(np.int64)
        return x[rank * rows:(rank + 1) * rows], tgt[rank * rows:(rank + 1) * rows]
    rng = np.random.Generator(np.random.PCG64(1234 + rank))
    x = rng.standard_normal((rows, T, I)).astype(np.float32)
    tgt = rng.integers(0, CLASSES, size=(rows,)).astype(np.int64)
    return x, tgt


def spawn_ranks(n, argv):
    """Parent of a multi-GPU run started as plain `python bench.py --gpus N`: one fresh process per GPU through
    torchrun.  Nothing in this process has touched the GPU (torch.cuda.device_count() does not initialise it); the
    children's stderr passes through, rank 0's JSON line is relayed, a failing child fails the run."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if os.environ.get("VMLMF_BENCH_REHEARSAL") == "1":
        have = n       # every rank on GPU 0 over gloo (see main): a rehearsal of the N > 1 code path, not a measurement
    if have < n:
        print(f"[bench] --gpus {n} asked for, {have} visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver only supports dmabuf IPC; without it RCCL's intra-node transport
    # fails at hipIpcGetMemHandle (the image exports it already; it is only defaulted here, never overridden)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    import signal
    import threading
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, start_new_session=True)
    # a rank that hangs (a collective one rank never entered) must not hang the caller: after `limit` seconds the whole
    # process group of the children is killed and the run fails
    limit = float(os.environ.get("VMLMF_BENCH_RANK_TIMEOUT", "1500"))
    timed_out = []

    def reap():
        timed_out.append(True)
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except OSError:
            pass

    timer = threading.Timer(limit, reap)
    timer.daemon = True
    timer.start()
    line = None
    for out in proc.stdout:
        if out.startswith("{") and '"metric"' in out:
            line = out
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    timer.cancel()
    if timed_out:
        print(f"[bench] the ranks did not finish within {limit:.0f} s: killed", file=sys.stderr)
        return 3
    if rc == 0 and line is None:
        print("[bench] the ranks finished without a result line", file=sys.stderr)
        rc = 1
    if rc == 0:
        sys.stdout.write(line)
        sys.stdout.flush()
    return rc


def usable_cores():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(budget_s=30.0):
    """Reference CPU path (port): oracle.literal_* (op-for-op PyTorch-CPU restatement of the reference cell
    and time loop, autograd backward) on the host cores, same shapes, same step definition.  The workload is
    dispatch-bound (~75 tiny ATen ops per timestep), so more threads are not faster: it is timed with 1
    thread and with all usable cores (capped at 16) and the FASTER one is reported."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vmlmf_oracle as O   # checker/baseline only; never on the product path
    P = O.to_torch(numpy_params(3), requires_grad=True)
    g = torch.Generator().manual_seed(0)
    lw = (0.01 * torch.randn(18, H, generator=g)).requires_grad_(True)
    lb = torch.full((18,), 0.1, requires_grad=True)
    x, tgt = synthetic_batch(0)
    xt, tt = torch.tensor(x), torch.tensor(tgt)

    def one():
        t0 = time.perf_counter()
        for p in list(P.values()) + [lw, lb]:
            p.grad = None
        loss, _ = O.literal_train_step_har(P, lw, lb, xt, tt)
        loss.backward()
        return time.perf_counter() - t0

    cores = usable_cores()
    results = {}
    for nt in sorted({1, min(cores, 16)}):
        torch.set_num_threads(nt)
        for _ in range(3):                      # 3 warm-up + >= 5 timed steps, median
            one()
        times, t_start = [], time.perf_counter()
        while len(times) < 5 or (len(times) < 9 and (time.perf_counter() - t_start) < budget_s / 4):
            times.append(one())
        results[nt] = float(np.median(times))
    best = min(results, key=results.get)
    cpu_model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": T / results[best], "unit": "RNN timesteps/s", "cores": best, "kind": "port", "cpu_model": cpu_model,
            "host_cores_usable": cores,
            "sample": f"full steps of the bench workload (B={B_PER_GPU} T={T} I={I} H={H} r={RU}), 3 warm-up + 5-9 "
                      f"timed per thread count, median; s/step by threads: "
                      + ", ".join(f"{k}: {v:.3f}" for k, v in results.items())
                      + f"; host has {cores} usable cores",
            "s_per_step": results[best]}


def other_configs(iters=100):
    """Outside the metric: BASELINE configs[2] in fp32 (two VMLMF layers of 256, rank 24, B 128, T 24, I 77), forward +
    backward of the RNN stack replayed from a hipGraph - as the wavefront launches (all layers in one launch per direction,
    DESIGN.md section 4f) and as the chained per-layer kernels (VMLMF_STACK=0)."""
    import torch
    from vmlmf_amd import MyLSTM, MyVMLMFCell
    out = {}
    prev = os.environ.get("VMLMF_STACK")
    try:
        for key, mode in (("ms_per_step", "auto"), ("chained_ms_per_step", "0")):
            os.environ["VMLMF_STACK"] = mode
            torch.manual_seed(0)
            rnn = MyLSTM(77, hidden_layer_sizes=[256, 256], batch_first=True, w_rank=24, u_ranks=[24], cell=MyVMLMFCell).cuda()
            x = torch.randn(128, 24, 77, device="cuda")

            def fb():
                rnn.zero_grad(set_to_none=True)
                y, _ = rnn(x)
                y[:, -1].sum().backward()

            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    fb()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fb()
            for _ in range(10):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                g.replay()
            torch.cuda.synchronize()
            out[key] = round((time.perf_counter() - t0) / iters * 1e3, 4)
    finally:
        if prev is None:
            os.environ.pop("VMLMF_STACK", None)
        else:
            os.environ["VMLMF_STACK"] = prev
    out["timesteps_per_s"] = round(24 / (out["ms_per_step"] * 1e-3), 1)
    out["workload"] = "BASELINE configs[2] in fp32: 2 x MyVMLMFCell(256), rank 24, B 128, T 24, I 77; RNN stack forward + backward, hipGraph replay"
    res = {"C_fp32": out}
    # the same workload with dtype bf16 (set_compute_dtype): what configs[2] names.  Timed here every round
    try:
        from vmlmf_amd import set_compute_dtype
        torch.manual_seed(0)
        rnn_b = MyLSTM(77, hidden_layer_sizes=[256, 256], batch_first=True, w_rank=24, u_ranks=[24], cell=MyVMLMFCell).cuda()
        set_compute_dtype(rnn_b, "bf16")
        xb = torch.randn(128, 24, 77, device="cuda")

        def fbb():
            rnn_b.zero_grad(set_to_none=True)
            yb, _ = rnn_b(xb)
            yb[:, -1].sum().backward()

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fbb()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gb):
            fbb()
        for _ in range(10):
            gb.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            gb.replay()
        torch.cuda.synchronize()
        msb = (time.perf_counter() - t0) / iters * 1e3
        res["C_bf16"] = {"ms_per_step": round(msb, 4), "timesteps_per_s": round(24 / (msb * 1e-3), 1),
                         "vs_fp32": round(msb / out["ms_per_step"], 2),
                         "workload": "BASELINE configs[2] with dtype bf16 (set_compute_dtype): below 4096 rows the same wavefront launches "
                                     "with fp32 arithmetic and a bf16 GATE TAPE (8 instead of 16 bytes per unit and step; forward "
                                     "bit-identical to fp32, gradients within the derived bound of tests/test_gpu_bf16.py); the "
                                     "bf16-MFMA row-block kernels take over from 4096 rows, where they win; hipGraph replay"}
    except Exception as e:   # never at the expense of the line
        res["C_bf16"] = {"error": f"{type(e).__name__}: {e}"}
    # BASELINE configs[4] on one GPU: two PTB group layers (H 650, ranks 32 / [32, 32]), B 256, T 35 (clustered row-block kernels)
    from vmlmf_amd import MyVMLSTMGroup
    torch.manual_seed(0)
    layers = [MyVMLSTMGroup(650, 650, w_rank=32, u_ranks=[32, 32]).cuda() for _ in range(2)]
    for l in layers:
        for p in l.parameters():
            torch.nn.init.uniform_(p, -0.05, 0.05)
    xe = 0.05 * torch.randn(35, 256, 650, device="cuda")
    states = [(torch.zeros(256, 650, device="cuda"), torch.zeros(256, 650, device="cuda")) for _ in layers]

    def fbe():
        for l in layers:
            l.zero_grad(set_to_none=True)
        h = xe
        for l, st in zip(layers, states):
            h, _ = l(h, st)
        h.sum().backward()

    for _ in range(3):
        fbe()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fbe()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    res["E_1gpu"] = {"ms_per_step": round(ms, 4), "timesteps_per_s": round(35 / (ms * 1e-3), 1),
                     "workload": "BASELINE configs[4] on one GPU: 2 x MyVMLSTMGroup(650), ranks 32 / [32, 32], B 256, T 35; layers forward + backward, eager launches"}
    # ... and what one GPU of an 8-GPU node gets of it (32 rows): the same layers, the clustered kernels with 4 live rows per workgroup
    try:
        xe32 = 0.05 * torch.randn(35, 32, 650, device="cuda")
        st32 = [(torch.zeros(32, 650, device="cuda"), torch.zeros(32, 650, device="cuda")) for _ in layers]

        from vmlmf_amd.lm import stack_layers

        def fbe_rows(xin, sts):   # Model.features' layer loop: one launch per direction where the clusters of both layers are co-resident
            for l in layers:
                l.zero_grad(set_to_none=True)
            out = stack_layers(layers, xin, sts)
            if out is not None:
                h = out[0]
            else:
                h = xin
                for l, st in zip(layers, sts):
                    h, _ = l(h, st)
            h.sum().backward()

        def fbe32():
            fbe_rows(xe32, st32)

        for _ in range(3):
            fbe32()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fbe32()
        torch.cuda.synchronize()
        ms32 = (time.perf_counter() - t0) / 20 * 1e3
        res["E_32rows"] = {"ms_per_step": round(ms32, 4), "timesteps_per_s": round(35 / (ms32 * 1e-3), 1),
                           "workload": "the same two layers at 32 rows (configs[4]'s share of one GPU on an 8-GPU node), eager launches; both "
                                       "layers in ONE launch per direction (csrc/vmlmf_rbx.hip), as Model.features runs them"}
        for rows in (64, 128):   # the 4-GPU and 2-GPU shares
            xr = 0.05 * torch.randn(35, rows, 650, device="cuda")
            sr = [(torch.zeros(rows, 650, device="cuda"), torch.zeros(rows, 650, device="cuda")) for _ in layers]
            for _ in range(3):
                fbe_rows(xr, sr)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fbe_rows(xr, sr)
            torch.cuda.synchronize()
            msr = (time.perf_counter() - t0) / 20 * 1e3
            res[f"E_{rows}rows"] = {"ms_per_step": round(msr, 4), "timesteps_per_s": round(35 / (msr * 1e-3), 1),
                                    "workload": f"the same two layers at {rows} rows, eager launches, one launch per direction"}
    except Exception as e:   # never at the expense of the line
        res["E_32rows"] = {"error": f"{type(e).__name__}: {e}"}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the measurements outside the metric (other_configs)")
    ap.add_argument("--no-graph", action="store_true", help="time eager launches only")
    ap.add_argument("--graph-collective", action="store_true", help="capture the gradient all-reduce inside the hipGraph too")
    ap.add_argument("--force-collective", action="store_true", help="run the RCCL gradient all-reduce even with one rank")
    ap.add_argument("--torch-loss", action="store_true", help="torch.nn.functional.cross_entropy instead of vmlmf_amd.cross_entropy")
    ap.add_argument("--separate-loss", action="store_true",
                    help="criterion(net(x), target) as two calls (vmlmf_amd.cross_entropy: a launch of its own) instead of Net.loss, "
                         "where the criterion rides on the forward recurrence's launch")
    ap.add_argument("--repack", action="store_true", help="(the default since round 3; kept for old command lines)")
    ap.add_argument("--keep-images", action="store_true",
                    help="value = the step with kept parameter images (no pack_kernel while the parameters are unchanged: "
                         "inference / gradient accumulation; a training loop re-packs every step, which is the default)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: this many rows in total, split contiguously over the ranks (configs[3]: 512)")
    ap.add_argument("--config", choices=("A", "E"), default="A",
                    help="A: the headline (BASELINE configs[1], and [3] with --global-batch).  E: BASELINE configs[4], the PTB LM "
                         "network data-parallel (tools/bench_lm.py: run_config_e) - global batch 256 unless --global-batch / "
                         "--batch-per-gpu say otherwise; its line is NOT the graded metric")
    ap.add_argument("--batch-per-gpu", type=int, default=0, help="config E: weak scaling with this many rows per GPU")
    ap.add_argument("--plain-layers", action="store_true", help="config E: MyVMLSTM layers instead of MyVMLSTMGroup")
    ap.add_argument("--transport", choices=("auto", "cabi", "torch", "p2p"), default="auto",
                    help="gradient all-reduce through the C ABI (vmlmf_flat_allreduce_group, RCCL) or torch.distributed "
                         "(backend nccl = RCCL).  auto: torch.distributed with more than one rank (the C-ABI communicator has "
                         "only ever run in a group of one: no multi-GPU box was available to the builder), the C ABI in the "
                         "one-rank self-test of --force-collective")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    # Only the JSON line may reach stdout: libraries print there too (RCCL's version banner at N > 1), so fd 1 is
    # pointed at stderr for the rest of the process and the result goes out through a private copy of stdout.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size is used", file=sys.stderr)
    strong = args.global_batch > 0
    if args.config == "A" and strong and args.global_batch % world != 0:
        raise SystemExit(f"--global-batch {args.global_batch} is not divisible by {world} ranks")
    rows_gpu = args.global_batch // world if strong else B_PER_GPU      # batch rows of this rank
    batches_per_step = (args.global_batch / B_PER_GPU) if strong else world   # 64-row batches one step processes
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # VMLMF_BENCH_REHEARSAL=1: every rank on GPU 0, gradients exchanged over gloo - RCCL refuses two ranks on one device, and
    # the builder's boxes have one GPU; this walks the N > 1 code path (spawn, sharding, exchange, barriers, max over ranks,
    # both scaling modes, the JSON line) end to end.  The line says "rehearsal": its numbers are NOT a scaling measurement.
    rehearsal = os.environ.get("VMLMF_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    # --force-collective: run the gradient all-reduce (RCCL) even on one GPU, to exercise the N > 1 code path
    collective = world > 1 or args.force_collective
    if collective:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # a bounded timeout: a rank that never joins a collective makes the others fail instead of waiting for ever
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=300))

    if args.config == "E":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_lm

        def e_barrier():
            if collective:
                dist.barrier()
            torch.cuda.synchronize()

        bench_lm.run_config_e(args, {"world": world, "rank": rank, "dev": dev, "collective": collective, "rehearsal": rehearsal,
                                     "result_fd": result_fd, "barrier": e_barrier,
                                     "log": lambda m: rank == 0 and print(f"[bench] {m}", file=sys.stderr, flush=True)})
        if collective:
            dist.destroy_process_group()
        return

    from vmlmf_amd import MyLSTM, MyVMLMFCell, Net, _lib
    from vmlmf_amd.dp import FlatGradAllReduce, broadcast_parameters

    torch.manual_seed(0)
    net = Net(I, layer_sizes=[H], w_rank=RW, u_rank=[RU], model=MyLSTM, cell=MyVMLMFCell)
    P = numpy_params(3)
    with torch.no_grad():
        for k, v in P.items():
            getattr(net.rnn.rnncells[0], k).copy_(torch.tensor(v))
    net = net.to(dev)
    broadcast_parameters(net)
    x_np, tgt_np = synthetic_batch(rank, rows_gpu, args.global_batch if strong else None)
    x = torch.tensor(x_np, device=dev)
    tgt = torch.tensor(tgt_np, device=dev)
    # The timed step has no optimizer in it, so the parameters do not change between its repetitions: the
    # layers keep their packed parameter images (C ABI vmlmf_pack_params / *_packed) instead of re-packing identical
    # values every forward (pack_kernel, 6 us).  --repack measures the step with the packing in it; train_step_ms
    # (optimizer inside the graph) re-packs every step by construction.
    import vmlmf_amd as _pkg
    if args.keep_images:
        _pkg.cache_packed_parameters(net, True)
    transport = args.transport if args.transport != "auto" else ("cabi" if world == 1 else "torch")
    reducer = FlatGradAllReduce(net.parameters(), op="avg", transport=transport if collective else "torch")
    reducer.always = args.force_collective
    lib = _lib.lib()

    # the criterion of the reference's loop (nn.CrossEntropyLoss, train.py:58-65): the package's fused kernels,
    # or the stock library op with --torch-loss (same values, six launches instead of two)
    import vmlmf_amd
    criterion = torch.nn.functional.cross_entropy if args.torch_loss else vmlmf_amd.cross_entropy

    # d(loss)/d(loss) = 1 as the package's constant tensor: loss.backward() alone makes autograd fill a fresh
    # ones_like(loss) every step (a 4 us launch between the loss and its backward), and the fused criterion returns the
    # gradient its forward kernel already wrote when it is handed this tensor; same values either way
    one = vmlmf_amd.unit_gradient(dev)

    fused_loss = not (args.torch_loss or args.separate_loss)

    def fwd_bwd():
        net.zero_grad(set_to_none=True)
        if fused_loss:
            loss = net.loss(x, tgt)          # = criterion(net(x), tgt), the criterion inside the forward launch (train.py:61-63)
        else:
            loss = criterion(net(x), tgt)
        loss.backward(one)
        return loss

    def step():
        loss = fwd_bwd()
        if collective:
            reducer.reduce()
        return loss

    def barrier():
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    log(f"world={world} device={torch.cuda.get_device_name(dev)}")
    for _ in range(args.warmup):
        step()
    barrier()
    log("warm-up done")

    # ---- timed region 1 (eager launches): HIP event pairs around the two serial recurrent kernels (rocprof
    # names them rec_fwd_kernel / rec_bwd_kernel), recorded on the stream they are launched on.  Eager PyTorch
    # is HOST-bound at this size (~0.4 ms of Python/dispatcher per step vs ~0.3 ms of GPU work), so this region
    # supplies the per-kernel durations for the roofline, and region 2 supplies `value`.
    lib.vmlmf_profile_enable((1 << 2) | (1 << 3))
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt_eager = time.perf_counter() - t0
    usec = (ctypes.c_float * _lib.NKERNELS)()
    cnt = (ctypes.c_int32 * _lib.NKERNELS)()
    lib.vmlmf_profile_read(usec, cnt, 1)
    lib.vmlmf_profile_enable(0)
    rec = {lib.vmlmf_kernel_name(k).decode(): usec[k] / max(cnt[k], 1) for k in (2, 3)}
    log(f"eager timed region done: {dt_eager / args.steps * 1e3:.4f} ms/step")

    # ---- timed region 2 (hipGraph): forward + loss + backward captured ONCE into a HIP graph and replayed;
    # the gradient all-reduce (N > 1) stays an eager RCCL call after each replay.
    launch_mode, dt = "eager", dt_eager
    loss = step().detach().clone()      # keep no reference into the autograd graph across the capture
    if not args.no_graph:
        import gc
        gc.collect()
        def capture(body):
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(3):
                    body()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            net.zero_grad(set_to_none=True)
            # with a process group alive its watchdog thread may call into the runtime while this thread captures: thread-local
            # capture mode keeps such calls from invalidating the capture (the capturing thread's own rules are unchanged)
            with torch.cuda.graph(graph, capture_error_mode="thread_local" if collective else "global"):
                out = body()
            return graph, out

        def agree(ok):      # every rank must take the same path (the replay loop may contain a collective)
            if not collective:
                return ok
            flag = torch.tensor([1 if ok else 0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(flag.item())

        # forward + loss + backward replay from one graph and the gradient all-reduce is an eager RCCL group call
        # after each replay (--graph-collective: try to capture it as well); fallback: the eager region above
        graph, g_loss, reduce_in_graph = None, None, False
        # (the collective is captured only on request: an RCCL launch inside a replayed graph saves ~3 % of the
        # step at N > 1, and a capture that misbehaves on some RCCL build would hang instead of failing)
        choices = ((step, True), (fwd_bwd, False)) if (collective and args.graph_collective) else ((fwd_bwd, False),)
        for body, with_reduce in choices:
            try:
                graph, g_loss = capture(body)
                ok = True
            except Exception as e:
                log(f"hipGraph capture ({'with' if with_reduce else 'without'} the all-reduce) failed "
                    f"({type(e).__name__}: {e})")
                torch.cuda.synchronize()
                graph, ok = None, False
            if agree(ok):
                reduce_in_graph = with_reduce
                break
            graph = None
        captured = graph is not None
        if captured:
            def gstep():
                graph.replay()
                if collective and not reduce_in_graph:
                    reducer.reduce()
                return g_loss

            for _ in range(args.warmup):
                gstep()
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                gstep()
            barrier()
            dt_graph = time.perf_counter() - t0
            loss = g_loss.detach().clone()
            log(f"hipGraph timed region done: {dt_graph / args.steps * 1e3:.4f} ms/step")
            launch_mode = "hipgraph" + ("+allreduce" if reduce_in_graph else "")
            dt = dt_graph

    # ---- outside the metric: the same K steps with kept parameter images (pack_kernel leaves the step: what inference or
    # gradient accumulation sees), or - under --keep-images - with the packing inside
    alt_ms = None
    if launch_mode.startswith("hipgraph"):
        try:
            _pkg.cache_packed_parameters(net, not args.keep_images)
            for _ in range(3):
                fwd_bwd()
            agraph, _ = capture(fwd_bwd)

            def astep():
                agraph.replay()
                if collective and not reduce_in_graph:
                    reducer.reduce()

            for _ in range(args.warmup):
                astep()
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                astep()
            barrier()
            ta = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
            if collective:
                dist.all_reduce(ta, op=dist.ReduceOp.MAX)
            alt_ms = float(ta.item()) / args.steps * 1e3
            del agraph
        except Exception as e:
            log(f"alternate parameter-image mode failed ({type(e).__name__}: {e})")
            torch.cuda.synchronize()
        finally:
            _pkg.cache_packed_parameters(net, args.keep_images)
            for _ in range(2):
                fwd_bwd()
            torch.cuda.synchronize()

    # ---- outside the metric, N > 1: the OTHER scaling mode in the same run (weak run: global batch 512 split over the ranks =
    # BASELINE configs[3]; strong run: 64 rows per GPU), K graph-replayed steps each, MAX over ranks
    other_mode = None
    if world > 1 and launch_mode.startswith("hipgraph"):
        try:
            o_strong = not strong
            o_global = 512 if o_strong else None
            if o_strong and 512 % world != 0:
                raise ValueError("512 rows do not split evenly")
            o_rows = 512 // world if o_strong else B_PER_GPU
            xo_np, to_np = synthetic_batch(rank, o_rows, o_global)
            xo, to = torch.tensor(xo_np, device=dev), torch.tensor(to_np, device=dev)

            def fb_other():
                net.zero_grad(set_to_none=True)
                lo = net.loss(xo, to) if fused_loss else criterion(net(xo), to)
                lo.backward(one)
                return lo

            for _ in range(5):
                fb_other()
                reducer.reduce()
            ograph, _ = capture(fb_other)
            for _ in range(args.warmup):
                ograph.replay()
                reducer.reduce()
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                ograph.replay()
                reducer.reduce()
            barrier()
            to_ = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
            dist.all_reduce(to_, op=dist.ReduceOp.MAX)
            o_dt = float(to_.item())
            o_batches = (512 / B_PER_GPU) if o_strong else world
            other_mode = {"scaling": "strong" if o_strong else "weak", "global_batch": o_rows * world, "batch_per_gpu": o_rows,
                          "ms_per_step": round(o_dt / args.steps * 1e3, 4), "value": round(o_batches * T * args.steps / o_dt, 1),
                          "unit": "RNN timesteps/s (64-row batches x T per second)"}
            del ograph
            for _ in range(2):
                step()
            torch.cuda.synchronize()
        except Exception as e:
            log(f"other scaling mode failed ({type(e).__name__}: {e})")
            torch.cuda.synchronize()

    # untimed extra pass: every internal kernel bracketed, for the breakdown
    lib.vmlmf_profile_enable((1 << _lib.NKERNELS) - 1)
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    lib.vmlmf_profile_read(usec, cnt, 1)
    lib.vmlmf_profile_enable(0)
    kern = {lib.vmlmf_kernel_name(k).decode(): round(usec[k] / max(cnt[k], 1), 2) for k in range(_lib.NKERNELS)}

    # the gradient all-reduce on its own ("all-reduce time isolated"): the same in-place RCCL group call
    # the timed steps make, on the gradients of the last step, between barriers
    allreduce_ms = None
    if collective:
        for _ in range(5):
            reducer.reduce()
        barrier()
        t1 = time.perf_counter()
        for _ in range(50):
            reducer.reduce()
        barrier()
        tar = torch.tensor([(time.perf_counter() - t1) / 50 * 1e3], device=dev, dtype=torch.float64)
        dist.all_reduce(tar, op=dist.ReduceOp.MAX)
        allreduce_ms = float(tar.item())

    # every rank must hold the SAME reduced gradients after the exchange: the norm each rank computes has to agree bit for bit
    grads_equal = None
    if collective:
        step()
        gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in net.parameters() if p.grad is not None)).reshape(1)
        ghi, glo = gn.clone(), gn.clone()
        dist.all_reduce(ghi, op=dist.ReduceOp.MAX)
        dist.all_reduce(glo, op=dist.ReduceOp.MIN)
        grads_equal = bool(torch.isfinite(ghi).item() and ghi.item() == glo.item())

    # optimizer, outside the metric (train.py:47,65): the stock one and the package's single-launch one, and the
    # whole training step (forward + loss + backward + optimizer) replayed from one hipGraph
    def time_opt(opt):
        step()
        opt.step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(20):
            opt.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t1) / 20 * 1e3

    adam_ms = time_opt(torch.optim.Adam(net.parameters(), lr=0.002))
    fused_opt = vmlmf_amd.optim.Adam(net.parameters(), lr=0.002)
    fused_adam_ms = time_opt(fused_opt)
    train_step_ms = None
    if launch_mode.startswith("hipgraph") and (not collective or reduce_in_graph):
        try:
            def train_step():
                out = step()
                fused_opt.step()
                return out
            tgraph, _ = capture(train_step)
            for _ in range(args.warmup):
                tgraph.replay()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                tgraph.replay()
            torch.cuda.synchronize()
            train_step_ms = (time.perf_counter() - t1) / args.steps * 1e3
        except Exception as e:
            log(f"training-step capture failed ({type(e).__name__}: {e})")
            torch.cuda.synchronize()

    # what a maintainer's loop gets (verdict r4 item 7), one GPU, after everything the metric needs has been measured - these
    # loops move the parameters: (a) train.py:47,58-65 as written - eager launches, nn.CrossEntropyLoss, torch.optim.Adam; (b) the
    # same loop with the package's pieces, still eager (host-bound: ~20 Python-level calls per step); (c) the two-line opt-in of
    # vmlmf_amd.optim.Adam + GraphedTrainStep (every step through its __call__, input copies included)
    harness = None
    if world == 1 and not args.no_extra:
        def loop_ms(body, n=100):
            for _ in range(10):
                body()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n):
                body()
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / n * 1e3
        try:
            crit_t, opt_t = torch.nn.CrossEntropyLoss(), torch.optim.Adam(net.parameters(), lr=0.002)

            def literal():
                opt_t.zero_grad()
                loss = crit_t(net(x), tgt)
                loss.backward()
                opt_t.step()
            literal_ms = loop_ms(literal)
            opt_p = vmlmf_amd.optim.Adam(net.parameters(), lr=0.002)

            def eager_pkg():
                opt_p.zero_grad(set_to_none=True)
                loss = net.loss(x, tgt)
                loss.backward(one)
                opt_p.step()
            eager_pkg_ms = loop_ms(eager_pkg)
            gstep = vmlmf_amd.GraphedTrainStep(net, vmlmf_amd.CrossEntropyLoss(), vmlmf_amd.optim.Adam(net.parameters(), lr=0.002), x, tgt)
            optin_ms = loop_ms(lambda: gstep(x, tgt))
            harness = {"unchanged_loop_ms": round(literal_ms, 4), "unchanged_loop": "train.py:47,58-65 as written: eager launches, nn.CrossEntropyLoss, torch.optim.Adam",
                       "eager_package_loop_ms": round(eager_pkg_ms, 4), "eager_package_loop": "eager launches, Net.loss, vmlmf_amd.optim.Adam",
                       "two_line_opt_in_ms": round(optin_ms, 4), "two_line_opt_in": "vmlmf_amd.optim.Adam + vmlmf_amd.GraphedTrainStep called per batch (INTEGRATION.md section 1)",
                       "recaptures": getattr(gstep, "recaptures", 0)}
            del gstep
        except Exception as e:
            log(f"harness loops failed ({type(e).__name__}: {e})")
            torch.cuda.synchronize()

    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    tmin = tmax.clone()
    if collective:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
    dt, dt_min = float(tmax.item()), float(tmin.item())
    rccl_ranks, ranks_counted_by = reducer.exchange_ranks() if collective else (None, None)
    collectives_per_step = reducer.last_collectives if collective else 0
    ms_per_step = dt / args.steps * 1e3
    value = batches_per_step * T * args.steps / dt

    if rank == 0:
        dom = max(rec, key=rec.get)                      # dominant kernel by measured time
        # HBM traffic of that kernel: PMC counters cannot be read from inside the process; they were collected
        # with rocprofv3 in separate passes on this same command and committed under profiles/
        traffic, traffic_note, util, util_note = None, None, None, None
        # (rocprof reports the kernels by their own names: the backward recurrence of this shape is rec3_bwd_kernel)
        inrow = kern.get("wgrad_mfma_kernel", 0.0) == 0.0 and rows_gpu > 64     # rec4_bwd_kernel: weight gradients inside the rows' workgroups
        prof_names = {"rec_bwd_kernel": (("rec4_bwd_kernel",) if inrow else ()) + ("rec3_bwd_kernel", "rec_bwd_kernel"),
                      "rec_fwd_kernel": ("rec_fwd_kernel", "rec3_fwd_kernel")}

        def prof_entry(kernels):
            for nm in prof_names.get(dom, (dom,)):
                if nm in kernels:
                    return kernels[nm]
            raise KeyError(dom)

        b256 = strong and rows_gpu == 256
        for name in (("r06_pmc_traffic_b256.json", "r05_pmc_traffic_b256.json", "r04_pmc_traffic_b256.json") if b256 else ()) + ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_zz4_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
                traffic = prof_entry(pmc["kernels"])["hbm_bytes_per_launch"]
                traffic_note = f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950 2x FETCH correction)"
                break
            except (OSError, KeyError, ValueError):
                pass
        try:   # SQ counters of the same command (two --pmc passes), per launch of the dominant kernel
            util_file = next(f for f in ((("r06_pmc_util_b256.json", "r05_pmc_util_b256.json", "r04_pmc_util_b256.json") if b256 else ()) + ("r06_pmc_util.json", "r05_pmc_util.json", "r04_pmc_util.json", "r03_pmc_util.json", "r02_zz4_pmc_util.json", "r02_pmc_util.json"))
                             if os.path.exists(os.path.join(ROOT, "profiles", f)))
            u = prof_entry(json.load(open(os.path.join(ROOT, "profiles", util_file)))["kernels"])["derived"]
            util = {k: u.get(k) for k in ("valu_active_frac", "mfma_busy_frac", "wait_frac", "issue_stall_frac",
                                          "lds_conflict_frac", "valu_insts_per_wave", "mfma_insts_per_wave")}
            util_note = f"profiles/{util_file} (rocprofv3 --pmc SQ_*; fractions of SQ_WAVE_CYCLES resp. of busy-CU cycles)"
        except (OSError, KeyError, ValueError):
            pass
        rows = rows_gpu * T                              # sample-timesteps one launch processes
        # 10 H ru per sample-step in either recurrent kernel; a backward launch that carries the weight-gradient workers
        # (no wgrad_mfma_kernel launch in the breakdown pass) also does their products: dpre^T x, dpre^T Q, h^T dQ
        riding = dom == "rec_bwd_kernel" and kern.get("wgrad_mfma_kernel", 0.0) == 0.0     # (riding workers, or the in-row form)
        F_WG = 2 * 4 * H * I + 2 * 4 * H * RU + 2 * H * RU
        # a forward launch whose x-projection wave forms the input side itself (no xproj_kernel launch in the breakdown pass) does
        # the forward's whole algorithmic work, section 8d's F = 2 I rw + 8 H rw + 10 H ru per sample-step
        x_inside = dom == "rec_fwd_kernel" and kern.get("xproj_kernel", 0.0) == 0.0
        flops = rows * (F_H + (F_WG if riding else 0) + (F_X if x_inside else 0))
        achieved = flops / (rec[dom] * 1e-6) / 1e12
        achieved_rec = rows * F_H / (rec[dom] * 1e-6) / 1e12      # the recurrence's own 10 H ru per sample-step only
        # workgroups of the dominant launch, one per CU (the launch asks for more than half a CU's LDS): the rows' (vmlmf_query)
        # and, when the weight gradients ride, the workers' (launch geometry of vmlmf_api.hip: plan_wride; not a counter)
        nwork = 0
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        if riding and not inrow:
            wpw = (192 + 128) // 64
            ntg = -(-(192 // 8 + (H + 31) // 32) // wpw)
            nwork = min(32, (cus - 8 - rows_gpu) // ntg) * ntg
        launch_wgs = rows_gpu + nwork
        out = {
            "metric": "RNN timesteps/sec (fwd+bwd) at B=64 T=128 hid=180 r=16; 1/2/4/8 GPU",
            "value": round(value, 1), "unit": "RNN timesteps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            # beside it: the same step in the other parameter-image mode, and the whole training step (forward + loss + backward
            # + fused Adam in one graph: packs every step by construction) - config.workload says which one `value` is
            ("ms_per_step_repack" if args.keep_images else "ms_per_step_kept_images"): None if alt_ms is None else round(alt_ms, 4),
            "train_step_ms": None if train_step_ms is None else round(train_step_ms, 4),
            "harness": harness,
            # the riding weight-gradient workers at the end of the run: False when VMLMF_WRIDE=0 / vmlmf_tune("wride", 0) switched them
            # off or a launch gave up a bounded wait on a shared GPU and the library fell back to the stand-alone kernel (then
            # ms_per_step is that form's, not a regression of the riding one)
            "riding_workers": {"armed": bool(_lib.tune_get("wride")), "stand_alone_weight_gradient_launch_in_step": kern.get("wgrad_mfma_kernel", 0.0) > 0.0,
                               "status": lib.vmlmf_check_status()},
            "ms_per_step_min_over_ranks": round(dt_min / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if not rehearsal else "synthetic; REHEARSAL: all ranks share GPU 0 over gloo - not a scaling measurement",
            "config": {"workload": ("BASELINE configs[3]: UCI-HAR shape, global batch %d split contiguously over the ranks, "
                                    % args.global_batch if strong else "BASELINE configs[1]: UCI-HAR shape, ")
                                   + f"Net(MyLSTM[MyVMLMFCell]) 1 layer, B={rows_gpu}/GPU T=128 I=9 H=180 w_rank=16 "
                                     "u_rank=16, CE loss, fwd+bwd; "
                                   + ("kept parameter images (--keep-images): no pack_kernel in the step" if args.keep_images else
                                      "parameters packed inside every step, as a training loop sees it (with kept parameter "
                                      "images: ms_per_step_kept_images)")
                                   + (", in-place RCCL all-reduce (AVG) of the flat gradient buffers" if collective else ""),
                       "global_batch": rows_gpu * world, "batch_per_gpu": rows_gpu, "seq_len": T,
                       "parallelism": f"dp{world}", "scaling": "strong" if strong else "weak",
                       "value_counts": "64-row batches x T timesteps per second",
                       "allreduce_transport": reducer.transport_used() if collective else None,
                       # ranks of the gradient exchange and WHO counted them: RCCL itself (ncclCommCount, C-ABI transport) or only
                       # the size of the torch.distributed group (backend named: over gloo there is no RCCL communicator)
                       "exchange_ranks": rccl_ranks, "exchange_ranks_counted_by": ranks_counted_by,
                       # (only a count RCCL itself gave: "ncclCommCount ..."; the torch transport's label names its backend, "nccl" too)
                       "rccl_ranks": rccl_ranks if (ranks_counted_by or "").startswith("ncclCommCount") else None,
                       "collectives_per_step": collectives_per_step,
                       "reduced_grad_norm_equal_across_ranks": grads_equal,
                       "launch": launch_mode,
                       "parameter_images": "kept while the parameters are unchanged (--keep-images)" if args.keep_images else
                                           "packed inside every step (what a training loop sees)",
                       "criterion": "torch.nn.functional.cross_entropy" if args.torch_loss else
                                    ("vmlmf_amd.cross_entropy" if args.separate_loss else
                                     "vmlmf_amd.Net.loss (cross-entropy riding on the forward launch, C ABI vmlmf_ce)")},
            "eager_ms_per_step": round(dt_eager / args.steps * 1e3, 4),
            "sample_timesteps_per_s": round(value * B_PER_GPU, 1),
            "step_flops": rows * F_STEP * world,
            "step_tflops": round(rows * F_STEP * world / (ms_per_step * 1e-3) / 1e12, 3),
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 3),
                         "peak": F32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / F32_MATRIX_PEAK_TFLOPS, 5),
                         "recurrence_only_achieved": round(achieved_rec, 3),
                         "recurrence_only_frac": round(achieved_rec / F32_MATRIX_PEAK_TFLOPS, 5), "traffic": traffic,
                         "traffic_unit": "bytes per launch", "traffic_source": traffic_note,
                         "launch_us": round(rec[dom], 2), "us_per_timestep": round(rec[dom] / T, 4),
                         # one batch row per CU: rows_gpu of 256 CUs are busy; the same rate against THEIR share of the peak
                         "launch_workgroups_one_per_cu": min(launch_wgs, cus),
                         "frac_of_those_cus": round(achieved / (F32_MATRIX_PEAK_TFLOPS * min(launch_wgs, cus) / float(cus)), 5),
                         "utilisation": util, "utilisation_source": util_note,
                         "flops_per_launch": flops,
                         "contains": ("recurrence (10 H ru per sample-step) + the weight-gradient products formed inside the rows' workgroups "
                                      "(rec4_bwd_kernel: 8 H I + 8 H ru + 2 H ru per sample-step on fp32 MFMA, operands from LDS)") if (riding and inrow) else
                                     ("recurrence (10 H ru per sample-step) + the weight-gradient products riding on the launch "
                                      "(8 H I + 8 H ru + 2 H ru per sample-step, on otherwise idle CUs)") if riding else
                                     ("the forward's algorithmic work of SURVEY section 8d: input side 2 I rw + 8 H rw (formed by the "
                                      "launch's x-projection wave) + recurrence 10 H ru per sample-step") if x_inside else
                                     "recurrence (10 H ru per sample-step)",
                         "measured": "HIP event pairs on the launch stream over the eager timed region of the same K "
                                     "steps (events cannot be read inside a replayed hipGraph)",
                         "note": "fp32: MFMA peak == vector peak on gfx950; the kernel is a 2T-long dependent "
                                 "chain on 64 of 256 CUs (one batch row per CU), see DESIGN.md"},
            "kernels_us": kern,
            "kernels_us_region": "EAGER launches bracketed by HIP event pairs in an untimed breakdown pass - not the replayed hipGraph `value` / "
                                 "`ms_per_step` come from (a graph node cannot be bracketed), so they need not add up to ms_per_step; "
                                 "roofline.launch_us is from the eager timed region too (its fraction is the conservative one)",
            "allreduce_ms": None if allreduce_ms is None else round(allreduce_ms, 4),
            "allreduce_bytes": 4 * reducer.numel() if collective else 0,
            "adam_ms": round(adam_ms, 4),
            "fused_adam_ms": round(fused_adam_ms, 4),
            "other_scaling_mode": other_mode,
            "loss": round(float(loss.item()), 6),
        }
        if world == 1 and not strong and not args.no_extra:
            try:
                out["other_configs"] = other_configs()
            except Exception as e:   # never at the expense of the line itself
                log(f"other_configs failed ({type(e).__name__}: {e})")
        if world == 1 and not strong and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["speedup_vs_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
