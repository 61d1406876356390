"""The whole language-model step of lm_test.py:196-209 at BASELINE config E's shape: Model (2 VMLMF layers, H 650,
rank 32, vocabulary 10 000, T 35) -> nll_loss -> backward -> clip + SGD.

  python tools/bench_lm.py [B]            one MI355X: fused loss / update against the reference's own formulations in stock ops
  python tools/bench_lm.py [B] --dropout 0.5   the shipped step at the reference's dropout (lm_test.py: --dropout 0.5): p = 0, p with the
                                          package's mask-free dropout (in the embedding gather and the layers' own launches), p with
                                          nn.Dropout's launches (Model.stock_dropout)
  python bench.py --config E [--gpus N]   run_config_e() below: BASELINE configs[4] data-parallel (global batch 256 split
                                          contiguously over the ranks, bucketed SUM all-reduce overlapping the recurrent
                                          layers' backward, clip after the reduce, rank-local state carry)
Not the graded bench line (bench.py without --config is)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
import torch
from vmlmf_amd import Model, MyVMLSTMGroup, nll_loss, optim

DEV = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() and __name__ == "__main__" else 256
T, H, V = 35, 650, 10000
RW, RU = 32, [32, 32]
# algorithmic flops per sample-timestep and layer, forward: 2 I rw + 8 H rw + 10 H sum(ru); fwd + bwd = 3 F
F_LAYER = 2 * H * RW + 8 * H * RW + 10 * H * sum(RU)
F_HEAD = 2 * H * V                      # the vocabulary projection, per token, forward


def build_e_model(dev, group=True, seed=0):
    """configs[4]'s network.  The reference's Model cannot build the group layers (constructor quirk, vmlmf_lm.py:387-392):
    they are put in by hand, then initialised as Model.reset_parameters does (U(-winit, winit), winit 0.05)."""
    torch.manual_seed(seed)
    model = Model(V, H, 2, 0.0, 0.05, w_rank=RW, u_ranks=[RU[0]], lstm_type="vmlmf")
    if group:
        model.rnns = torch.nn.ModuleList([MyVMLSTMGroup(H, H, w_rank=RW, u_ranks=list(RU)) for _ in range(2)])
        model.reset_parameters()
    return model.to(dev)


def synthetic_tokens(global_batch, n_minibatches=2, seed=4321):
    """(T, B) int64 inputs and targets for a few consecutive minibatches: seeded numpy PCG64, the same on every rank."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return [(rng.integers(0, V, size=(T, global_batch)), rng.integers(0, V, size=(T, global_batch))) for _ in range(n_minibatches)]


def run_config_e(args, ctx):
    """bench.py --config E.  ctx: world, rank, dev, collective, rehearsal, result_fd, log, barrier (made by bench.py).
    A step = lm_test.py:196-204 on this rank's columns: detach states -> Model.forward -> nll_loss -> backward with the
    bucketed SUM all-reduce running underneath -> wait.  K timed steps between barriers, MAX over ranks.  `train_step_ms`
    adds clip_grad_norm_ on the reduced gradients + the SGD update (lm_test.py:204-207)."""
    import torch.distributed as dist
    from vmlmf_amd.dp import LmDataParallel
    world, rank, dev = ctx["world"], ctx["rank"], ctx["dev"]
    collective, log, barrier = ctx["collective"], ctx["log"], ctx["barrier"]
    per_gpu = getattr(args, "batch_per_gpu", 0)
    gb = per_gpu * world if per_gpu else (args.global_batch or 256)
    if gb % world:
        raise SystemExit(f"global batch {gb} is not divisible by {world} ranks")
    bl = gb // world
    model = build_e_model(dev, group=not getattr(args, "plain_layers", False))
    dp = LmDataParallel(model, lr=1.0, max_norm=5.0, transport="cabi" if args.transport == "cabi" else "torch")
    dp.reducer.always = bool(args.force_collective)
    batches = [(dp.shard(torch.tensor(x, device=dev)), dp.shard(torch.tensor(y, device=dev))) for x, y in synthetic_tokens(gb)]
    assert batches[0][0].shape == (T, bl)
    states = model.state_init(bl)
    log(f"config E: world={world} global batch {gb} ({bl}/rank) device={torch.cuda.get_device_name(dev)}")

    k = [0]

    def fwd_bwd():
        nonlocal states
        x, y = batches[k[0] % len(batches)]
        k[0] += 1
        loss, states = dp.forward_backward(x, y, states)
        return loss

    def train():
        nonlocal states
        x, y = batches[k[0] % len(batches)]
        k[0] += 1
        loss, norm, states = dp.step(x, y, states)
        return loss, norm

    def timed(fn, n):
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        barrier()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        lo = t.clone()
        if collective:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        return float(t.item()), float(lo.item()), out

    for _ in range(args.warmup):
        fwd_bwd()
    dt, dt_min, loss = timed(fwd_bwd, args.steps)
    coll, over, nbytes = dp.reducer.last_collectives, dp.reducer.last_overlapped, dp.reducer.bytes_per_step
    # every rank must hold the same reduced gradients: their norm, computed per rank, has to agree bit for bit
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)).reshape(1)
    gmax, gmin = gn.clone(), gn.clone()
    if collective:
        dist.all_reduce(gmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(gmin, op=dist.ReduceOp.MIN)
    gloss = float(dp.global_loss(loss))
    # the exchange alone, on the gradients of the last step (nothing to hide behind): what the overlap has to cover
    exch_ms = None
    if collective:
        def exch():
            dp.reducer.arm()
            dp.reducer.wait()
        for _ in range(3):
            exch()
        e_dt, _, _ = timed(exch, 10)
        exch_ms = e_dt / 10 * 1e3
    for _ in range(min(args.warmup, 3)):
        train()
    tt, _, (tloss, tnorm) = timed(train, args.steps)
    ranks, counted_by = dp.reducer.exchange_ranks()
    if rank == 0:
        ms = dt / args.steps * 1e3
        flops = 3 * (2 * F_LAYER + F_HEAD) * T * gb
        out = {
            "metric": "RNN timesteps/sec (fwd+bwd), PTB LM network of BASELINE configs[4]",
            "value": round(T * args.steps / dt, 1), "unit": "RNN timesteps/s (minibatches of the global batch x T per second)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "ms_per_step_min_over_ranks": round(dt_min / args.steps * 1e3, 4),
            "train_step_ms": round(tt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak" if per_gpu else "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (uniform random tokens, random-init weights)" if not ctx["rehearsal"] else
                    "synthetic; REHEARSAL: all ranks share GPU 0 over gloo - not a scaling measurement",
            "config": {"workload": f"BASELINE configs[4]: PTB LM network (vmlmf_lm.py Model: Embed 10000x650 -> 2 x "
                                   f"{'MyVMLSTMGroup g=2 ranks 32/[32,32]' if not getattr(args, 'plain_layers', False) else 'MyVMLSTM rank 32'}"
                                   f" -> Linear 650->10000 -> nll_loss), T=35, global batch {gb} split contiguously over the "
                                   f"ranks ({bl}/GPU), fwd+bwd + bucketed SUM all-reduce; clip + SGD in train_step_ms",
                       "global_batch": gb, "batch_per_gpu": bl, "seq_len": T, "parallelism": f"dp{world}",
                       "allreduce_transport": dp.reducer.transport_used() if collective else None,
                       "exchange_ranks": ranks, "exchange_ranks_counted_by": counted_by,
                       "collectives_per_step": coll, "collectives_started_inside_backward": over,
                       "buckets": "fc.w, fc.b | rnns.1 | rnns.0 | embed.w (backward order)"},
            "words_per_s": round(T * gb * args.steps / dt, 1),
            "step_tflops": round(flops / (ms * 1e-3) / 1e12, 2),
            "allreduce_bytes": nbytes, "allreduce_alone_ms": None if exch_ms is None else round(exch_ms, 4),
            "loss_global": round(gloss, 5), "loss_per_token": round(gloss * T / (T * gb), 5),
            "reduced_grad_norm": float(gmax.item()), "reduced_grad_norm_equal_across_ranks": bool(gmax.item() == gmin.item()),
            "train_loss_local": round(float(tloss), 5), "train_clip_norm": round(float(tnorm), 5),
        }
        os.write(ctx["result_fd"], (json.dumps(out) + "\n").encode())
    dp.reducer.close()


def stock_nll(scores, y):          # lm_test.py:140-153 as written
    batch_size = y.size(1)
    expscores = scores.exp()
    probabilities = expscores / expscores.sum(1, keepdim=True)
    answerprobs = probabilities[range(len(y.reshape(-1))), y.reshape(-1)]
    return torch.mean(-torch.log(answerprobs) * batch_size)


def run(tag, group, fused, head=False, dropout=0.0, stock_dropout=False):
    """head: Model.loss (projection + loss with the gradient formed in place, tuned GEMM forms, the package's embedding
    gradient) instead of model(x) -> nll_loss(scores, y)."""
    import vmlmf_amd
    torch.manual_seed(0)
    model = Model(V, H, 2, dropout, 0.05, w_rank=32, u_ranks=[32], lstm_type="vmlmf")
    model.stock_dropout = stock_dropout
    if group:   # the reference's Model cannot build the group layers (constructor quirk): put them in by hand
        model.rnns = torch.nn.ModuleList([MyVMLSTMGroup(H, H, w_rank=32, u_ranks=[32, 32]) for _ in range(2)])
        model.reset_parameters()
    model = model.to(DEV)
    x = torch.randint(0, V, (T, B), device=DEV)
    y = torch.randint(0, V, (T, B), device=DEV)
    states = model.state_init(B)

    def step():
        model.zero_grad(set_to_none=True)
        st = [(h.detach(), c.detach()) for h, c in states]
        if head:
            loss, _ = model.loss(x, y, st)
            loss.backward(vmlmf_amd.unit_gradient(DEV))
        else:
            scores, _ = model(x, st)
            loss = nll_loss(scores, y) if fused else stock_nll(scores, y)
            loss.backward()
        if fused:
            optim.clip_sgd_step(model.parameters(), lr=1e-3, max_norm=5.0)
        else:
            with torch.no_grad():
                torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
                for p in model.parameters():
                    p -= 1e-3 * p.grad

    for _ in range(6):   # (3 left a one-off - GEMM form timing of a new shape, allocator growth - inside the timed steps: 2.9 ms once at B = 32)
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(json.dumps({"config": tag, "B": B, "T": T, "vocab": V, "fused_loss_and_update": fused, "head_in_place": head,
                      "dropout": dropout, "dropout_launches": "none" if dropout == 0 else ("nn.Dropout" if stock_dropout else "package (mask-free)"),
                      "ms_per_step_eager": round(ms, 3), "words_per_s": round(T * B / ms * 1e3)}), flush=True)
    return ms


if __name__ == "__main__":
    if "--dropout" in sys.argv:
        p = float(sys.argv[sys.argv.index("--dropout") + 1])
        tag = "E-model: Embed + 2 x MyVMLSTMGroup + Linear + nll"
        base = run(tag, True, True, head=True)
        ours = run(tag, True, True, head=True, dropout=p)
        stock = run(tag, True, True, head=True, dropout=p, stock_dropout=True)
        print(json.dumps({"dropout": p, "ms_p0": round(base, 3), "ms_package": round(ours, 3), "ms_nn_dropout": round(stock, 3),
                          "package_over_p0": round(ours / base, 4), "nn_dropout_over_p0": round(stock / base, 4)}), flush=True)
        sys.exit(0)
    run("E-model: Embed + 2 x MyVMLSTM + Linear + nll", False, True, head=True)
    if "--only-head" in sys.argv:   # (for a profile of the shipped step alone)
        sys.exit(0)
    run("E-model: Embed + 2 x MyVMLSTMGroup + Linear + nll", True, True, head=True)
    run("E-model: Embed + 2 x MyVMLSTM + Linear + nll", False, True)
    run("E-model: Embed + 2 x MyVMLSTM + Linear + nll", False, False)
    run("E-model: Embed + 2 x MyVMLSTMGroup + Linear + nll", True, True)
