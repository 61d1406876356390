"""GPU: the N > 1 code path of bench.py, run by the driver's GPU test tier on its one-GPU box (verdict r3, item 1).

RCCL refuses two ranks on one device, so the ranks share GPU 0 and exchange their gradients over gloo
(VMLMF_BENCH_REHEARSAL=1): everything else is what an 8-GPU run executes - the parent spawning one process per rank, contiguous
sharding, the flat (HAR) / bucketed (LM) gradient exchange inside the step, barriers, MAX over ranks, both scaling modes, ONE
JSON line from rank 0.  The lines say REHEARSAL: their numbers measure nothing.  Also here: the one-rank self-tests of the
RCCL paths that the one-GPU box CAN run on hardware (the C-ABI transport and the all-reduce captured inside the hipGraph).
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, rehearsal=True, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    if rehearsal:
        env["VMLMF_BENCH_REHEARSAL"] = "1"
    env["VMLMF_BENCH_RANK_TIMEOUT"] = str(timeout - 60)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + ["--no-cpu-baseline", "--no-extra"],
                       capture_output=True, text=True, env=env, timeout=timeout)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE JSON line, nothing else on stdout
    return json.loads(lines[0]), r.stderr


@pytest.mark.parametrize("mode", ["weak", "strong"])
def test_two_ranks_walk_the_headline_bench(mode):
    """`python bench.py --gpus 2` (the driver's plain form: the parent starts the ranks itself), weak (64 rows per rank) and
    strong (BASELINE configs[3]: global batch 512 split contiguously)."""
    args = ["--gpus", "2", "--steps", "8", "--warmup", "3"] + (["--global-batch", "512"] if mode == "strong" else [])
    j, err = _bench(args)
    assert j["n_gpus"] == 2 and j["scaling"] == mode and "REHEARSAL" in j["data"]
    c = j["config"]
    assert c["collectives_per_step"] == 1                          # ONE exchange per step
    assert c["exchange_ranks"] == 2 and "gloo" in c["exchange_ranks_counted_by"] and c["rccl_ranks"] is None
    assert c["batch_per_gpu"] == (256 if mode == "strong" else 64) and c["global_batch"] == 2 * c["batch_per_gpu"]
    assert c["reduced_grad_norm_equal_across_ranks"] is True       # both ranks hold the same averaged gradient
    assert j["loss"] == j["loss"] and 0.0 < j["loss"] < 10.0 and j["value"] > 0 and j["ms_per_step"] > 0
    assert j["ms_per_step_min_over_ranks"] <= j["ms_per_step"]      # MAX over ranks is what `value` is computed from
    assert j["other_scaling_mode"] is None or j["other_scaling_mode"]["scaling"] != mode


def test_two_ranks_walk_the_lm_network_of_configs4():
    """`python bench.py --config E --gpus 2`: BASELINE configs[4]'s network data-parallel - contiguous column shards of the
    (T, B) token batch, bucketed SUM all-reduce started from inside the backward pass (vocabulary projection first, embedding
    last), clip_grad_norm_ on the REDUCED gradients, rank-local state carry (vmlmf_amd.dp.LmDataParallel; lm_test.py:196-207)."""
    j, err = _bench(["--config", "E", "--gpus", "2", "--global-batch", "64", "--steps", "4", "--warmup", "2"])
    c = j["config"]
    assert j["n_gpus"] == 2 and "REHEARSAL" in j["data"] and "configs[4]" in c["workload"]
    assert c["global_batch"] == 64 and c["batch_per_gpu"] == 32
    assert c["collectives_per_step"] == 5 and c["collectives_started_inside_backward"] >= 4
    assert c["exchange_ranks"] == 2 and "gloo" in c["exchange_ranks_counted_by"]
    assert j["reduced_grad_norm_equal_across_ranks"] is True and j["reduced_grad_norm"] > 0
    layer = 650 * 32 + 4 * 650 * 32 + 8 * 650 + 2 * 650 + 2 * (2 * 325 * 32 + 2 * 32 * 1300)       # 318 500 per group layer
    assert j["allreduce_bytes"] == 4 * (2 * 10000 * 650 + 10000 + 2 * layer)                       # every gradient, once
    for k in ("loss_global", "train_loss_local", "train_clip_norm", "ms_per_step", "train_step_ms"):
        assert j[k] == j[k] and j[k] > 0, k
    assert 8.0 < j["loss_per_token"] < 10.5                         # ln(10000) = 9.21 at random initialisation


def test_two_processes_exchange_over_the_peer_to_peer_path():
    """vmlmf_p2p_* (ABI 13; SURVEY section 8e: the one-shot exchange for the HAR network's 121 KiB of gradients, train.py:64-65):
    two processes on this box's ONE GPU map each other's staging areas over hipIpc, write their buffers into them and sum in rank
    order - every exchange bit-equal to the same buffers reduced over gloo, for buffer sizes that are no multiple of four, both
    parities of the staging area, SUM and AVG, a run of 200 exchanges of random sizes without a host synchronisation in between, and
    through vmlmf_amd.dp.FlatGradAllReduce(transport="p2p")."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0", VMLMF_WRIDE="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "p2p_worker.py")], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0 and "P2P-OK 9" in out, out[-1000:] + err[-3000:]


def test_two_ranks_walk_the_headline_bench_over_the_peer_to_peer_transport():
    """`python bench.py --gpus 2 --transport p2p` (rehearsal: both ranks on GPU 0): the step's one exchange runs through vmlmf_p2p_* -
    the gloo group only carries the handles - and both ranks end up with the same averaged gradient."""
    j, err = _bench(["--gpus", "2", "--steps", "8", "--warmup", "3", "--transport", "p2p"])
    c = j["config"]
    assert j["n_gpus"] == 2 and "REHEARSAL" in j["data"]
    assert c["allreduce_transport"].startswith("p2p"), c["allreduce_transport"]
    assert c["collectives_per_step"] == 1 and c["exchange_ranks"] == 2 and "vmlmf_p2p_connect" in c["exchange_ranks_counted_by"]
    assert c["reduced_grad_norm_equal_across_ranks"] is True
    assert j["loss"] == j["loss"] and 0.0 < j["loss"] < 10.0 and j["value"] > 0


def test_one_rank_self_tests_of_the_rccl_paths():
    """What one GPU can run of RCCL itself: the C-ABI communicator (ncclCommCount says 1) with the all-reduce eager and
    captured inside the hipGraph (`--graph-collective`), and the LM network's bucketed exchange on the C-ABI side stream."""
    j, _ = _bench(["--force-collective", "--transport", "cabi", "--steps", "20", "--warmup", "5"], rehearsal=False)
    c = j["config"]
    assert c["allreduce_transport"].startswith("cabi") and c["rccl_ranks"] == 1 and "ncclCommCount" in c["exchange_ranks_counted_by"]
    assert c["collectives_per_step"] == 1 and c["launch"] == "hipgraph"
    j, _ = _bench(["--force-collective", "--transport", "cabi", "--graph-collective", "--steps", "20", "--warmup", "5"], rehearsal=False)
    assert j["config"]["launch"] == "hipgraph+allreduce" and j["config"]["rccl_ranks"] == 1 and j["train_step_ms"] is not None
    j, _ = _bench(["--force-collective", "--graph-collective", "--transport", "torch", "--steps", "20", "--warmup", "5"], rehearsal=False)
    assert j["config"]["launch"].startswith("hipgraph") and "nccl" in j["config"]["exchange_ranks_counted_by"]
    j, _ = _bench(["--config", "E", "--force-collective", "--transport", "cabi", "--global-batch", "32", "--steps", "4", "--warmup", "2"],
                  rehearsal=False)
    c = j["config"]
    assert c["allreduce_transport"].startswith("cabi") and c["exchange_ranks"] == 1 and c["collectives_per_step"] == 5
    assert j["reduced_grad_norm_equal_across_ranks"] is True and 8.0 < j["loss_per_token"] < 10.5


def test_the_default_single_gpu_line_keeps_its_contract():
    """`python bench.py` as the driver runs it at N = 1 (shortened): ONE JSON line with the graded fields, the roofline and
    cpu_baseline objects, the round-5 additions (harness, riding_workers) - and the headline step as three launches: no pack /
    criterion / reduce / finish launch in the per-kernel breakdown, the finishing launch present."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "5"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 30 and j["warmup"] == 5 and j["higher_is_better"] is True and j["vs_baseline"] is None
    assert j["dtype"] == "f32" and j["data"] == "synthetic" and "configs[1]" in j["config"]["workload"] and j["value"] > 0
    assert abs(j["value"] - 128 * 1e3 / j["ms_per_step"]) <= 1e-3 * j["value"]          # 64-row batches x T per second
    rf = j["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    assert rf["traffic"] is None or rf["traffic"] > 0
    cb = j["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    k = j["kernels_us"]
    assert k["pack_kernel"] == 0 and k["ce_fwd_kernel"] == 0 and k["reduce_cg_kernel"] == 0 and k["finish_kernel"] == 0, k
    assert k["rec_fwd_kernel"] > 0 and k["rec_bwd_kernel"] > 0 and k["finish2_kernel"] > 0, k
    h = j["harness"]
    assert h["two_line_opt_in_ms"] < h["eager_package_loop_ms"] <= h["unchanged_loop_ms"] * 1.05, h
    rw = j["riding_workers"]
    assert rw["armed"] is True and rw["stand_alone_weight_gradient_launch_in_step"] is False and rw["status"] == 0, rw
