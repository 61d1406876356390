"""CPU: bench.py's multi-GPU front end (the driver runs `python bench.py --gpus N` as well as the torchrun line)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_spawns_ranks_and_refuses_when_the_devices_are_missing():
    """Without a torchrun environment `--gpus 2` must try to start two ranks itself - here there is no GPU, so the
    parent reports that and exits non-zero BEFORE anything touches a device (it is not a silently ignored flag)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2 and "--gpus 2 asked for" in r.stderr and r.stdout == ""


def test_strong_scaling_shards_tile_the_global_minibatch():
    sys.path.insert(0, ROOT)
    import bench
    xg, tg = bench.synthetic_batch(0, 512, 512)
    for world in (2, 4, 8):
        rows = 512 // world
        xs = [bench.synthetic_batch(r, rows, 512) for r in range(world)]
        assert np.array_equal(np.concatenate([a for a, _ in xs]), xg)
        assert np.array_equal(np.concatenate([b for _, b in xs]), tg)
    # weak mode: every rank draws its own batch
    assert not np.array_equal(bench.synthetic_batch(0)[0], bench.synthetic_batch(1)[0])
