"""First-contact diagnostic on a GPU box: error table for a few cases + raw timings (not a test)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import vmlmf_oracle as O
from hip_util import run_hip, run_literal


def report(tag, got, ref):
    for k in ("y", "hT", "cT", "dx", "dh0", "dc0"):
        if k in got and k in ref:
            e = np.abs(got[k] - ref[k]).max()
            print(f"  {tag:10s} {k:5s} maxerr {e:.3e}  (|ref| {np.abs(ref[k]).max():.3e}) finite={np.isfinite(got[k]).all()}")
    if "G" in got and "G" in ref:
        for k in ref["G"]:
            e = np.abs(got["G"][k] - ref["G"][k]).max()
            print(f"  {tag:10s} G.{k:7s} maxerr {e:.3e}  (|ref| {np.abs(ref['G'][k]).max():.3e})")


def case(variant, B, T, I, H, rw, ru, tm, state, seed=0):
    rng = np.random.Generator(np.random.PCG64(seed + 5))
    P = O.make_params(variant, I, H, rw, ru if variant in (O.V2, O.V4) else ru[0], seed=3)
    shp = (T, B, I) if tm else (B, T, I)
    x = rng.standard_normal(shp).astype(np.float32)
    h0 = (0.4 * rng.standard_normal((B, H))).astype(np.float32) if state else None
    c0 = (0.4 * rng.standard_normal((B, H))).astype(np.float32) if state else None
    dy = rng.standard_normal(shp[:2] + (H,)).astype(np.float32)
    dhT = rng.standard_normal((B, H)).astype(np.float32)
    dcT = rng.standard_normal((B, H)).astype(np.float32)
    print(f"case v{variant} B{B} T{T} I{I} H{H} rw{rw} ru{ru} tm{tm} state{state}")
    try:
        got = run_hip(variant, P, x, h0, c0, dy, dhT, dcT, time_major=tm)
    except Exception as e:
        print("  HIP FAILED:", repr(e)[:300])
        return
    ref = run_literal(variant, P, x, h0, c0, dy, dhT, dcT, time_major=tm)
    report("hip-vs-f64", got, ref)


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    case(O.V1, 2, 1, 4, 16, 2, [3], False, True)
    case(O.V1, 2, 3, 4, 16, 2, [3], False, False)
    case(O.V1, 4, 6, 9, 180, 16, [16], False, True)
    case(O.V1, 3, 4, 12, 130, 32, [32], True, True)
    case(O.V2, 4, 5, 6, 20, 3, [2, 5], False, True)
    case(O.V3, 6, 5, 24, 24, 4, [6], True, True)
    case(O.V4, 9, 4, 20, 20, 3, [4, 2], True, True)
    case(O.V1, 300, 3, 6, 40, 4, [4], False, True)
    # timing, config A
    import ctypes
    from vmlmf_amd import _lib, MyLSTM, MyVMLMFCell
    torch.manual_seed(0)
    rnn = MyLSTM(9, hidden_layer_sizes=[180], batch_first=True, w_rank=16, u_ranks=[16], cell=MyVMLMFCell).cuda()
    x = torch.randn(64, 128, 9, device="cuda")
    for it in range(3):
        rnn.zero_grad()
        y, _ = rnn(x)
        y.sum().backward()
    torch.cuda.synchronize()
    L = _lib.lib()
    L.vmlmf_profile_enable(0xff)
    t0 = time.perf_counter()
    n = 20
    for it in range(n):
        rnn.zero_grad()
        y, _ = rnn(x)
        y.sum().backward()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    us = (ctypes.c_float * 8)()
    cnt = (ctypes.c_int32 * 8)()
    L.vmlmf_profile_read(us, cnt, 1)
    L.vmlmf_profile_enable(0)
    print(f"config A fwd+bwd wall (profiling on): {(t1 - t0) / n * 1e3:.3f} ms/step")
    for k in range(8):
        if cnt[k]:
            print(f"  {L.vmlmf_kernel_name(k).decode():16s} {us[k] / cnt[k]:9.1f} us  x{cnt[k]}")
    t0 = time.perf_counter()
    for it in range(n):
        rnn.zero_grad()
        y, _ = rnn(x)
        y.sum().backward()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"config A fwd+bwd wall (profiling off): {(t1 - t0) / n * 1e3:.3f} ms/step")
