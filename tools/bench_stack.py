"""Durations of the two wavefront launches (HIP events around them, vmlmf_profile_enable) for L layers and T steps:
the slope over T is the time per step, the offset over L the lag per layer boundary.
   python tools/bench_stack.py [H=256] [rank=24] [B=128] [I=77]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VMLMF_STACK", "1")   # VMLMF_STACK=0: the chained per-layer kernels (durations are then per layer launch)
import ctypes
import torch
from vmlmf_amd import MyLSTM, MyVMLMFCell, _lib

H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
r = int(sys.argv[2]) if len(sys.argv) > 2 else 24
B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
I = int(sys.argv[4]) if len(sys.argv) > 4 else 77
lib = _lib.lib()
for L in (1, 2, 3, 4):
    if L * B > 256 and L > 1:
        continue
    for T in (24, 48, 96):
        torch.manual_seed(0)
        rnn = MyLSTM(I, hidden_layer_sizes=[H] * L, batch_first=True, w_rank=r, u_ranks=r, cell=MyVMLMFCell).cuda()
        x = torch.randn(B, T, I, device="cuda")
        for it in range(3):
            rnn.zero_grad(set_to_none=True)
            y, _ = rnn(x)
            y[:, -1].sum().backward()
        torch.cuda.synchronize()
        lib.vmlmf_profile_enable(0b1100)
        lib.vmlmf_profile_read(None, None, 1)
        n = 10
        for it in range(n):
            rnn.zero_grad(set_to_none=True)
            y, _ = rnn(x)
            y[:, -1].sum().backward()
        torch.cuda.synchronize()
        us = (ctypes.c_float * _lib.NKERNELS)()
        cnt = (ctypes.c_int32 * _lib.NKERNELS)()
        lib.vmlmf_profile_read(us, cnt, 1)
        lib.vmlmf_profile_enable(0)
        print(f"L {L} T {T:3d}: fwd {us[2] / max(cnt[2], 1):7.2f} us  bwd {us[3] / max(cnt[3], 1):7.2f} us   ({cnt[2]} + {cnt[3]} launches)", flush=True)
