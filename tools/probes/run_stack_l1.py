"""Single wide-input layer (H 180, I 77, r 16, B 64): wavefront launch (in-kernel x side) against the chained kernels
(xproj + recurrence + dqx_dx) over the sequence length: where the x-team's per-step cost overtakes the launches it saves."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from vmlmf_amd import MyLSTM, MyVMLMFCell
H, I, r, B = 180, 77, 16, 64
if len(sys.argv) > 4:
    H, I, r = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
BS = [int(v) for v in sys.argv[4].split(',')] if len(sys.argv) > 4 else [B]
TS = [int(v) for v in sys.argv[5].split(',')] if len(sys.argv) > 5 else [16, 32, 64, 128, 256]
LS = [int(v) for v in sys.argv[6].split(',')] if len(sys.argv) > 6 else [1, 2]
for L, B in [(l, b) for l in LS for b in BS]:
    for T in TS:
        res = {}
        for mode in ("0", "1"):
            os.environ["VMLMF_STACK"] = mode
            torch.manual_seed(0)
            rnn = MyLSTM(I, hidden_layer_sizes=[H] * L, batch_first=True, w_rank=r, u_ranks=r, cell=MyVMLMFCell).cuda()
            x = torch.randn(B, T, I, device="cuda")
            def step():
                rnn.zero_grad(set_to_none=True)
                y, _ = rnn(x)
                y[:, -1].sum().backward()
            side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3): step()
            torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g): step()
            for _ in range(20): g.replay()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(100): g.replay()
            torch.cuda.synchronize(); res[mode] = (time.perf_counter() - t0) / 100 * 1e3
        print(f"L {L} B {B:4d} T {T:4d}: chained {res['0']:.4f} ms  wavefront {res['1']:.4f} ms", flush=True)
