import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch, numpy as np
import vmlmf_oracle as O
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), flush=True)
os.system("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Core|Socket' ; cat /sys/fs/cgroup/cpu.max 2>/dev/null")
P = O.to_torch(O.make_params(O.V1, 9, 180, 16, 16, seed=3), requires_grad=True)
x, tgt = O.synthetic_batch(64, 128, 9)
lw = torch.randn(18, 180, requires_grad=True); lb = torch.zeros(18, requires_grad=True)
for nt in (int(sys.argv[1]) if len(sys.argv) > 1 else 8, 1, 4, 16):
    torch.set_num_threads(nt)
    t0 = time.perf_counter()
    loss, _ = O.literal_train_step_har(P, lw, lb, torch.tensor(x), torch.tensor(tgt))
    loss.backward()
    print("threads", nt, "step s", round(time.perf_counter() - t0, 3), flush=True)
