"""Host time of the eager training step at config A, piece by piece (no GPU sync inside the timed loops): zero_grad, Net.loss
(forward), backward, the package's Adam; and the same for the unchanged loop (nn.CrossEntropyLoss + torch.optim.Adam)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import vmlmf_amd
from vmlmf_amd import MyLSTM, MyVMLMFCell, Net

torch.manual_seed(0)
net = Net(9, layer_sizes=[180], w_rank=16, u_rank=[16], model=MyLSTM, cell=MyVMLMFCell).cuda()
x = torch.randn(64, 128, 9, device="cuda")
t = torch.randint(0, 18, (64,), device="cuda")
one = vmlmf_amd.unit_gradient("cuda")
opt = vmlmf_amd.optim.Adam(net.parameters(), lr=2e-3)
N = 400


def timed(fn, n=N):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3


def full():
    net.zero_grad(set_to_none=True)
    loss = net.loss(x, t)
    loss.backward(one)


def full_opt():
    opt.zero_grad(set_to_none=True)
    loss = net.loss(x, t)
    loss.backward(one)
    opt.step()


acc = {"zero": 0.0, "fwd": 0.0, "bwd": 0.0, "opt": 0.0}


def pieces():
    a = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    b = time.perf_counter()
    loss = net.loss(x, t)
    c = time.perf_counter()
    loss.backward(one)
    d = time.perf_counter()
    opt.step()
    e = time.perf_counter()
    acc["zero"] += b - a; acc["fwd"] += c - b; acc["bwd"] += d - c; acc["opt"] += e - d


print("fwd+bwd            host %.4f  wall %.4f ms" % timed(full))
print("fwd+bwd+Adam       host %.4f  wall %.4f ms" % timed(full_opt))
for k in acc: acc[k] = 0.0
h, w = timed(pieces)
tot = N + 30
print("pieces (host ms): " + ", ".join(f"{k} {v / tot * 1e3:.4f}" for k, v in acc.items()), " wall %.4f" % w)
with torch.no_grad():
    print("forward only (no_grad) host %.4f wall %.4f" % timed(lambda: net.loss(x, t)))
topt = torch.optim.Adam(net.parameters(), lr=2e-3)
crit = torch.nn.CrossEntropyLoss()


def unchanged():
    topt.zero_grad()
    loss = crit(net(x), t)
    loss.backward()
    topt.step()


print("unchanged loop     host %.4f  wall %.4f ms" % timed(unchanged))
if "--prof" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(300): full_opt()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)

# ---- where the forward's host time goes: the C++ op alone (same arguments Net.loss ends up passing) vs the Python above it
from vmlmf_amd import functional as F
ops = F.torch_ops()
cell = net.rnn.rnncells[0]
if ops is not None:
    params = list(cell.kernel_params())
    unit, ticket = F.unit_gradient(x.device), F.ce_ticket(x.device)

    def op_only():
        return ops.sequence_loss(x, None, None, params, cell.variant, 1, 16, [16], False, 0, None, net.lin.weight, net.lin.bias, t, -100, unit, ticket)

    print("C++ op alone, grad mode   host %.4f wall %.4f" % timed(op_only))
    with torch.no_grad():
        print("C++ op alone, no_grad     host %.4f wall %.4f" % timed(op_only))

    def op_bwd():
        out = op_only()
        out[4].backward(one)

    print("C++ op + backward         host %.4f wall %.4f" % timed(op_bwd))
lib = F._lib.lib()
st = torch.zeros(2, dtype=torch.int64, device="cuda")
sn = torch.zeros(2, dtype=torch.int64, device="cuda")
stream = F._lib.raw_stream(x.device)
print("one tiny launch through ctypes (drop_advance) host %.4f wall %.4f" % timed(lambda: lib.vmlmf_dropout_advance(st.data_ptr(), sn.data_ptr(), stream)))
e = torch.empty(16, device="cuda")
print("torch.empty(16)           host %.4f" % timed(lambda: torch.empty(16, device="cuda"))[0])
print("e.zero_() (one launch)    host %.4f wall %.4f" % timed(lambda: e.zero_()))
