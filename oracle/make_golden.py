"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE (build container only).

TEST INFRASTRUCTURE.  Run from the repo root:   python oracle/make_golden.py
Needs /root/reference (absent on the GPU box: nothing at test/bench time reads it; only the .npz
fixtures travel).  Fixtures are data only: inputs, parameters (or the seed of the repo's own numpy
generator that makes them) and the reference's outputs / autograd gradients.

Reference entry points exercised (V/ = /root/reference/rnn_compression_factorization_vmlmf/):
  MyVMLMFCell        V/src/models/vmlmf.py:38-125      bare cell and through MyLSTM (241-316)
  MyVMLMFCellg2      V/src/models/vmlmf_group.py:37-155  bare cell and through MyLSTM
  MyVMLSTM           V/src/models/vmlmf_lm.py:178-280
  MyVMLSTMGroup      V/src/models/vmlmf_lm.py:53-174   (B=40 only: scratch rows hard-coded, 112-113)
  Net                V/src/models/vmlmf.py:319-355 + the train.py:58-65 loop (3 Adam steps)
  MyLSTMCell         V/src/models/vmlmf.py:127-238     low-rank mode, bare cell and through MyLSTM
  MyVMLMFgCellg2     V/src/models/vmlmf_group.py:158-251  bare cell and through MyLSTM

`python oracle/make_golden.py NAME...` regenerates only the named fixtures.
"""
import os
import sys

import numpy as np
import torch

REF_SRC = "/root/reference/rnn_compression_factorization_vmlmf/src"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REF_SRC)
sys.path.insert(0, HERE)

from models.vmlmf import MyVMLMFCell, MyLSTMCell, MyLSTM, Net  # noqa: E402  (reference)
from models.vmlmf_group import MyVMLMFCellg2, MyVMLMFgCellg2   # noqa: E402  (reference)
from models.vmlmf_lm import MyVMLSTM, MyVMLSTMGroup          # noqa: E402  (reference)
import vmlmf_oracle as O                                     # noqa: E402

torch.set_num_threads(8)


def rng_of(seed):
    return np.random.Generator(np.random.PCG64(seed))


def load_into(module, P, prefix=""):
    """Copy the seeded parameter dict into a reference module (names are the reference's own)."""
    named = dict(module.named_parameters())
    with torch.no_grad():
        for k, v in P.items():
            cands = [prefix + k, prefix + "layers." + k]
            name = next(c for c in cands if c in named)
            named[name].copy_(torch.tensor(v))
    return named


def grads_of(module, P, prefix=""):
    named = dict(module.named_parameters())
    out = {}
    for k in P:
        name = next(c for c in [prefix + k, prefix + "layers." + k] if c in named)
        out[k] = named[name].grad.detach().numpy().copy()
    return out


def save(name, **arrs):
    flat = {}
    for k, v in arrs.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat[f"{k}/{kk}"] = np.asarray(vv)
        else:
            flat[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **flat)
    print(f"{name:28s} {os.path.getsize(path) / 1024:8.1f} KiB")


def make_cell(variant, I, H, rw, ru, g=2):
    if variant == O.V1:
        return MyVMLMFCell(I, H, w_rank=rw, u_ranks=ru if not isinstance(ru, list) else ru[0])
    if variant == O.V2:
        return MyVMLMFCellg2(I, H, w_rank=rw, u_ranks=ru, g=g)
    if variant == O.V3:
        return MyVMLSTM(I, H, w_rank=rw, u_ranks=ru if not isinstance(ru, list) else ru[0])
    if variant == O.V5:
        return MyLSTMCell(I, H, w_rank=rw, u_ranks=ru if not isinstance(ru, list) else ru[0])
    if variant == O.V6:
        return MyVMLMFgCellg2(I, H, w_rank=rw, u_ranks=ru, g=g)
    return MyVMLSTMGroup(I, H, w_rank=rw, u_ranks=ru, g=g)


HAR_CELL = {O.V1: MyVMLMFCell, O.V2: MyVMLMFCellg2, O.V5: MyLSTMCell, O.V6: MyVMLMFgCellg2}


def case_bare_cell(name, variant, B, I, H, rw, ru, seed):
    """One step through the reference cell with random (h, c) and random upstream grads."""
    r = rng_of(seed)
    P = O.make_params(variant, I, H, rw, ru, seed=seed + 1)
    cell = make_cell(variant, I, H, rw, ru)
    load_into(cell, P)
    x = torch.tensor(r.standard_normal((B, I)).astype(np.float32), requires_grad=True)
    h = torch.tensor((0.5 * r.standard_normal((B, H))).astype(np.float32), requires_grad=True)
    c = torch.tensor((0.5 * r.standard_normal((B, H))).astype(np.float32), requires_grad=True)
    dh = r.standard_normal((B, H)).astype(np.float32)
    dc = r.standard_normal((B, H)).astype(np.float32)
    if variant not in (O.V3, O.V4):
        hn, cn = cell(x, (h, c))
    else:
        hn, cn = cell.lstm_step(x, h, c)
    ((hn * torch.tensor(dh)).sum() + (cn * torch.tensor(dc)).sum()).backward()
    save(name, meta=np.array([variant, B, 1, I, H, rw] + list(np.atleast_1d(ru))), P=P,
         x=x.detach().numpy(), h0=h.detach().numpy(), c0=c.detach().numpy(), dh=dh, dc=dc,
         h1=hn.detach().numpy(), c1=cn.detach().numpy(), dx=x.grad.numpy(), dh0=h.grad.numpy(),
         dc0=c.grad.numpy(), G=grads_of(cell, P))


def case_har_seq(name, variant, B, T, I, H, rw, ru, seed):
    """One layer through the reference MyLSTM (batch-first, zero initial state), full tensors."""
    r = rng_of(seed)
    P = O.make_params(variant, I, H, rw, ru, seed=seed + 1)
    rnn = MyLSTM(I, hidden_layer_sizes=[H], batch_first=True, w_rank=rw, u_ranks=ru, cell=HAR_CELL[variant])
    load_into(rnn, P, prefix="rnncells.0.")
    x = torch.tensor(r.standard_normal((B, T, I)).astype(np.float32), requires_grad=True)
    dy = r.standard_normal((B, T, H)).astype(np.float32)
    dhT = r.standard_normal((B, H)).astype(np.float32)
    y, hcat = rnn(x)
    ((y * torch.tensor(dy)).sum() + (hcat * torch.tensor(dhT)).sum()).backward()
    save(name, meta=np.array([variant, B, T, I, H, rw] + list(np.atleast_1d(ru))), P=P,
         x=x.detach().numpy(), dy=dy, dhT=dhT, y=y.detach().numpy(), hT=hcat.detach().numpy(),
         dx=x.grad.numpy(), G=grads_of(rnn, P, prefix="rnncells.0."))


def case_lm_seq(name, variant, B, T, H, rw, ru, seed, scale=0.1, full=True, xscale=1.0):
    """One LM layer through the reference forward(x, states), time-major, non-zero initial state."""
    r = rng_of(seed)
    P = O.make_params(variant, H, H, rw, ru, seed=seed + 1, scale=scale)
    layer = make_cell(variant, H, H, rw, ru)
    load_into(layer, P)
    x = torch.tensor((xscale * r.standard_normal((T, B, H))).astype(np.float32), requires_grad=True)
    h0 = torch.tensor((0.3 * r.standard_normal((B, H))).astype(np.float32), requires_grad=True)
    c0 = torch.tensor((0.3 * r.standard_normal((B, H))).astype(np.float32), requires_grad=True)
    dy = r.standard_normal((T, B, H)).astype(np.float32)
    dhT = r.standard_normal((B, H)).astype(np.float32)
    dcT = r.standard_normal((B, H)).astype(np.float32)
    y, (hT, cT) = layer(x, (h0, c0))
    ((y * torch.tensor(dy)).sum() + (hT * torch.tensor(dhT)).sum() + (cT * torch.tensor(dcT)).sum()).backward()
    arrs = dict(meta=np.array([variant, B, T, H, H, rw] + list(np.atleast_1d(ru))),
                hT=hT.detach().numpy(), cT=cT.detach().numpy(), dh0=h0.grad.numpy(), dc0=c0.grad.numpy(),
                G=grads_of(layer, P))
    if full:
        arrs.update(P=P, x=x.detach().numpy(), h0=h0.detach().numpy(), c0=c0.detach().numpy(),
                    dy=dy, dhT=dhT, dcT=dcT, y=y.detach().numpy(), dx=x.grad.numpy())
    else:  # large shape: inputs are re-generated from (seed, scale, xscale); keep strided samples
        arrs.update(seed=np.array([seed]), scale=np.array([scale, xscale]),
                    y_s=y.detach().numpy()[::4, ::4], dx_s=x.grad.numpy()[::4, ::4])
    save(name, **arrs)


def regen_lm_inputs(seed, B, T, H, xscale=1.0):
    """Same draw order as case_lm_seq (used by the tests for the large fixtures)."""
    r = rng_of(seed)
    x = (xscale * r.standard_normal((T, B, H))).astype(np.float32)
    h0 = (0.3 * r.standard_normal((B, H))).astype(np.float32)
    c0 = (0.3 * r.standard_normal((B, H))).astype(np.float32)
    dy = r.standard_normal((T, B, H)).astype(np.float32)
    dhT = r.standard_normal((B, H)).astype(np.float32)
    dcT = r.standard_normal((B, H)).astype(np.float32)
    return x, h0, c0, dy, dhT, dcT


def case_lm_carry(name, seed):
    """V3 layer, two consecutive minibatches with detached state carry (lm_test.py:196-203) and the
    reference's nll-style scaling replaced by a fixed linear functional (the LM head is out of scope)."""
    B, T, H, rw, ru = 6, 5, 8, 3, 3
    r = rng_of(seed)
    P = O.make_params(O.V3, H, H, rw, ru, seed=seed + 1)
    layer = make_cell(O.V3, H, H, rw, ru)
    load_into(layer, P)
    xs = [r.standard_normal((T, B, H)).astype(np.float32) for _ in range(2)]
    ws = [r.standard_normal((T, B, H)).astype(np.float32) for _ in range(2)]
    states = (torch.zeros(B, H), torch.zeros(B, H))
    out = dict(meta=np.array([O.V3, B, T, H, H, rw, ru]), P=P)
    for i in range(2):
        layer.zero_grad()
        states = (states[0].detach(), states[1].detach())
        y, states = layer(torch.tensor(xs[i]), states)
        loss = torch.mean(y * torch.tensor(ws[i])) * B
        loss.backward()
        out[f"x{i}"], out[f"w{i}"] = xs[i], ws[i]
        out[f"loss{i}"] = np.array([loss.item()])
        out[f"hT{i}"], out[f"cT{i}"] = states[0].detach().numpy(), states[1].detach().numpy()
        out[f"G{i}"] = grads_of(layer, P)
    save(name, **out)


def case_config_a(name="cfgA_v1_uci"):
    """BASELINE config A/B: UCI-HAR shape, MyVMLMFCell, B=64 T=128 I=9 H=180 rank 16, through MyLSTM."""
    B, T, I, H, rw, ru = 64, 128, 9, 180, 16, 16
    P = O.make_params(O.V1, I, H, rw, ru, seed=3)
    x_np, _ = O.synthetic_batch(B, T, I, seed=1234)
    r = rng_of(4321)
    dy = r.standard_normal((B, T, H)).astype(np.float32)
    rnn = MyLSTM(I, hidden_layer_sizes=[H], batch_first=True, w_rank=rw, u_ranks=[ru], cell=MyVMLMFCell)
    load_into(rnn, P, prefix="rnncells.0.")
    x = torch.tensor(x_np, requires_grad=True)
    y, hcat = rnn(x)
    (y * torch.tensor(dy)).sum().backward()
    save(name, meta=np.array([O.V1, B, T, I, H, rw, ru]), seeds=np.array([3, 1234, 4321]),
         x_head=x_np[:2, :4], y_s=y.detach().numpy()[:, ::16], hT=hcat.detach().numpy(),
         dx=x.grad.numpy(), G=grads_of(rnn, P, prefix="rnncells.0."))


def case_config_a_group(name="cfgA_v2_uci"):
    """UCI-HAR shape with the group cell, ranks [16,16]."""
    B, T, I, H, rw, ru = 64, 128, 9, 180, 16, [16, 16]
    P = O.make_params(O.V2, I, H, rw, ru, seed=3)
    x_np, _ = O.synthetic_batch(B, T, I, seed=1234)
    r = rng_of(4321)
    dy = r.standard_normal((B, T, H)).astype(np.float32)
    rnn = MyLSTM(I, hidden_layer_sizes=[H], batch_first=True, w_rank=rw, u_ranks=ru, cell=MyVMLMFCellg2)
    load_into(rnn, P, prefix="rnncells.0.")
    x = torch.tensor(x_np, requires_grad=True)
    y, hcat = rnn(x)
    (y * torch.tensor(dy)).sum().backward()
    save(name, meta=np.array([O.V2, B, T, I, H, rw] + ru), seeds=np.array([3, 1234, 4321]),
         y_s=y.detach().numpy()[:, ::16], hT=hcat.detach().numpy(),
         dx=x.grad.numpy(), G=grads_of(rnn, P, prefix="rnncells.0."))


def case_config_a_novm(name, variant):
    """UCI-HAR shape with the two comparison cells (plain low-rank LSTM rank 16; group cell without vm [16,16])."""
    B, T, I, H, rw = 64, 128, 9, 180, 16
    ru = [16] if variant == O.V5 else [16, 16]
    P = O.make_params(variant, I, H, rw, ru, seed=3)
    x_np, _ = O.synthetic_batch(B, T, I, seed=1234)
    r = rng_of(4321)
    dy = r.standard_normal((B, T, H)).astype(np.float32)
    rnn = MyLSTM(I, hidden_layer_sizes=[H], batch_first=True, w_rank=rw, u_ranks=ru, cell=HAR_CELL[variant])
    load_into(rnn, P, prefix="rnncells.0.")
    x = torch.tensor(x_np, requires_grad=True)
    y, hcat = rnn(x)
    (y * torch.tensor(dy)).sum().backward()
    save(name, meta=np.array([variant, B, T, I, H, rw] + ru), seeds=np.array([3, 1234, 4321]),
         y_s=y.detach().numpy()[:, ::16], hT=hcat.detach().numpy(),
         dx=x.grad.numpy(), G=grads_of(rnn, P, prefix="rnncells.0."))


def case_config_c(name="cfgC_v1_opp2"):
    """BASELINE config C shape: Opportunity, 2-layer MyVMLMFCell H=256 r=24, B=128 T=24 I=77 (fp32 reference)."""
    B, T, I, H, rw, ru = 128, 24, 77, 256, 24, 24
    P0 = O.make_params(O.V1, I, H, rw, ru, seed=3)
    P1 = O.make_params(O.V1, H, H, rw, ru, seed=5)
    x_np, _ = O.synthetic_batch(B, T, I, seed=1234, classes=18)
    r = rng_of(4321)
    dy = r.standard_normal((B, T, H)).astype(np.float32)
    rnn = MyLSTM(I, hidden_layer_sizes=[H, H], batch_first=True, w_rank=rw, u_ranks=[ru], cell=MyVMLMFCell)
    load_into(rnn, P0, prefix="rnncells.0.")
    load_into(rnn, P1, prefix="rnncells.1.")
    x = torch.tensor(x_np, requires_grad=True)
    y, hcat = rnn(x)
    (y * torch.tensor(dy)).sum().backward()
    save(name, meta=np.array([O.V1, B, T, I, H, rw, ru]), seeds=np.array([3, 5, 1234, 4321]),
         y_s=y.detach().numpy()[:, ::6], hT=hcat.detach().numpy(), dx_s=x.grad.numpy()[::4],
         G0=grads_of(rnn, P0, prefix="rnncells.0."), G1=grads_of(rnn, P1, prefix="rnncells.1."))


def case_unequal_sizes(name="seq_v1_h128_h256"):
    """MyLSTM with hidden_layer_sizes=[128, 256] (vmlmf.py:283-292: layer 1 reads 128 inputs), Opportunity's 77 features, fp32 reference."""
    B, T, I, Hs, rw, ru = 32, 12, 77, [128, 256], 24, 24
    P0 = O.make_params(O.V1, I, Hs[0], rw, ru, seed=13)
    P1 = O.make_params(O.V1, Hs[0], Hs[1], rw, ru, seed=15)
    x_np, _ = O.synthetic_batch(B, T, I, seed=2345, classes=18)
    r = rng_of(5432)
    dy = r.standard_normal((B, T, Hs[1])).astype(np.float32)
    rnn = MyLSTM(I, hidden_layer_sizes=Hs, batch_first=True, w_rank=rw, u_ranks=[ru], cell=MyVMLMFCell)
    load_into(rnn, P0, prefix="rnncells.0.")
    load_into(rnn, P1, prefix="rnncells.1.")
    x = torch.tensor(x_np, requires_grad=True)
    y, hcat = rnn(x)
    (y * torch.tensor(dy)).sum().backward()
    save(name, meta=np.array([O.V1, B, T, I, Hs[0], Hs[1], rw, ru]), seeds=np.array([13, 15, 2345, 5432]),
         y_s=y.detach().numpy()[:, ::4], hT=hcat.detach().numpy(), dx_s=x.grad.numpy()[::4],
         G0=grads_of(rnn, P0, prefix="rnncells.0."), G1=grads_of(rnn, P1, prefix="rnncells.1."))


def case_net_adam(name="cfgA_net_adam3"):
    """Net (MyLSTM + Linear(H,18)) at config A, 3 steps of the train.py:58-65 loop (Adam lr=.002, CE)."""
    B, T, I, H, rw, ru = 64, 128, 9, 180, 16, 16
    P = O.make_params(O.V1, I, H, rw, ru, seed=3)
    x_np, tgt = O.synthetic_batch(B, T, I, seed=1234)
    torch.manual_seed(0)
    net = Net(I, layer_sizes=[H], w_rank=rw, u_rank=[ru], model=MyLSTM, cell=MyVMLMFCell)
    load_into(net, P, prefix="rnn.rnncells.0.")
    lin_w = net.lin.weight.detach().numpy().copy()
    lin_b = net.lin.bias.detach().numpy().copy()
    opt = torch.optim.Adam(net.parameters(), lr=0.002)
    losses, logits = [], []
    for _ in range(3):
        net.zero_grad()
        out = net(torch.tensor(x_np))
        loss = torch.nn.functional.cross_entropy(out, torch.tensor(tgt).long())
        loss.backward()
        opt.step()
        losses.append(loss.item())
        logits.append(out.detach().numpy().copy())
    final = {k: v.detach().numpy().copy() for k, v in net.named_parameters() if k.startswith("rnn.")}
    save(name, meta=np.array([O.V1, B, T, I, H, rw, ru]), seeds=np.array([3, 1234]), lin_w=lin_w, lin_b=lin_b,
         losses=np.array(losses), logits=np.stack(logits), final=final,
         unused_cell_has_grad=np.array([int(any(p.grad is not None for p in net.cell.parameters()))]))


def case_state_dict_names(name="state_dict_names"):
    """Parameter names+shapes the reference exposes (checkpoint compatibility, save_load.py:47,64-65)."""
    out = {}
    nets = {
        "net_v1": Net(77, layer_sizes=[180], w_rank=8, u_rank=[6], model=MyLSTM, cell=MyVMLMFCell),
        "net_v2": Net(77, layer_sizes=[180], w_rank=8, u_rank=[2, 4], model=MyLSTM, cell=MyVMLMFCellg2),
        "lm_v3": MyVMLSTM(16, 16, w_rank=4, u_ranks=4),
        "lm_v4": MyVMLSTMGroup(16, 16, w_rank=4, u_ranks=[2, 3]),
        # a list u_rank raises in the reference here (vmlmf.py:177 gets the list through Net.cell, vmlmf.py:349-350)
        "net_v5": Net(77, layer_sizes=[180], w_rank=8, u_rank=6, model=MyLSTM, cell=MyLSTMCell),
        "net_v6": Net(77, layer_sizes=[180], w_rank=8, u_rank=[2, 4], model=MyLSTM, cell=MyVMLMFgCellg2),
    }
    for tag, m in nets.items():
        for k, v in m.state_dict().items():
            out[f"{tag}/{k}"] = np.array(v.shape, dtype=np.int64)
    x = torch.randn(81, 24, 77)
    out["net_v1_out_shape"] = np.array(nets["net_v1"](x).shape)
    out["net_v2_out_shape"] = np.array(nets["net_v2"](x).shape)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name:28s} ok")


def case_flop_counts(name="flop_counts"):
    """Numbers and printed text of the harness reports (V/src/utils/compression_cal.py), as main.py:143-157 calls them."""
    import contextlib
    import io
    import types
    from utils import compression_cal as CC
    rows, texts = [], []
    grid = [("vmlmf", MyVMLMFCell, 9, [180], 16, [16], 64, 128), ("vmlmf", MyVMLMFCell, 77, [256, 256], 24, [24], 128, 24),
            ("mylstm", MyLSTMCell, 9, [180], None, None, 64, 128), ("vmlmf", MyVMLMFCell, 77, [180], 8, [6], 81, 24)]
    for tag, cell, I, layers, rw, ru, B, T in grid:
        net = Net(I, layer_sizes=layers, w_rank=rw, u_rank=ru, model=MyLSTM, cell=cell)
        args = types.SimpleNamespace(batch_size=B, model=tag)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            CC.print_model_parm_nums(net)
            CC.print_model_parm_flops(net, T, args, modeltype="mylstm" if tag == "mylstm" else "vmmodel")
        rows.append([CC.count_lstm(net, T, B, tag), CC.count_linear(net, 18), sum(p.numel() for p in net.parameters())])
        texts.append(buf.getvalue())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), counts=np.array(rows, dtype=np.int64),
                        text=np.array(texts))
    print(f"{name:28s} ok")


def case_lm_model(name="lm_model_v3", lstm_type="vmlmf"):
    """The whole LM network of lm_test.py (vmlmf_lm.py:366-440, lstm_type "vmlmf" - or "custom", the dense baseline layer -
    dropout 0) on two consecutive minibatches of the training loop (lm_test.py:196-209): scores, nll_loss, gradients,
    clip + SGD step, carried states."""
    from models.vmlmf_lm import Model
    from train_test.lm_test import nll_loss
    V, H, L, B, T = 60, 16, 2, 4, 5
    torch.manual_seed(7)
    model = Model(V, H, L, 0.0, 0.1, w_rank=4, u_ranks=[5], lstm_type=lstm_type)
    init = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    r = rng_of(61)
    out = dict(meta=np.array([V, H, L, B, T, 4, 5]), init=init)
    states = model.state_init(B)
    lr, max_norm = 1.0, 0.25
    for i in range(2):
        x = torch.tensor(r.integers(0, V, size=(T, B)))
        y = torch.tensor(r.integers(0, V, size=(T, B)))
        model.zero_grad()
        states = model.detach(states)
        scores, states = model(x, states)
        loss = nll_loss(scores, y)
        loss.backward()
        out[f"x{i}"], out[f"y{i}"] = x.numpy(), y.numpy()
        out[f"scores{i}"], out[f"loss{i}"] = scores.detach().numpy(), np.array([loss.item()])
        out[f"G{i}"] = {k: p.grad.numpy().copy() for k, p in model.named_parameters()}
        with torch.no_grad():
            norm = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
            for p in model.parameters():
                p -= lr * p.grad
        out[f"norm{i}"] = np.array([float(norm)])
    out["final"] = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    out["hT"] = np.stack([s[0].detach().numpy() for s in states])
    save(name, **out)


def case_nll(name="nll_v10000"):
    """nll_loss (lm_test.py:140-153) at the PTB vocabulary width: R = T*B = 70 rows of 10 000 scores (seeded; the
    fixture keeps the loss and a strided sample of the autograd gradient)."""
    from train_test.lm_test import nll_loss
    T, B, V = 7, 10, 10000
    r = rng_of(71)
    scores = torch.tensor((2.0 * r.standard_normal((T * B, V))).astype(np.float32), requires_grad=True)
    y = torch.tensor(r.integers(0, V, size=(T, B)))
    loss = nll_loss(scores, y)
    (3.0 * loss).backward()
    g = scores.grad.numpy()
    rows = np.arange(T * B)
    save(name, meta=np.array([T, B, V, 71]), loss=np.array([loss.item()]), y=y.numpy(), g_s=g[:, ::97],
         g_target=g[rows, y.numpy().reshape(-1)], upstream=np.array([3.0]))


CASES = {
    "cell_v1": lambda n: case_bare_cell(n, O.V1, 4, 5, 8, 3, 3, 11),
    "cell_v1_b1": lambda n: case_bare_cell(n, O.V1, 1, 5, 8, 3, 2, 12),
    "cell_v1_ieqh": lambda n: case_bare_cell(n, O.V1, 3, 8, 8, 4, 5, 13),
    "cell_v2": lambda n: case_bare_cell(n, O.V2, 4, 5, 12, 3, [2, 3], 14),
    "cell_v3": lambda n: case_bare_cell(n, O.V3, 4, 8, 8, 3, 3, 15),
    "cell_v4": lambda n: case_bare_cell(n, O.V4, 40, 12, 12, 3, [2, 3], 16),
    "cell_v5": lambda n: case_bare_cell(n, O.V5, 4, 5, 8, 3, 3, 17),
    "cell_v5_iwide": lambda n: case_bare_cell(n, O.V5, 3, 12, 8, 3, 4, 18),   # I > H: fine without vm_x
    "cell_v6": lambda n: case_bare_cell(n, O.V6, 4, 5, 12, 3, [2, 3], 19),
    "seq_v1": lambda n: case_har_seq(n, O.V1, 4, 6, 5, 8, 3, [3], 21),
    "seq_v1_wide": lambda n: case_har_seq(n, O.V1, 5, 7, 20, 70, 5, [7], 22),       # H not a multiple of 64, odd ranks
    "seq_v2": lambda n: case_har_seq(n, O.V2, 4, 6, 5, 12, 3, [2, 3], 23),
    "seq_v2_demo": lambda n: case_har_seq(n, O.V2, 6, 8, 77, 180, 8, [2, 4], 24),   # demo.sh:10 shapes, short T
    "seq_v1_demo": lambda n: case_har_seq(n, O.V1, 6, 8, 77, 180, 8, [6], 25),      # demo.sh:7 shapes, short T
    "seq_v5": lambda n: case_har_seq(n, O.V5, 4, 6, 5, 8, 3, [3], 26),
    "seq_v5_wide": lambda n: case_har_seq(n, O.V5, 5, 7, 20, 70, 5, [7], 27),
    "seq_v6": lambda n: case_har_seq(n, O.V6, 4, 6, 5, 12, 3, [2, 3], 28),
    "seq_v6_demo": lambda n: case_har_seq(n, O.V6, 6, 8, 77, 180, 8, [2, 4], 29),
    "seq_v3": lambda n: case_lm_seq(n, O.V3, 4, 6, 8, 3, 3, 31),
    "seq_v4": lambda n: case_lm_seq(n, O.V4, 40, 5, 12, 3, [2, 3], 32),
    "lm_v3_carry": lambda n: case_lm_carry(n, 41),
    "cfgA_v1_uci": lambda n: case_config_a(n),
    "cfgA_v2_uci": lambda n: case_config_a_group(n),
    "cfgA_v5_uci": lambda n: case_config_a_novm(n, O.V5),
    "cfgA_v6_uci": lambda n: case_config_a_novm(n, O.V6),
    "cfgC_v1_opp2": lambda n: case_config_c(n),
    "seq_v1_h128_h256": lambda n: case_unequal_sizes(n),
    "cfgA_net_adam3": lambda n: case_net_adam(n),
    # BASELINE config E shape, one MyVMLSTMGroup layer at the only batch the reference executes (40)
    "cfgE_v4_b40": lambda n: case_lm_seq(n, O.V4, 40, 35, 650, 32, [32, 32], 51, scale=0.05, full=False, xscale=0.05),
    "cfgE_v3_b64": lambda n: case_lm_seq(n, O.V3, 64, 35, 650, 32, 32, 52, scale=0.05, full=False, xscale=0.05),
    "state_dict_names": lambda n: case_state_dict_names(n),
    "flop_counts": lambda n: case_flop_counts(n),
    "lm_model_v3": lambda n: case_lm_model(n),
    "lm_model_custom": lambda n: case_lm_model(n, lstm_type="custom"),
    "nll_v10000": lambda n: case_nll(n),
}


def main(argv):
    os.makedirs(OUT, exist_ok=True)
    for name in (argv or list(CASES)):
        CASES[name](name)


if __name__ == "__main__":
    main(sys.argv[1:])
