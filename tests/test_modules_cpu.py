"""CPU: the nn.Module mirrors expose the reference's constructor signatures, attributes, parameter names and
shapes (state_dict compatibility, save_load.py:47,64-65), and refuse to run without a HIP device."""
import numpy as np
import os

import pytest
import torch

import vmlmf_amd
from vmlmf_amd import (MyLSTM, MyLSTMCell, MyVMLMFCell, MyVMLMFCellg2, MyVMLMFgCellg2, MyVMLSTM, MyVMLSTMGroup,
                       Net)
from conftest import load_golden


def test_state_dict_names_and_shapes_match_reference():
    ref = load_golden("state_dict_names")
    nets = {
        "net_v1": Net(77, layer_sizes=[180], w_rank=8, u_rank=[6], model=MyLSTM, cell=MyVMLMFCell),
        "net_v2": Net(77, layer_sizes=[180], w_rank=8, u_rank=[2, 4], model=MyLSTM, cell=MyVMLMFCellg2),
        "lm_v3": MyVMLSTM(16, 16, w_rank=4, u_ranks=4),
        "lm_v4": MyVMLSTMGroup(16, 16, w_rank=4, u_ranks=[2, 3]),
        "net_v5": Net(77, layer_sizes=[180], w_rank=8, u_rank=6, model=MyLSTM, cell=MyLSTMCell),
        "net_v6": Net(77, layer_sizes=[180], w_rank=8, u_rank=[2, 4], model=MyLSTM, cell=MyVMLMFgCellg2),
    }
    for tag, m in nets.items():
        mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        theirs = {k: tuple(int(x) for x in v) for k, v in ref[tag].items()}
        assert list(mine) == list(theirs), tag          # same names, same registration order
        assert mine == theirs, tag


def test_reference_unit_test_shape_assertions():
    """The six shape checks of V/src/unittest/unit_test.py:63-93 on our modules."""
    c = Net(77, layer_sizes=[180], w_rank=8, u_rank=[6], model=MyLSTM, cell=MyVMLMFCell)
    g = Net(77, layer_sizes=[180], w_rank=8, u_rank=[2, 4], model=MyLSTM, cell=MyVMLMFCellg2)
    assert c.cell.dia_x.shape == (1, 77) and c.cell.dia_h.shape == (1, 180)
    assert c.cell.u_x.shape == (77, 8) and c.cell.u_h.shape == (180, 6)
    assert c.cell.v_x.shape == (720, 8) and c.cell.v_h.shape == (720, 6)
    assert g.cell.layers['dia_x'].shape == (1, 77) and g.cell.layers['dia_h'].shape == (1, 180)
    assert g.cell.layers['u_h_0'].shape == (2, 90, 2) and g.cell.layers['u_h_1'].shape == (2, 90, 4)
    assert g.cell.layers['v_h_0'].shape == (2, 2, 360) and g.cell.layers['v_h_1'].shape == (2, 4, 360)


def test_seeded_construction_draws_the_same_values_as_the_reference():
    """Parameter creation order equals the reference's, so torch.manual_seed reproduces its init: the golden
    Net fixture stores lin.weight drawn after the cells under seed 0."""
    d = load_golden("cfgA_net_adam3")
    torch.manual_seed(0)
    net = Net(9, layer_sizes=[180], w_rank=16, u_rank=[16], model=MyLSTM, cell=MyVMLMFCell)
    assert np.array_equal(net.lin.weight.detach().numpy(), d["lin_w"])


def test_attributes_other_reference_code_reads():
    net = Net(9, layer_sizes=[180], w_rank=16, u_rank=[16], model=MyLSTM, cell=MyVMLMFCell)
    # compression_cal.py:128-132,144
    assert (net.rnn.input_size, net.rnn.hidden_layer_sizes, net.rnn.w_rank, net.rnn.u_ranks) == (9, [180], 16, 16)
    assert MyVMLSTM(8, 8, w_rank=2, u_ranks=2).hidden_size == 8     # vmlmf_lm.py:418


def test_no_cpu_fallback():
    cell = MyVMLMFCell(5, 8, w_rank=3, u_ranks=3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cell(torch.randn(2, 5), (torch.zeros(2, 8), torch.zeros(2, 8)))
    rnn = MyLSTM(5, hidden_layer_sizes=[8], w_rank=3, u_ranks=[3], cell=MyVMLMFCell)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rnn(torch.randn(2, 4, 5))
    lm = MyVMLSTMGroup(8, 8, w_rank=2, u_ranks=[2, 2])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lm(torch.randn(3, 2, 8), (torch.zeros(2, 8), torch.zeros(2, 8)))


def test_product_package_never_imports_the_oracle():
    import os
    root = os.path.dirname(vmlmf_amd.__file__)
    for fn in os.listdir(root):
        if fn.endswith(".py"):
            src = open(os.path.join(root, fn)).read()
            assert "vmlmf_oracle" not in src and "oracle" not in src.replace("# oracle", ""), fn


def test_baseline_cell_still_runs_inside_mylstm_on_cpu():
    """MyLSTMCell keeps the reference's Python loop of stock ops wherever it is not the HIP path: vanilla mode
    anywhere, low-rank mode on CPU tensors (on a HIP device low-rank mode runs the sequence kernels)."""
    rnn = MyLSTM(5, hidden_layer_sizes=[8, 8], w_rank=3, u_ranks=[3], cell=MyLSTMCell)
    y, hc = rnn(torch.randn(2, 4, 5))
    assert y.shape == (2, 4, 8) and hc.shape == (2, 16)
    vanilla = MyLSTM(5, hidden_layer_sizes=[8], cell=MyLSTMCell)
    assert vanilla(torch.randn(2, 4, 5))[0].shape == (2, 4, 8)
    with pytest.raises(RuntimeError, match="vanilla cell is not on the HIP path"):
        vanilla.rnncells[0].sequence(torch.randn(2, 4, 5))


def test_low_rank_baseline_cell_rejects_a_rank_list_like_the_reference():
    """vmlmf.py:177: torch.randn([hidden_size, u_ranks]) with the raw list argument raises TypeError; through
    MyLSTM a one-element list is unwrapped first (vmlmf.py:269) and works."""
    with pytest.raises(TypeError):
        MyLSTMCell(5, 8, w_rank=3, u_ranks=[3])
    MyLSTMCell(5, 8, w_rank=3, u_ranks=3)


def test_ablation_group_cell_has_no_cpu_path():
    rnn = MyLSTM(5, hidden_layer_sizes=[8], w_rank=3, u_ranks=[2, 2], cell=MyVMLMFgCellg2)
    assert not any("dia" in k for k in rnn.state_dict())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rnn(torch.randn(2, 4, 5))


def test_widened_rows_have_no_cpu_path_either():
    """SURVEY section 8f rows (classifier head, loss, optimizers): stock ops where they are not the product
    (head / loss dispatch), a loud error where they are (optimizers never touch the CPU)."""
    import pytest
    import vmlmf_amd
    z = torch.randn(4, 5, requires_grad=True)
    t = torch.tensor([0, 1, 2, 3])
    assert torch.allclose(vmlmf_amd.cross_entropy(z, t), torch.nn.functional.cross_entropy(z, t))
    w, b = torch.randn(3, 5), torch.randn(3)
    assert torch.allclose(vmlmf_amd.head_linear(z, w, b), torch.nn.functional.linear(z, w, b))
    p = torch.nn.Parameter(torch.randn(3))
    p.grad = torch.randn(3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        vmlmf_amd.optim.Adam([p]).step()
    with pytest.raises(RuntimeError, match="no CPU path"):
        vmlmf_amd.optim.clip_sgd_step([p], lr=0.1, max_norm=1.0)


def test_harness_reports_match_reference_numbers_and_text(capsys):
    """compression_cal.py:33-145 counterparts: parameter count, FLOP accounting, printed lines (main.py:143-157)."""
    import types
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import compression_cal as CC   # out of the product package (out of scope); kept as a tool
    ref = load_golden("flop_counts")
    grid = [("vmlmf", MyVMLMFCell, 9, [180], 16, [16], 64, 128), ("vmlmf", MyVMLMFCell, 77, [256, 256], 24, [24], 128, 24),
            ("mylstm", MyLSTMCell, 9, [180], None, None, 64, 128), ("vmlmf", MyVMLMFCell, 77, [180], 8, [6], 81, 24)]
    for row, (tag, cell, I, layers, rw, ru, B, T) in enumerate(grid):
        net = Net(I, layer_sizes=layers, w_rank=rw, u_rank=ru, model=MyLSTM, cell=cell)
        args = types.SimpleNamespace(batch_size=B, model=tag)
        capsys.readouterr()
        CC.print_model_parm_nums(net)
        CC.print_model_parm_flops(net, T, args, modeltype="mylstm" if tag == "mylstm" else "vmmodel")
        assert capsys.readouterr().out == str(ref["text"][row])
        got = [CC.count_lstm(net, T, B, tag), CC.count_linear(net, 18), sum(p.numel() for p in net.parameters())]
        assert got == [int(v) for v in ref["counts"][row]]
    # the uncounted model types: message, no number (and the reference's TypeError when only args.model says so)
    net = Net(9, layer_sizes=[16], w_rank=4, u_rank=[2, 2], model=MyLSTM, cell=MyVMLMFCellg2)
    CC.print_model_parm_flops(net, 8, types.SimpleNamespace(batch_size=2, model="vmlmf_group"), modeltype="vmlmf_group")
    assert capsys.readouterr().out == "Not Implemented\n"
    with pytest.raises(TypeError):
        CC.print_model_parm_flops(net, 8, types.SimpleNamespace(batch_size=2, model="vmlmf_group"))


def test_lm_network_surface_matches_reference():
    """Model / Embed / Linear / LSTM of vmlmf_lm.py: names, shapes, registration order, constructor quirks."""
    from vmlmf_amd import Model, nll_loss
    d = load_golden("lm_model_v3")
    V, H, L, B, T, rw, ru = (int(v) for v in d["meta"])
    torch.manual_seed(7)
    m = Model(V, H, L, 0.0, 0.1, w_rank=rw, u_ranks=[ru], lstm_type="vmlmf")
    assert list(m.state_dict()) == list(d["init"])
    for k, v in m.state_dict().items():          # same creation order + reset_parameters => same seeded values
        assert np.array_equal(v.numpy(), d["init"][k]), k
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(T, B, dtype=torch.int64), m.state_init(B))
    # the reference's constructor quirks (vmlmf_lm.py:390-402)
    with pytest.raises(TypeError):
        Model(V, H, L, 0.0, 0.1, w_rank=rw, u_ranks=[2, 2], lstm_type="vmgroup")
    assert type(Model(V, H, 1, 0.0, 0.1, w_rank=rw, u_ranks=[2, 2], lstm_type="vm_group").rnns[0]) is torch.nn.LSTM
    # ... and the package's own way to the group network the reference cannot construct (not part of the reference's interface)
    from vmlmf_amd import MyVMLSTMGroup
    mg = Model.with_group_layers(V, H, 2, 0.5, 0.1, w_rank=rw, u_ranks=[2, 3])
    assert all(type(r) is MyVMLSTMGroup for r in mg.rnns) and mg.lstm_type == "vmgroup" and mg.dropout.p == 0.5
    assert mg.rnns[1].u_h[1].shape == (2, H // 2, 3) and mg.rnns[0].v_h[0].shape == (2, 2, 4 * (H // 2))
    assert all(float(p.abs().max()) <= 0.1 and float(p.abs().max()) > 0 for p in mg.parameters())
    assert [tuple(s.shape) for s in mg.state_init(3)[0]] == [(3, H), (3, H)]
    # dense_layer= overrides the class built for lstm_type="custom" (default: vmlmf_amd.LSTM, tested against the reference below)

    class Dense(torch.nn.Module):          # stand-in with the reference's layer interface
        def __init__(self, i, h):
            super().__init__()
            self.hidden_size = h
            self.w = torch.nn.Parameter(torch.zeros(i, h))

        def forward(self, x, states):
            y = torch.tanh(x @ self.w)
            return y, (y[-1], states[1])

    c = Model(V, H, L, 0.0, 0.1, lstm_type="custom", dense_layer=Dense)
    scores, st = c(torch.tensor(d["x0"]), c.state_init(B))
    assert scores.shape == (T * B, V) and st[0][0].shape == (B, H)
    z = torch.randn(T * B, V)
    y = torch.tensor(d["y0"])
    want = -torch.log_softmax(z, 1)[torch.arange(T * B), y.reshape(-1)].mean() * B
    assert torch.allclose(nll_loss(z, y), want, rtol=1e-5)


def test_dense_baseline_network_matches_the_reference():
    """Model(lstm_type="custom") (lm_test.py:52 offers it) builds the dense baseline layer of vmlmf_lm.py:283-339 - stock ops,
    off the VMLMF path, runs wherever the tensors are: two minibatches of the LM loop against the reference's own run."""
    from vmlmf_amd import LSTM, Model, nll_loss
    d = load_golden("lm_model_custom")
    V, H, L, B, T, _, _ = (int(v) for v in d["meta"])
    torch.manual_seed(7)
    m = Model(V, H, L, 0.0, 0.1, lstm_type="custom")
    assert all(type(r) is LSTM for r in m.rnns) and list(m.state_dict()) == list(d["init"])
    for k, v in m.state_dict().items():
        assert np.array_equal(v.numpy(), d["init"][k]), k
    states = m.state_init(B)
    for i in range(2):
        m.zero_grad()
        states = m.detach(states)
        scores, states = m(torch.tensor(d[f"x{i}"]), states)
        loss = nll_loss(scores, torch.tensor(d[f"y{i}"]))
        loss.backward()
        assert np.allclose(scores.detach().numpy(), d[f"scores{i}"], atol=2e-6, rtol=1e-5)
        assert abs(loss.item() - float(d[f"loss{i}"][0])) < 1e-5 * abs(loss.item())
        for k, p in m.named_parameters():
            g = d[f"G{i}"][k]
            assert np.abs(p.grad.numpy() - g).max() <= 1e-5 * max(np.abs(g).max(), 1e-6), k
        with torch.no_grad():
            norm = torch.nn.utils.clip_grad_norm_(m.parameters(), 0.25)
            for p in m.parameters():
                p -= 1.0 * p.grad
        assert abs(float(norm) - float(d[f"norm{i}"][0])) < 1e-5 * float(norm)
    assert np.allclose(torch.stack([s[0] for s in states]).detach().numpy(), d["hT"], atol=2e-6, rtol=1e-4)
