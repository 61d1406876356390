"""Parameter and FLOP reports of the reference's harness (V/src/utils/compression_cal.py:33-145), host logic only.

Same function names, arguments, printed text and returned numbers as the reference, so `main.py:143-157` runs on
these modules unchanged; the counts are the reference's own accounting (e.g. the "mylstm" type is always priced
with dense gate matrices, whatever its ranks), checked against numbers captured from the reference
(tests/golden/flop_counts.npz).
"""
from __future__ import annotations

_UNCOUNTED = ("vmlmf_group", "vmlmf_lm")   # compression_cal.py:49,127


def print_model_parm_nums(model):
    """compression_cal.py:33-39."""
    total = sum(p.numel() for p in model.parameters())
    print(f" + Number of params:{(total / 1e3):.2f}K")


def _count_lstm_cell(modeltype, input_size, hidden_size, w_rank=None, u_rank=None, bias=True):
    """FLOPs of one cell call for one sample (compression_cal.py:74-115)."""
    I, H = input_size, hidden_size
    if isinstance(u_rank, list):
        u_rank = u_rank[0]

    def dot(n):          # an n-term dot product: n multiplies, n - 1 adds
        return 2 * n - 1

    if modeltype == "mylstm":
        gate = dot(I) * H + dot(H) * H + H            # two dense products and their sum
    else:
        low_x = dot(I) * w_rank + dot(w_rank) * H     # (x U) V^T
        low_h = dot(H) * u_rank + dot(u_rank) * H
        strip_x = dot(w_rank) * I + H                 # rowsum(U * V) and its subtraction
        strip_h = dot(u_rank) * H + H
        gate = low_x + low_h + I + H + 3 * H + strip_x + strip_h   # + the two vector products and three sums
    if bias:
        gate += H
    return 4 * gate + 3 * H + H                       # c' = f*c + i*g (3H), h' = o*tanh(c') (H)


def count_lstm(model, seq_len, batch_size, modeltype):
    """compression_cal.py:117-136."""
    if modeltype in _UNCOUNTED:
        print("Not Implemented")
        return None
    rnn = model.rnn
    sizes = [rnn.input_size] + list(rnn.hidden_layer_sizes)
    total = sum(_count_lstm_cell(modeltype, sizes[i], sizes[i + 1], rnn.w_rank, rnn.u_ranks, bias=True)
                for i in range(len(sizes) - 1))
    return total * seq_len * batch_size


def count_linear(model, output_size):
    """compression_cal.py:138-145."""
    return model.rnn.hidden_layer_sizes[-1] * output_size * 2


def print_model_parm_flops(model, seq_len, args, modeltype="vmmodel"):
    """compression_cal.py:41-57.  `args` needs .batch_size and .model.  As in the reference, an `args.model` of
    "vmlmf_group" passes the first check (it tests the `modeltype` argument), prints "Not Implemented" from
    count_lstm and then fails with TypeError on `None + int`."""
    if modeltype in _UNCOUNTED:
        print("Not Implemented")
        return
    modeltype = args.model.lower() if modeltype != "mylstm" else "mylstm"
    total_ops = count_lstm(model, seq_len, args.batch_size, modeltype)
    total_ops += count_linear(model, 18)
    print(f"  + Number of FLOPs: {(total_ops / 1e6):.2f}M")
    print(total_ops)


def print_model_parm_names(model):
    """compression_cal.py:59-72."""
    for idx, m in enumerate(model.modules()):
        print(idx, '->', m)
    print("Model's state_dict:")
    for name, tensor in model.state_dict().items():
        print(name, "\t", tensor.size())
