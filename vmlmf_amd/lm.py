"""Drop-in counterparts of the reference's language-model layers, backed by the HIP kernels.

  MyVMLSTM       V/src/models/vmlmf_lm.py:178-280   forward(x:(T,B,X), (h,c)) -> (y, (h,c))
  MyVMLSTMGroup  V/src/models/vmlmf_lm.py:53-174    (the reference only runs at batch 40: its scratch is
                                                     hard-coded, 112-113; this implementation has no limit)
Parameter names, shapes and registration order follow the reference (state_dict compatible).
"""
from __future__ import annotations

import torch
from torch import nn

from . import _lib
from .functional import vmlmf_sequence


class MyVMLSTM(nn.Module):
    variant = _lib.V3_LM

    def __init__(self, input_size, hidden_size, dropout=0, w_rank=None, u_ranks=None):
        super().__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.dropout = dropout
        self.w_rank = w_rank
        self.u_ranks = u_ranks
        # the reference allocates uninitialised storage and lets Model.reset_parameters fill it
        # (vmlmf_lm.py:407-410); zeros here: no RNG draw, so a seeded Model init stays in step
        self.u_x = nn.Parameter(torch.zeros(input_size, w_rank))
        self.u_h = nn.Parameter(torch.zeros(hidden_size, u_ranks))
        self.w_x = nn.Parameter(torch.zeros(4 * hidden_size, w_rank))
        self.w_h = nn.Parameter(torch.zeros(4 * hidden_size, u_ranks))
        self.b_x = nn.Parameter(torch.zeros(4 * hidden_size))
        self.b_h = nn.Parameter(torch.zeros(4 * hidden_size))
        self.dia_x = nn.Parameter(torch.zeros(1, input_size))
        self.dia_h = nn.Parameter(torch.zeros(1, hidden_size))
        self.cnt = 0

    def __repr__(self):
        return f"LSTM(input: {self.input_size}, hidden: {self.hidden_size})"

    def kernel_params(self):
        return (self.dia_x, self.dia_h, self.u_x, self.w_x, self.b_x, self.b_h, self.u_h, self.w_h)

    def _run(self, x, h, c):
        return vmlmf_sequence(self.variant, x, h, c, self.kernel_params(), self.w_rank, [self.u_ranks],
                              g=1, time_major=True)

    def lstm_step(self, x, h, c):
        """One timestep (vmlmf_lm.py:222-269): T = 1 of the sequence kernels."""
        _, hn, cn = self._run(x.unsqueeze(0), h, c)
        return hn, cn

    def forward(self, x, states):
        h, c = states
        y, hT, cT = self._run(x, h, c)
        return y, (hT, cT)


class MyVMLSTMGroup(nn.Module):
    variant = _lib.V4_LM_GROUP

    def __init__(self, input_size, hidden_size, dropout=0, w_rank=None, u_ranks=None, g=2):
        super().__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.dropout = dropout
        self.g = g
        self.w_rank = w_rank
        self.u_ranks = u_ranks
        self.u_x = nn.Parameter(torch.zeros(input_size, w_rank))
        self.w_x = nn.Parameter(torch.zeros(4 * hidden_size, w_rank))
        self.u_h = nn.ParameterList([nn.Parameter(torch.zeros(g, int(hidden_size / g), u_ranks[s]))
                                     for s in range(self.g)])
        self.v_h = nn.ParameterList([nn.Parameter(torch.zeros(g, u_ranks[s], 4 * int(hidden_size / g)))
                                     for s in range(self.g)])
        self.b_x = nn.Parameter(torch.zeros(4 * hidden_size))
        self.b_h = nn.Parameter(torch.zeros(4 * hidden_size))
        self.dia_x = nn.Parameter(torch.zeros(1, input_size))
        self.dia_h = nn.Parameter(torch.zeros(1, hidden_size))
        self.cnt = 0

    def __repr__(self):
        return f"LSTM(input: {self.input_size}, hidden: {self.hidden_size})"

    def kernel_params(self):
        out = [self.dia_x, self.dia_h, self.u_x, self.w_x, self.b_x, self.b_h]
        for s in range(self.g):
            out += [self.u_h[s], self.v_h[s]]
        return tuple(out)

    def _run(self, x, h, c):
        return vmlmf_sequence(self.variant, x, h, c, self.kernel_params(), self.w_rank, list(self.u_ranks),
                              g=self.g, time_major=True)

    def lstm_step(self, x, h, c):
        _, hn, cn = self._run(x.unsqueeze(0), h, c)
        return hn, cn

    def forward(self, x, states):
        h, c = states
        y, hT, cT = self._run(x, h, c)
        return y, (hT, cT)
