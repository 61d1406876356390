"""Drop-in counterparts of the reference's language-model layers, backed by the HIP kernels.

  MyVMLSTM       V/src/models/vmlmf_lm.py:178-280   forward(x:(T,B,X), (h,c)) -> (y, (h,c))
  MyVMLSTMGroup  V/src/models/vmlmf_lm.py:53-174    (the reference only runs at batch 40: its scratch is
                                                     hard-coded, 112-113; this implementation has no limit)
  Embed, Linear, Model   V/src/models/vmlmf_lm.py:33-51, 345-364, 366-440: the rest of the LM network around those layers
                         (SURVEY section 8f rank 3).  Embedding lookup and the vocabulary projection (one library GEMM) are
                         stock ops; the loss that consumes the scores is vmlmf_amd.nll_loss (fused kernels).  The reference's
                         dense "custom" LSTM layer (283-339) is the uncompressed baseline, off the VMLMF path: class LSTM
                         below keeps it available in stock library ops (Model(lstm_type="custom")); dense_layer= overrides it.
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

    def _run(self, x, h, c, drop=None):
        return vmlmf_sequence(self.variant, x, h, c, self.kernel_params(), self.w_rank, [self.u_ranks],
                              g=1, time_major=True, dtype=getattr(self, "compute_dtype", "f32"), pack_cache=getattr(self, "_pack_cache", None),
                              drop=drop)

    def lstm_step(self, x, h, c):
        """One timestep (vmlmf_lm.py:222-269): T = 1 of the sequence kernels."""
        _, hn, cn = self._run(x.unsqueeze(0), h, c)
        return hn, cn

    def forward(self, x, states, drop=None):
        """drop = (p, snapshot, site): the returned y went through Model's dropout (vmlmf_lm.py:438-439) - inside this layer's own
        launches where the library covers it (functional.vmlmf_sequence)."""
        h, c = states
        y, hT, cT = self._run(x, h, c, drop)
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

    def _run(self, x, h, c, drop=None):
        return vmlmf_sequence(self.variant, x, h, c, self.kernel_params(), self.w_rank, list(self.u_ranks),
                              g=self.g, time_major=True, dtype=getattr(self, "compute_dtype", "f32"), pack_cache=getattr(self, "_pack_cache", None),
                              drop=drop)

    def lstm_step(self, x, h, c):
        _, hn, cn = self._run(x.unsqueeze(0), h, c)
        return hn, cn

    def forward(self, x, states, drop=None):
        """drop = (p, snapshot, site): the returned y went through Model's dropout (vmlmf_lm.py:438-439) - inside this layer's own
        launches where the library covers it (functional.vmlmf_sequence)."""
        h, c = states
        y, hT, cT = self._run(x, h, c, drop)
        return y, (hT, cT)


class LSTM(nn.Module):
    """The reference's dense "custom" layer (vmlmf_lm.py:283-339): the uncompressed baseline Model(lstm_type="custom")
    builds.  Not part of the VMLMF path, so stock library ops on whatever device the tensors live on: the input side of all
    T steps is one GEMM ahead of the time loop, the recurrence one addmm per step.  Same parameter names and shapes."""

    def __init__(self, input_size, hidden_size, dropout=0):
        super().__init__()
        self.input_size, self.hidden_size, self.dropout = input_size, hidden_size, dropout
        self.w_x = nn.Parameter(torch.zeros(4 * hidden_size, input_size))
        self.w_h = nn.Parameter(torch.zeros(4 * hidden_size, hidden_size))
        self.b_x = nn.Parameter(torch.zeros(4 * hidden_size))
        self.b_h = nn.Parameter(torch.zeros(4 * hidden_size))

    def __repr__(self):
        return f"LSTM(input: {self.input_size}, hidden: {self.hidden_size})"

    def forward(self, x, states):
        h, c = states
        T, B, _ = x.shape
        H = self.hidden_size
        gx = torch.addmm(self.b_x + self.b_h, x.reshape(T * B, -1), self.w_x.t()).view(T, B, 4 * H)
        w_ht = self.w_h.t()
        ys = []
        for t in range(T):
            i, f, o, n = torch.addmm(gx[t], h, w_ht).split(H, 1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(n)
            h = torch.sigmoid(o) * torch.tanh(c)
            ys.append(h)
        return torch.stack(ys), (h, c)


class Embed(nn.Module):
    """Embedding table indexed by token id (vmlmf_lm.py:33-51)."""

    def __init__(self, vocab_size, embed_size):
        super().__init__()
        self.vocab_size = vocab_size
        self.embed_size = embed_size
        self.w = nn.Parameter(torch.zeros(vocab_size, embed_size))

    def forward(self, x):
        from .functional import embedding
        return embedding(self.w, x)      # the gather is the stock op; the table's gradient is the package's kernel on HIP tensors

    def __repr__(self):
        return f"Embedding(vocab: {self.vocab_size}, embedding: {self.embed_size})"


class Linear(nn.Module):
    """Vocabulary projection (vmlmf_lm.py:345-364): (T, B, H) -> (T*B, V) scores, one library GEMM."""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.w = nn.Parameter(torch.zeros(hidden_size, input_size))
        self.b = nn.Parameter(torch.zeros(hidden_size))

    def forward(self, x):
        return torch.addmm(self.b, x.view(-1, x.size(2)), self.w.t())

    def __repr__(self):
        return f"FC(input: {self.input_size}, output: {self.hidden_size})"


class Model(nn.Module):
    """The language model of lm_test.py (vmlmf_lm.py:366-440): Embed -> dropout -> layer_num x (LSTM layer ->
    dropout) -> Linear.  Constructor logic is the reference's, quirks included: `u_ranks` is reduced to its last
    element unless lstm_type is the string "vm_group", while the group layers are only built for the string
    "vmgroup" -- so, as in the reference, "vmgroup" with a rank list fails in MyVMLSTMGroup's constructor and
    "vm_group" silently builds torch.nn.LSTM layers.  Build MyVMLSTMGroup layers directly for the group variant."""

    def __init__(self, vocab_size, hidden_size, layer_num, dropout, winit, w_rank=None, u_ranks=None,
                 lstm_type="pytorch", dense_layer=None):
        """dense_layer: overrides the class built for lstm_type="custom" (default: LSTM above, the reference's dense baseline
        layer in stock ops); any class with the signature (input_size, hidden_size) and forward(x, states)."""
        super().__init__()
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.layer_num = layer_num
        self.winit = winit
        self.lstm_type = lstm_type
        self.embed = Embed(vocab_size, hidden_size)
        if u_ranks is not None and lstm_type != "vm_group":
            u_ranks = u_ranks[-1]
        if lstm_type == "vmgroup":
            rnns = [MyVMLSTMGroup(hidden_size, hidden_size, w_rank=w_rank, u_ranks=u_ranks) for _ in range(layer_num)]
        elif lstm_type == "custom":
            rnns = [(dense_layer or LSTM)(hidden_size, hidden_size) for _ in range(layer_num)]
        elif lstm_type != "vmlmf":
            rnns = [nn.LSTM(hidden_size, hidden_size) for _ in range(layer_num)]
        else:
            rnns = [MyVMLSTM(hidden_size, hidden_size, w_rank=w_rank, u_ranks=u_ranks) for _ in range(layer_num)]
        self.rnns = nn.ModuleList(rnns)
        self.fc = Linear(hidden_size, vocab_size)
        self.dropout = nn.Dropout(p=dropout)
        self.reset_parameters()

    def reset_parameters(self):
        for param in self.parameters():
            nn.init.uniform_(param, -self.winit, self.winit)

    @classmethod
    def with_group_layers(cls, vocab_size, hidden_size, layer_num, dropout, winit, w_rank, u_ranks, g=2):
        """The network of BASELINE configs[4] - Model with MyVMLSTMGroup layers - which the reference's constructor cannot build
        (lstm_type "vmgroup" reduces the rank list to its last element and MyVMLSTMGroup then fails, vmlmf_lm.py:387-392; "vm_group",
        the CLI's spelling, builds nn.LSTM).  NOT part of the reference's interface: the constructor above keeps the reference's
        behaviour, this puts the group layers in by hand (as a maintainer has to) and initialises every parameter as
        Model.reset_parameters does.  u_ranks: one rank per group rotation, e.g. [32, 32]."""
        model = cls(vocab_size, hidden_size, layer_num, dropout, winit, w_rank=w_rank, u_ranks=[list(u_ranks)[-1]], lstm_type="vmlmf")
        model.rnns = nn.ModuleList([MyVMLSTMGroup(hidden_size, hidden_size, w_rank=w_rank, u_ranks=list(u_ranks), g=g)
                                    for _ in range(layer_num)])
        model.lstm_type = "vmgroup"
        model.reset_parameters()
        return model

    def state_init(self, batch_size):
        dev = next(self.parameters()).device
        flat = self.lstm_type in ["custom", "vmlmf", "vmgroup", "hmd"]
        shape = (lambda layer: (batch_size, layer.hidden_size)) if flat else \
            (lambda layer: (1, batch_size, layer.hidden_size))
        return [(torch.zeros(*shape(layer), device=dev), torch.zeros(*shape(layer), device=dev)) for layer in self.rnns]

    def detach(self, states):
        return [(h.detach(), c.detach()) for (h, c) in states]

    def _stack(self, x, states):
        """The layer loop of forward() (vmlmf_lm.py:437-439) as ONE wavefront launch per direction with the carried states as
        initial states (functional.vmlmf_stack), when that gives the same values: no dropout between the layers (p = 0 or
        eval mode), MyVMLSTM layers of one configuration, and a stack the wavefront kernels cover (hidden sizes up to 256).
        None otherwise: the caller loops over the layers."""
        if (self.training and self.dropout.p > 0) or not x.is_cuda or len(self.rnns) < 2:
            return None
        if not all(type(r) is MyVMLSTM for r in self.rnns):
            return None
        r0 = self.rnns[0]
        if any((r.input_size, r.hidden_size, r.w_rank, r.u_ranks) != (r0.input_size, r0.hidden_size, r0.w_rank, r0.u_ranks)
               for r in self.rnns):
            return None
        if any(getattr(r, "compute_dtype", "f32") != "f32" for r in self.rnns):
            return None
        from .functional import vmlmf_stack
        h0 = torch.stack([st[0] for st in states])
        c0 = torch.stack([st[1] for st in states])
        ur = r0.u_ranks if isinstance(r0.u_ranks, (list, tuple)) else [r0.u_ranks]
        out = vmlmf_stack(variant=r0.variant, x=x, layer_params=[r.kernel_params() for r in self.rnns], w_rank=r0.w_rank,
                          u_ranks=list(ur), g=1, time_major=True, h0=h0, c0=c0)
        if out is None:
            return None
        y, hs, cs = out
        return y, [(hs[i], cs[i]) for i in range(len(self.rnns))]

    def forward(self, x, states):
        x, states = self.features(x, states)
        scores = self.fc(x)
        return scores, states

    def loss(self, x, y, states):
        """nll_loss(self(x, states)[0], y) of the training loop (lm_test.py:200-202) without ever handing out the scores: the
        projection's output is overwritten by its own gradient inside the loss (functional.lm_head_loss), which is what the two
        backward GEMMs read.  Returns (loss, states); same values as the two-call form."""
        from .functional import lm_head_loss
        h, states = self.features(x, states)
        return lm_head_loss(h, self.fc.w, self.fc.b, y), states

    def features(self, x, states):
        """Everything of forward() in front of the vocabulary projection (vmlmf_lm.py:434-439): (T, B, H) activations, states.
        Training with p > 0 on a HIP device: the three dropouts run without mask tensors (functional.dropout_*: Philox factors
        regenerated in the backward) - the embedding's inside its gather, a VMLMF layer's inside the layer's launches; with
        self.stock_dropout = True they are nn.Dropout's launches as in the reference."""
        p = self.dropout.p
        if self.training and p > 0 and x.is_cuda and not getattr(self, "stock_dropout", False):
            from .functional import dropout, dropout_advance, embedding_dropout
            snap = dropout_advance(self.dropout_state())
            x = embedding_dropout(self.embed.w, x, p, snap, 0)
            for i, rnn in enumerate(self.rnns):
                if isinstance(rnn, (MyVMLSTM, MyVMLSTMGroup)):
                    x, states[i] = rnn(x, states[i], drop=(p, snap, i + 1))
                else:
                    x, states[i] = rnn(x, states[i])
                    x = dropout(x, p, snap, i + 1)
            return x, states
        x = self.embed(x)
        x = self.dropout(x)
        stacked = self._stack(x, states)
        if stacked is not None:
            x, new_states = stacked
            for i, st in enumerate(new_states):
                states[i] = st
            x = self.dropout(x)
        else:
            for i, rnn in enumerate(self.rnns):
                x, states[i] = rnn(x, states[i])
                x = self.dropout(x)
        return x, states

    def dropout_state(self, seed=None):
        """{seed, offset} of this model's dropout generator on its device (created on first use; seed=None draws it from torch's CPU
        generator).  Create it BEFORE capturing a training step into a hipGraph (any eager warm-up step does)."""
        dev = self.embed.w.device
        st = getattr(self, "_drop_state", None)
        if st is None or st.device != dev or seed is not None:
            from .functional import dropout_state
            st = self._drop_state = dropout_state(dev, seed)
        return st
