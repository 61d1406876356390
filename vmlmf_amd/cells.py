"""Drop-in counterparts of the reference's HAR modules, backed by the HIP kernels.

Same constructor signatures, forward signatures, attribute names and parameter names/shapes as
  MyVMLMFCell    V/src/models/vmlmf.py:38-125
  MyVMLMFCellg2  V/src/models/vmlmf_group.py:37-155
  MyVMLMFgCellg2 V/src/models/vmlmf_group.py:158-251   (ablation: the group cell without vm)
  MyLSTMCell     V/src/models/vmlmf.py:127-238   (baseline cell; its low-rank mode runs the same kernels,
                                                  its vanilla mode is stock GEMMs and not the hot path)
  MyLSTM         V/src/models/vmlmf.py:241-316
  Net            V/src/models/vmlmf.py:319-355
so a reference checkpoint loads with load_state_dict and train.py / test.py run unchanged.
What differs is how forward is evaluated: a VMLMF layer is ONE sequence-level kernel pipeline
(vmlmf_amd.functional.vmlmf_sequence) instead of T cell calls of ~75 ATen ops each.
"""
from __future__ import annotations

import torch
from torch import nn

from . import _lib
from .functional import head_linear, stack_mode, vmlmf_sequence, vmlmf_stack

TIME_STEPS = 128
RECURRENT_MAX = pow(2, 1 / TIME_STEPS)
RECURRENT_MIN = pow(1 / 2, 1 / TIME_STEPS)


def _delist(u_ranks):
    return u_ranks[-1] if isinstance(u_ranks, list) and len(u_ranks) < 2 else u_ranks


class MyVMLMFCell(nn.Module):
    """VMLMF LSTM cell: diag(d) + (U V^T with its diagonal removed), shared d across the four gates."""

    variant = _lib.V1_CELL

    def __init__(self, input_size, hidden_size, w_rank=None, u_ranks=None):
        super().__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.w_rank = w_rank
        self.u_ranks = _delist(u_ranks)
        r = self.u_ranks
        # creation order == the reference's, so a seeded construction draws identical values
        self.u_x = nn.Parameter(0.1 * torch.randn([input_size, w_rank]))
        self.u_h = nn.Parameter(0.1 * torch.randn([hidden_size, r]))
        self.v_x = nn.Parameter(0.1 * torch.randn([4 * hidden_size, w_rank]))
        self.v_h = nn.Parameter(0.1 * torch.randn([4 * hidden_size, r]))
        self.b_x = nn.Parameter(0.1 * torch.randn([4 * hidden_size]))
        self.b_h = nn.Parameter(0.1 * torch.randn([4 * hidden_size]))
        self.dia_x = nn.Parameter(0.1 * torch.randn([1, input_size]))
        self.dia_h = nn.Parameter(0.1 * torch.randn([1, hidden_size]))
        self.cnt = 0

    def __repr__(self):
        return (f"LSTM_FINAL(input: {self.input_size}, hidden: {self.hidden_size}, "
                f"w_rank: {self.w_rank}, u_ranks: {self.u_ranks})")

    # kernel-facing view of the parameters (order fixed by functional.py)
    def kernel_params(self):
        return (self.dia_x, self.dia_h, self.u_x, self.v_x, self.b_x, self.b_h, self.u_h, self.v_h)

    def kernel_cfg(self):
        return dict(variant=self.variant, w_rank=self.w_rank, u_ranks=[self.u_ranks], g=1,
                    dtype=getattr(self, "compute_dtype", "f32"), pack_cache=getattr(self, "_pack_cache", None))

    def sequence(self, x, h0=None, c0=None, time_major=False, head=None, target=None, ignore_index=-100):
        """Whole-sequence evaluation: (y, hT, cT) (+ logits with a classifier `head`, + loss with a `target` for it)."""
        return vmlmf_sequence(x=x, h0=h0, c0=c0, params=self.kernel_params(), time_major=time_major, head=head,
                              target=target, ignore_index=ignore_index, **self.kernel_cfg())

    def forward(self, x, hidden_states):
        """One step: x (B, I), (h, c) each (B, H) -> (h_next, c_next).  T = 1 of the same kernels."""
        (h, c) = hidden_states
        if x.dim() == 1:
            x = x.unsqueeze(0)
        _, h_next, c_next = self.sequence(x.unsqueeze(1), h, c)
        return h_next, c_next


class MyVMLMFCellg2(nn.Module):
    """Group-low-rank VMLMF cell (g groups, shift s couples group j to group (j+s) mod g)."""

    variant = _lib.V2_GROUP_CELL

    def __init__(self, input_size, hidden_size, w_rank=None, u_ranks=None, g=2,
                 recurrent_init=None, hidden_init=None):
        super().__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.recurrent_init = recurrent_init
        self.hidden_init = hidden_init
        self.w_rank = w_rank
        self.u_ranks = u_ranks
        self.g = g
        self.layers = nn.ParameterDict()
        self.layers['dia_x'] = nn.Parameter(0.1 * torch.randn([1, input_size]))
        self.layers['dia_h'] = nn.Parameter(0.1 * torch.randn([1, hidden_size]))
        self.layers['u_x'] = nn.Parameter(0.1 * torch.randn([input_size, w_rank]))
        self.layers['v_x'] = nn.Parameter(0.1 * torch.randn([4 * hidden_size, w_rank]))
        for s in range(self.g):
            self.layers[f'u_h_{s}'] = nn.Parameter(0.1 * torch.randn([g, int(hidden_size / g), u_ranks[s]]))
            self.layers[f'v_h_{s}'] = nn.Parameter(0.1 * torch.randn([g, u_ranks[s], 4 * int(hidden_size / g)]))
        for vec in ['x', 'h']:
            self.layers[f'bias_{vec}'] = nn.Parameter(torch.ones([1, 4 * hidden_size]))

    def __repr__(self):
        return (f"LSTM VM Group (input:{self.input_size}, hidden:{self.hidden_size}, "
                f"w_rank:{self.w_rank}, u_ranks:{self.u_ranks}")

    def kernel_params(self):
        L = self.layers
        out = [L['dia_x'], L['dia_h'], L['u_x'], L['v_x'], L['bias_x'], L['bias_h']]
        for s in range(self.g):
            out += [L[f'u_h_{s}'], L[f'v_h_{s}']]
        return tuple(out)

    def kernel_cfg(self):
        return dict(variant=self.variant, w_rank=self.w_rank, u_ranks=list(self.u_ranks), g=self.g,
                    dtype=getattr(self, "compute_dtype", "f32"), pack_cache=getattr(self, "_pack_cache", None))

    def sequence(self, x, h0=None, c0=None, time_major=False, head=None, target=None, ignore_index=-100):
        return vmlmf_sequence(x=x, h0=h0, c0=c0, params=self.kernel_params(), time_major=time_major, head=head,
                              target=target, ignore_index=ignore_index, **self.kernel_cfg())

    def forward(self, x, hidden_states):
        (h, c) = hidden_states
        if x.dim() == 1:
            x = x.unsqueeze(0)
        _, h_next, c_next = self.sequence(x.unsqueeze(1), h, c)
        return h_next, c_next


class MyVMLMFgCellg2(MyVMLMFCellg2):
    """The group cell without the vector multiplication (the reference's ablation): no dia_x / dia_h, no
    diagonal removal, and BOTH sides chunk their pre-activations as (f, i, n, o) (vmlmf_group.py:211,232)."""

    variant = _lib.V6_GROUP_NOVM

    def __init__(self, input_size, hidden_size, w_rank=None, u_ranks=None, g=2,
                 recurrent_init=None, hidden_init=None):
        nn.Module.__init__(self)
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.recurrent_init = recurrent_init
        self.hidden_init = hidden_init
        self.w_rank = w_rank
        self.u_ranks = u_ranks
        self.g = g
        self.layers = nn.ParameterDict()
        self.layers['u_x'] = nn.Parameter(0.1 * torch.randn([input_size, w_rank]))
        self.layers['v_x'] = nn.Parameter(0.1 * torch.randn([4 * hidden_size, w_rank]))
        for s in range(self.g):
            self.layers[f'u_h_{s}'] = nn.Parameter(0.1 * torch.randn([g, int(hidden_size / g), u_ranks[s]]))
            self.layers[f'v_h_{s}'] = nn.Parameter(0.1 * torch.randn([g, u_ranks[s], 4 * int(hidden_size / g)]))
        for vec in ['x', 'h']:
            self.layers[f'bias_{vec}'] = nn.Parameter(torch.ones([1, 4 * hidden_size]))

    def __repr__(self):
        return (f"LSTM VM Group (input:{self.input_size}, hidden:{self.hidden_size}, "
                f"w_rank:{self.w_rank}, u_ranks:{self.u_ranks})")

    def kernel_params(self):
        L = self.layers
        out = [L['u_x'], L['v_x'], L['bias_x'], L['bias_h']]
        for s in range(self.g):
            out += [L[f'u_h_{s}'], L[f'v_h_{s}']]
        return tuple(out)


class MyLSTMCell(nn.Module):
    """Vanilla / plain low-rank LSTM cell of the reference (the baselines VMLMF is compared with).

    Low-rank mode (w_rank and u_ranks given) is the VMLMF recurrence with d = 0 and no diagonal removal: on a HIP
    device it runs the same sequence kernels (variant 5, per-gate V factors passed as they are).  Vanilla mode
    (dense (I,H)/(H,H) gate matrices) is outside the hot path and stays stock GEMMs through rocBLAS."""

    variant = _lib.V5_LMF_CELL

    def __init__(self, input_size, hidden_size, w_rank=None, u_ranks=None,
                 recurrent_init=None, hidden_init=None):
        super().__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.recurrent_init = recurrent_init
        self.hidden_init = hidden_init
        self.w_rank = w_rank
        self.u_ranks = u_ranks[0] if isinstance(u_ranks, list) else u_ranks
        I, H = input_size, hidden_size

        def mk(*shape):
            return nn.Parameter(0.1 * torch.randn(list(shape)))

        if w_rank is None:
            self.w1, self.w2, self.w3, self.w4 = mk(I, H), mk(I, H), mk(I, H), mk(I, H)
        else:
            self.w = mk(I, w_rank)
            self.w1, self.w2, self.w3, self.w4 = mk(w_rank, H), mk(w_rank, H), mk(w_rank, H), mk(w_rank, H)
        if u_ranks is None:
            self.u1, self.u2, self.u3, self.u4 = mk(H, H), mk(H, H), mk(H, H), mk(H, H)
        else:
            r = u_ranks   # the raw argument, as the reference does: a list raises TypeError here (vmlmf.py:177)
            self.u = mk(H, r)
            self.u1, self.u2, self.u3, self.u4 = mk(r, H), mk(r, H), mk(r, H), mk(r, H)
        self.bias_f = nn.Parameter(torch.ones([1, H]))
        self.bias_i = nn.Parameter(torch.ones([1, H]))
        self.bias_c = nn.Parameter(torch.ones([1, H]))
        self.bias_o = nn.Parameter(torch.ones([1, H]))

    @property
    def low_rank(self):
        return self.w_rank is not None and self.u_ranks is not None

    def kernel_params(self):
        return (self.w, self.u, self.w1, self.w2, self.w3, self.w4, self.u1, self.u2, self.u3, self.u4,
                self.bias_i, self.bias_f, self.bias_o, self.bias_c)

    def kernel_cfg(self):
        return dict(variant=self.variant, w_rank=self.w_rank, u_ranks=[self.u_ranks], g=1,
                    dtype=getattr(self, "compute_dtype", "f32"), pack_cache=getattr(self, "_pack_cache", None))

    def sequence(self, x, h0=None, c0=None, time_major=False, head=None, target=None, ignore_index=-100):
        """Whole-sequence evaluation on the HIP kernels (low-rank mode only): (y, hT, cT)."""
        if not self.low_rank:
            raise RuntimeError("vmlmf_amd: MyLSTMCell.sequence needs w_rank and u_ranks (the vanilla cell is "
                               "not on the HIP path)")
        return vmlmf_sequence(x=x, h0=h0, c0=c0, params=self.kernel_params(), time_major=time_major, head=head,
                              target=target, ignore_index=ignore_index, **self.kernel_cfg())

    def forward(self, x, hidden_states):
        (h, c) = hidden_states
        if self.low_rank and x.is_cuda:
            if x.dim() == 1:
                x = x.unsqueeze(0)
            _, h_next, c_next = self.sequence(x.unsqueeze(1), h, c)
            return h_next, c_next
        xin = x if self.w_rank is None else torch.matmul(x, self.w)
        hin = h if self.u_ranks is None else torch.matmul(h, self.u)
        pre = [torch.matmul(xin, w) + torch.matmul(hin, u)
               for w, u in ((self.w1, self.u1), (self.w2, self.u2), (self.w3, self.u3), (self.w4, self.u4))]
        i = torch.sigmoid(pre[0] + self.bias_i)
        f = torch.sigmoid(pre[1] + self.bias_f)
        o = torch.sigmoid(pre[2] + self.bias_o)
        c_tilda = torch.tanh(pre[3] + self.bias_c)
        c_next = f * c + i * c_tilda
        return o * torch.tanh(c_next), c_next


class MyLSTM(nn.Module):
    """Stack of layers over a sequence.  VMLMF cells (and the low-rank baseline cell on a HIP device) run one fused
    sequence pipeline per layer; any other cell class keeps the reference's per-timestep loop over that cell's
    own forward."""

    def __init__(self, input_size, hidden_layer_sizes=None, batch_first=True,
                 recurrent_inits=None, hidden_inits=None, w_rank=None, u_ranks=None,
                 cell=MyLSTMCell, **kwargs):
        super().__init__()
        if hidden_layer_sizes is None:
            hidden_layer_sizes = [32, 32]
        self.input_size = input_size
        self.hidden_layer_sizes = hidden_layer_sizes
        self.batch_first = batch_first
        self.w_rank = w_rank
        self.drop = nn.Dropout(p=0.5)   # defined, never applied (vmlmf.py:268)
        self.cell = cell
        self.u_ranks = u_ranks[0] if isinstance(u_ranks, list) and len(u_ranks) < 2 else u_ranks
        self.time_index, self.batch_index = (1, 0) if batch_first else (0, 1)
        cells = []
        in_size = input_size
        for i, hidden_size in enumerate(hidden_layer_sizes):
            if recurrent_inits is not None:
                kwargs["recurrent_init"] = recurrent_inits[i]
            if hidden_inits is not None:
                kwargs["hidden_init"] = hidden_inits[i]
            cells.append(self.cell(in_size, hidden_size, w_rank=self.w_rank, u_ranks=self.u_ranks, **kwargs))
            in_size = hidden_size
        self.rnncells = nn.ModuleList(cells)

    def run_layers(self, x, head=None, target=None, ignore_index=-100):
        """(output sequence of the last layer, [final h of every layer]); with `head` = (weight, bias) of a classifier on the
        last layer's final hidden state also its logits (None when the last layer cannot carry it); with a `target` for that
        classifier also the mean cross-entropy of the logits (None when the launch that formed them could not carry it)."""
        hiddens, logits, loss = [], None, None
        # every layer in one wavefront launch per direction when the stack is covered (same cell type and sizes above the
        # first layer; include/vmlmf_hip.h: vmlmf_stack_*); a classifier rides on the top layer's workgroups
        cells = list(self.rnncells)
        # (hidden sizes may differ from layer to layer, vmlmf.py:283-292: the library decides what its launches cover - a VMLMF cell needs
        #  input_size <= hidden_size, so the sizes of such a stack never shrink)
        if (x.is_cuda and all(type(c) is type(cells[0]) and hasattr(c, "kernel_cfg") for c in cells)
                and getattr(cells[0], "low_rank", True)):
            cfg = cells[0].kernel_cfg()
            cfg.pop("pack_cache", None)
            if all({k: v for k, v in c.kernel_cfg().items() if k != "pack_cache"} == cfg for c in cells[1:]):
                # (a single layer whose kept parameter images were asked for stays on the per-layer call, which uses them)
                kept = len(cells) == 1 and getattr(cells[0], "_pack_cache", None) is not None and stack_mode() != "1"
                out = None if kept else vmlmf_stack(x=x, layer_params=[c.kernel_params() for c in cells],
                                                    time_major=not self.batch_first, head=head, **cfg)
                if out is not None:
                    if head is not None:
                        return (out[0], out[1], out[3]) + ((None,) if target is not None else ())
                    return out[0], out[1]
        for i, cell in enumerate(self.rnncells):
            fused = hasattr(cell, "sequence") and (not isinstance(cell, MyLSTMCell) or (cell.low_rank and x.is_cuda))
            if fused and head is not None and i == len(self.rnncells) - 1:
                if target is not None:
                    x, h, _, logits, loss = cell.sequence(x, None, None, time_major=not self.batch_first, head=head, target=target,
                                                          ignore_index=ignore_index)
                else:
                    x, h, _, logits = cell.sequence(x, None, None, time_major=not self.batch_first, head=head)
            elif fused:
                x, h, _ = cell.sequence(x, None, None, time_major=not self.batch_first)
            else:
                B = x.size(self.batch_index)
                h = torch.zeros(B, self.hidden_layer_sizes[i], device=x.device)
                c = torch.zeros(B, self.hidden_layer_sizes[i], device=x.device)
                outs = []
                for x_t in torch.unbind(x, self.time_index):
                    h, c = cell(x_t, (h, c))
                    outs.append(h)
                x = torch.stack(outs, self.time_index)
            hiddens.append(h)
        if head is not None:
            return (x, hiddens, logits) + ((loss,) if target is not None else ())
        return x, hiddens

    def forward(self, x):
        x, hiddens = self.run_layers(x)
        return x, torch.cat(hiddens, -1)


class Net(nn.Module):
    """MyLSTM + Linear(H, 18) classifier on the last timestep (18 classes hard-coded, vmlmf.py:345)."""

    def __init__(self, input_size, layer_sizes=None, w_rank=None, u_rank=None, model=MyLSTM, cell=MyLSTMCell):
        super().__init__()
        if layer_sizes is None:
            layer_sizes = [32, 32]
        self.rnn = model(input_size, hidden_layer_sizes=layer_sizes, batch_first=True,
                         w_rank=w_rank, u_ranks=u_rank, cell=cell)
        self.lin = nn.Linear(layer_sizes[-1], 18)
        self.lin.bias.data.fill_(.1)
        self.lin.weight.data.normal_(0, .01)
        # the reference keeps an extra, never-trained cell "for unit_test" (vmlmf.py:349-350); its
        # parameters are part of every reference checkpoint, so it is kept for state_dict compatibility
        u = u_rank[-1] if cell == MyVMLMFCell else u_rank
        self.cell = cell(input_size, layer_sizes[-1], w_rank=w_rank, u_ranks=u)

    def _fast(self, x):
        """The per-call plan of the headline family - ONE VMLMF layer with a narrow input (its x side rides inside the recurrent
        launch) under a classifier of at most 32 classes, fp32 on a HIP device, the C++ binding loaded: everything run_layers() and
        vmlmf_sequence() would find out again on every call, found out once.  (cell's parameter dict, constants of the C++ op) or None.
        Invalidated by anything that changes the answer: another device / dtype, kept parameter images, a compute dtype, the
        VMLMF_STACK mode (read at build time only for mode "1", which routes single layers to the wavefront launch)."""
        plan = self.__dict__.get("_fast_plan")
        if plan is None:
            plan = False
            rnn = self.rnn
            if (type(rnn) is MyLSTM and rnn.batch_first and len(rnn.rnncells) == 1 and type(rnn.rnncells[0]) is MyVMLMFCell
                    and stack_mode() != "1"):
                cell = rnn.rnncells[0]
                from .functional import torch_ops
                ops = torch_ops()
                if (ops is not None and cell.input_size <= 16 and cell.hidden_size <= 192 and self.lin.weight.shape[0] <= _lib.HEAD_MAX_CLASSES
                        and self.lin.weight.dtype == torch.float32):
                    plan = (ops, cell, cell._parameters, self.lin._parameters, int(cell.w_rank), [int(cell.u_ranks)])
            self.__dict__["_fast_plan"] = plan
        if plan is False or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 3:
            return None
        cd = plan[1].__dict__
        if cd.get("_pack_cache") is not None or cd.get("compute_dtype", "f32") != "f32":
            return None
        return plan

    def _apply(self, fn, *args, **kwargs):   # .to() / .cuda() / .float(): the plan is looked up again
        self.__dict__.pop("_fast_plan", None)
        return super()._apply(fn, *args, **kwargs)

    def forward(self, x):
        plan = self._fast(x)
        if plan is not None:
            ops, cell, cp, lp, rw, ur = plan
            out = ops.sequence(x, None, None, [cp["dia_x"], cp["dia_h"], cp["u_x"], cp["v_x"], cp["b_x"], cp["b_h"], cp["u_h"], cp["v_h"]],
                               cell.variant, 1, rw, ur, False, 0, None, lp["weight"], lp["bias"])
            return out[3].squeeze(1)
        if isinstance(self.rnn, MyLSTM) and self.rnn.batch_first:
            # y[:, -1] IS the last layer's final h (same kernel value): taking it from there keeps autograd
            # from materialising a zero (B,T,H) gradient for y just to carry its last slice
            # ... and the classifier rides on the last layer's kernels when it can (HIP tensors, <= 32 classes): its logits come
            # out of the recurrence's epilogue, its backward is folded into the layer's backward (no launches of its own)
            ride = (x.is_cuda and x.dtype == torch.float32 and self.lin.weight.shape[0] <= _lib.HEAD_MAX_CLASSES
                    and self.lin.weight.dtype == torch.float32)
            if ride:
                _, hiddens, logits = self.rnn.run_layers(x, head=(self.lin.weight, self.lin.bias))
                if logits is not None:
                    return logits.squeeze(1)
            else:
                _, hiddens = self.rnn.run_layers(x)
            last = hiddens[-1]
        else:
            y, _ = self.rnn(x)
            last = y[:, -1]
        return head_linear(last, self.lin.weight, self.lin.bias).squeeze(1)

    def loss(self, x, target, ignore_index=-100, return_logits=False):
        """criterion(self(x), target) for criterion = nn.CrossEntropyLoss(ignore_index=...) with its other arguments at their
        defaults - the pair of lines `output = model(data); loss = criterion(output, target)` of the reference's loop
        (V/src/train_test/train.py:61-63) as ONE call, so that the criterion can ride on the launch that forms the logits: a batch
        row's logits, log-sum-exp, loss term and d(loss)/d(logits) come out of the forward recurrence's epilogue, and the backward
        needs no criterion launch either.  Same values as the two lines (the mean's summation order differs); wherever the classifier
        cannot ride (CPU tensors, other dtypes, more than 32 classes, stacks on the wavefront launches) it IS the two lines."""
        plan = self._fast(x) if (target.dtype == torch.int64 and target.dim() == 1) else None
        if plan is not None:
            from .functional import ce_ticket, unit_gradient
            ops, cell, cp, lp, rw, ur = plan
            out = ops.sequence_loss(x, None, None, [cp["dia_x"], cp["dia_h"], cp["u_x"], cp["v_x"], cp["b_x"], cp["b_h"], cp["u_h"], cp["v_h"]],
                                    cell.variant, 1, rw, ur, False, 0, None, lp["weight"], lp["bias"], target, int(ignore_index),
                                    unit_gradient(x.device), ce_ticket(x.device))
            return (out[4], out[3]) if return_logits else out[4]
        ride = (isinstance(self.rnn, MyLSTM) and self.rnn.batch_first and x.is_cuda and x.dtype == torch.float32
                and self.lin.weight.shape[0] <= _lib.HEAD_MAX_CLASSES and self.lin.weight.dtype == torch.float32
                and target.dtype == torch.int64 and target.dim() == 1)
        if ride:
            _, _, logits, loss = self.rnn.run_layers(x, head=(self.lin.weight, self.lin.bias), target=target, ignore_index=ignore_index)
            if logits is not None and loss is not None:
                return (loss, logits) if return_logits else loss
            if logits is not None:
                from .functional import cross_entropy
                loss = cross_entropy(logits.squeeze(1), target, ignore_index)
                return (loss, logits) if return_logits else loss
        from .functional import cross_entropy
        logits = self.forward(x)
        loss = cross_entropy(logits, target, ignore_index)
        return (loss, logits) if return_logits else loss
