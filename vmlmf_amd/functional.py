"""torch.autograd bridge over the C ABI: one call = one layer over the whole sequence.

PyTorch supplies device memory (caching allocator), the current HIP stream and autograd bookkeeping; all
arithmetic of the hot path happens in libvmlmf_hip.so.  Tensors must live on a HIP device.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib

# order in which parameter tensors are passed to VmlmfSeqFn (and gradients come back)
#   V1-V4: dia_x dia_h u_x v_x b_x b_h u_h[0] v_h[0] (u_h[1] v_h[1])
#   V6:    u_x v_x b_x b_h u_h[0] v_h[0] u_h[1] v_h[1]                       (no vm vectors)
#   V5:    w u w1 w2 w3 w4 u1 u2 u3 u4 bias_i bias_f bias_o bias_c          (gate order of the kernels: i, f, o, c~)
N_FIXED = 6


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


# ---- C++ binding (csrc/torch_binding.cpp: TORCH_LIBRARY "vmlmf" with C++ autograd functions over the same C ABI) --------
# Loaded when vmlmf_amd/lib/libvmlmf_torch.so has been built (`make -C vmlmf_amd/csrc torch`, done by
# __graft_entry__.build()).  Same kernels, same results; it only takes the per-call bookkeeping out of Python (eager
# steps of the unchanged reference loop are host-bound at the UCI-HAR shape).  VMLMF_PYBIND=ctypes keeps the classes below.
_OPS = None


def torch_ops():
    """torch.ops.vmlmf, or None when the C++ binding is not built / disabled."""
    global _OPS
    if _OPS is None:
        _OPS = False
        import os
        path = os.path.join(os.path.dirname(_lib.LIB_PATH), "libvmlmf_torch.so")
        if os.environ.get("VMLMF_PYBIND", "") != "ctypes" and os.path.exists(path) and "VMLMF_LIB" not in os.environ:
            _lib.lib()                       # libvmlmf_hip.so first (the binding links against it)
            try:
                torch.ops.load_library(path)
                _OPS = torch.ops.vmlmf
            except OSError:
                _OPS = False
    return _OPS or None


def _params_struct(tensors, g, variant=_lib.V1_CELL):
    p = _lib.Params()
    if variant == _lib.V5_LMF_CELL:
        p.u_x, p.u_h[0] = tensors[0].data_ptr(), tensors[1].data_ptr()
        for k in range(4):
            p.w_gate[k] = tensors[2 + k].data_ptr()
            p.u_gate[k] = tensors[6 + k].data_ptr()
            p.b_gate[k] = tensors[10 + k].data_ptr()
        return p
    names = ["dia_x", "dia_h", "u_x", "v_x", "b_x", "b_h"]
    if variant == _lib.V6_GROUP_NOVM:
        names = names[2:]
    for name, t in zip(names, tensors[:len(names)]):
        setattr(p, name, t.data_ptr())
    for s in range(g):
        p.u_h[s] = tensors[len(names) + 2 * s].data_ptr()
        p.v_h[s] = tensors[len(names) + 2 * s + 1].data_ptr()
    return p


def _hidden_size(variant, params):
    if variant == _lib.V5_LMF_CELL:
        return params[1].shape[0]              # u (H, ru)
    if variant == _lib.V6_GROUP_NOVM:
        return params[2].shape[-1] // 4        # bias_x (1, 4H)
    return params[1].shape[-1]                 # dia_h (1, H)


# (variant, g, w_rank, u_ranks, time_major, dtype, B, T, I, H, training) -> (Desc, Sizes): host-side descriptor cache
_DESC_CACHE = {}
# one grow-only scratch buffer per device: workspace contents never outlive the call that fills them and all
# calls on a device are serialised on the current stream
_WORKSPACE = {}


def _desc_for(cfg, B, T, I, H, training):
    key = cfg + (B, T, I, H, training)
    hit = _DESC_CACHE.get(key)
    if hit is None:
        variant, g, w_rank, u_ranks, time_major, dtype = cfg
        desc = _lib.make_desc(variant, B, T, I, H, w_rank, u_ranks, g=g, time_major=time_major, training=training, dtype=dtype)
        hit = (desc, _lib.query(desc))
        _DESC_CACHE[key] = hit
    return hit


def _workspace(dev, nbytes):
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(nbytes, device=dev, dtype=torch.uint8)      # graph-private pool
    key = (dev.index, _lib.raw_stream(dev).value)
    buf = _WORKSPACE.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        _WORKSPACE[key] = buf
    return buf


class PackCache:
    """Kept parameter images of one layer (C ABI: vmlmf_pack_params / vmlmf_seq_*_packed), opt-in through
    vmlmf_amd.cache_packed_parameters(module).  The library packs the reference-layout parameters into the kernels' register
    images on every forward (6 us of the 185 us headline step); while the parameters are unchanged the images can be
    reused - inference, evaluation, gradient accumulation, a loop without an optimizer step.  "Unchanged" is judged by
    (data_ptr, _version) of every parameter tensor: in-place updates under torch.no_grad() (torch.optim, the reference's
    `param -= lr * grad`, vmlmf_amd.optim) are seen; writes through `param.data` are NOT (they do not bump the version) -
    that is why the cache is opt-in.  During a stream capture the cache is only read, never filled."""
    __slots__ = ("key", "packed", "hits", "fills")

    def __init__(self):
        self.key, self.packed, self.hits, self.fills = None, None, 0, 0

    def get(self, cfg, params, x, B, T, I, H):
        if self.key == "unsupported":
            return None
        key = (cfg, B, I, H, x.device.index, _lib.lib().vmlmf_tune_generation(),
               tuple((p.data_ptr(), p._version) for p in params))
        if key == self.key:
            self.hits += 1
            return self.packed
        if torch.cuda.is_current_stream_capturing():
            return None                       # packed inside the captured call, as without a cache
        variant, g, w_rank, u_ranks, time_major, dtype = cfg
        desc = _lib.make_desc(variant, B, T, I, H, w_rank, u_ranks, g=g, time_major=time_major, training=True, dtype=dtype)
        n = ctypes.c_size_t()
        rc = _lib.lib().vmlmf_pack_bytes(ctypes.byref(desc), ctypes.byref(n))
        if rc == _lib.E_UNSUPPORTED:          # step-wise / clustered layers keep per-call state in their image
            self.key, self.packed = "unsupported", None
            return None
        _lib.check(rc)
        packed = torch.empty(n.value, device=x.device, dtype=torch.uint8)   # a NEW buffer: an earlier forward's backward may still need the old one
        ps = _params_struct(params, g, variant)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().vmlmf_pack_params(ctypes.byref(desc), ctypes.byref(ps), _ptr(packed), _lib.raw_stream(x.device)))
        self.key, self.packed = key, packed
        self.fills += 1
        return packed


def cache_packed_parameters(module, enable=True):
    """Let every VMLMF layer under `module` keep its packed parameter images between forward calls while its parameters are
    unchanged (see PackCache for what "unchanged" can and cannot see).  Returns the number of layers switched."""
    n = 0
    for m in module.modules():
        if hasattr(m, "kernel_params"):
            m._pack_cache = PackCache() if enable else None
            n += 1
    return n


def _require_hip(t, what):
    if not t.is_cuda:
        raise RuntimeError(
            f"vmlmf_amd: {what} is on {t.device}; the VMLMF hot path runs only as HIP kernels on an MI355X "
            "(no CPU fallback). Move the module and inputs to 'cuda'.")
    if t.dtype != torch.float32:
        raise RuntimeError(f"vmlmf_amd: {what} must be float32, got {t.dtype}")


_TICKET = {}
_TICKET_CAPTURE_SLOTS = 16


def ce_ticket(device):
    """The ticket words of the criterion that rides on the forward launch (include/vmlmf_hip.h: vmlmf_ce.ticket): two zeroed
    int64; every launch leaves them zero.  Launches that share a ticket must be ordered on ONE stream, so the pair is kept per
    (device, stream).  Nothing is allocated inside a stream capture (the allocation would land in the graph's private pool and a
    memset node in the graph): a captured forward takes the next pair of a block of 16 that the first call outside a capture set
    aside for the device, so different graphs - replayed on whatever streams - do not meet on one pair either.  Only a process whose
    very first forward is captured pays a zeroing node in that graph (the pair is then the graph's own and is not kept here)."""
    device = torch.device(device)
    capturing = torch.cuda.is_current_stream_capturing()
    block = _TICKET.get((device.index, "capture"))
    if block is None and not capturing:
        block = _TICKET[(device.index, "capture")] = [torch.zeros(2 * _TICKET_CAPTURE_SLOTS, device=device, dtype=torch.int64), 0]
    if capturing:
        if block is None:
            return torch.zeros(2, device=device, dtype=torch.int64)
        i = block[1] % _TICKET_CAPTURE_SLOTS
        block[1] += 1
        return block[0][2 * i:2 * i + 2]
    key = (device.index, _lib.raw_stream(device).value)
    t = _TICKET.get(key)
    if t is None:
        t = _TICKET[key] = torch.zeros(2, device=device, dtype=torch.int64)
    return t


class VmlmfSeqFn(torch.autograd.Function):
    """y, hT, cT, logits, loss = f(x, h0, c0, *params) for one layer.  cfg = (variant, g, w_rank, u_ranks, time_major, dtype).
    target (with a head): mean cross-entropy of the logits against it as the fifth output (vmlmf_ce: inside the same launch)."""

    @staticmethod
    def forward(ctx, cfg, packed, x, h0, c0, head_w, head_b, target, ignore_index, drop, *params):
        variant, g, w_rank, u_ranks, time_major, _ = cfg
        ctx.set_materialize_grads(False)
        _require_hip(x, "input")
        for p in params:
            _require_hip(p, "parameter")
        x = x.contiguous()
        params = tuple(p.contiguous() for p in params)
        if time_major:
            T, B, I = x.shape
        else:
            B, T, I = x.shape
        H = _hidden_size(variant, params)
        training = bool(any(ctx.needs_input_grad))   # False under torch.no_grad(): inference kernels
        ctx.packed = packed                           # kept parameter images (PackCache) or None
        desc, sizes = _desc_for(cfg, B, T, I, H, training)
        dev = x.device
        y = torch.empty((T, B, H) if time_major else (B, T, H), device=dev, dtype=torch.float32)
        hT = torch.empty((B, H), device=dev, dtype=torch.float32)
        cT = torch.empty((B, H), device=dev, dtype=torch.float32)
        ws = _workspace(dev, sizes.workspace_bytes)
        reserve = torch.empty(sizes.reserve_bytes, device=dev, dtype=torch.uint8) if training else None
        h0c = None if h0 is None else h0.contiguous()
        c0c = None if c0 is None else c0.contiguous()
        ps = _params_struct(params, g, variant)
        stream = _lib.raw_stream(dev)
        # a classifier riding on the final hidden state (Net.lin): its logits are the 4th output
        hw = None if head_w is None else head_w.contiguous()
        hb = None if head_b is None else head_b.contiguous()
        logits = torch.empty((B, hw.shape[0]) if hw is not None else (0,), device=dev, dtype=torch.float32)
        hd = _lib.Head()
        if hw is not None:
            _require_hip(hw, "head weight")
            hd.classes, hd.weight, hd.logits = hw.shape[0], hw.data_ptr(), logits.data_ptr()
            hd.bias = None if hb is None else hb.data_ptr()
        ex = _lib.Extra()
        ex.packed = None if packed is None else packed.data_ptr()
        ex.head = ctypes.pointer(hd) if hw is not None else None
        # the criterion on those logits: loss | nvalid | lse[B] in one allocation, the unit gradient of the logits beside it
        stats = dz_unit = None
        ce = _lib.Ce()
        if target is not None:
            if hw is None:
                raise RuntimeError("vmlmf_amd: a criterion rides on the classifier's logits (head)")
            tg = target.contiguous()
            stats = torch.empty(2 + B, device=dev, dtype=torch.float32)
            dz_unit = torch.empty_like(logits) if training else None
            base = stats.data_ptr()
            ce.target, ce.ignore_index = tg.data_ptr(), int(ignore_index)
            ce.loss, ce.nvalid, ce.lse = base, base + 4, base + 8
            ce.dlogits_unit = None if dz_unit is None else dz_unit.data_ptr()
            ce.ticket = ce_ticket(dev).data_ptr()
            ex.ce = ctypes.pointer(ce)
        # dropout of the layer's output inside its launches (vmlmf_dropout, ABI 11): drop = (p, snapshot, site); the first output is
        # then the DROPPED copy, y itself stays with the backward
        yd = dr = None
        if drop is not None:
            yd = torch.empty_like(y)
            dr = _lib.Dropout(float(drop[0]), int(drop[2]), drop[1].data_ptr(), yd.data_ptr())
            ex.drop = ctypes.pointer(dr)
        with _lib.on_device(dev):
            _lib.check(_lib.lib().vmlmf_seq_forward_ex(
                ctypes.byref(desc), ctypes.byref(ps), _ptr(x), _ptr(h0c), _ptr(c0c), _ptr(y), _ptr(hT),
                _ptr(cT), _ptr(reserve), _ptr(ws), sizes.workspace_bytes, stream, ctypes.byref(ex)))
        if training:
            ctx.cfg, ctx.desc, ctx.sizes = cfg, desc, sizes
            ctx.has_h0, ctx.has_c0 = h0 is not None, c0 is not None
            ctx.has_head, ctx.has_head_b = hw is not None, hb is not None
            ctx.has_ce = dz_unit is not None
            ctx.drop = None if drop is None else (float(drop[0]), int(drop[2]))
            ctx.save_for_backward(x, y, reserve, *params, *([h0c] if h0 is not None else []),
                                  *([c0c] if c0 is not None else []), *([hw] if hw is not None else []),
                                  *([dz_unit] if dz_unit is not None else []), *([drop[1]] if drop is not None else []))
            ctx.nparams = len(params)
        loss = stats[0] if stats is not None else torch.empty((0,), device=dev, dtype=torch.float32)
        return (y if yd is None else yd), hT, cT, logits, loss

    @staticmethod
    def backward(ctx, dy, dhT, dcT, dlogits, dloss):
        variant, g, w_rank, u_ranks, time_major, _ = ctx.cfg
        saved = ctx.saved_tensors
        x, y, reserve = saved[0], saved[1], saved[2]
        params = saved[3:3 + ctx.nparams]
        rest = list(saved[3 + ctx.nparams:])
        h0 = rest.pop(0) if ctx.has_h0 else None
        c0 = rest.pop(0) if ctx.has_c0 else None
        hw = rest.pop(0) if ctx.has_head else None
        dz = rest.pop(0) if ctx.has_ce else None   # (popped whether or not the loss takes part: what follows it is the dropout snapshot)
        if dz is not None and dloss is not None:
            # the criterion's share of d(logits): what the forward launch wrote for d(loss) = 1 - as it is when the incoming
            # gradient IS the package's constant one (vmlmf_amd.unit_gradient), scaled otherwise
            unit = _UNIT.get(dz.device)
            if not (unit is not None and dloss.data_ptr() == unit.data_ptr()):
                dz = dz * dloss
            dlogits = dz if dlogits is None else dlogits + dz
        snap = rest.pop(0) if ctx.drop is not None else None
        dev = x.device
        desc, sizes = ctx.desc, ctx.sizes
        dy = None if dy is None else dy.contiguous()
        dhT = None if dhT is None else dhT.contiguous()
        dcT = None if dcT is None else dcT.contiguous()
        need_dx = ctx.needs_input_grad[2]   # (cfg, packed, x, h0, c0, head_w, head_b, target, ignore_index, drop, *params)
        dx = torch.empty_like(x) if need_dx else None
        B, H = y.shape[1 if time_major else 0], y.shape[2]
        dh0 = torch.empty((B, H), device=dev, dtype=torch.float32) if ctx.has_h0 else None
        dc0 = torch.empty((B, H), device=dev, dtype=torch.float32) if ctx.has_c0 else None
        # one flat buffer for all parameter gradients (views are returned): a single allocation, and the
        # gradients of a layer are contiguous for the data-parallel all-reduce
        # (the classifier's weight and bias gradients are the tail of the same allocation: ONE flat buffer, ONE all-reduce per
        # step)
        use_head = hw is not None and dlogits is not None
        total = sum(p.numel() for p in params)
        flat = torch.empty(total + ((hw.shape[0] * H + hw.shape[0]) if use_head else 0), device=dev, dtype=torch.float32)
        grads, o = [], 0
        for p in params:
            grads.append(flat[o:o + p.numel()].view(p.shape))
            o += p.numel()
        grads = tuple(grads)
        ws = _workspace(dev, sizes.workspace_bytes)
        ps = _params_struct(params, g, variant)
        gs = _params_struct(grads, g, variant)
        stream = _lib.raw_stream(dev)
        dW = db = None
        hd = _lib.Head()
        if use_head:
            dl = dlogits.contiguous()
            C = hw.shape[0]
            dW = flat[total:total + C * H].view(C, H)
            db = flat[total + C * H:] if ctx.has_head_b else None
            hd.classes, hd.weight, hd.dlogits, hd.dweight = C, hw.data_ptr(), dl.data_ptr(), dW.data_ptr()
            hd.dbias = None if db is None else db.data_ptr()
        ex = _lib.Extra()
        ex.packed = None if ctx.packed is None else ctx.packed.data_ptr()
        ex.head = ctypes.pointer(hd) if use_head else None
        if snap is not None:   # dy is the gradient of the dropped copy: the launch regenerates the forward's factors
            dr = _lib.Dropout(ctx.drop[0], ctx.drop[1], snap.data_ptr(), None)
            ex.drop = ctypes.pointer(dr)
        with _lib.on_device(dev):
            _lib.check(_lib.lib().vmlmf_seq_backward_ex(
                ctypes.byref(desc), ctypes.byref(ps), _ptr(x), _ptr(h0), _ptr(c0), _ptr(y), _ptr(reserve),
                _ptr(dy), _ptr(dhT), _ptr(dcT), _ptr(dx), _ptr(dh0), _ptr(dc0), ctypes.byref(gs),
                _ptr(ws), sizes.workspace_bytes, stream, ctypes.byref(ex)))
        return (None, None, dx, dh0, dc0, dW, db, None, None, None) + grads


def vmlmf_sequence(variant, x, h0, c0, params, w_rank, u_ranks, g=1, time_major=False, dtype="f32", pack_cache=None,
                   head=None, target=None, ignore_index=-100, drop=None):
    """Run one VMLMF layer over a whole sequence on the GPU.

    params: tensors in the order dia_x, dia_h, u_x, v_x, b_x, b_h, u_h[0], v_h[0] (, u_h[1], v_h[1]),
    each in the reference's layout (the cells without vm: see the table at the top of this file).
    dtype: "f32" (the reference's arithmetic) or "bf16" (bf16 MFMA with fp32 accumulation and state, bf16 tapes; all
    tensors stay float32 - include/vmlmf_hip.h: vmlmf_desc.dtype).  Returns (y, hT, cT).
    head: (weight (C, H), bias (C) or None) of a classifier on the layer's final hidden state (Net.lin); the call then
    returns (y, hT, cT, logits) and neither the logits nor their backward cost a launch of their own on the VALU kernels.
    target (with head): (B,) int64 class indices - the mean cross-entropy of the logits against them (nn.CrossEntropyLoss() with
    default arguments, train.py:58-65) comes back as a fifth element, formed inside the same forward launch.
    drop: (p, snapshot, site) - nn.Dropout(p) behind the layer (vmlmf_lm.py:438-439) without a mask tensor: the returned y is the
    dropped activation; inside the layer's own launches where the library takes it (vmlmf_dropout_fused: the row-block kernels,
    time-major), one launch of the package's otherwise (dropout()).
    """
    dt = _lib.DTYPES[dtype] if isinstance(dtype, str) else int(dtype)
    if drop is not None and head is None and x.is_cuda:
        ur_ = tuple(u_ranks) if isinstance(u_ranks, (list, tuple)) else (int(u_ranks),)
        cfg = (variant, g, int(w_rank), ur_, bool(time_major), dt)
        T, B = (x.shape[0], x.shape[1]) if time_major else (x.shape[1], x.shape[0])
        training = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        desc, _ = _desc_for(cfg, B, T, x.shape[2], _hidden_size(variant, params), training)
        if _lib.lib().vmlmf_dropout_fused(ctypes.byref(desc)) == 1:
            return VmlmfSeqFn.apply(cfg, None, x, h0, c0, None, None, None, ignore_index, tuple(drop), *params)[:3]
    if drop is not None:
        y, hT, cT = vmlmf_sequence(variant, x, h0, c0, params, w_rank, u_ranks, g=g, time_major=time_major, dtype=dtype,
                                   pack_cache=pack_cache)
        return dropout(y, drop[0], drop[1], drop[2]), hT, cT
    packed = None
    if pack_cache is not None and x.is_cuda:
        T, B = (x.shape[0], x.shape[1]) if time_major else (x.shape[1], x.shape[0])
        cfg0 = (variant, g, int(w_rank), tuple(u_ranks) if isinstance(u_ranks, (list, tuple)) else (int(u_ranks),), bool(time_major), dt)
        packed = pack_cache.get(cfg0, params, x, B, T, x.shape[2], _hidden_size(variant, params))
    ur = tuple(u_ranks) if isinstance(u_ranks, (list, tuple)) else (int(u_ranks),)
    ops = torch_ops()
    if ops is not None:
        if target is not None:
            return ops.sequence_loss(x, h0, c0, list(params), variant, g, int(w_rank), list(ur), bool(time_major), dt, packed,
                                     head[0], head[1], target, int(ignore_index), unit_gradient(x.device), ce_ticket(x.device))
        out = ops.sequence(x, h0, c0, list(params), variant, g, int(w_rank), list(ur), bool(time_major), dt, packed,
                           None if head is None else head[0], None if head is None else head[1])
        return out if head is not None else out[:3]
    cfg = (variant, g, int(w_rank), ur, bool(time_major), dt)
    out = VmlmfSeqFn.apply(cfg, packed, x, h0, c0, None if head is None else head[0], None if head is None else head[1],
                           target, ignore_index, None, *params)
    if target is not None:
        return out
    return out[:4] if head is not None else out[:3]


# ---- stacked layers: one wavefront launch per direction (C ABI 7: vmlmf_stack_*) ----------------------------------------
# (cfg, L, B, T, I, H, training) -> None (not covered: chain the per-layer calls) or (descs, reserve bytes, workspace bytes)
_STACK_CACHE = {}


def _stack_plan(cfg, L, B, T, I, H, training):
    """H: the layers' hidden size, or one per layer (round 6: MyLSTM builds any hidden_layer_sizes, vmlmf.py:283-292)."""
    Hs = tuple(H) if isinstance(H, (list, tuple)) else (int(H),) * L
    key = (cfg, L, B, T, I, Hs, training, _lib.lib().vmlmf_tune_generation())
    if key in _STACK_CACHE:
        return _STACK_CACHE[key]
    variant, g, w_rank, u_ranks, time_major, dtype = cfg
    plan = None
    if 1 <= L <= _lib.STACK_MAX and g in (1, 2) and dtype in (_lib.DT_F32, _lib.DT_BF16):
        layers = (_lib.StackLayer * L)()
        descs = []
        for l in range(L):
            d = _lib.make_desc(variant, B, T, I if l == 0 else Hs[l - 1], Hs[l], w_rank, u_ranks, g=g, time_major=time_major,
                               training=training, dtype=dtype)
            layers[l].desc = d
            descs.append(d)
        rb = (ctypes.c_size_t * L)()
        wb = ctypes.c_size_t()
        rc = _lib.lib().vmlmf_stack_query(L, ctypes.addressof(layers), ctypes.addressof(rb), ctypes.addressof(wb))
        if rc == 0:
            plan = (descs, [int(v) for v in rb], int(wb.value))
        elif rc not in (_lib.E_UNSUPPORTED, _lib.E_SHAPE):
            _lib.check(rc)
    _STACK_CACHE[key] = plan
    return plan


def stack_takes_dropout(cfg, L, B, T, I, H, training):
    """Does the stack run in the clustered form (include/vmlmf_hip.h: vmlmf_stack_dropout_fused), whose launches apply the dropout
    between the layers themselves?"""
    plan = _stack_plan(cfg, L, B, T, I, H, training)
    if plan is None:
        return False
    layers = (_lib.StackLayer * L)()
    for l in range(L):
        layers[l].desc = plan[0][l]
    return _lib.lib().vmlmf_stack_dropout_fused(L, ctypes.addressof(layers)) == 1


class VmlmfStackFn(torch.autograd.Function):
    """y_top, hT_0 .. hT_{L-1}, cT_0 .. cT_{L-1} = f(x, params of layer 0, ..., params of layer L-1): every layer of a stack in
    one launch per direction (include/vmlmf_hip.h: vmlmf_stack_forward / vmlmf_stack_backward).  Initial states are zero
    (MyLSTM.forward, vmlmf.py:296-298)."""

    @staticmethod
    def forward(ctx, cfg, L, x, head_w, head_b, h0, c0, drops, *params):
        # drops: None, or one entry per layer - None / (p, snapshot, site): nn.Dropout(p) behind that layer inside the stack's launches
        # (the clustered form only: vmlmf_stack_dropout_fused); the first output is then the top layer's DROPPED copy
        variant, g, w_rank, u_ranks, time_major, _ = cfg
        ctx.set_materialize_grads(False)
        _require_hip(x, "input")
        x = x.contiguous()
        params = tuple(p.contiguous() for p in params)
        nper = len(params) // L
        if time_major:
            T, B, I = x.shape
        else:
            B, T, I = x.shape
        Hs = tuple(_hidden_size(variant, params[l * nper:(l + 1) * nper]) for l in range(L))
        training = bool(any(ctx.needs_input_grad))
        plan = _stack_plan(cfg, L, B, T, I, Hs, training)
        if plan is None:
            raise RuntimeError("vmlmf_amd: this stack is not covered by the wavefront kernels (vmlmf_stack_supported)")
        descs, rbytes, wbytes = plan
        dev = x.device
        ys = [torch.empty((T, B, Hs[l]) if time_major else (B, T, Hs[l]), device=dev, dtype=torch.float32) for l in range(L)]
        # final states of every layer: one allocation when the layers are alike (what the LM carries), one per layer otherwise
        hc = [torch.empty((2, B, Hs[l]), device=dev, dtype=torch.float32) for l in range(L)] if len(set(Hs)) > 1 else \
            list(torch.empty((L, 2, B, Hs[0]), device=dev, dtype=torch.float32).unbind(0))
        reserves = [torch.empty(rbytes[l], device=dev, dtype=torch.uint8) if training else None for l in range(L)]
        ws = _workspace(dev, wbytes)
        h0c = None if h0 is None else h0.contiguous()      # initial states of every layer, (L, B, H), or None = zeros
        c0c = None if c0 is None else c0.contiguous()
        layers = (_lib.StackLayer * L)()
        keep = []
        yds = [None] * L
        for l in range(L):
            ps = _params_struct(params[l * nper:(l + 1) * nper], g, variant)
            keep.append(ps)
            ly = layers[l]
            ly.desc, ly.params = descs[l], ctypes.pointer(ps)
            if drops is not None and drops[l] is not None:
                yds[l] = torch.empty_like(ys[l])
                dr = _lib.Dropout(float(drops[l][0]), int(drops[l][2]), drops[l][1].data_ptr(), yds[l].data_ptr())
                keep.append(dr)
                ly.drop = ctypes.pointer(dr)
            ly.y, ly.hT, ly.cT = ys[l].data_ptr(), hc[l][0].data_ptr(), hc[l][1].data_ptr()
            ly.reserve = None if reserves[l] is None else reserves[l].data_ptr()
            ly.h0 = None if h0c is None else h0c[l].data_ptr()
            ly.c0 = None if c0c is None else c0c[l].data_ptr()
        # a classifier riding on the top layer's final hidden state (Net.lin): its logits are the last output
        hw = None if head_w is None else head_w.contiguous()
        hb = None if head_b is None else head_b.contiguous()
        logits = torch.empty((B, hw.shape[0]) if hw is not None else (0,), device=dev, dtype=torch.float32)
        hd = _lib.Head()
        if hw is not None:
            _require_hip(hw, "head weight")
            hd.classes, hd.weight, hd.logits = hw.shape[0], hw.data_ptr(), logits.data_ptr()
            hd.bias = None if hb is None else hb.data_ptr()
        with _lib.on_device(dev):
            _lib.check(_lib.lib().vmlmf_stack_forward(L, ctypes.addressof(layers), x.data_ptr(),
                                                      ctypes.addressof(hd) if hw is not None else None, ws.data_ptr(), wbytes,
                                                      _lib.raw_stream(dev)))
        if training:
            ctx.cfg, ctx.L, ctx.nper, ctx.plan = cfg, L, nper, plan
            ctx.has_head, ctx.has_head_b = hw is not None, hb is not None
            ctx.has_h0, ctx.has_c0 = h0c is not None, c0c is not None
            ctx.drops = None if drops is None else [None if d is None else (float(d[0]), int(d[2])) for d in drops]
            dsaved = [] if drops is None else [t for l in range(L) if drops[l] is not None for t in (yds[l], drops[l][1])]
            ctx.save_for_backward(x, *ys, *reserves, *params, *([hw] if hw is not None else []),
                                  *([h0c] if h0c is not None else []), *([c0c] if c0c is not None else []), *dsaved)
        top = ys[-1] if yds[-1] is None else yds[-1]
        return (top,) + tuple(hc[l][0] for l in range(L)) + tuple(hc[l][1] for l in range(L)) + (logits,)

    @staticmethod
    def backward(ctx, dy, *dstates):
        variant, g, w_rank, u_ranks, time_major, _ = ctx.cfg
        L, nper = ctx.L, ctx.nper
        descs, rbytes, wbytes = ctx.plan
        saved = ctx.saved_tensors
        x, ys, reserves, params = saved[0], saved[1:1 + L], saved[1 + L:1 + 2 * L], list(saved[1 + 2 * L:])
        dsaved = {}
        if ctx.drops is not None:     # (dropped copy, snapshot) of every layer with a dropout, saved behind everything else
            for l in reversed(range(L)):
                if ctx.drops[l] is not None:
                    snap = params.pop()
                    yd = params.pop()
                    dsaved[l] = (yd, snap)
        c0 = params.pop() if ctx.has_c0 else None
        h0 = params.pop() if ctx.has_h0 else None
        hw = params.pop() if ctx.has_head else None
        dlogits = dstates[2 * L]
        dev = x.device
        dy = None if dy is None else dy.contiguous()
        dhT = [None if d is None else d.contiguous() for d in dstates[:L]]
        dcT = [None if d is None else d.contiguous() for d in dstates[L:2 * L]]
        need_dx = ctx.needs_input_grad[2]
        dx = torch.empty_like(x) if need_dx else None
        # one flat buffer for the parameter gradients of the whole stack and of the classifier (views are returned)
        use_head = hw is not None and dlogits is not None
        total = sum(p.numel() for p in params)
        flat = torch.empty(total + ((hw.shape[0] * hw.shape[1] + hw.shape[0]) if use_head else 0), device=dev, dtype=torch.float32)
        grads, o = [], 0
        for p in params:
            grads.append(flat[o:o + p.numel()].view(p.shape))
            o += p.numel()
        ws = _workspace(dev, wbytes)
        B, H = ys[0].shape[1 if time_major else 0], ys[0].shape[2]
        dh0 = torch.empty((L, B, H), device=dev, dtype=torch.float32) if ctx.has_h0 else None
        dc0 = torch.empty((L, B, H), device=dev, dtype=torch.float32) if ctx.has_c0 else None
        layers = (_lib.StackLayer * L)()
        keep = []
        for l in range(L):
            ps = _params_struct(params[l * nper:(l + 1) * nper], g, variant)
            gs = _params_struct(grads[l * nper:(l + 1) * nper], g, variant)
            keep += [ps, gs]
            ly = layers[l]
            ly.desc, ly.params, ly.grads = descs[l], ctypes.pointer(ps), ctypes.pointer(gs)
            if l in dsaved:   # the launch regenerates the forward's factors; the layer above's weight gradients read the dropped copy
                dr = _lib.Dropout(ctx.drops[l][0], ctx.drops[l][1], dsaved[l][1].data_ptr(), dsaved[l][0].data_ptr())
                keep.append(dr)
                ly.drop = ctypes.pointer(dr)
            ly.y, ly.reserve = ys[l].data_ptr(), reserves[l].data_ptr()
            ly.h0 = None if h0 is None else h0[l].data_ptr()
            ly.c0 = None if c0 is None else c0[l].data_ptr()
            ly.dh0 = None if dh0 is None else dh0[l].data_ptr()
            ly.dc0 = None if dc0 is None else dc0[l].data_ptr()
            ly.dhT = None if dhT[l] is None else dhT[l].data_ptr()
            ly.dcT = None if dcT[l] is None else dcT[l].data_ptr()
        dW = db = None
        hd = _lib.Head()
        if use_head:
            dl = dlogits.contiguous()
            C, H = hw.shape
            dW = flat[total:total + C * H].view(C, H)
            db = flat[total + C * H:] if ctx.has_head_b else None
            hd.classes, hd.weight, hd.dlogits, hd.dweight = C, hw.data_ptr(), dl.data_ptr(), dW.data_ptr()
            hd.dbias = None if db is None else db.data_ptr()
        with _lib.on_device(dev):
            _lib.check(_lib.lib().vmlmf_stack_backward(L, ctypes.addressof(layers), x.data_ptr(), _ptr(dy), _ptr(dx),
                                                       ctypes.addressof(hd) if use_head else None, ws.data_ptr(), wbytes,
                                                       _lib.raw_stream(dev)))
        return (None, None, dx, dW, db, dh0, dc0, None) + tuple(grads)


def stack_mode():
    """VMLMF_STACK: "auto" (default: every covered stack of two or more layers - measured faster than the chained kernels
    from B = 64 to 2048, profiles/r02_stack_vs_chained_over_batch.txt; a single layer only when its input is wider than 16,
    i.e. when its x-projection would otherwise be a launch of its own, and T <= 96), "0" (never: chain the per-layer
    calls), "1" (whenever the wavefront kernels cover the stack)."""
    import os
    return os.environ.get("VMLMF_STACK", "auto")


def vmlmf_stack(variant, x, layer_params, w_rank, u_ranks, g=1, time_major=False, dtype="f32", head=None, h0=None, c0=None, drops=None):
    """Run a stack of VMLMF layers (zero initial states) in one wavefront launch per direction.  layer_params: one parameter
    tuple per layer, in vmlmf_sequence's order.  Returns (y of the top layer, [hT per layer], [cT per layer]) or None when
    the stack is not covered / not worth it (the caller then chains vmlmf_sequence calls).  head: (weight (C, H), bias or
    None) of a classifier on the top layer's final hidden state (<= 32 classes): a fourth element, its logits, is returned.
    h0 / c0: initial states of every layer as (L, B, H) tensors (None = zeros, MyLSTM.forward; the LM network carries them
    from batch to batch, vmlmf_lm.py:421-439)."""
    mode = stack_mode()
    L = len(layer_params)
    if mode == "0" or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 3:
        return None
    dt = _lib.DTYPES[dtype] if isinstance(dtype, str) else int(dtype)
    ur = tuple(u_ranks) if isinstance(u_ranks, (list, tuple)) else (int(u_ranks),)
    cfg = (variant, g, int(w_rank), ur, bool(time_major), dt)
    T, B = (x.shape[0], x.shape[1]) if time_major else (x.shape[1], x.shape[0])
    Hs = tuple(_hidden_size(variant, ps) for ps in layer_params)
    H = Hs[0]
    if len(set(Hs)) > 1:
        if h0 is not None or c0 is not None or drops is not None:
            return None     # (carried states / dropout: the LM's stacks, whose layers are alike)
        H = Hs
    xwave = x.shape[2] <= 16 and Hs[0] <= 192 and g == 1       # (vg_xwave_ok of the C side: <= 3 waves of units, one group)
    if mode != "1" and L == 1 and (xwave or T > 96):
        return None          # (a single layer with a narrow input already forms its x side inside the recurrent kernel; with a wide
                             #  one the x-team's per-step cost overtakes the two launches it saves at T ~ 128:
                             #  profiles/r02_stack_vs_chained_over_T.txt)
    # (the classifier's weight and bias count: with a frozen RNN under a trainable head the launch runs in training mode, and the
    #  coverage check must be made for that mode - otherwise an uncovered stack raised instead of falling back to chained layers)
    training = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for ps in layer_params for p in ps)
                                            or (h0 is not None and h0.requires_grad) or (c0 is not None and c0.requires_grad)
                                            or (head is not None and any(t is not None and t.requires_grad for t in head)))
    if _stack_plan(cfg, L, B, T, x.shape[2], H, training) is None:
        return None
    if drops is not None and any(d is not None for d in drops):
        # drops: per layer None / (p, snapshot, site) - only the clustered form applies dropout inside its launches
        if not stack_takes_dropout(cfg, L, B, T, x.shape[2], H, training):
            return None
    else:
        drops = None
    flat = [p for ps in layer_params for p in ps]
    hw, hb = (None, None) if head is None else head
    ops = torch_ops()
    if ops is not None and h0 is None and c0 is None and dt == _lib.DT_F32 and drops is None and len(set(Hs)) == 1:      # (initial states, the bf16 tape, dropout, unequal sizes: the ctypes form below)
        y, hT, cT, logits = ops.stack(x, flat, L, variant, int(w_rank), list(ur), int(g), bool(time_major), hw, hb)
        out = (y, list(hT.unbind(0)), list(cT.unbind(0)))
        return out + (logits,) if head is not None else out
    res = VmlmfStackFn.apply(cfg, L, x, hw, hb, h0, c0, drops, *flat)
    out = (res[0], list(res[1:1 + L]), list(res[1 + L:1 + 2 * L]))
    return out + (res[1 + 2 * L],) if head is not None else out


class HeadLinearFn(torch.autograd.Function):
    """logits = h @ W^T + bias for the classifier on the last timestep (Net.lin, V/src/models/vmlmf.py:345,
    353-355): two latency-sized kernels instead of three library GEMM launches and a bias-gradient reduction."""

    @staticmethod
    def forward(ctx, h, weight, bias):
        ctx.set_materialize_grads(False)
        _require_hip(h, "head input")
        _require_hip(weight, "head weight")
        if h.stride(-1) != 1:
            h = h.contiguous()
        weight = weight.contiguous()
        bias_c = None if bias is None else bias.contiguous()
        B, H = h.shape
        C = weight.shape[0]
        out = torch.empty((B, C), device=h.device, dtype=torch.float32)
        stream = _lib.raw_stream(h.device)
        with _lib.on_device(h.device):
            _lib.check(_lib.lib().vmlmf_head_forward(B, H, C, _ptr(h), h.stride(0), _ptr(weight), _ptr(bias_c),
                                                     _ptr(out), stream))
        ctx.save_for_backward(h, weight)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, dl):
        if dl is None:
            return None, None, None
        h, weight = ctx.saved_tensors
        dl = dl.contiguous()
        B, H = h.shape
        C = weight.shape[0]
        need_h, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dh = torch.empty((B, H), device=h.device, dtype=torch.float32) if need_h else None
        # weight and bias gradients share one allocation (contiguous for the data-parallel all-reduce)
        flat = torch.empty(C * H + C, device=h.device, dtype=torch.float32) if (need_w or need_b) else None
        dW = flat[:C * H].view(C, H) if need_w else None
        db = flat[C * H:] if need_b else None
        stream = _lib.raw_stream(h.device)
        with _lib.on_device(h.device):
            _lib.check(_lib.lib().vmlmf_head_backward(B, H, C, _ptr(h), h.stride(0), _ptr(weight), _ptr(dl),
                                                      _ptr(dh), _ptr(dW), _ptr(db), stream))
        return dh, dW, db


def set_compute_dtype(module, dtype):
    """Select the arithmetic of every VMLMF layer under `module`: "f32" (default, the reference's) or "bf16" (bf16 MFMA in
    the recurrence with fp32 accumulation and state, bf16 tapes; BASELINE configs[2]).  Parameters, inputs, outputs and
    gradients stay float32 tensors either way; a layer the bf16 kernels do not cover raises when it is run."""
    if dtype not in _lib.DTYPES:
        raise ValueError(f"dtype must be one of {sorted(_lib.DTYPES)}")
    n = 0
    for m in module.modules():
        if hasattr(m, "kernel_params"):
            m.compute_dtype = dtype
            n += 1
    return n


def head_linear(h, weight, bias):
    """nn.Linear on (B, H) rows through the head kernels when they apply (HIP fp32, <= 32 classes); the stock
    library op otherwise (it is not part of the VMLMF path and has no CPU restriction of its own)."""
    if (h.is_cuda and h.dim() == 2 and h.dtype == torch.float32 and weight.dtype == torch.float32
            and weight.shape[0] <= _lib.HEAD_MAX_CLASSES):
        ops = torch_ops()
        if ops is not None:
            return ops.head_linear(h, weight, bias)
        return HeadLinearFn.apply(h, weight, bias)
    return torch.nn.functional.linear(h, weight, bias)


_UNIT = {}


def unit_gradient(device):
    """The constant 1.0 as a 0-d fp32 tensor on `device`, one object per device, never written.  `loss.backward()` makes
    autograd fill a fresh ones_like(loss) in every step; `loss.backward(vmlmf_amd.unit_gradient(loss.device))` is the
    same gradient without that launch, and the fused criteria recognise this very tensor as "scale 1" and return
    the gradient their forward kernel already wrote instead of launching a backward kernel."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    t = _UNIT.get(device)
    if t is None:
        t = torch.ones((), device=device, dtype=torch.float32)
        _UNIT[device] = t
    return t


class CrossEntropyFn(torch.autograd.Function):
    """Mean cross-entropy of (B, C) logits against int64 targets, forward and backward in one launch each (and no
    backward launch at all when the incoming gradient is unit_gradient(device))."""

    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        _require_hip(logits, "logits")
        logits = logits.contiguous()
        target = target.contiguous()
        B, C = logits.shape
        dev = logits.device
        stats = torch.empty(B + 2, device=dev, dtype=torch.float32)     # loss | nvalid | lse[B]
        dz_unit = torch.empty_like(logits) if ctx.needs_input_grad[0] else None
        stream = _lib.raw_stream(dev)
        with _lib.on_device(dev):
            _lib.check(_lib.lib().vmlmf_ce_forward(B, C, _ptr(logits), _ptr(target), int(ignore_index),
                                                   stats.data_ptr(), stats.data_ptr() + 8, stats.data_ptr() + 4,
                                                   _ptr(dz_unit), stream))
        ctx.save_for_backward(logits, target, stats, *([dz_unit] if dz_unit is not None else []))
        ctx.ignore_index = int(ignore_index)
        ctx.leaf_logits = logits.grad_fn is None
        return stats[0]

    @staticmethod
    def backward(ctx, dloss):
        logits, target, stats = ctx.saved_tensors[:3]
        unit = _UNIT.get(logits.device)
        if unit is not None and dloss.data_ptr() == unit.data_ptr() and len(ctx.saved_tensors) == 4:
            # d(loss) is the package's constant one: forward wrote this.  Logits produced by another node (the classifier
            # head) just pass it on.  When the logits are a LEAF, AccumulateGrad adopts what it is handed as .grad (views
            # included), and in-place work on that gradient (zero_grad(set_to_none=False), clip_grad_norm_, a further
            # backward through a retained graph) would then write into the saved buffer: such callers get a copy.
            dz = ctx.saved_tensors[3]
            return (dz.clone() if ctx.leaf_logits else dz), None, None
        B, C = logits.shape
        dloss = dloss.contiguous()
        dz = torch.empty_like(logits)
        stream = _lib.raw_stream(logits.device)
        with _lib.on_device(logits.device):
            _lib.check(_lib.lib().vmlmf_ce_backward(B, C, _ptr(logits), _ptr(target), ctx.ignore_index,
                                                    stats.data_ptr() + 8, stats.data_ptr() + 4, _ptr(dloss),
                                                    _ptr(dz), stream))
        return dz, None, None


def cross_entropy(input, target, ignore_index=-100):
    """Drop-in for torch.nn.functional.cross_entropy(input, target) with the default arguments (mean reduction,
    class-index targets, no weights, no label smoothing) — the criterion of the reference's training loop
    (V/src/train_test/train.py:58-65).  Classifier-sized (B, C) fp32 logits on a HIP device take the fused
    kernels; anything else goes to the library op."""
    if (input.is_cuda and input.dim() == 2 and input.dtype == torch.float32 and target.dtype == torch.int64
            and target.dim() == 1 and input.numel() <= 65536):
        ops = torch_ops()
        if ops is not None:
            return ops.cross_entropy(input, target, int(ignore_index), unit_gradient(input.device))
        return CrossEntropyFn.apply(input, target, ignore_index)
    return torch.nn.functional.cross_entropy(input, target, ignore_index=ignore_index)


class CrossEntropyLoss(torch.nn.Module):
    """nn.CrossEntropyLoss() with default arguments, on the fused kernels (see cross_entropy)."""

    def __init__(self, ignore_index=-100):
        super().__init__()
        self.ignore_index = ignore_index

    def forward(self, input, target):
        return cross_entropy(input, target, self.ignore_index)


class NllLossFn(torch.autograd.Function):
    """loss = mean over rows of -log softmax(scores)[row, y[row]] * batch_size (lm_test.py:140-153), one read of
    `scores` forward, one read and one write backward."""

    @staticmethod
    def forward(ctx, scores, y, batch_size):
        ctx.set_materialize_grads(False)
        _require_hip(scores, "scores")
        scores = scores.contiguous()
        yrow = y.reshape(-1).contiguous()
        R, V = scores.shape
        scale = float(batch_size) / float(R)
        stats = torch.empty(1 + 2 * R, device=scores.device, dtype=torch.float32)   # loss | lse | rowloss
        stream = _lib.raw_stream(scores.device)
        base = stats.data_ptr()
        with _lib.on_device(scores.device):
            _lib.check(_lib.lib().vmlmf_nll_forward(R, V, _ptr(scores), _ptr(yrow), scale, base, base + 4,
                                                    base + 4 * (1 + R), stream))
        ctx.save_for_backward(scores, yrow, stats)
        ctx.scale = scale
        return stats[0]

    @staticmethod
    def backward(ctx, dloss):
        if dloss is None:
            return None, None, None
        scores, yrow, stats = ctx.saved_tensors
        R, V = scores.shape
        dloss = dloss.contiguous()
        dz = torch.empty_like(scores)
        stream = _lib.raw_stream(scores.device)
        with _lib.on_device(scores.device):
            _lib.check(_lib.lib().vmlmf_nll_backward(R, V, _ptr(scores), _ptr(yrow), ctx.scale, stats.data_ptr() + 4,
                                                     _ptr(dloss), _ptr(dz), stream))
        return dz, None, None


def nll_loss(scores, y):
    """Drop-in for the reference's language-model loss `nll_loss(scores, y)` (V/src/train_test/lm_test.py:140-153):
    scores (T*B, V) fp32, y (T, B) int64 -> scalar, scaled by batch_size as there.  On a HIP device the fused
    kernels run (stable around the row maximum: where the reference's plain exp overflows, this does not);
    CPU tensors take the reference's own formulation in stock ops."""
    batch_size = y.size(1)
    if scores.is_cuda and scores.dim() == 2 and scores.dtype == torch.float32 and y.dtype == torch.int64:
        return NllLossFn.apply(scores, y, batch_size)
    expscores = scores.exp()
    probabilities = expscores / expscores.sum(1, keepdim=True)
    answerprobs = probabilities[range(len(y.reshape(-1))), y.reshape(-1)]
    return torch.mean(-torch.log(answerprobs) * batch_size)


class LinearNllFn(torch.autograd.Function):
    """Vocabulary projection + log-softmax + NLL of the LM loop (Linear, vmlmf_lm.py:355-358; nll_loss, lm_test.py:140-153)
    WITHOUT the (T*B, V) score tensor in HBM: the rows are cut into chunks whose scores (chunk x V) live in one reused buffer
    that stays in the memory-side cache; every chunk is one library GEMM + the fused NLL kernel.  The backward recomputes a
    chunk's scores (a fourth GEMM-sized product) before it forms dscores, dh and accumulates dW, db.
    Measured at config E's shape (tools/bench_lm_head.py, profiles/r03_lm_head_fusion.jsonl): the forward alone is faster than
    GEMM + loss over the full tensor; forward + backward is slower (the recomputed product costs more than the three round
    trips of 358 MB it saves) - so vmlmf_amd.linear_nll takes this path only when no gradient is needed."""

    @staticmethod
    def forward(ctx, h, w, b, y, chunk_rows):
        _require_hip(h, "h")
        h2 = h.reshape(-1, h.shape[-1]).contiguous()
        R, V = h2.shape[0], w.shape[0]
        batch_size = y.size(1)
        yrow = y.reshape(-1).contiguous()
        scale = float(batch_size) / float(R)
        C = min(int(chunk_rows), R)
        buf = torch.empty((C, V), device=h.device, dtype=torch.float32)
        stats = torch.empty(1 + 2 * R, device=h.device, dtype=torch.float32)       # (chunk loss) | lse | rowloss
        losses = torch.empty((R + C - 1) // C, device=h.device, dtype=torch.float32)
        stream = _lib.raw_stream(h.device)
        lib = _lib.lib()
        with _lib.on_device(h.device):
            for ci, r0 in enumerate(range(0, R, C)):
                n = min(C, R - r0)
                sc = buf[:n]
                torch.addmm(b, h2[r0:r0 + n], w.t(), out=sc)
                _lib.check(lib.vmlmf_nll_forward(n, V, _ptr(sc), yrow.data_ptr() + 8 * r0, scale, losses.data_ptr() + 4 * ci,
                                                 stats.data_ptr() + 4 * (1 + r0), stats.data_ptr() + 4 * (1 + R + r0), stream))
        ctx.save_for_backward(h2, w, b, yrow, stats)
        ctx.scale, ctx.C, ctx.hshape = scale, C, h.shape
        return losses.sum()

    @staticmethod
    def backward(ctx, dloss):
        h2, w, b, yrow, stats = ctx.saved_tensors
        R, V, C = h2.shape[0], w.shape[0], ctx.C
        dloss = dloss.contiguous()
        buf = torch.empty((C, V), device=h2.device, dtype=torch.float32)
        dzb = torch.empty((C, V), device=h2.device, dtype=torch.float32)
        dh = torch.empty_like(h2)
        dw = torch.zeros_like(w)
        db = torch.zeros_like(b)
        stream = _lib.raw_stream(h2.device)
        lib = _lib.lib()
        with _lib.on_device(h2.device):
            for r0 in range(0, R, C):
                n = min(C, R - r0)
                sc, dz = buf[:n], dzb[:n]
                torch.addmm(b, h2[r0:r0 + n], w.t(), out=sc)                     # the scores again
                _lib.check(lib.vmlmf_nll_backward(n, V, _ptr(sc), yrow.data_ptr() + 8 * r0, ctx.scale,
                                                  stats.data_ptr() + 4 * (1 + r0), _ptr(dloss), _ptr(dz), stream))
                torch.mm(dz, w, out=dh[r0:r0 + n])
                dw.addmm_(dz.t(), h2[r0:r0 + n])
                db.add_(dz.sum(0))
        return dh.view(ctx.hshape), dw, db, None, None


# ---- the LM head for TRAINING: projection + loss with the gradient of the scores formed in place -------------------------------
# (verdict r3 item 5.)  scores = h W^T is one library GEMM (fp32: rocBLAS / hipBLASLt run it at 120-140 TFLOP/s, 0.77-0.9 of the
# fp32 matrix peak - profiles/r03_lm_head_fusion.jsonl: a hand-written kernel would have to match that to pay); everything
# between the three GEMMs is this package's: vmlmf_nll_forward_grad adds the bias, takes the loss and overwrites the scores
# with their own gradient in ONE pass (the bias gradient falls out as column sums), so the backward's two GEMMs read the
# gradient where the forward left it - no second 358 MB matrix, no separate bias reduction, no backward loss kernel.
# Which library and which operand layout serves each of the three products best differs by 10-25 % (tools/experiments/
# gemm_probe.py: dW as dz^T h runs at 96-104 TFLOP/s, as (h^T dz)^T at 119 on rocBLAS); the forms are timed once per shape on the
# device and the fastest is kept (VMLMF_HEAD_TUNE=0: the first form of each list).
_HEAD_FORMS = {}


def _blas_libs():
    libs = [None]
    if hasattr(torch.backends.cuda, "preferred_blas_library"):
        libs += ["hipblaslt", "cublas"]          # ("cublas" is rocBLAS on ROCm)
    return libs


class _with_blas:
    def __init__(self, lib):
        self.lib, self.prev = lib, None

    def __enter__(self):
        if self.lib is not None:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                self.prev = torch.backends.cuda.preferred_blas_library()
                torch.backends.cuda.preferred_blas_library(self.lib)

    def __exit__(self, *exc):
        if self.prev is not None:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                torch.backends.cuda.preferred_blas_library(self.prev)
        return False


def transposed(m):
    """m^T as a dense tensor (vmlmf_transpose: LDS tiles; torch's strided copy takes 43 us for the 26 MB of the PTB head's dW)."""
    if not (m.is_cuda and m.dtype == torch.float32 and m.dim() == 2 and m.is_contiguous()):
        return m.t().contiguous()
    out = torch.empty((m.shape[1], m.shape[0]), device=m.device, dtype=torch.float32)
    with _lib.on_device(m.device):
        _lib.check(_lib.lib().vmlmf_transpose(m.shape[0], m.shape[1], _ptr(m), _ptr(out), _lib.raw_stream(m.device)))
    return out


def _head_products():
    """name -> [(label, fn)] candidate forms; every fn returns a contiguous result."""
    return {
        "fwd": [("mm(h, W^T)", lambda h, w: torch.mm(h, w.t()))],
        "dh": [("mm(dz, W)", lambda dz, w: torch.mm(dz, w))],
        "dw": [("mm(dz^T, h)", lambda dz, h: torch.mm(dz.t(), h)),
               ("mm(h^T, dz)^T", lambda dz, h: transposed(torch.mm(h.t(), dz)))],
    }


def head_forms(R, H, V, device):
    """The (library, form) chosen for each of the LM head's three GEMMs at this shape; timed on first use."""
    import os
    key = (R, H, V, device.index)
    hit = _HEAD_FORMS.get(key)
    if hit is not None:
        return hit
    prods = _head_products()
    tune = os.environ.get("VMLMF_HEAD_TUNE", "1") != "0" and not torch.cuda.is_current_stream_capturing()
    chosen = {k: (None, v[0][0], v[0][1], None) for k, v in prods.items()}
    if tune:
        # with the user's TunableOp on (PYTORCH_TUNABLEOP_ENABLED=1) the solutions of both libraries are candidates of every call:
        # no preferred-library switch, under which the other library's tuned solutions are not found again
        tunable = getattr(torch.cuda, "tunable", None)
        libs = [None] if (tunable is not None and tunable.is_enabled()) else _blas_libs()
        with torch.no_grad():
            a = {"fwd": (torch.randn(R, H, device=device), torch.randn(V, H, device=device)),
                 "dh": (torch.randn(R, V, device=device), torch.randn(V, H, device=device)),
                 "dw": (torch.randn(R, V, device=device), torch.randn(R, H, device=device))}
            for name, forms in prods.items():
                best = None
                for lib in libs:
                    for label, fn in forms:
                        try:
                            with _with_blas(lib):
                                fn(*a[name])
                                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                                e0.record()
                                for _ in range(3):
                                    fn(*a[name])
                                e1.record()
                            e1.synchronize()
                            ms = e0.elapsed_time(e1) / 3
                        except RuntimeError:
                            continue
                        if best is None or ms < best[3]:
                            best = (lib, label, fn, ms)
                if best is not None:
                    chosen[name] = best
            del a
    if tune or not torch.cuda.is_current_stream_capturing():
        _HEAD_FORMS[key] = chosen
    return chosen


def _run_form(form, *args):
    lib, _, fn, _ = form
    with _with_blas(lib):
        return fn(*args)


class LmHeadLossFn(torch.autograd.Function):
    """loss = nll_loss(Linear(h), y) (vmlmf_lm.py:355-358 + lm_test.py:140-153) for training; see the block comment above."""

    @staticmethod
    def forward(ctx, h, w, b, y):
        _require_hip(h, "h")
        _require_hip(w, "fc.w")
        H = h.shape[-1]
        h2 = h.reshape(-1, H).contiguous()
        w = w.contiguous()
        R, V = h2.shape[0], w.shape[0]
        forms = head_forms(R, H, V, h.device)
        scores = _run_form(forms["fwd"], h2, w)                       # (R, V), no bias
        yrow = y.reshape(-1).contiguous()
        scale = float(y.size(1)) / float(R)
        stats = torch.empty(1 + R, device=h.device, dtype=torch.float32)          # loss | rowloss
        dbias = torch.empty(V, device=h.device, dtype=torch.float32) if b is not None else None
        lib = _lib.lib()
        scratch = _workspace(h.device, 4 * lib.vmlmf_nll_grad_scratch_floats(R, V))
        with _lib.on_device(h.device):
            _lib.check(lib.vmlmf_nll_forward_grad(R, V, _ptr(scores), _ptr(None if b is None else b.contiguous()), _ptr(yrow), scale,
                                                  stats.data_ptr(), stats.data_ptr() + 4, _ptr(dbias), scratch.data_ptr(),
                                                  _lib.raw_stream(h.device)))
        ctx.save_for_backward(h2, w, scores, *([dbias] if dbias is not None else []))
        ctx.hshape, ctx.forms, ctx.has_b = h.shape, forms, b is not None
        return stats[0]

    @staticmethod
    def backward(ctx, dloss):
        h2, w, dz = ctx.saved_tensors[:3]
        dbias = ctx.saved_tensors[3] if ctx.has_b else None
        unit = _UNIT.get(dz.device)
        scaled = not (unit is not None and dloss.data_ptr() == unit.data_ptr())
        dh = _run_form(ctx.forms["dh"], dz, w) if ctx.needs_input_grad[0] else None
        dw = _run_form(ctx.forms["dw"], dz, h2) if ctx.needs_input_grad[1] else None
        db = dbias if (ctx.has_b and ctx.needs_input_grad[2]) else None
        if scaled:                                   # d(loss) is not the package's constant one: the stored gradient is for 1
            dh = None if dh is None else dh.mul_(dloss)
            dw = None if dw is None else dw.mul_(dloss)
            db = None if db is None else db * dloss
        return (None if dh is None else dh.view(ctx.hshape)), dw, db, None


def lm_head_loss(h, weight, bias, y):
    """Training form of nll_loss(Linear(h), y): h (T, B, H), weight (V, H), bias (V), y (T, B) int64 -> scalar loss whose
    backward reads the scores' gradient where the forward left it.  Call loss.backward(vmlmf_amd.unit_gradient(device)) to spare
    the three scalings a foreign d(loss) tensor costs.  Vocabulary widths the in-register loss does not cover (not a multiple
    of four, beyond 12288) and CPU tensors take projection + nll_loss."""
    V = weight.shape[0]
    if h.is_cuda and h.dtype == torch.float32 and y.dtype == torch.int64 and V % 4 == 0 and V <= 12288:
        return LmHeadLossFn.apply(h, weight, bias, y)
    return nll_loss(torch.addmm(bias, h.reshape(-1, h.shape[-1]), weight.t()), y)


class EmbedFn(torch.autograd.Function):
    """x = w[tokens] (Embed, vmlmf_lm.py:46-48) whose backward is the package's deterministic scatter-add (vmlmf_embed_backward:
    every row of the table's gradient summed in position order, no float atomics, the zero fill in the same pass)."""

    @staticmethod
    def forward(ctx, w, tokens):
        ctx.save_for_backward(tokens)
        ctx.wshape = w.shape
        return torch.nn.functional.embedding(tokens, w)

    @staticmethod
    def backward(ctx, dy):
        (tokens,) = ctx.saved_tensors
        V, H = ctx.wshape
        dy2 = dy.reshape(-1, H).contiguous()
        R = dy2.shape[0]
        tok = tokens.reshape(-1).contiguous()
        dw = torch.empty((V, H), device=dy.device, dtype=torch.float32)
        lib = _lib.lib()
        nbytes = lib.vmlmf_embed_backward_scratch_bytes(R, V)
        if torch.cuda.is_current_stream_capturing():
            scratch = torch.empty(nbytes, device=dy.device, dtype=torch.uint8)
        else:
            key = ("embed", dy.device.index, _lib.raw_stream(dy.device).value)
            scratch = _WORKSPACE.get(key)
            if scratch is None or scratch.numel() < nbytes:
                scratch = _WORKSPACE[key] = torch.empty(nbytes, device=dy.device, dtype=torch.uint8)
        with _lib.on_device(dy.device):
            _lib.check(lib.vmlmf_embed_backward(R, H, V, _ptr(tok), _ptr(dy2), _ptr(dw), scratch.data_ptr(), nbytes,
                                                _lib.raw_stream(dy.device)))
        return dw, None


# ---- dropout of the LM network without mask tensors (C ABI 11: vmlmf_dropout_*; csrc/vmlmf_dropout.h) --------------------------
def dropout_state(device, seed=None):
    """{seed, offset} of the package's dropout generator on `device` (int64[2]).  seed=None: drawn from torch's CPU generator, so
    torch.manual_seed() makes the masks repeatable."""
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        # data-parallel replicas seeded alike (torch.manual_seed(s) on every rank) still draw different masks for their shards
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            seed = (seed + torch.distributed.get_rank() * 0x9E3779B97F4A7C15) & (2 ** 62 - 1)
    return torch.tensor([int(seed), 0], dtype=torch.int64, device=device)


def dropout_advance(state):
    """This forward's snapshot of `state`; the state moves on by one (one tiny launch: a node of a captured step, so every replay
    draws fresh factors)."""
    if not state.is_cuda or state.dtype != torch.int64 or state.numel() != 2:
        raise RuntimeError("vmlmf_amd: the dropout state is two int64 words on a HIP device (dropout_state())")
    snap = torch.empty_like(state)
    with _lib.on_device(state.device):
        _lib.check(_lib.lib().vmlmf_dropout_advance(state.data_ptr(), snap.data_ptr(), _lib.raw_stream(state.device)))
    return snap


class DropoutFn(torch.autograd.Function):
    """y = x * factor(snapshot, site) over the rows of x (..., H): nn.Dropout(p) in one launch per direction, the factors
    regenerated in the backward (no mask is kept)."""

    @staticmethod
    def forward(ctx, x, p, snap, site):
        _require_hip(x, "input")
        x = x.contiguous()
        y = torch.empty_like(x)
        H = x.shape[-1]
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().vmlmf_dropout_apply(x.numel() // H, H, _ptr(x), _ptr(y), float(p), snap.data_ptr(), int(site),
                                                      _lib.raw_stream(x.device)))
        ctx.save_for_backward(snap)
        ctx.p, ctx.site = float(p), int(site)
        return y

    @staticmethod
    def backward(ctx, dy):
        (snap,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        H = dy.shape[-1]
        with _lib.on_device(dy.device):
            _lib.check(_lib.lib().vmlmf_dropout_apply(dy.numel() // H, H, _ptr(dy), _ptr(dx), ctx.p, snap.data_ptr(), ctx.site,
                                                      _lib.raw_stream(dy.device)))
        return dx, None, None, None


def dropout(x, p, snap, site):
    """nn.Dropout(p)(x) in training mode with the package's generator: x (..., H) float32 on a HIP device."""
    if p <= 0.0:
        return x
    if x.dtype != torch.float32:
        raise RuntimeError("vmlmf_amd.dropout: float32 activations")
    return DropoutFn.apply(x, p, snap, site)


def dropout_factors(rows, H, p, snap, site, layer_desc=None):
    """The (rows, H) factors (0 or 1/(1-p)) the kernels apply for (snapshot, site) - for a layer whose launches apply them
    themselves pass its descriptor (vmlmf_dropout_factors).  The parity tests multiply a CPU restatement by this tensor."""
    out = torch.empty((rows, H), device=snap.device, dtype=torch.float32)
    with _lib.on_device(snap.device):
        _lib.check(_lib.lib().vmlmf_dropout_factors(None if layer_desc is None else ctypes.byref(layer_desc), rows, H, float(p),
                                                     snap.data_ptr(), int(site), out.data_ptr(), _lib.raw_stream(snap.device)))
    return out


class EmbedDropFn(torch.autograd.Function):
    """dropout(w[tokens]) of vmlmf_lm.py:434-435 as one gather launch; backward: EmbedFn's scatter-add with the factors regenerated
    on the fly."""

    @staticmethod
    def forward(ctx, w, tokens, p, snap, site):
        V, H = w.shape
        tok = tokens.reshape(-1).contiguous()
        out = torch.empty(tuple(tokens.shape) + (H,), device=w.device, dtype=torch.float32)
        with _lib.on_device(w.device):
            _lib.check(_lib.lib().vmlmf_embed_dropout_forward(tok.numel(), H, V, _ptr(tok), _ptr(w), _ptr(out), float(p), snap.data_ptr(),
                                                              int(site), _lib.raw_stream(w.device)))
        ctx.save_for_backward(tok, snap)
        ctx.wshape, ctx.p, ctx.site = w.shape, float(p), int(site)
        return out

    @staticmethod
    def backward(ctx, dy):
        tok, snap = ctx.saved_tensors
        V, H = ctx.wshape
        dy2 = dy.reshape(-1, H).contiguous()
        R = dy2.shape[0]
        dw = torch.empty((V, H), device=dy.device, dtype=torch.float32)
        lib = _lib.lib()
        nbytes = lib.vmlmf_embed_backward_scratch_bytes(R, V)
        scratch = _embed_scratch(dy.device, nbytes)
        with _lib.on_device(dy.device):
            _lib.check(lib.vmlmf_embed_dropout_backward(R, H, V, _ptr(tok), _ptr(dy2), _ptr(dw), scratch.data_ptr(), nbytes, ctx.p,
                                                        snap.data_ptr(), ctx.site, _lib.raw_stream(dy.device)))
        return dw, None, None, None, None


def _embed_scratch(dev, nbytes):
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(nbytes, device=dev, dtype=torch.uint8)
    key = ("embed", dev.index, _lib.raw_stream(dev).value)
    scratch = _WORKSPACE.get(key)
    if scratch is None or scratch.numel() < nbytes:
        scratch = _WORKSPACE[key] = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    return scratch


def _embed_covered(weight, tokens):
    return (weight.is_cuda and weight.dtype == torch.float32 and weight.dim() == 2 and weight.shape[1] <= 1024 and tokens.dtype == torch.int64
            and weight.is_contiguous() and weight.shape[0] * ((tokens.numel() + 31) // 32) * 4 <= (64 << 20))


def embedding_dropout(weight, tokens, p, snap, site=0):
    """dropout(weight[tokens]) - the first two lines of Model.forward (vmlmf_lm.py:434-435) - as one launch per direction on HIP fp32
    tables up to 1024 wide; embedding() followed by dropout() otherwise."""
    if p <= 0.0:
        return embedding(weight, tokens)
    if _embed_covered(weight, tokens):
        return EmbedDropFn.apply(weight, tokens, p, snap, site)
    return dropout(embedding(weight, tokens), p, snap, site)


def embedding(weight, tokens):
    """weight[tokens] with the package's backward on HIP fp32 tables up to 1024 wide and 64 MB of position bits; the stock op otherwise."""
    if (weight.is_cuda and weight.dtype == torch.float32 and weight.dim() == 2 and weight.shape[1] <= 1024 and tokens.dtype == torch.int64
            and weight.is_contiguous() and weight.shape[0] * ((tokens.numel() + 31) // 32) * 4 <= (64 << 20)):
        return EmbedFn.apply(weight, tokens)
    return weight[tokens]


def linear_nll(h, weight, bias, y, chunk_rows=2048, fused=None):
    """loss = nll_loss(Linear(h), y) of the LM loop (vmlmf_lm.py:355-358 + lm_test.py:140-153): h (T, B, H), weight (V, H), bias
    (V), y (T, B) int64.  fused=None: the chunked form without the score tensor when no gradient is needed (evaluation,
    perplexity), GEMM + loss over the full tensor when one is (measured: LinearNllFn's docstring); True / False force one."""
    need_grad = torch.is_grad_enabled() and (h.requires_grad or weight.requires_grad or bias.requires_grad)
    use = (not need_grad) if fused is None else bool(fused)
    if use and h.is_cuda:
        return LinearNllFn.apply(h, weight, bias, y, chunk_rows)
    return nll_loss(torch.addmm(bias, h.reshape(-1, h.shape[-1]), weight.t()), y)
