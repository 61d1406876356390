"""Optimizer steps of the reference's two training loops as single launches over all parameter tensors.

  Adam            drop-in for torch.optim.Adam(model.parameters(), lr=...)      V/src/train_test/train.py:47,65
  clip_sgd_step   clip_grad_norm_(params, max_norm) + `param -= lr * param.grad` V/src/train_test/lm_test.py:203-209

The models are 10-20 small tensors: the stock optimizer costs ~2 ms of dispatcher time per step at the headline
shape, ten times the forward + backward it follows.  Here one kernel walks every tensor (pointers travel in the
kernel arguments); the step counters live on the device, so a whole training step can sit in one hipGraph.
SURVEY.md section 8f ("next" row).  HIP tensors only; anything else raises (no CPU path).
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib


def _tensor_lists(pairs, offsets, step_index=None):
    """[(param, grad)] -> ctypes TensorList structs of at most MAX_TENSORS entries each."""
    out = []
    for lo in range(0, len(pairs), _lib.MAX_TENSORS):
        tl = _lib.TensorList()
        chunk = pairs[lo:lo + _lib.MAX_TENSORS]
        for i, (p, g) in enumerate(chunk):
            tl.param[i], tl.grad[i] = p.data_ptr(), g.data_ptr()
            tl.numel[i], tl.state_offset[i] = p.numel(), offsets[lo + i]
            tl.step_index[i] = 0 if step_index is None else step_index[lo + i]
        tl.count = len(chunk)
        out.append(tl)
    return out


def _check(p, g):
    if not (p.is_cuda and g.is_cuda):
        raise RuntimeError("vmlmf_amd.optim: parameters and gradients must live on a HIP device (no CPU path)")
    if p.dtype != torch.float32 or g.dtype != torch.float32 or not p.is_contiguous():
        raise RuntimeError("vmlmf_amd.optim: dense float32 parameters only")


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (betas, eps, L2 weight_decay; no amsgrad / maximize), one launch per step and
    parameter group.  state[p] holds 'step', 'exp_avg', 'exp_avg_sq' like the stock optimizer (the moments are
    views of two flat buffers per group).
    Non-finite gradients of the VMLMF layers never reach the parameters: the library's backward marks a per-device health word
    when it writes an Inf / NaN parameter gradient (a launch that gave up a bounded wait, VMLMF_E_PROTOCOL, leaves NaN gradients, and
    a replayed hipGraph cannot ask the host), this optimizer's tick launch reads it and skips the whole step; skipped_steps()
    counts.  vmlmf_amd._lib.tune("adam_guard", 2) scans every gradient instead (any source, one more launch); guard=False: the
    unguarded launch."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, guard=True):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._guarded = bool(guard)

    def skipped_steps(self):
        """Optimizer steps the device-side gate refused because a gradient was not finite (synchronises)."""
        g = getattr(self, "_guard", None)
        return 0 if g is None else int(g[_lib.GUARD_SKIPPED].item())

    def _group_state(self, gi, group):
        """Flat moment buffers covering every parameter of the group (allocated at the first step; kept out of
        param_groups so that state_dict() stays what torch.optim.Adam's is)."""
        if not hasattr(self, "_flat"):
            self._flat = {}
        gs = self._flat.get(gi)
        if gs is None:
            ps = group["params"]
            dev = ps[0].device
            total, offs, sidx = 0, {}, {}
            for k, p in enumerate(ps):
                offs[p], sidx[p] = total, k
                total += p.numel()
            gs = dict(m=torch.zeros(total, device=dev), v=torch.zeros(total, device=dev),
                      steps=torch.zeros(len(ps), device=dev), offs=offs, sidx=sidx)
            # ONE guard block for the whole optimizer: the verdict on a step's gradients is taken once, by the first launch of the
            # step, and holds for every tensor list and parameter group of that step
            if getattr(self, "_guarded", True) and getattr(self, "_guard", None) is None:
                self._guard = torch.zeros(_lib.GUARD_WORDS, device=dev, dtype=torch.int32)
            self._flat[gi] = gs
            for p in ps:
                o = offs[p]
                new = dict(step=gs["steps"][sidx[p]], exp_avg=gs["m"][o:o + p.numel()].view_as(p),
                           exp_avg_sq=gs["v"][o:o + p.numel()].view_as(p))
                # state that exists already (load_state_dict of a checkpoint, also one written by torch.optim.Adam)
                # moves into the flat buffers the kernel reads
                old = self.state.get(p)
                if old:
                    for k in ("exp_avg", "exp_avg_sq"):
                        if k in old:
                            new[k].copy_(old[k].to(device=dev, dtype=torch.float32).view_as(p))
                    if "step" in old:
                        new["step"].fill_(float(old["step"]))
                self.state[p] = new
        return gs

    def load_state_dict(self, state_dict):
        """As torch.optim.Optimizer.load_state_dict; the flat buffers are rebuilt from the loaded moments and step
        counts at the next step()."""
        super().load_state_dict(state_dict)
        # converted NOW, not at the next step(): torch hands the checkpoint's own tensors through when device and dtype
        # already match, so a lazily read state would follow whatever the checkpoint's owner does to them meanwhile
        self._flat = {}
        for gi, group in enumerate(self.param_groups):
            if any(self.state.get(p) for p in group["params"]):
                self._group_state(gi, group)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.lib()
        # every launch of the step first (tensor lists of every group), then FIRST on the first and LAST on the last of them: one
        # verdict on the step's gradients for all of them (include/vmlmf_hip.h: vmlmf_adam_step_ex)
        calls = []
        for gi, group in enumerate(self.param_groups):
            live = [p for p in group["params"] if p.grad is not None]
            if not live:
                continue
            gs = self._group_state(gi, group)
            pairs = []
            for p in live:
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                _check(p, g)
                pairs.append((p, g))
            for tl in _tensor_lists(pairs, [gs["offs"][p] for p in live], [gs["sidx"][p] for p in live]):
                calls.append((tl, gs, group, live))
        guard = getattr(self, "_guard", None)
        for ci, (tl, gs, group, live) in enumerate(calls):
            dev = live[0].device
            b1, b2 = group["betas"]
            flags = (_lib.ADAM_FIRST if ci == 0 else 0) | (_lib.ADAM_LAST if ci == len(calls) - 1 else 0)
            with _lib.on_device(dev):
                _lib.check(lib.vmlmf_adam_step_ex(ctypes.byref(tl), gs["m"].data_ptr(), gs["v"].data_ptr(), gs["steps"].data_ptr(),
                                                  float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                  float(group["weight_decay"]), None if guard is None else guard.data_ptr(), flags,
                                                  _lib.raw_stream(dev)))
            _bump_versions(live)
        return loss


def _bump_versions(params):
    """The kernels write parameters through raw pointers: tell autograd's version counters (kept parameter images,
    functional.PackCache, and any saved-tensor check key on them) that the tensors changed in place."""
    for p in params:
        torch.autograd.graph.increment_version(p)


class _Scratch:
    bufs = {}

    @classmethod
    def get(cls, dev):
        b = cls.bufs.get(dev)
        if b is None:
            b = cls.bufs[dev] = torch.empty(_lib.MAX_TENSORS * 64 + 1, device=dev)
        return b


@torch.no_grad()
def clip_sgd_step(parameters, lr, max_norm):
    """The LM loop's update (lm_test.py:203-209): norm = clip_grad_norm_(parameters, max_norm); p -= lr * p.grad.
    Returns the total gradient norm before clipping as a 0-d device tensor (what clip_grad_norm_ returns).
    A norm that is not finite (NaN gradients of a launch that gave up a bounded wait; an overflow) skips the step on the device:
    parameters and gradients keep their values and the returned norm says so."""
    live = [p for p in parameters if p.grad is not None]
    if not live:
        return torch.zeros(())
    if len(live) > _lib.MAX_TENSORS:
        raise RuntimeError(f"clip_sgd_step: at most {_lib.MAX_TENSORS} tensors (the norm spans all of them)")
    for p in live:
        _check(p, p.grad)
        if not p.grad.is_contiguous():
            raise RuntimeError("clip_sgd_step scales the gradients in place: they must be contiguous")
    dev = live[0].device
    scratch = _Scratch.get(dev)
    norm = torch.empty((), device=dev)
    tl = _tensor_lists([(p, p.grad) for p in live], [0] * len(live))[0]
    stream = _lib.raw_stream(dev)
    with _lib.on_device(dev):
        _lib.check(_lib.lib().vmlmf_sgd_clip_step(ctypes.byref(tl), float(lr), float(max_norm), norm.data_ptr(),
                                                  scratch.data_ptr(), stream))
    _bump_versions(live)
    return norm
