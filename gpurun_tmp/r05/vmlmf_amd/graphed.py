"""One training step (forward + loss + backward + optimizer) captured once into a hipGraph and replayed.

Eager PyTorch needs about 0.36 ms of host time to enqueue the ~20 launches of a step at the UCI-HAR shape, the GPU
0.2 ms to run them: the reference's loop (V/src/train_test/train.py:58-65) is host-bound on an MI355X unless the
step is replayed from a graph.  The kernels, the fused loss and vmlmf_amd.optim.Adam are all capture-safe (no
host synchronisation, device-side step counters).

    step = GraphedTrainStep(model, vmlmf_amd.cross_entropy, vmlmf_amd.optim.Adam(model.parameters(), lr), x0, t0)
    for data, target in loader:
        loss = step(data.to(dev), target.to(dev))        # 0-d device tensor, valid until the next call

Constructing it leaves the model and the optimizer as they were (the warm-up steps the capture needs run on a
snapshot that is restored).  A batch whose shape differs from the example's (the last, shorter DataLoader batch) runs
the same body eagerly.  Scalars the kernels take by value are frozen in a capture (the learning rate of
vmlmf_amd.optim.Adam): when a group's lr / betas / eps / weight_decay has changed since the capture, the step is captured again.

A launch inside the graph that gives up a bounded wait (include/vmlmf_hip.h: VMLMF_E_PROTOCOL - a GPU shared with other
processes can starve the riding weight-gradient workers) leaves NaN gradients.  vmlmf_amd.optim.Adam's device-side gate skips
the update of such a step, so the parameters stay intact; the host reads the library's status word in front of every replay
(a host memory read, no GPU call), and when an earlier replay failed - or the library's kernel selection changed since the
capture (vmlmf_tune_generation) - the step is captured again: the library has switched to the stand-alone weight-gradient
kernel by then, and the poisoned launch is not replayed any more.  `failed_steps` / `recaptures` count.  Replays are
asynchronous: the host learns of a failed replay when it prepares a later one, and the replays queued in between ran the same
poisoned graph - each of their updates was skipped on the device too, so up to the queue depth of steps can be lost (their count
is `optimizer.skipped_steps()`), never a parameter.  That protection is vmlmf_amd.optim.Adam's (guard=True, the default): another
optimizer would apply the NaN gradients of such a step, which is why the constructor warns when it is handed one while the riding
weight-gradient workers are on.

When the criterion is the package's cross-entropy and the model offers `loss(x, target)` (vmlmf_amd.Net), the step calls that:
the criterion then rides on the forward recurrence's launch (one launch less in each direction).
"""
from __future__ import annotations

import gc

import warnings

import torch

from . import _lib
from . import functional as _F
from .functional import unit_gradient


class GraphedTrainStep:
    def __init__(self, model, criterion, optimizer, example_input, example_target, warmup=3):
        """The `warmup` steps (allocator warm-up, creation of the optimizer state) run on the example batch; parameters
        and optimizer state are put back to their values from before afterwards, so the first call is step 1 of the
        reference loop (train.py:58-65).  The capture itself executes nothing."""
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.x = example_input.clone()
        self.t = example_target.clone()
        dev = self.x.device
        # root gradient of the scalar loss, kept across replays (a bare loss.backward() fills a new ones_like(loss)
        # inside every step: one more launch on the critical path)
        self._one = unit_gradient(dev)
        if warmup < 1:
            raise ValueError("at least one warm-up step: optimizer state must exist before the capture")
        if dev.type != "cuda":
            raise RuntimeError("GraphedTrainStep needs HIP tensors")
        params = [p for group in optimizer.param_groups for p in group["params"]]
        self._params = list(params)
        # the criterion riding on the model's own launches (vmlmf_amd.Net.loss) when it is the package's cross-entropy
        self._fused_loss = hasattr(model, "loss") and (criterion is _F.cross_entropy or isinstance(criterion, _F.CrossEntropyLoss))
        self._ignore_index = getattr(criterion, "ignore_index", -100)
        from .optim import Adam as _Adam
        if not (isinstance(optimizer, _Adam) and getattr(optimizer, "_guarded", True)):
            warnings.warn("vmlmf_amd.GraphedTrainStep: only vmlmf_amd.optim.Adam (guard=True) keeps the NaN gradients of a launch that "
                          "gave up a bounded wait (VMLMF_E_PROTOCOL) away from the parameters inside a replayed graph; with this "
                          "optimizer run with VMLMF_WRIDE=0 on a GPU that is shared with other processes", RuntimeWarning)
        seen = {id(p) for p in params}
        params += [p for p in model.parameters() if id(p) not in seen]
        with torch.no_grad():
            p_before = [p.detach().clone() for p in params]
            s_before = {(id(p), k): (v.detach().clone() if torch.is_tensor(v) else v)
                        for p, st in optimizer.state.items() for k, v in st.items()}
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):          # allocator warm-up and optimizer state, outside the capture
                self._body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        # undo the warm-up IN PLACE (the state tensors must keep their addresses: the capture records them): state
        # that did not exist before goes back to zero, which is what a fresh optimizer starts from
        with torch.no_grad():
            for p, v in zip(params, p_before):
                p.copy_(v)
            for p, st in optimizer.state.items():
                for k, v in st.items():
                    old = s_before.get((id(p), k))
                    if torch.is_tensor(v):
                        v.copy_(old) if torch.is_tensor(old) else v.zero_()
                    elif old is not None:
                        st[k] = old
        torch.cuda.synchronize(dev)
        self._capture()

    def _hyper(self):
        return tuple((float(g["lr"]), tuple(g.get("betas", ())), g.get("eps"), g.get("weight_decay"))
                     for g in self.optimizer.param_groups)

    def _capture(self):
        gc.collect()                         # no autograd graph of earlier steps may outlive this point
        self.graph = torch.cuda.CUDAGraph()
        self.model.zero_grad(set_to_none=True)
        self._captured_hyper = self._hyper()
        self._captured_generation = _lib.lib().vmlmf_tune_generation()
        with torch.cuda.graph(self.graph):
            self.loss = self._body()

    def _body(self, x=None, t=None):
        self.model.zero_grad(set_to_none=True)
        xin, tin = self.x if x is None else x, self.t if t is None else t
        if self._fused_loss:
            loss = self.model.loss(xin, tin, ignore_index=self._ignore_index)
        else:
            loss = self.criterion(self.model(xin), tin)
        loss.backward(self._one if loss.dim() == 0 and loss.dtype == self._one.dtype else None)
        self.optimizer.step()
        return loss.detach()

    def __call__(self, x, target):
        if x.shape != self.x.shape or target.shape != self.t.shape:
            return self._body(x, target)     # e.g. the last, shorter batch of a DataLoader: same step, eager launches
        if self._hyper() != self._captured_hyper:
            self._capture()                  # a scheduler changed lr (by-value kernel argument): capture it again
        # did a launch of an EARLIER replay give up a bounded wait?  (the status word is host memory: no GPU call)
        lib = _lib.lib()
        with _lib.on_device(self.x.device):
            rc = lib.vmlmf_check_status()
        if rc == _lib.E_PROTOCOL:
            # every replay queued since the failing one was skipped on the device as well: the optimizer counted them
            skipped = getattr(self.optimizer, "skipped_steps", None)
            lost = int(skipped()) - getattr(self, "_skipped_seen", 0) if callable(skipped) else 1
            self._skipped_seen = getattr(self, "_skipped_seen", 0) + max(lost, 0)
            self.failed_steps = getattr(self, "failed_steps", 0) + max(lost, 1)
            warnings.warn("vmlmf_amd.GraphedTrainStep: " + lib.vmlmf_last_error().decode() + " - that step's update was skipped "
                          "on the device; the step is captured again without the launch that failed", RuntimeWarning)
        elif rc != 0:
            _lib.check(rc)
        if rc != 0 or lib.vmlmf_tune_generation() != self._captured_generation:
            torch.cuda.synchronize(self.x.device)
            self.recaptures = getattr(self, "recaptures", 0) + 1
            self._capture()
        self.x.copy_(x, non_blocking=True)
        self.t.copy_(target, non_blocking=True)
        self.graph.replay()
        # the replayed optimizer kernels changed the parameters on the device; the host-side version counters only moved
        # while the step was captured: move them now, so that nothing keyed on them (kept parameter images) goes stale
        for p in self._params:
            torch.autograd.graph.increment_version(p)
        return self.loss
