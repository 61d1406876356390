"""One training step (forward + loss + backward + optimizer) captured once into a hipGraph and replayed.

Eager PyTorch needs about 0.36 ms of host time to enqueue the ~20 launches of a step at the UCI-HAR shape, the GPU
0.2 ms to run them: the reference's loop (V/src/train_test/train.py:58-65) is host-bound on an MI355X unless the
step is replayed from a graph.  The kernels, the fused loss and vmlmf_amd.optim.Adam are all capture-safe (no
host synchronisation, device-side step counters).

    step = GraphedTrainStep(model, vmlmf_amd.cross_entropy, vmlmf_amd.optim.Adam(model.parameters(), lr), x0, t0)
    for data, target in loader:
        loss = step(data.to(dev), target.to(dev))        # 0-d device tensor, valid until the next call
"""
from __future__ import annotations

import gc

import torch

from .functional import unit_gradient


class GraphedTrainStep:
    def __init__(self, model, criterion, optimizer, example_input, example_target, warmup=3):
        """The `warmup` steps are real training steps on the example batch (they create the optimizer state and
        settle the allocator); the capture itself executes nothing."""
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
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):          # allocator warm-up and optimizer state, outside the capture
                self._body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        gc.collect()                         # no autograd graph of the warm-up may outlive this point
        self.graph = torch.cuda.CUDAGraph()
        model.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.loss = self._body()

    def _body(self):
        self.model.zero_grad(set_to_none=True)
        loss = self.criterion(self.model(self.x), self.t)
        loss.backward(self._one if loss.dim() == 0 and loss.dtype == self._one.dtype else None)
        self.optimizer.step()
        return loss.detach()

    def __call__(self, x, target):
        self.x.copy_(x, non_blocking=True)
        self.t.copy_(target, non_blocking=True)
        self.graph.replay()
        return self.loss
