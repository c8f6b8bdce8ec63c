"""hipGraph capture of a whole training step.

The train-mode plan issues ~500 kernel launches plus ~600 small torch ops per step; eagerly that
is ~35 ms of host time on the reference's single host thread, more than the ~30 ms the GPU needs
for 128 clips - the step is launch-bound.  Shapes are static (fixed batch, 10 s clips), so the
forward + loss + backward + optimizer update are captured once into a hipGraph
(`torch.cuda.CUDAGraph`: our kernels are launched on torch's capture stream through the C ABI, so
they are recorded like any other node) and replayed with one host call per step.

The log-mel front-end stays outside the graph: in train mode its mel basis changes every step
(fmin/fmax jitter drawn on the host, models/preprocess.py:45-55) and is uploaded from host memory.
"""
import torch


class GraphedTrainStep:
    """step(x_mel, target) -> loss, replaying a captured fwd + loss + bwd + optimizer.step().

    `optimizer` must be capturable (e.g. torch.optim.Adam(..., capturable=True)).  `loss_fn(logits,
    target)` must be made of capturable torch ops."""

    def __init__(self, model, optimizer, loss_fn, x_example, y_example, warmup=3):
        self.model, self.opt, self.loss_fn = model, optimizer, loss_fn
        self.x = x_example.clone()
        self.y = y_example.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        self.opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.loss = self._eager(zero=False)

    def _eager(self, zero=True):
        if zero:
            self.opt.zero_grad(set_to_none=True)
        logits, _ = self.model(self.x)
        loss = self.loss_fn(logits, self.y)
        loss.backward()
        self.opt.step()
        return loss.detach()

    def __call__(self, x, y):
        self.x.copy_(x)
        self.y.copy_(y)
        self.graph.replay()
        cache = getattr(self.model, "_cache", None)
        if cache is not None:            # a replay updates the weights without bumping their version counters
            cache.invalidate()
        return self.loss
