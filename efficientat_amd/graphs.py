"""hipGraph capture of a whole training step.

The train-mode plan issues ~500 kernel launches plus ~600 small torch ops per step; eagerly that
is ~35 ms of host time on the reference's single host thread, more than the ~30 ms the GPU needs
for 128 clips - the step is launch-bound.  Shapes are static (fixed batch, 10 s clips), so the
forward + loss + backward + optimizer update are captured once into a hipGraph
(`torch.cuda.CUDAGraph`: our kernels are launched on torch's capture stream through the C ABI, so
they are recorded like any other node) and replayed with one host call per step.

The log-mel front-end stays outside the graph: in train mode its mel basis changes every step
(fmin/fmax jitter drawn on the host, models/preprocess.py:45-55) and is uploaded from host memory.

`GraphedForward` does the same for the eval path (log-mel + network, ~60 launches per batch): the batch is cut into
sub-batches issued on concurrent HIP streams inside the captured graph, so that the latency-bound kernels of one
sub-batch (SE / head GEMMs, kernel tails) overlap with the bandwidth-bound kernels of the other.
"""
import torch
import torch.distributed as dist


def _capture_mode():
    """"thread_local" while a process group exists (its watchdog thread polls events during our capture), else "global"."""
    return "thread_local" if dist.is_available() and dist.is_initialized() else "global"


class GraphedForward:
    """logits, features = fwd(wave): hipGraph replay of eval-mode `model(mel(wave).unsqueeze(1))`.

    Static shapes: every call must pass a (B, L) batch of the example's shape (it is copied into the captured input
    buffer; pass nothing / the buffer itself - `fwd.wave` - to skip the copy).  The weights are folded / packed at
    construction: rebuild the object after a parameter update.  `streams` sub-batches run concurrently."""

    def __init__(self, model, mel, wave_example, streams=2):
        if model.training or mel.training:
            raise RuntimeError("GraphedForward captures the eval path: call model.eval() / mel.eval() first")
        self.model, self.mel = model, mel
        self.wave = wave_example.detach().clone()
        n = max(1, min(int(streams), self.wave.shape[0]))
        self.chunks = list(self.wave.chunk(n))
        self.streams = [torch.cuda.Stream() for _ in self.chunks] if len(self.chunks) > 1 else []
        self.logits = self.features = None
        self._issue()                                    # folds / packs the weights, builds the mel tables
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._issue()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(self.graph, capture_error_mode=_capture_mode()):
            self._issue()

    def _issue(self):
        with torch.no_grad():
            if not self.streams:
                self.logits, self.features = self.model(self.mel(self.wave).unsqueeze(1))
                return
            cur = torch.cuda.current_stream()
            res = []
            for st, wv in zip(self.streams, self.chunks):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    res.append(self.model(self.mel(wv).unsqueeze(1)))
            for st in self.streams:
                cur.wait_stream(st)
            self.logits = torch.cat([r[0] for r in res])
            self.features = torch.cat([r[1] for r in res])

    def replay(self):
        self.graph.replay()

    def __call__(self, wave=None):
        if wave is not None and wave.data_ptr() != self.wave.data_ptr():
            if wave.shape != self.wave.shape:
                raise ValueError(f"GraphedForward was captured for {tuple(self.wave.shape)}, got {tuple(wave.shape)}")
            self.wave.copy_(wave)
        self.graph.replay()
        return self.logits, self.features


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
        # With a process group (data-parallel step: the bucketed RCCL all-reduces are captured too) torch's ProcessGroupNCCL
        # watchdog THREAD polls the events of the warm-up steps' collectives with hipEventQuery.  In the default "global"
        # capture mode any such call from ANY thread while this thread captures fails with
        # hipErrorStreamCaptureUnsupported, the watchdog throws and the process aborts (seen in ~1 of 8 runs of
        # tests/rccl_reducer_case.py: the poll has to land inside the ~0.3 s capture).  "thread_local" restricts only the
        # capturing thread, which is what is wanted here; the device is drained first so that the warm-up's works are
        # complete before the capture begins.
        torch.cuda.synchronize()
        with torch.cuda.graph(self.graph, capture_error_mode=_capture_mode()):
            self.loss = self._eager(zero=False)
        from . import ops
        ops.zero_arena.end("dymn_step")      # (a step arena left open by an unfinished pass must not serve later callers)

    def _eager(self, zero=True):
        if zero:
            self.opt.zero_grad(set_to_none=True)
        logits, _ = self.model(self.x)
        loss = self.loss_fn(logits, self.y)
        loss.backward()
        self.opt.step()
        return loss.detach()

    def __call__(self, x, y):
        # (a producer that wrote straight into the captured input buffers - e.g. `mel(wave, out=step.x)` - passes them back)
        if x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x)
        if y.data_ptr() != self.y.data_ptr():
            self.y.copy_(y)
        self.graph.replay()
        cache = getattr(self.model, "_cache", None)
        if cache is not None:            # a replay updates the weights without bumping their version counters
            cache.invalidate()
        return self.loss
