"""Data-parallel gradient exchange: bucketed all-reduce (RCCL over xGMI via torch.distributed's
"nccl" backend) overlapped with the backward pass.

What it must equal: the reference's only multi-GPU behaviour, PyTorch-Lightning DDP
(`ex_pl_audioset.py:287-293`): one process per GPU, per-GPU minibatch, local BatchNorm statistics,
gradients averaged over ranks.

The network's backward is one autograd Function that produces parameter gradients in reverse
layer order; it hands each one to `GradReducer.push`.  Gradients are packed into flat buckets
(~4 MB: mn10's 19.5 MB gradient makes ~5 buckets; xGMI is point-to-point, ring all-reduce is
per-link bound, so a few MB per collective amortises the launch while still overlapping); a full
bucket is all-reduced asynchronously (RCCL runs on its own stream, ordered after the kernels that
produced the bucket) while the remaining layers' backward kernels keep the compute stream busy.
`finish()` waits for the collectives and returns the averaged gradients.

Round 5: no packed copy for the large gradients.  The backward asks the reducer for the MEMORY of a weight gradient before
it launches the kernel that produces it (`alloc`): consecutive requests are carved out of one flat, zero-filled buffer per
step, in production order, so a bucket is a contiguous range of that buffer and is all-reduced IN PLACE; `finish()` hands
out views of the same memory.  Only what is produced elsewhere (BatchNorm / bias / depthwise gradients: a few hundred KB
per step) still goes through `torch.cat`.
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, process_group=None, bucket_bytes=4 << 20, local=False, force_buckets=False):
        """local=True: pass-through (no collective) whatever the state of torch.distributed - what a model that was
        never handed to `enable_data_parallel` uses, so that wrapping it in torch DDP, or running backward on a
        subset of ranks, neither reduces twice nor deadlocks.  force_buckets=True: pack / all-reduce / unpack even
        with a single rank (exercises the bucket path on one GPU; tests and EAT_BENCH_FORCE_DIST)."""
        self.group = process_group
        self.bucket_bytes = bucket_bytes
        self.last_stats = None      # {"buckets", "bytes", "bucket_bytes"} of the last finished pass (bench.py's `rccl` object)
        self.world = 1 if local or not dist.is_initialized() else dist.get_world_size(process_group)
        self.bucketed = (self.world > 1 or force_buckets) and not local
        # gloo (CPU tests) has no AVG: SUM + one scale there
        self._avg = self.bucketed and dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self._op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        if self.bucketed and not dist.is_initialized():
            raise RuntimeError("GradReducer(force_buckets=True) needs an initialised torch.distributed process group")
        self._reset()

    def _reset(self):
        self.cur, self.cur_bytes = [], 0
        self.inflight = []      # (work, flat, [(name, shape, numel)])
        self.out = {}
        # flat gradient buffer of the pass (`alloc`): its size is learnt from the first pass (which falls back to `push`ed
        # copies); slots = [name, offset, numel, shape, pushed] in allocation order, `done` = first slot not yet reduced
        self.need = max(getattr(self, "need", 0), getattr(self, "req", 0))
        self.gbuf, self.req, self.slots, self.done = None, 0, [], 0

    def alloc(self, name, shape, device, dtype=torch.float32):
        """Zero-filled memory for the gradient `name` inside the pass's flat bucket buffer, or None (not bucketed / size not
        learnt yet / not fp32): the producer then allocates for itself and the gradient is packed by `push` as before."""
        if not self.bucketed or dtype != torch.float32:
            return None
        numel = 1
        for d in shape:
            numel *= int(d)
        n_al = (numel + 63) // 64 * 64                      # 256-byte aligned slots
        off = self.req
        self.req += n_al
        if self.need < self.req:
            return None
        if self.gbuf is None:
            self.gbuf = torch.zeros((self.need,), device=device, dtype=torch.float32)
        if self.gbuf.device != device:
            return None
        self.slots.append([name, off, numel, tuple(shape), False])
        return self.gbuf[off:off + numel].view(shape)

    def push(self, name, grad):
        if not self.bucketed:
            self.out[name] = grad
            return
        if self.gbuf is not None and grad.dtype == torch.float32:
            base = self.gbuf.data_ptr()
            for slot in self.slots[self.done:]:
                if slot[0] == name and grad.data_ptr() == base + 4 * slot[1] and grad.numel() == slot[2]:
                    slot[3], slot[4] = tuple(grad.shape), True      # (handed out 2-D, pushed in the parameter's shape)
                    self._flush_flat()
                    return
        self.cur.append((name, grad))
        self.cur_bytes += grad.numel() * grad.element_size()
        if self.cur_bytes >= self.bucket_bytes:
            self._flush()

    def _flush_flat(self, force=False):
        """All-reduce, in place, the longest prefix of produced slots once it fills a bucket."""
        end = self.done
        while end < len(self.slots) and self.slots[end][4]:
            end += 1
        if end == self.done:
            return
        lo = self.slots[self.done][1]
        hi = self.slots[end - 1][1] + (self.slots[end - 1][2] + 63) // 64 * 64
        if not force and (hi - lo) * 4 < self.bucket_bytes:
            return
        flat = self.gbuf[lo:hi]                             # (the alignment padding between slots is zeros on every rank)
        work = dist.all_reduce(flat, op=self._op, group=self.group, async_op=True)
        meta = [(s[0], s[3], s[2], s[1] - lo) for s in self.slots[self.done:end]]
        self.inflight.append((work, flat, meta))
        self.done = end

    def _flush(self):
        if not self.cur:
            return
        # one packed copy per bucket (4 MB: ~1 us of HBM time; the weight gradients of the train plan come out of one zero
        # arena in production order, but BatchNorm / SE / head gradients live elsewhere, so a bucket is not contiguous)
        flat = torch.cat([g.reshape(-1) for _, g in self.cur])
        meta = [(n, g.shape, g.numel()) for n, g in self.cur]
        # AVG: the 1 / world scaling happens inside the collective (RCCL's ncclAvg) - no separate pass over the bucket
        work = dist.all_reduce(flat, op=self._op, group=self.group, async_op=True)
        self.inflight.append((work, flat, meta))
        self.cur, self.cur_bytes = [], 0

    def begin_pass(self):
        """Drop whatever an ABORTED pass left behind (an exception inside the backward, e.g. an out-of-memory error the
        training loop catches and skips): without this its slots / in-flight works would leak into the next pass and
        `finish()` would raise "never pushed" on every later step.  Called at the start of every backward."""
        if self.slots or self.cur or self.inflight or self.out:
            for work, _, _ in self.inflight:
                work.wait()
            self._reset()

    def finish(self):
        """-> {name: averaged gradient}; blocks the compute stream (not the host) on the collectives."""
        if self.gbuf is not None:
            if any(not s[4] for s in self.slots[self.done:]):
                missing = [s[0] for s in self.slots if not s[4]]
                self._reset()                                 # (the reducer stays usable: the next pass starts clean)
                raise RuntimeError(f"GradReducer: gradient memory was handed out but never pushed: {missing[:4]}")
            self._flush_flat(force=True)
        self._flush()
        if self.bucketed:
            self.last_stats = {"buckets": len(self.inflight), "bytes": int(sum(f.numel() * f.element_size() for _, f, _ in self.inflight)),
                               "bucket_bytes": self.bucket_bytes}
        inv = 1.0 / self.world
        for work, flat, meta in self.inflight:
            work.wait()
            if not self._avg:
                flat.mul_(inv)
            off = 0
            for m in meta:
                if len(m) == 4:                              # in-place member of the flat buffer: (name, shape, numel, offset)
                    self.out[m[0]] = flat[m[3]:m[3] + m[2]].view(m[1])
                    continue
                name, shape, n = m
                self.out[name] = flat[off:off + n].view(shape)
                off += n
        out = self.out
        self._reset()
        return out


def enable_data_parallel(model, process_group=None, bucket_bytes=4 << 20, broadcast=True, force_buckets=False):
    """Attach a GradReducer to `model` (used by its train-mode backward) and, like DDP, broadcast
    rank 0's parameters and buffers so all replicas start identical."""
    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised (one process per GPU, backend 'nccl' = RCCL)")
    if broadcast:
        with torch.no_grad():
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t, src=0, group=process_group)
    reducer = GradReducer(process_group, bucket_bytes, force_buckets=force_buckets)
    if getattr(model, "_monolithic_backward", False):
        # MN: the single backward Function pushes gradients itself as it produces them (mn_train.py) ...
        model._grad_reducer = reducer
        # ... except in trunk mode (non-default heads, train-mode return_fmaps), where the head runs under torch autograd
        # on top of the trunk Function: its parameters get post-accumulate hooks feeding a reducer of their own, active
        # only while `forward_train` has flagged trunk mode (in the default mode the Function returns - and has already
        # reduced - the head's gradients: a second reduction would average twice)
        head = [n for n, _ in model.named_parameters() if n.startswith("classifier.")]
        install_grad_hooks(model, GradReducer(process_group, bucket_bytes, force_buckets=force_buckets), only=head,
                           active=lambda: getattr(model, "_eat_trunk_active", False))
    else:
        # DyMN and any autograd-driven module: per-parameter post-accumulate hooks feed the same
        # bucketed reducer; a callback queued on the autograd engine averages at the end of backward
        install_grad_hooks(model, reducer)
    return model


def install_grad_hooks(model, reducer, only=None, active=None):
    """only: restrict the hooks to these parameter names; active: callable - the hooks pass while it returns False."""
    names = {p: n for n, p in model.named_parameters()}
    state = {"pending": False}
    only = set(only) if only is not None else None

    def finalize():
        out = reducer.finish()
        state["pending"] = False
        with torch.no_grad():
            for n, p in model.named_parameters():
                g = out.get(n)
                if g is not None and g.data_ptr() != p.grad.data_ptr():
                    p.grad.copy_(g)

    def hook(p):
        if not reducer.bucketed or (active is not None and not active()):
            return
        if not state["pending"]:
            state["pending"] = True
            torch.autograd.Variable._execution_engine.queue_callback(finalize)
        reducer.push(names[p], p.grad)

    for n, p in model.named_parameters():
        if p.requires_grad and (only is None or n in only):
            p.register_post_accumulate_grad_hook(hook)
    model._grad_hooks_reducer = reducer
    return model
