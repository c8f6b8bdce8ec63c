"""The optimizer of the reference's training loop on the HIP path: `torch.optim.Adam` / `AdamW` (ex_audioset.py:86-91) with the
same constructor arguments and state-dict layout (`exp_avg`, `exp_avg_sq`, `step` per parameter), whose `step()` is ONE launch
of `eat_adam_multi` over every parameter (+ a one-thread counter kernel in the capturable form) instead of torch's multi-tensor
chunks.  SURVEY 8(f) row f1 / K17 allow torch's fused optimizer; this is the in-library form of the same update."""
import numpy as np
import torch

from . import _lib

_CHUNK = 4096


class FusedAdam(torch.optim.Optimizer):
    """Adam (decoupled=False: L2 weight decay, `torch.optim.Adam`) or AdamW (decoupled=True).  fp32 CUDA parameters with fp32
    gradients; `capturable=True` keeps the step counter on the device (required inside a hipGraph capture); `lr` may be a
    0-dim / 1-element float32 CUDA tensor that a scheduler writes (then it is read on the device).  No amsgrad / maximize."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled=False, capturable=False):
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError(f"invalid betas {betas}")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, decoupled=decoupled,
                                      capturable=capturable))
        self._tables = {}
        self._spare = {}
        self._counters = {}
        self._captured = []
        self._layouts = {}

    def _state(self, p, capturable):
        st = self.state[p]
        if not st:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.EatHipError("FusedAdam: run one eager step before capturing (the moment buffers and the step counter "
                                       "must exist outside the graph - a captured initialisation would re-run on every replay)")
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device if capturable else "cpu")
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def _table(self, gi, ps):
        """Chunk table of a parameter group (device tensor), rebuilt when a parameter or gradient moved.  Outside a capture the
        upload is a synchronous copy; inside a stream capture (the gradients of a captured step live at new addresses) it goes
        through a pinned buffer allocated by the first eager step - the captured copy node re-reads that buffer on every replay,
        so it is used for ONE capture only (`_spare`)."""
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in ps)
        cached = self._tables.get(gi)
        if cached is not None and cached[0] == key:
            return cached[1], cached[2]
        # the chunk layout depends on the parameters only: offsets / lengths / parameter and moment addresses are built once per
        # parameter set, a rebuild (gradients at new addresses: an eager data-parallel step hands out a fresh bucket buffer
        # every pass) only adds the gradients' base addresses - vectorised, no Python loop over ~1500 chunks
        lay = self._layouts.get(gi)
        pkey = tuple(k[0] for k in key)
        if lay is None or lay[0] != pkey:
            idx, offs, lens = [], [], []
            for i, p in enumerate(ps):
                st = self.state[p]
                if p.dtype != torch.float32 or not p.is_cuda:
                    raise _lib.EatHipError("FusedAdam: fp32 CUDA parameters and gradients only")
                if not p.is_contiguous():
                    raise _lib.EatHipError("FusedAdam: parameters and gradients must be contiguous")
                n = p.numel()
                o = np.arange(0, n, _CHUNK, dtype=np.int64)
                idx.append(np.full(o.shape, i, dtype=np.int64))
                offs.append(o)
                lens.append(np.minimum(_CHUNK, n - o))
            idx, offs, lens = np.concatenate(idx), np.concatenate(offs), np.concatenate(lens).astype(np.int32)
            pb = np.array([p.data_ptr() for p in ps], dtype=np.uint64)[idx] + (4 * offs).astype(np.uint64)
            mb = np.array([self.state[p]["exp_avg"].data_ptr() for p in ps], dtype=np.uint64)[idx] + (4 * offs).astype(np.uint64)
            vb = np.array([self.state[p]["exp_avg_sq"].data_ptr() for p in ps], dtype=np.uint64)[idx] + (4 * offs).astype(np.uint64)
            lay = self._layouts[gi] = (pkey, idx, (4 * offs).astype(np.uint64), lens, pb, mb, vb)
        _, idx, boffs, lens, pb, mb, vb = lay
        for p in ps:
            if p.grad.dtype != torch.float32 or not p.grad.is_contiguous():
                raise _lib.EatHipError("FusedAdam: fp32 contiguous gradients only")
        gb = np.array([k[1] for k in key], dtype=np.uint64)[idx] + boffs
        tab = np.zeros((lens.shape[0],), dtype=[("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i4"), ("pad", "<i4")])
        tab["p"], tab["g"], tab["m"], tab["v"], tab["n"] = pb, gb, mb, vb, lens
        recs = tab
        raw = torch.from_numpy(tab.view(np.uint8).copy())
        if torch.cuda.is_current_stream_capturing():
            spare = self._spare.get(gi)
            if not spare or spare[0].numel() != raw.numel():
                raise _lib.EatHipError("FusedAdam: the gradients moved inside a stream capture and no staging buffer is left - run "
                                       "one eager step with this set of parameters before capturing, and capture once per optimizer")
            host, dev_tab = spare
            self._spare[gi] = None
            host.copy_(raw)
            dev_tab.copy_(host, non_blocking=True)
            self._tables[gi] = (key, dev_tab, len(recs))
            self._captured.append((host, dev_tab))      # the graph's copy node re-reads them on every replay: never freed
        else:
            dev_tab = raw.to(ps[0].device)
            self._tables[gi] = (key, dev_tab, len(recs))
            if not self._spare.get(gi):                                # staging for the next capture (same parameters => same size)
                self._spare[gi] = (torch.empty_like(raw).pin_memory(), torch.empty_like(dev_tab))
        return dev_tab, len(recs)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            cap = bool(group["capturable"])
            sts = [self._state(p, cap) for p in ps]
            tab, n = self._table(gi, ps)
            lr = group["lr"]
            lr_ptr, lr_val = (lr.data_ptr(), 0.0) if (torch.is_tensor(lr) and lr.is_cuda) else (None, float(lr))
            # one counter per group on the device (capturable): every parameter's `step` is a view of it after the first step
            if cap:
                ctr = self._counters.get(gi)
                if ctr is None:
                    if torch.cuda.is_current_stream_capturing():
                        raise _lib.EatHipError("FusedAdam: run one eager step before capturing")
                    ctr = self._counters[gi] = torch.zeros((1,), dtype=torch.float32, device=ps[0].device)
                    ctr.fill_(float(sts[0]["step"]))
                step_ptr, step_val = ctr.data_ptr(), 0.0
            else:
                step_ptr, step_val = None, float(sts[0]["step"])
            b1, b2 = group["betas"]
            _lib.call("eat_adam_multi", tab.data_ptr(), n, lr_ptr, lr_val, step_ptr, step_val, float(b1), float(b2),
                      float(group["eps"]), float(group["weight_decay"]), 1 if group["decoupled"] else 0, float(grad_scale),
                      torch.cuda.current_stream().cuda_stream)
            for st in sts:
                if cap:
                    st["step"] = ctr[0]
                else:
                    st["step"] = st["step"] + 1
        return loss
