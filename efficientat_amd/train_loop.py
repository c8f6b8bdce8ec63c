"""The KD training step of the reference's `ex_audioset.py` (lines 139-199) as one device-resident engine.

What the reference does per step on the host thread: mel -> mixup of the log-mel and the labels (~8 torch ops) ->
model -> BCE on mixed labels + knowledge-distillation BCE against gathered, mixed teacher probabilities (~20 torch ops
incl. a CPU-side index lookup and a host->device copy of the gathered rows) -> three `.cpu()` scalar reads (each a full
device sync) -> backward -> Adam.  Here:

  * the teacher table lives on the GPU; the file-name -> row lookup stays a host dict (names are Python strings) but only
    the (B,) int64 index vector travels;
  * mixup of the log-mel is one kernel (`eat_mixup_fwd`), the label / teacher mixing is folded into the loss kernel;
  * `eat_kd_loss_fwd_bwd` computes the loss terms AND d loss / d logits in one pass over the (B, 527) logits; the three
    statistics are accumulated in a device buffer and read ONCE per epoch (`epoch_stats`);
  * forward / backward are the HIP plans of mn_train.py / dymn_train.py; with `enable_data_parallel` the gradients are
    reduced in buckets while backward runs; the optimizer is whatever the caller built (fused Adam recommended).

Host RNG draws (`mixup`: torch.randperm, then numpy beta) happen in the reference's order, so a seeded run mixes the same
pairs with the same lambdas as the reference loop.
"""
import numpy as np
import torch

from . import ops
from .input_pipeline import I16_SCALE


def mixup(size, alpha):
    """helpers/utils.py:90-95: permutation + per-sample lambda = max(l, 1 - l), l ~ Beta(alpha, alpha)."""
    perm = torch.randperm(size)
    lam = np.random.beta(alpha, alpha, size).astype(np.float32)
    return perm, torch.from_numpy(np.maximum(lam, 1.0 - lam))


class _KDLoss(torch.autograd.Function):
    """loss scalar (device) whose backward hands the pre-computed d loss / d logits to the network's backward."""

    @staticmethod
    def forward(ctx, logits, y, perm, lam, teacher, tidx, kd_lambda, sums):
        logits = logits.contiguous()
        # the step's three terms go to their own zeroed buffer (the epoch accumulator grows to the hundreds: a difference
        # of two such fp32 numbers would lose 3-4 digits of the step loss); the accumulator is updated from it
        step = torch.zeros(3, device=logits.device, dtype=torch.float32)
        ctx.save_for_backward(ops.kd_loss_fwd_bwd(logits, y, perm, lam, teacher, tidx, kd_lambda, step))
        sums += step.to(sums.dtype)
        return step[0]

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return dlogits * g, None, None, None, None, None, None, None


def kd_loss(logits, y, perm=None, lam=None, teacher=None, teacher_idx=None, kd_lambda=1.0, sums=None):
    """Fused loss of ex_audioset.py:149-189 (see include/eat_hip.h: eat_kd_loss_fwd_bwd).  Returns the scalar loss as a
    device tensor that supports .backward(); `sums` (3,) accumulates (loss, label part, KD part) across calls."""
    if sums is None:
        sums = torch.zeros(3, device=logits.device, dtype=torch.float64)
    return _KDLoss.apply(logits, y.contiguous().float(), perm, lam, teacher, teacher_idx, float(kd_lambda), sums)


class KDTrainer:
    """step(wave, names, y) = one iteration of the reference's training loop (ex_audioset.py:139-199) without host syncs.

    model / mel: the HIP-backed modules; optimizer: e.g. torch.optim.Adam(model.parameters(), lr, fused=True);
    teacher_preds: (N, 527) tensor of teacher LOGITS (as stored in passt_enemble_logits_mAP_495.npy) or None;
    fname_to_index: dict file name -> row of teacher_preds."""

    def __init__(self, model, mel, optimizer, teacher_preds=None, fname_to_index=None, kd_lambda=0.1, temperature=1.0,
                 mixup_alpha=0.3):
        assert 0 <= kd_lambda <= 1, "Lambda for Knowledge Distillation must be between 0 and 1."
        self.model, self.mel, self.opt = model, mel, optimizer
        self.kd_lambda, self.mixup_alpha = float(kd_lambda), mixup_alpha
        dev = next(model.parameters()).device
        self.teacher = None
        if teacher_preds is not None and kd_lambda > 0:
            self.teacher = torch.sigmoid(torch.as_tensor(teacher_preds).float() / temperature).to(dev).contiguous()
        self.fname_to_index = fname_to_index or {}
        if self.teacher is not None and self.fname_to_index:
            # validated once on the host: the loss kernel gathers teacher rows by these indices on the device
            bad = [(f, i) for f, i in self.fname_to_index.items() if not (-1 <= int(i) < self.teacher.shape[0])]
            if bad:
                raise ValueError(f"fname_to_index holds {len(bad)} indices outside the teacher table of {self.teacher.shape[0]} "
                                 f"rows, e.g. {bad[0]}")
        self.sums = torch.zeros(3, device=dev, dtype=torch.float64)
        self.steps = 0

    def _teacher_rows(self, names):
        return torch.tensor([self.fname_to_index.get(f, -1) for f in names], dtype=torch.int64)

    def loss_and_backward(self, x, names, y):
        """mel -> mixup -> model -> KD loss -> backward (ex_audioset.py:139-196): leaves the gradients in `.grad` (averaged
        over the ranks when the model was handed to `enable_data_parallel`) and returns the loss as a device scalar."""
        dev = x.device
        bs = x.size(0)
        if x.dtype == torch.int16:                                             # 16-bit transport (input_pipeline.py)
            x = ops.wave_i16_to_f32(x.reshape(bs, -1).contiguous(), scale=1.0 / I16_SCALE)
        spec = self.mel(x.reshape(bs, -1)).unsqueeze(1)                        # _mel_forward, ex_audioset.py:223-228
        perm = lam = None
        if self.mixup_alpha:
            rn, lm = mixup(bs, self.mixup_alpha)                               # host draws, reference order
            perm, lam = rn.to(dev, torch.int32, non_blocking=True), lm.to(dev, non_blocking=True)
            spec = ops.mixup_fwd(spec, perm, lam)
        tidx = None
        if self.teacher is not None:
            tidx = self._teacher_rows(names).to(dev, non_blocking=True)
        y_hat, _ = self.model(spec)
        # (the reference's kd_lambda == 0 branch skips the KD term, i.e. loss = hard-label BCE: lambda 1 here)
        loss = kd_loss(y_hat, y, perm, lam, self.teacher, tidx, self.kd_lambda if self.teacher is not None else 1.0, self.sums)
        loss.backward()
        return loss.detach()

    def step(self, x, names, y):
        """x (B, 1, L) or (B, L) waveforms and y (B, 527) targets on the device; names: the B file names."""
        loss = self.loss_and_backward(x, names, y)
        self.opt.step()
        self.opt.zero_grad()
        self.steps += 1
        return loss                                                            # device scalar: no sync

    def epoch_stats(self):
        """Mean (train_loss, label_loss, distillation_loss) since the last call: the ONE host sync of the epoch."""
        s = (self.sums / max(1, self.steps)).cpu().tolist()
        self.sums.zero_()
        self.steps = 0
        return dict(train_loss=s[0], label_loss=s[1], distillation_loss=s[2])


class _HostRing:
    """Pinned staging for the small per-step host draws of a captured step (permutation, lambdas, teacher rows): `put`
    copies a host tensor into the next pinned slot and from there into the graph's static device buffer (asynchronous H2D on
    the current stream, i.e. ordered before the replay that follows); a slot is reused only after its upload completed."""

    def __init__(self, dev_buf, ring=4):
        self.dev = dev_buf
        self.host = [torch.empty(dev_buf.shape, dtype=dev_buf.dtype, pin_memory=True) for _ in range(ring)]
        self.ev = [None] * ring
        self.i = 0

    def put(self, t):
        i = self.i
        self.i = (i + 1) % len(self.host)
        if self.ev[i] is not None:
            self.ev[i].synchronize()
        self.host[i].copy_(t)
        self.dev.copy_(self.host[i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.ev[i] = ev


class GraphedKDTrainer(KDTrainer):
    """`KDTrainer` with the whole iteration - log-mel, mixup, forward, KD loss, backward, [bucketed RCCL all-reduce], optimizer -
    captured ONCE into a hipGraph and replayed with one host call per step (the eager loop issues ~450 launches per step
    from the reference's single host thread and is launch-bound).  What changes from step to step enters the graph through
    static device buffers, refreshed before each replay:

        wave (B, L) / y (B, 527)      the batch (device-to-device copy from the prefetcher's slot, or the int16 -> fp32
                                      conversion of a 16-bit batch written straight into the buffer);
        perm (B) int32, lam (B)       the mixup draw (helpers/utils.py:90-95), drawn on the host in the reference's order;
        tidx (B) int64                rows of the teacher table (the file-name lookup stays a host dict);
        the mel basis                 band table of the step's (fmin, fmax) jitter (models/preprocess.py:45-55), fixed shape
                                      (`AugmentMelSTFT.static_tables`).

    Host RNG: mel draws, then the mixup draws - the order of `KDTrainer.step` and of ex_audioset.py:139-146, so a seeded
    run replays the same augmentation as the eager trainer.  SpecAugment masks (freqm / timem != 0) are scalar launch
    arguments of the mel kernel: the mel then runs eagerly in front of the graph, writing the graph's input buffer.

    The optimizer must be capturable (torch.optim.Adam(..., capturable=True[, fused=True]); a tensor `lr` lets a scheduler
    change the rate without re-capturing).  DyMN: `model.update_params(epoch)` changes Python-side temperatures that are
    launch constants - call `recapture()` after it.  Batches of another size (a last partial batch) fall back to the eager
    step."""

    def __init__(self, model, mel, optimizer, batch_size, clip_samples, n_classes=527, teacher_preds=None,
                 fname_to_index=None, kd_lambda=0.1, temperature=1.0, mixup_alpha=0.3, warmup=2):
        super().__init__(model, mel, optimizer, teacher_preds, fname_to_index, kd_lambda, temperature, mixup_alpha)
        dev = next(model.parameters()).device
        self.B, self.L = int(batch_size), int(clip_samples)
        self.wave = torch.zeros((self.B, self.L), device=dev)
        self.y = torch.zeros((self.B, n_classes), device=dev)
        self._perm = _HostRing(torch.arange(self.B, device=dev, dtype=torch.int32)) if mixup_alpha else None
        self._lam = _HostRing(torch.ones(self.B, device=dev)) if mixup_alpha else None
        self._tidx = _HostRing(torch.full((self.B,), -1, device=dev, dtype=torch.int64)) if self.teacher is not None else None
        self.mel_in_graph = not (mel.freqm or mel.timem)
        T = 1 + (self.L - 1) // mel.hopsize
        self.spec = torch.empty((self.B, 1, mel.n_mels, T), device=dev)
        mel.static_tables(dev)
        mel.stage_tables(mel.fmin, mel.fmax)
        self.warmup = warmup
        self.graph = None
        self.loss = None
        self.recapture()

    # the captured sequence (everything reads / writes static buffers)
    def _issue(self):
        if self.mel_in_graph:
            self.mel.forward_static(self.wave, out=self.spec)
        spec = self.spec
        perm = lam = None
        if self._perm is not None:
            perm, lam = self._perm.dev, self._lam.dev
            spec = ops.mixup_fwd(spec, perm, lam)
        y_hat, _ = self.model(spec)
        loss = kd_loss(y_hat, self.y, perm, lam, self.teacher, None if self._tidx is None else self._tidx.dev,
                       self.kd_lambda if self.teacher is not None else 1.0, self.sums)
        loss.backward()
        self.opt.step()
        return loss.detach()

    def recapture(self):
        from .graphs import _capture_mode
        if not self.mel_in_graph:
            self.mel.forward_static(self.wave, out=self.spec)
        keep = self.sums.clone()
        state = _snapshot(self.model, self.opt)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):                 # (allocator / pack-plan warm-up on a side stream, as torch recommends)
                self.opt.zero_grad(set_to_none=True)
                self._issue()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        self.opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        with torch.cuda.graph(self.graph, capture_error_mode=_capture_mode()):
            self.loss = self._issue()
        # the warm-up steps trained on the zero batch: put parameters, BatchNorm buffers, optimizer state and the loss sums back
        _restore(self.model, self.opt, state)
        self.sums.copy_(keep)
        cache = getattr(self.model, "_cache", None)
        if cache is not None:
            cache.invalidate()

    def step(self, x, names, y):
        bs = x.size(0)
        if bs != self.B or x.numel() != self.B * self.L:
            return super().step(x, names, y)             # e.g. the last, partial batch of an epoch
        fmin, fmax, fmask, tmask = self.mel.draw(self.L)                     # host draws, reference order: mel first
        self.mel.stage_tables(fmin, fmax)
        if x.dtype == torch.int16:                                           # 16-bit transport (input_pipeline.py)
            ops.wave_i16_to_f32(x.reshape(bs, -1).contiguous(), out=self.wave, scale=1.0 / I16_SCALE)
        else:
            self.wave.copy_(x.reshape(bs, -1), non_blocking=True)
        self.y.copy_(y, non_blocking=True)
        if not self.mel_in_graph:
            self.mel.forward_static(self.wave, out=self.spec, fmask=fmask, tmask=tmask)
        if self._perm is not None:
            rn, lm = mixup(bs, self.mixup_alpha)
            self._perm.put(rn.to(torch.int32))
            self._lam.put(lm)
        if self._tidx is not None:
            self._tidx.put(self._teacher_rows(names))
        self.graph.replay()
        cache = getattr(self.model, "_cache", None)
        if cache is not None:            # a replay updates the weights without bumping their version counters
            cache.invalidate()
        self.steps += 1
        return self.loss


def _snapshot(model, opt):
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    od = opt.state_dict()
    ost = {k: {n: (t.detach().clone() if torch.is_tensor(t) else t) for n, t in st.items()} for k, st in od["state"].items()}
    return sd, ost


def _restore(model, opt, state):
    """In place (the captured graph holds the addresses of the parameters and of the optimizer's moment buffers)."""
    sd, ost = state
    with torch.no_grad():
        for k, v in model.state_dict().items():
            v.copy_(sd[k])
        cur = opt.state_dict()["state"]
        for k, st in cur.items():
            for n, t in st.items():
                if not torch.is_tensor(t):
                    continue
                if k in ost and n in ost[k]:
                    t.copy_(ost[k][n])
                else:
                    t.zero_()                            # state created by the warm-up (first use of the optimizer)
