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

    def step(self, x, names, y):
        """x (B, 1, L) or (B, L) waveforms and y (B, 527) targets on the device; names: the B file names."""
        dev = x.device
        bs = x.size(0)
        spec = self.mel(x.reshape(bs, -1)).unsqueeze(1)                        # _mel_forward, ex_audioset.py:223-228
        perm = lam = None
        if self.mixup_alpha:
            rn, lm = mixup(bs, self.mixup_alpha)                               # host draws, reference order
            perm, lam = rn.to(dev, torch.int32, non_blocking=True), lm.to(dev, non_blocking=True)
            spec = ops.mixup_fwd(spec, perm, lam)
        tidx = None
        if self.teacher is not None:
            tidx = torch.tensor([self.fname_to_index.get(f, -1) for f in names], dtype=torch.int64).to(dev, non_blocking=True)
        y_hat, _ = self.model(spec)
        # (the reference's kd_lambda == 0 branch skips the KD term, i.e. loss = hard-label BCE: lambda 1 here)
        loss = kd_loss(y_hat, y, perm, lam, self.teacher, tidx, self.kd_lambda if self.teacher is not None else 1.0, self.sums)
        loss.backward()
        self.opt.step()
        self.opt.zero_grad()
        self.steps += 1
        return loss.detach()                                                   # device scalar: no sync

    def epoch_stats(self):
        """Mean (train_loss, label_loss, distillation_loss) since the last call: the ONE host sync of the epoch."""
        s = (self.sums / max(1, self.steps)).cpu().tolist()
        self.sums.zero_()
        self.steps = 0
        return dict(train_loss=s[0], label_loss=s[1], distillation_loss=s[2])
