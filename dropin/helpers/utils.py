"""`helpers.utils` of the reference (helpers/utils.py): name->width table, AudioSet label list read
from ./metadata/class_labels_indices.csv at import (same CWD-relative behaviour), the LR schedule
helpers and the host-side mixup / mixstyle draws.  Host logic only - nothing here is accelerated."""
import csv

import numpy as np
import torch

from efficientat_amd.utils import NAME_TO_WIDTH, exp_rampup, exp_warmup_linear_down, linear_rampdown  # noqa: F401

with open("metadata/class_labels_indices.csv", "r") as _f:
    _rows = list(csv.reader(_f, delimiter=","))[1:]
ids = [r[1] for r in _rows]
labels = [r[2] for r in _rows]
classes_num = len(labels)


def mixup(size, alpha):
    """Permutation + per-sample lambda = max(l, 1-l), l ~ Beta(alpha, alpha) (same RNG draw order)."""
    perm = torch.randperm(size)
    lam = np.random.beta(alpha, alpha, size).astype(np.float32)
    return perm, torch.FloatTensor(np.maximum(lam, 1.0 - lam))


def mixstyle(x, p=0.4, alpha=0.4, eps=1e-6, mix_labels=False):
    """Frequency-wise MixStyle: mix per-(b,f) mean/std with a permuted sample's."""
    if np.random.rand() > p:
        return x
    n = x.size(0)
    mu = x.mean(dim=[1, 3], keepdim=True).detach()
    sig = (x.var(dim=[1, 3], keepdim=True) + eps).sqrt().detach()
    lmda = torch.distributions.beta.Beta(alpha, alpha).sample((n, 1, 1, 1)).to(x.device)
    perm = torch.randperm(n).to(x.device)
    out = (x - mu) / sig * (sig * lmda + sig[perm] * (1 - lmda)) + (mu * lmda + mu[perm] * (1 - lmda))
    return (out, perm, lmda) if mix_labels else out
