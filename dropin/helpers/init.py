"""`helpers.init.worker_init_fn` of the reference: per-DataLoader-worker seeding from
(torch.initial_seed(), worker id) via numpy SeedSequence (helpers/init.py:6-33)."""
import random

import numpy as np
import torch


def _spawn(seq, as_int):
    state = seq.spawn(1)[0].generate_state(2, dtype=np.uint32)
    return sum(int(s) << (32 * i) for i, s in enumerate(state)) if as_int else state


def worker_init_fn(wid):
    seq = np.random.SeedSequence([torch.initial_seed(), wid])
    torch.random.manual_seed(_spawn(seq, True))
    np.random.seed(_spawn(seq, False))
    random.seed(_spawn(seq, True))
