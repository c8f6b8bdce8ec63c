"""f1: the device-side training-loop glue (efficientat_amd/train_loop.py, csrc/train_glue.hip) against the torch-op
formulation the reference's loop uses (ex_audioset.py:142-189), written out here on CPU tensors as the oracle."""
import contextlib
import io

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("no GPU", allow_module_level=True)

from efficientat_amd import ops  # noqa: E402
from efficientat_amd.mn import get_model  # noqa: E402
from efficientat_amd.preprocess import AugmentMelSTFT  # noqa: E402
from efficientat_amd.train_loop import KDTrainer, kd_loss  # noqa: E402

DEV = torch.device("cuda:0")


def _reference_loss(y_hat, y, rn, lam, teacher_probs, indices, kd_lambda):
    """ex_audioset.py:142-189 verbatim in meaning, on CPU tensors."""
    bs = y_hat.shape[0]
    distillation_loss = nn.BCEWithLogitsLoss(reduction="none")
    if rn is not None:
        y_mix = y * lam.reshape(bs, 1) + y[rn] * (1. - lam.reshape(bs, 1))
    else:
        y_mix = y
    label_loss = F.binary_cross_entropy_with_logits(y_hat, y_mix, reduction="none").mean()
    if kd_lambda > 0 and teacher_probs is not None:
        unknown = indices == -1
        soft = teacher_probs[indices]
        if rn is not None:
            st = distillation_loss(y_hat, soft).mean(dim=1) * lam.reshape(bs) + \
                distillation_loss(y_hat, soft[rn]).mean(dim=1) * (1. - lam.reshape(bs))
        else:
            st = distillation_loss(y_hat, soft)
            st = st.mean(dim=1)
        st = torch.where(unknown, torch.zeros_like(st), st).mean()
        label_loss = kd_lambda * label_loss
        st = (1 - kd_lambda) * st
    else:
        st = torch.tensor(0.)
    return label_loss + st, label_loss, st


@pytest.mark.parametrize("mix,kd", [(True, True), (True, False), (False, True), (False, False)])
def test_fused_kd_loss_and_gradient(mix, kd):
    g = torch.Generator().manual_seed(3)
    B, C, N = 9, 527, 40
    z = (torch.randn(B, C, generator=g) * 3).requires_grad_(True)
    y = (torch.rand(B, C, generator=g) < 0.01).float()
    rn = torch.randperm(B, generator=g) if mix else None
    lam = (torch.rand(B, generator=g) * 0.5 + 0.5) if mix else None
    teacher = torch.sigmoid(torch.randn(N, C, generator=g) * 2 - 5)
    idx = torch.randint(0, N, (B,), generator=g)
    idx[2] = -1
    idx[7] = -1
    kd_lambda = 0.1 if kd else 1.0
    loss, lab, st = _reference_loss(z, y, rn, lam, teacher if kd else None, idx, 0.1 if kd else 0.0)
    loss.backward()
    zd = z.detach().to(DEV).requires_grad_(True)
    sums = torch.zeros(3, device=DEV)
    got = kd_loss(zd, y.to(DEV), None if rn is None else rn.to(DEV, torch.int32), None if lam is None else lam.to(DEV),
                  teacher.to(DEV) if kd else None, idx.to(DEV) if kd else None, kd_lambda, sums)
    got.backward()
    assert abs(got.item() - loss.item()) < 2e-6 * max(1.0, abs(loss.item()))
    s = sums.cpu()
    assert abs(s[1].item() - lab.detach().item()) < 2e-6 and abs(s[2].item() - float(st.detach())) < 2e-6
    assert float((zd.grad.cpu() - z.grad).abs().max()) < 1e-8 + 1e-5 * float(z.grad.abs().max())


@pytest.mark.parametrize("shape", [(5, 1, 128, 1000), (3, 1, 7, 9)])
def test_mixup_kernel(shape):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(*shape, generator=g)
    rn, lam = torch.randperm(shape[0], generator=g), torch.rand(shape[0], generator=g)
    ref = x * lam.reshape(-1, 1, 1, 1) + x[rn] * (1. - lam.reshape(-1, 1, 1, 1))
    got = ops.mixup_fwd(x.to(DEV), rn.to(DEV, torch.int32), lam.to(DEV))
    assert float((got.cpu() - ref).abs().max()) < 1e-6


def test_kd_trainer_step_equals_the_reference_loop_formulation():
    """One KDTrainer.step (device glue) vs the same step spelled as the reference loop does (torch ops, same host RNG
    seeds -> same mix-up pairs): identical loss and parameter updates."""
    from efficientat_amd.train_loop import mixup
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        mel = AugmentMelSTFT(freqm=0, timem=0).to(DEV)
        m1, m2 = get_model(width_mult=0.4), get_model(width_mult=0.4)
    with torch.no_grad():
        for mod in m1.modules():
            if isinstance(mod, nn.Conv2d):
                fan_in = mod.weight.shape[1] * mod.weight.shape[2] * mod.weight.shape[3]
                mod.weight.normal_(0, (2.0 / fan_in) ** 0.5)
            elif isinstance(mod, nn.Linear):
                mod.weight.normal_(0, mod.weight.shape[1] ** -0.5)
    m2.load_state_dict(m1.state_dict())
    m1.to(DEV).train()
    m2.to(DEV).train()
    mel.train()
    B, N = 6, 20
    g = torch.Generator().manual_seed(5)
    wave = (0.1 * torch.randn(B, 1, 64000, generator=g)).to(DEV)
    y = (torch.rand(B, 527, generator=g) < 0.01).float().to(DEV)
    names = [f"f{i}" for i in range(B)]
    f2i = {f"f{i}": i for i in range(B) if i != 3}
    tlogits = torch.randn(N, 527, generator=g) * 2 - 5
    keep = torch.ones(B, m1.classifier[2].out_features, device=DEV)
    m1._drop_mask_override = m2._drop_mask_override = keep
    p0 = [p.detach().clone() for p in m1.parameters()]
    o1 = torch.optim.SGD(m1.parameters(), lr=1e-3)
    o2 = torch.optim.SGD(m2.parameters(), lr=1e-3)
    tr = KDTrainer(m1, mel, o1, teacher_preds=tlogits, fname_to_index=f2i, kd_lambda=0.1, mixup_alpha=0.3)
    torch.manual_seed(11); np.random.seed(11)
    l1 = tr.step(wave, names, y)
    # the reference formulation (ex_audioset.py:139-199) on the same modules
    torch.manual_seed(11); np.random.seed(11)
    teacher = torch.sigmoid(tlogits / 1.0)
    x = mel(wave.reshape(B, -1)).unsqueeze(1)
    rn, lam = mixup(B, 0.3)
    lam = lam.to(DEV)
    x = x * lam.reshape(B, 1, 1, 1) + x[rn] * (1. - lam.reshape(B, 1, 1, 1))
    y_hat, _ = m2(x)
    y_mix = y * lam.reshape(B, 1) + y[rn] * (1. - lam.reshape(B, 1))
    label_loss = F.binary_cross_entropy_with_logits(y_hat, y_mix, reduction="none").mean()
    indices = torch.tensor([f2i.get(f, -1) for f in names], dtype=torch.int64)
    soft = teacher[indices].to(DEV)
    bce = nn.BCEWithLogitsLoss(reduction="none")
    st = bce(y_hat, soft).mean(dim=1) * lam.reshape(B) + bce(y_hat, soft[rn]).mean(dim=1) * (1. - lam.reshape(B))
    st[indices == -1] = st[indices == -1] * 0
    loss = 0.1 * label_loss + 0.9 * st.mean()
    loss.backward()
    o2.step()
    assert abs(l1.item() - loss.item()) < 1e-5 * max(1.0, abs(loss.item()))
    # the parameter UPDATES (= -lr * gradient) agree like gradients do (activation-kink budget of test_gpu_train.py)
    d1 = [(p.detach() - q).double() for p, q in zip(m1.parameters(), p0)]
    d2 = [(p.detach() - q).double() for p, q in zip(m2.parameters(), p0)]
    dmax = max(float(d.norm()) for d in d2)
    rels = [float((a - b).norm() / b.norm()) for a, b in zip(d1, d2) if float(b.norm()) > 1e-5 * dmax]
    assert len(rels) > 50 and max(rels) < 5e-2 and float(np.median(rels)) < 1e-2, (max(rels), float(np.median(rels)))
    stats = tr.epoch_stats()
    assert abs(stats["train_loss"] - loss.item()) < 1e-5 and tr.steps == 0


# ------------------------------------------------------------------------------------------------ f2: input staging
def test_device_prefetcher_delivers_the_loader_batches_in_order():
    """DevicePrefetcher (efficientat_amd/input_pipeline.py): pinned, double-buffered H2D on a copy stream must hand out
    exactly the DataLoader's batches (datasets/audioset.py:138-161 tuple layout, file names passed through), in order,
    while the consumer keeps the compute stream busy; slots are recycled only after the consumer moved on."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dropin", "datasets", "audioset.py")
    os.environ["EAT_SYNTH_AUDIOSET"] = "1"                                          # explicit opt-in to the synthetic stand-in
    spec = importlib.util.spec_from_file_location("eat_dropin_audioset", path)    # (not `import datasets`: the
    audioset = importlib.util.module_from_spec(spec)                                 #  HuggingFace package has that name)
    spec.loader.exec_module(audioset)
    from torch.utils.data import DataLoader, Subset
    from efficientat_amd.input_pipeline import DevicePrefetcher

    ds = Subset(audioset.get_training_set(add_index=True), list(range(22)))       # 22 clips: a ragged last batch
    loader = DataLoader(ds, batch_size=4, shuffle=False, num_workers=0)
    want = [(w.clone(), list(n), y.clone(), i.clone()) for w, n, y, i in loader]
    assert want[0][0].shape == (4, 1, 320000) and want[-1][0].shape[0] == 2
    got = []
    burn = torch.randn(2048, 2048, device=DEV)
    for depth in (1, 2, 3):
        got.clear()
        for w, n, y, i in DevicePrefetcher(loader, DEV, depth=depth):
            assert w.is_cuda and y.is_cuda and i.is_cuda and isinstance(n[0], str)
            for _ in range(4):
                burn = torch.tanh(burn @ burn * 1e-3)                                # compute-stream work between batches
            got.append((w.float().sum(dim=(1, 2)), list(n), y.clone(), i.clone(), w[:, 0, 12345].clone()))
        assert len(got) == len(want)
        for (ws, n, y, i, probe), (w0, n0, y0, i0) in zip(got, want):
            assert n == n0
            assert torch.equal(y.cpu(), y0) and torch.equal(i.cpu(), i0)
            assert torch.equal(probe.cpu(), w0[:, 0, 12345])
            assert torch.allclose(ws.cpu(), w0.sum(dim=(1, 2)), rtol=1e-4, atol=1e-3)
