"""GPU parity at the configurations BASELINE.json names (the ones bench.py measures):

  configs[1]  mn10_as forward at batch 256        -> test_mn10_batch256_*: the 5 parity clips embedded at batch positions
              {0, 1, 127, 128, 255} of a 256-clip batch (1x1 GEMM tiles straddle samples, grids are 50x larger than in
              the 5-clip tests, the SE-scale slot logic changes regime) vs the oracle / golden logits and vs the same
              clips run as a batch of 5 (batch invariance)
  configs[2]  mn40_as train step (bf16 MFMA 1x1)  -> test_mn40_train_step_*: fp32 and bf16 arithmetic against torch-CPU
              autograd over the oracle on 8 full-length clips
  configs[3]  dymn20_as train step                -> test_dymn20_*: eval logits / fmaps and one train step vs the oracle

Tolerances: logits <= 1e-3 (BASELINE north_star); gradients rel-L2 per tensor as in test_gpu_train.py (activation
kinks make them discontinuous in forward rounding); bf16: SURVEY 8c "bf16-level tolerance (~1e-2 relative)".
"""
import contextlib
import io
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import eat_oracle as O
from oracle import synth

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("no GPU", allow_module_level=True)

from efficientat_amd import mn as mn_mod  # noqa: E402
from efficientat_amd.preprocess import AugmentMelSTFT  # noqa: E402

DEV = torch.device("cuda:0")
SLOTS = [0, 1, 127, 128, 255]


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _rel(got, ref):
    got, ref = got.detach().cpu().double().reshape(-1), ref.detach().double().reshape(-1)
    return float((got - ref).norm() / max(1e-30, float(ref.norm())))


def _ulp_noise(x, seed=4321):
    """x * (1 + 2^-23 N(0,1)): one-ulp input noise.  These synthetic networks amplify round-off by 10^3 ... 10^5 (activation
    kinks, batch statistics, DyMN's softmax attention at temperature 1: SURVEY 8c) - a second step on such an input measures
    the round-off floor of a gradient comparison on THIS network, tensor by tensor."""
    return x * (1.0 + 2.0 ** -23 * torch.randn(x.shape, generator=torch.Generator().manual_seed(seed)))


def _floor(noise, name, k=4.0):
    return k * noise.get(name, 0.0) if noise else 0.0


def _grad_state(sd, skip=("running_mean", "running_var", "num_batches_tracked", "lambdas", "init_v")):
    return {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and not k.endswith(skip) else v.clone())
            for k, v in sd.items()}


# ------------------------------------------------------------------ configs[1]: mn10 forward at batch 256
@pytest.fixture(scope="module")
def mn10_b256(golden_dir):
    g = np.load(os.path.join(golden_dir, "mn10_ref.npz"))
    sd = synth.synth_state(synth.mn_shapes(1.0), seed=0)
    for k in g.files:
        if k.startswith("bn/"):
            sd[k[3:]] = torch.from_numpy(g[k])
    clips = synth.parity_clips(320000, seed=1234)
    x_ref = O.mel_forward(clips).unsqueeze(1)
    with torch.no_grad():
        ref, _ = O.mn_forward(sd, x_ref)
    gen = torch.Generator().manual_seed(4242)
    wave = (0.1 * torch.randn(256, 320000, generator=gen)).clamp_(-1, 1)
    for i, s in enumerate(SLOTS):
        wave[s] = clips[i]
    return dict(sd=sd, clips=clips, x_ref=x_ref, ref=ref, wave=wave, golden=g["eval_logits"])


@pytest.mark.parametrize("pw_mode", ["fp32", "auto"])
def test_mn10_batch256_matches_oracle_and_is_batch_invariant(mn10_b256, pw_mode, monkeypatch):
    d = mn10_b256
    monkeypatch.setattr(mn_mod, "_PW_MODE", pw_mode)
    model = _quiet(mn_mod.get_model, width_mult=1.0)
    model.load_state_dict(d["sd"], strict=True)
    model.to(DEV).eval()
    mel = _quiet(AugmentMelSTFT, freqm=0, timem=0).to(DEV).eval()
    with torch.no_grad():
        m256 = mel(d["wave"].to(DEV))                          # HIP log-mel of the whole batch
        m5 = mel(d["clips"].to(DEV))
        # T2 at B = 256: model kernels on the oracle's mel in the 5 slots
        mx = m256.clone()
        mx[SLOTS] = d["x_ref"][:, 0].to(DEV)
        l_t2, f_t2 = model(mx.unsqueeze(1))
        l_t3, _ = model(m256.unsqueeze(1))                     # T3 at B = 256: waveform -> logits
        l5_t2, f5_t2 = model(d["x_ref"].to(DEV))
        l5_t3, _ = model(m5.unsqueeze(1))
    assert torch.isfinite(l_t3).all()
    # the front-end is batch-invariant bit for bit (no atomics, same tiles)
    assert torch.equal(m256[SLOTS], m5)
    got2, got3 = l_t2[SLOTS].cpu(), l_t3[SLOTS].cpu()
    assert float((got2 - d["ref"]).abs().max()) < 1e-3
    assert float((got3 - d["ref"]).abs().max()) < 1e-3
    assert np.abs(got2.numpy() - d["golden"]).max() < 1e-3      # stored output of the unmodified reference
    assert np.abs(got3.numpy() - d["golden"]).max() < 1e-3
    # batch invariance: the same clips inside a 256-batch and as a 5-batch.  Not bit-exact: the fused SE / head pools
    # accumulate with atomics, and in `auto` a last-bit difference of an SE sum can flip a bf16 rounding of the split
    # 1x1 kernels downstream (measured ~1e-6 / ~1e-5 on |logits| <= 3.6).
    tol = 1e-5 if pw_mode == "fp32" else 5e-5
    assert float((l_t2[SLOTS] - l5_t2).abs().max()) < tol, float((l_t2[SLOTS] - l5_t2).abs().max())
    assert float((l_t3[SLOTS] - l5_t3).abs().max()) < tol
    assert float((f_t2[SLOTS] - f5_t2).abs().max()) < tol


# ------------------------------------------------------------------ configs[3]: dymn20
def _dymn20_case(temp):
    wave = synth.parity_clips(320000, seed=31)[[0, 2, 3, 4]]     # noise, two-tone, silence+chirp, AM noise+tone
    x = O.mel_forward(wave).unsqueeze(1)
    fwd = lambda sd, xm, **k: O.dymn_forward(sd, xm, width_mult=2.0, temperature=temp, **k)
    sd = synth.calibrate(synth.synth_state(synth.dymn_shapes(2.0), seed=0), fwd, x)
    return dict(sd=sd, x=x, fwd=fwd, temp=temp)


@pytest.fixture(scope="module")
def dymn20_case():
    return _dymn20_case(1.0)


@pytest.fixture(scope="module")
def dymn20_case_t30():
    """The same network at the reference's STARTING temperature (models/dymn/dy_block.py:133-139: T_max = 30): the kernel
    attention is then close to uniform and the step well conditioned - the regime in which the tensors that need the
    round-off-floor escape at T = 1 (below) are held to the FIXED bars."""
    return _dymn20_case(30.0)


# parameters whose gradient the one-ulp control itself moves by more than the fixed bar at temperature 1: the Linear that
# produces the K = 4 kernel-attention logits of a DynamicConv (models/dymn/dy_block.py:68-71, `residuals.0`) - a softmax at
# T = 1 over logits that differ by O(10) amplifies round-off of the context vector into percent-level changes
_ATTENTION_HEAD = (".residuals.0.weight", ".residuals.0.bias")


def _dymn20(sd, temp):
    from efficientat_amd.dymn import get_model
    model = _quiet(get_model, width_mult=2.0)
    model.load_state_dict(sd, strict=True)
    for m in model.modules():
        if hasattr(m, "temperature"):
            m.temperature = temp
    return model.to(DEV)


def test_dymn20_eval_matches_oracle(dymn20_case):
    d = dymn20_case
    with torch.no_grad():
        ref_logits, ref_fmaps = d["fwd"](d["sd"], d["x"], return_fmaps=True)
    model = _dymn20(d["sd"], d["temp"]).eval()
    with torch.no_grad():
        logits, fmaps = model(d["x"].to(DEV), return_fmaps=True)
        logits2, feat = model(d["x"].to(DEV))
    assert len(fmaps) == 17 and feat.shape == (4, 1920)
    for i, (a, b) in enumerate(zip(fmaps, ref_fmaps)):
        assert a.shape == b.shape
        assert _rel(a, b) < 2e-4, (i, _rel(a, b))
    # north_star: logits within 1e-3 ABSOLUTE of the reference CPU path (|logit| reaches ~12 on this model)
    e1, e2 = np.abs(logits.cpu().numpy() - ref_logits.numpy()).max(), np.abs(logits2.cpu().numpy() - ref_logits.numpy()).max()
    print(f"dymn20 eval logits: max abs err {e1:.2e} / {e2:.2e} on |logit| <= {np.abs(ref_logits.numpy()).max():.1f}")
    assert e1 < 1e-3 and e2 < 1e-3, (e1, e2)


@pytest.mark.parametrize("temp", [1.0, 30.0])
@pytest.mark.parametrize("prec", ["fp32", "auto"])
def test_dymn20_train_step_matches_oracle(dymn20_case, dymn20_case_t30, prec, temp):
    """temp = 30 (the reference's initial temperature, a conditioned network): FIXED bars for every tensor, no round-off
    escape.  temp = 1: the bars yield to 4x the tensor's own one-ulp floor, but ONLY for the kernel-attention heads
    (`_ATTENTION_HEAD`) - the escape is a named list that cannot grow silently - and those very tensors are held to the
    fixed bars by the temp = 30 run.
    fp32: exact fp32 GEMMs - SURVEY 8c's gradient bar (rel-L2 <= 1e-2 per tensor).  auto (what bench.py times): split
    bf16 operands from C_in = 40 on, ~1e-5 relative noise per GEMM, i.e. ~100x as many activation-kink flips as fp32
    re-association: 5e-2 per tensor, 2e-2 median (see test_gpu_dymn.py::test_dymn10_train_step_matches_oracle).  Either bar
    yields to 4x the round-off floor of the tensor (`_ulp_noise`): at temperature 1 the kernel attention of this network
    turns one-ulp input noise into percent-level changes of the 4-value attention-bias gradients."""
    d = dymn20_case if temp == 1.0 else dymn20_case_t30
    y = (torch.rand(4, 527, generator=torch.Generator().manual_seed(5)) < 0.01).float()
    keep = (torch.rand(4, 2560, generator=torch.Generator().manual_seed(6)) < 0.8).float()
    sdr = _grad_state(d["sd"])
    stats = {}
    logits_ref, _ = d["fwd"](sdr, d["x"], train=True, stats=stats, drop_mask=keep)
    loss_ref = F.binary_cross_entropy_with_logits(logits_ref, y)
    loss_ref.backward()

    model = _dymn20(d["sd"], d["temp"]).train()
    model.train_precision = prec
    model._drop_mask_override = keep
    logits, emb = model(d["x"].to(DEV))
    loss = F.binary_cross_entropy_with_logits(logits, y.to(DEV))
    loss.backward()
    # round-off floor of this network: the same step on the input with one-ulp noise, against the step above
    ctrl = _dymn20(d["sd"], d["temp"]).train()
    ctrl.train_precision = prec
    ctrl._drop_mask_override = keep
    F.binary_cross_entropy_with_logits(ctrl(_ulp_noise(d["x"]).to(DEV))[0], y.to(DEV)).backward()
    noise = {n: _rel(pc.grad, p.grad.cpu()) for (n, p), (_, pc) in zip(model.named_parameters(), ctrl.named_parameters())}
    del ctrl
    assert abs(loss.item() - loss_ref.item()) < 1e-4 * max(1.0, abs(loss_ref.item()))
    lerr = float((logits.detach().cpu() - logits_ref.detach()).abs().max())
    assert lerr < 1e-3, lerr                                           # absolute, |logit| up to ~12
    gmax = max(float(v.grad.norm()) for v in sdr.values() if getattr(v, "grad", None) is not None)
    rels, bad, excused = [], [], []
    bar = 1e-2 if prec == "fp32" else 5e-2
    for name, p in model.named_parameters():
        ref = sdr[name].grad
        assert p.grad is not None, name
        if float(ref.norm()) < 1e-4 * gmax:      # zero-gradient BN biases, cancellation-dominated attention logits
            continue
        r = _rel(p.grad, ref)
        rels.append(r)
        if r <= bar:
            continue
        if name.endswith(_ATTENTION_HEAD):
            # the uncapped floor: at temperature 1 only (at T = 30 these very tensors must meet the fixed bar)
            if temp == 1.0 and r <= _floor(noise, name):
                excused.append((name, round(r, 4), round(noise[name], 4)))
                continue
        elif r <= min(_floor(noise, name), 2 * bar):
            # any other tensor: at most 2x the bar, and only where the one-ulp control moves it that much itself
            # (measured: one BatchNorm weight at 1.05e-2 with a floor of 4 x 1.04e-2 in the fp32 run at T = 30)
            excused.append((name, round(r, 4), round(noise[name], 4)))
            continue
        bad.append((name, r, noise[name]))
    nmed = float(np.median(list(noise.values())))
    print(f"dymn20 train step [{prec}, T = {temp}]: logits max abs err {lerr:.2e}, gradient rel-L2 median {np.median(rels):.2e}, "
          f"max {max(rels):.2e}  (one-ulp input noise on the same step: median {nmed:.2e}, max {max(noise.values()):.2e}); "
          f"{len(excused)} tensors above the fixed bar but inside their round-off floor (attention heads at T = 1: 4x the "
          f"one-ulp control; others: at most 2x the bar): {excused[:6]}")
    assert not bad, bad[:8]
    med_bar = 3e-3 if prec == "fp32" else 2e-2
    assert float(np.median(rels)) < (med_bar if temp != 1.0 else max(med_bar, 4 * nmed)), float(np.median(rels))
    msd = model.state_dict()
    for k, v in stats.items():
        assert _rel(msd[k], v) < 1e-4, k


# ------------------------------------------------------------------ configs[2]: mn40 train step, fp32 and bf16
@pytest.fixture(scope="module")
def mn40_case():
    wave = torch.cat([synth.parity_clips(320000, seed=21), synth.parity_clips(320000, seed=22)[[0, 2, 4]]])   # 8 clips
    x = O.mel_forward(wave).unsqueeze(1)
    fwd = lambda sd, xm, **k: O.mn_forward(sd, xm, width_mult=4.0, **k)
    sd = synth.calibrate(synth.synth_state(synth.mn_shapes(4.0), seed=0), fwd, x)
    y = (torch.rand(8, 527, generator=torch.Generator().manual_seed(2)) < 0.01).float()
    keep = (torch.rand(8, 5120, generator=torch.Generator().manual_seed(3)) < 0.8).float()
    sdr = _grad_state(sd)
    stats = {}
    logits, _ = fwd(sdr, x, train=True, stats=stats, drop_mask=keep)
    loss = F.binary_cross_entropy_with_logits(logits, y)
    loss.backward()
    grads = {k: v.grad for k, v in sdr.items() if getattr(v, "grad", None) is not None}
    return dict(sd=sd, x=x, y=y, keep=keep, loss=float(loss), logits=logits.detach(), grads=grads, stats=stats)


def _mn40_step(d, precision, act_storage="fp32"):
    model = _quiet(mn_mod.get_model, width_mult=4.0)
    model.load_state_dict(d["sd"], strict=True)
    model.to(DEV).train()
    model.train_precision = precision
    model.act_storage = act_storage
    model._drop_mask_override = d["keep"]
    logits, _ = model(d["x"].to(DEV))
    loss = F.binary_cross_entropy_with_logits(logits, d["y"].to(DEV))
    loss.backward()
    gmax = max(float(g.norm()) for g in d["grads"].values())
    rels = {}
    for name, p in model.named_parameters():
        ref = d["grads"][name]
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
        if float(ref.norm()) < 1e-5 * gmax:      # project-BN biases: true gradient is exactly zero
            continue
        rels[name] = _rel(p.grad, ref)
    return model, loss.item(), logits.detach().cpu(), rels


def test_mn40_train_step_fp32_matches_oracle(mn40_case):
    """Exact-fp32 arithmetic (train_precision='fp32': fp32 MFMA forward / data-gradient GEMMs AND the exact fp32 weight
    gradient kernel) against torch-CPU autograd over the oracle: same bars as mn10 (3 % per tensor, 1 % median)."""
    d = mn40_case
    model, loss, logits, rels = _mn40_step(d, "fp32")
    assert abs(loss - d["loss"]) < 1e-5 * max(1.0, abs(d["loss"]))
    assert float((logits - d["logits"]).abs().max()) < 1e-3 * max(1.0, float(d["logits"].abs().max()))
    bad = [(n, r) for n, r in rels.items() if r > 3e-2]
    assert not bad, bad[:8]
    assert float(np.median(list(rels.values()))) < 1e-2
    # SURVEY 8c's per-tensor bar is 1e-2; the fp32 CPU oracle itself sits 1 - 3e-2 away from its fp64 evaluation on a few
    # tensors of this random-weight net (activation kinks: DESIGN 5).  Which tensors exceed 1e-2 is printed, and there may
    # be only a handful of them - a systematic error moves the median, an indexing error moves everything
    above = sorted(((round(r, 4), n) for n, r in rels.items() if r > 1e-2), reverse=True)
    print(f"mn40 fp32 train step: {len(above)} of {len(rels)} gradient tensors above 1e-2 (max 3e-2 allowed): {above[:8]}")
    assert len(above) <= max(3, len(rels) // 20), above
    msd = model.state_dict()
    for k, v in d["stats"].items():
        assert _rel(msd[k], v) < 1e-5, k


def test_mn40_train_step_auto_matches_oracle(mn40_case):
    """The default training arithmetic ('auto': split-operand bf16x3 GEMMs from C_in >= 40, bf16x3 weight gradients)."""
    d = mn40_case
    _, loss, logits, rels = _mn40_step(d, "auto")
    assert abs(loss - d["loss"]) < 2e-5 * max(1.0, abs(d["loss"]))
    assert float((logits - d["logits"]).abs().max()) < 1e-3 * max(1.0, float(d["logits"].abs().max()))
    bad = [(n, r) for n, r in rels.items() if r > 3e-2]
    assert not bad, bad[:8]
    assert float(np.median(list(rels.values()))) < 1e-2


@pytest.mark.parametrize("storage", ["fp32", "bf16"])
def test_mn40_train_step_bf16_tracks_oracle(mn40_case, storage):
    """storage = "bf16": BASELINE configs[2] as stated - bf16 ACTIVATION STORAGE (`model.act_storage`; the wide tensors z_e,
    z_d, y_d and the gradients arriving at them live in bf16 in HBM) on top of the bf16 GEMM operands, against the oracle's
    emulation of exactly that (`O.emulate_bf16_pointwise(storage=True)`) and against the fp32 oracle.
    storage = "fp32": 1x1 GEMMs on plain bf16 operands (2^-9 relative round-off per operand), anchored on the
    ORACLE, not on our own fp32 path:
      (a) an oracle evaluation of the SAME arithmetic (`O.emulate_bf16_pointwise`: bf16-rounded operands, fp32
          accumulation; that the kernel computes exactly this per layer is test_pw_conv_bf16's tight check) - loss and
          logits must agree at the level where only rounding-boundary flips differ (measured on MI355X: a value within
          fp32 noise of a bf16 rounding boundary rounds the other way in ~2.5e-4 of the elements, which ~30 bf16 layers
          and the activation kinks amplify to ~3e-3 of the logit scale; the same mechanism as the fp32 gradient budget
          of SURVEY 8c, with a 2^16 times larger seed);
      (b) the fp32 oracle: bf16 noise of this 47-layer synthetic net is large (the emulated oracle's own gradients are
          ~20 % median rel-L2 away from the fp32 ones), so the criterion is "not further from the fp32 oracle than the
          oracle's bf16 evaluation is" (x1.25 + 1 %), for the gradients and for the logits."""
    d = mn40_case
    sdr = _grad_state(d["sd"])
    fwd = lambda sd, xm, **k: O.mn_forward(sd, xm, width_mult=4.0, **k)
    with O.emulate_bf16_pointwise(storage=storage == "bf16"):
        logits_e, _ = fwd(sdr, d["x"], train=True, stats={}, drop_mask=d["keep"])
        loss_e = F.binary_cross_entropy_with_logits(logits_e, d["y"])
        loss_e.backward()
    logits_e = logits_e.detach()
    gmax = max(float(g.norm()) for g in d["grads"].values())
    emu_vs_fp32 = {n: _rel(v.grad, d["grads"][n]) for n, v in sdr.items()
                   if getattr(v, "grad", None) is not None and float(d["grads"][n].norm()) >= 1e-5 * gmax}

    model, loss, logits, hip_vs_fp32 = _mn40_step(d, "bf16", act_storage=storage)
    if storage == "bf16":                 # every block really ran on bf16 storage
        from efficientat_amd import ops
        B, _, F0, T0 = d["x"].shape
        f, t = (F0 - 1) // 2 + 1, (T0 - 1) // 2 + 1
        for blk in model.features[1:-1]:
            c = blk.cnf
            assert ops.b16_block_ok(B, c.expanded_channels, f, t, c.kernel, c.stride), (f, t, c.kernel, c.stride)
            f, t = ops.conv_out(f, c.kernel, c.stride), ops.conv_out(t, c.kernel, c.stride)
    scale = float(d["logits"].abs().max())
    # (a) same arithmetic, oracle vs HIP
    assert abs(loss - float(loss_e)) < 2e-3 * abs(float(loss_e)), (loss, float(loss_e))
    assert float((logits - logits_e).abs().max()) < 2e-2 * scale
    hip_vs_emu = np.array([_rel(p.grad, sdr[n].grad) for n, p in model.named_parameters() if n in emu_vs_fp32])
    assert np.isfinite(hip_vs_emu).all()
    ev = np.array(list(emu_vs_fp32.values()))
    assert float(np.median(hip_vs_emu)) < float(np.median(ev)), (float(np.median(hip_vs_emu)), float(np.median(ev)))
    # (b) bf16 noise vs the fp32 oracle: not larger than the emulated oracle's own
    assert abs(loss - d["loss"]) < 2e-2 * abs(d["loss"]), (loss, d["loss"])
    hv = np.array(list(hip_vs_fp32.values()))
    assert float(np.median(hv)) < 1.25 * float(np.median(ev)) + 1e-2, (float(np.median(hv)), float(np.median(ev)))
    e_hip, e_emu = float((logits - d["logits"]).abs().max()), float((logits_e - d["logits"]).abs().max())
    assert e_hip < 1.5 * e_emu + 1e-2 * scale, (e_hip, e_emu)


# ------------------------------------------------------------------ train-step parity AT THE MEASURED BATCH SIZES
# bench.py times the mn10 step at 256 clips per GPU and the mn40 / dymn20 steps at 128.  Torch-CPU autograd over the
# oracle at those sizes needs minutes and tens of GB, so the large batch is built from COPIES of the small one: with the
# mean-reduced loss, a batch of r copies of n clips has exactly the batch statistics, per-sample activations, loss and
# parameter gradients of the n-clip batch (BatchNorm couples the samples only through statistics that the copies leave
# unchanged).  The n-clip step is pinned on the oracle; the r x n step runs every kernel in the regime the bench measures
# (multi-sample BN reducers, split-K weight gradients, G samples per wave, full-size grids) and must reproduce it.
def _tiled_step(model, x, y, keep, reps):
    model._drop_mask_override = keep.repeat(reps, 1)
    logits, _ = model(x.repeat(reps, 1, 1, 1).to(DEV))
    loss = F.binary_cross_entropy_with_logits(logits, y.repeat(reps, 1).to(DEV))
    loss.backward()
    return loss.item(), logits.detach().cpu(), {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()}


def _check_tiled(small, large, n, stats_small, stats_large, grad_tol=3e-2, fwd_tol=2e-5, med_tol=5e-3, noise=None, max_above=None):
    """Forward quantities (loss, logits, running statistics) must agree to round-off.  Gradients: the two runs sum in
    different orders (split-K, atomics, Gram-matrix statistics), so an activation within ~1e-7 of a ReLU / Hardswish kink
    may take the other branch in ALL copies at once - the same mechanism, and the same size (1e-3 ... 1e-2 of a tensor's
    norm per flipped element), as between the oracle in fp32 and in fp64: per-tensor bound 3 %, median 0.5 %."""
    loss_s, logits_s, g_s = small
    loss_l, logits_l, g_l = large
    assert abs(loss_l - loss_s) < fwd_tol * max(1.0, abs(loss_s)), (loss_l, loss_s)
    reps = logits_l.shape[0] // n
    scale = max(1.0, float(logits_s.abs().max()))
    for r in (0, reps // 2, reps - 1):                      # first, middle and last copy of the batch
        assert float((logits_l[r * n:(r + 1) * n] - logits_s).abs().max()) < 10 * fwd_tol * scale
    gmax = max(float(v.norm()) for v in g_s.values())
    rels = []
    for name, ref in g_s.items():
        if float(ref.norm()) < 1e-4 * gmax:                  # zero-gradient project-BN biases: round-off only
            continue
        rels.append((_rel(g_l[name], ref), name))
    # noise: {name: rel} of a second small-batch step on one-ulp-noisy input (`_ulp_noise`) - the bars yield to 4x that
    bad = [(r, nm) for r, nm in rels if r >= max(grad_tol, _floor(noise, nm))]
    assert not bad, max(bad)
    above = sorted(((round(r, 4), nm) for r, nm in rels if r > 1e-2), reverse=True)
    print(f"tiled batch vs oracle-pinned batch: {len(above)} of {len(rels)} gradient tensors above 1e-2: {above[:6]}")
    if max_above is not None:                                  # (fp32-class arithmetic of the MN plans; bf16 / DyMN at T = 1: printed)
        assert len(above) <= max_above, above
    nmed = float(np.median([noise[nm] for _, nm in rels])) if noise else 0.0
    assert float(np.median([r for r, _ in rels])) < max(med_tol, 4 * nmed), (float(np.median([r for r, _ in rels])), nmed)
    for k, v in stats_small.items():                         # running statistics (unbiased factor n/(n-1) differs by ~1e-6;
        assert _rel(stats_large[k], v) < 10 * fwd_tol, k     #  the context-generator norms of DyMN see n = 4 x 8 values per channel)


def test_mn10_train_step_at_batch_256_reproduces_the_oracle_pinned_batch(golden_dir):
    g = np.load(os.path.join(golden_dir, "mn10_ref.npz"))
    sd = synth.synth_state(synth.mn_shapes(1.0), seed=0)
    for k in g.files:
        if k.startswith("bn/"):
            sd[k[3:]] = torch.from_numpy(g[k])
    clips = torch.cat([synth.parity_clips(320000, seed=1234), synth.parity_clips(320000, seed=77)[[0, 2, 4]]])     # 8 clips
    x = O.mel_forward(clips).unsqueeze(1)
    y = (torch.rand(8, 527, generator=torch.Generator().manual_seed(2)) < 0.01).float()
    keep = (torch.rand(8, 1280, generator=torch.Generator().manual_seed(3)) < 0.8).float()
    # the 8-clip step against torch-CPU autograd over the oracle
    sdr = _grad_state(sd)
    stats = {}
    logits_ref, _ = O.mn_forward(sdr, x, train=True, stats=stats, drop_mask=keep)
    loss_ref = F.binary_cross_entropy_with_logits(logits_ref, y)
    loss_ref.backward()
    runs, bufs = {}, {}
    for reps in (1, 32, "ctrl"):                             # 8 and 256 clips; 8 clips with one-ulp input noise
        model = _quiet(mn_mod.get_model, width_mult=1.0)
        model.load_state_dict(sd, strict=True)
        model.to(DEV).train()
        model.train_precision = "auto"                       # the arithmetic bench.py times
        runs[reps] = _tiled_step(model, _ulp_noise(x) if reps == "ctrl" else x, y, keep, 1 if reps == "ctrl" else reps)
        bufs[reps] = {k: v.detach().cpu() for k, v in model.state_dict().items() if k.endswith(("running_mean", "running_var"))}
    loss8, logits8, g8 = runs[1]
    assert abs(loss8 - float(loss_ref)) < 2e-5 * max(1.0, abs(float(loss_ref)))
    assert float((logits8 - logits_ref.detach()).abs().max()) < 1e-3
    gmax = max(float(v.grad.norm()) for v in sdr.values() if getattr(v, "grad", None) is not None)
    rels = [_rel(g8[n], sdr[n].grad) for n in g8 if float(sdr[n].grad.norm()) >= 1e-5 * gmax]
    assert max(rels) < 3e-2 and float(np.median(rels)) < 1e-2, (max(rels), float(np.median(rels)))
    for k, v in stats.items():
        if k.endswith(("running_mean", "running_var")):
            assert _rel(bufs[1][k], v) < 1e-4, k
    noise = {n: _rel(runs["ctrl"][2][n], g) for n, g in g8.items()}
    _check_tiled(runs[1], runs[32], 8, bufs[1], bufs[32], noise=noise, max_above=16)      # (of 159 tensors)


@pytest.mark.parametrize("precision", ["auto", "bf16"])
def test_mn40_train_step_at_batch_128_reproduces_the_oracle_pinned_batch(mn40_case, precision):
    """configs[2] at its batch size: 16 copies of the 8 clips whose step test_mn40_train_step_* pin on the oracle."""
    d = mn40_case
    runs, bufs = {}, {}
    for reps in (1, 16):
        model = _quiet(mn_mod.get_model, width_mult=4.0)
        model.load_state_dict(d["sd"], strict=True)
        model.to(DEV).train()
        model.train_precision = precision
        runs[reps] = _tiled_step(model, d["x"], d["y"], d["keep"], reps)
        bufs[reps] = {k: v.detach().cpu() for k, v in model.state_dict().items() if k.endswith(("running_mean", "running_var"))}
        del model
        torch.cuda.empty_cache()
    # bf16 operands: a value within fp32 noise of a bf16 rounding boundary may round the other way in the other regime
    if precision == "auto":
        _check_tiled(runs[1], runs[16], 8, bufs[1], bufs[16], max_above=16)
    else:
        # gradients: bf16 round-off is amplified to ~20 % per tensor by this synthetic 47-layer net even between two
        # evaluations of the oracle (test_mn40_train_step_bf16_tracks_oracle); here only that the two regimes stay in
        # that class
        _check_tiled(runs[1], runs[16], 8, bufs[1], bufs[16], grad_tol=1.0, fwd_tol=1e-3, med_tol=0.25)


@pytest.mark.parametrize("prec", ["fp32", "auto"])
def test_dymn20_train_step_at_batch_128_reproduces_the_oracle_pinned_batch(dymn20_case, prec):
    """configs[3] at its batch size: 32 copies of the 4 clips of test_dymn20_train_step_matches_oracle."""
    d = dymn20_case
    y = (torch.rand(4, 527, generator=torch.Generator().manual_seed(5)) < 0.01).float()
    keep = (torch.rand(4, 2560, generator=torch.Generator().manual_seed(6)) < 0.8).float()
    runs, bufs = {}, {}
    for reps in (1, 32, "ctrl"):
        model = _dymn20(d["sd"], d["temp"]).train()
        model.train_precision = prec
        runs[reps] = _tiled_step(model, _ulp_noise(d["x"]) if reps == "ctrl" else d["x"], y, keep, 1 if reps == "ctrl" else reps)
        bufs[reps] = {k: v.detach().cpu() for k, v in model.state_dict().items() if k.endswith(("running_mean", "running_var"))}
        del model
        torch.cuda.empty_cache()
    # per-tensor bar 5e-2 as in the dymn10 / dymn20 oracle tests of the default arithmetic: the 4-value bias gradients of
    # the kernel-attention Linears are sums over the batch of softmax-Jacobian rows (which sum to zero over K) - the
    # tensors with the least signal per kink flip
    # (auto: the ~1e-5 noise of the split-operand GEMMs differs between the two regimes' tilings - more kink flips)
    noise = {n: _rel(runs["ctrl"][2][n], g) for n, g in runs[1][2].items()}
    _check_tiled(runs[1], runs[32], 4, bufs[1], bufs[32], grad_tol=5e-2, med_tol=5e-3 if prec == "fp32" else 1e-2, noise=noise)


# ------------------------------------------------------------------ DISTINCT clips at the measured batch sizes: permutation
# The tiled tests above fill the batch with copies, so an indexing error whose period divides the copy length (a kernel
# reading sample b +- 8, a wrong plane stride inside a several-samples-per-wave mode, a split-K slot mixing samples) returns
# the right values and stays invisible.  Here every clip of the batch is different: a permutation of the batch must permute
# the logits and leave the loss, every parameter gradient and every BatchNorm running statistic unchanged up to the order
# of the floating-point sums (plus, rarely, an activation within ~1e-7 of a kink changing side: SURVEY 8c).
def _distinct_batch(B, n_out, seed):
    g = torch.Generator().manual_seed(seed)
    amp = 10.0 ** (-2.0 + 2.0 * torch.rand(B, 1, generator=g))                    # 0.01 ... 1 per clip
    wave = (amp * torch.randn(B, 320000, generator=g)).clamp_(-1, 1)
    tone = torch.sin(torch.arange(320000)[None, :] * (2 * 3.14159265 * (50.0 + 200.0 * torch.arange(B)[:, None]) / 32000.0))
    wave = (0.7 * wave + 0.05 * tone * (torch.arange(B)[:, None] % 3 == 0)).clamp_(-1, 1)
    mel = AugmentMelSTFT(n_mels=128, sr=32000, win_length=800, hopsize=320, n_fft=1024, freqm=0, timem=0).to(DEV).eval()
    with torch.no_grad():
        x = mel(wave.to(DEV)).unsqueeze(1)
    y = (torch.rand(B, 527, generator=g) < 0.01).float().to(DEV)
    keep = (torch.rand(B, n_out, generator=g) < 0.8).float()
    return x, y, keep


def _permutation_check(make_model, B, n_hidden, seed):
    """Three steps from the same weights: the batch as it is, the batch permuted, and (control) the batch as it is with
    one-ulp noise on the input (x * (1 + 2^-23 * N(0,1))).  The synthetic networks amplify round-off by 10^3 ... 10^4
    (i.i.d. 1e-5 on the mel moves logits by up to 5e-4; batch statistics are perturbed coherently), so the
    absolute size of the permutation's effect says little; what it must NOT exceed is the effect of round-off-sized noise
    (x 5 + a floor).  A sample-indexing error puts a different clip's data into some sum: an O(1) error in the affected
    tensors, orders of magnitude above either."""
    x, y, keep = _distinct_batch(B, n_hidden, seed)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(seed + 1))
    ident = torch.arange(B)
    noise = 1.0 + 2.0 ** -23 * torch.randn(x.shape, generator=torch.Generator().manual_seed(seed + 2)).to(DEV)
    outs = []
    for order, xin in ((ident, x), (perm, x), (ident, x * noise)):
        model = make_model()
        model._drop_mask_override = keep[order]
        logits, _ = model(xin[order.to(DEV)].contiguous())
        loss = F.binary_cross_entropy_with_logits(logits, y[order.to(DEV)])
        loss.backward()
        grads = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()}
        bufs = {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if k.endswith(("running_mean", "running_var"))}
        outs.append((loss.item(), logits.detach().cpu(), grads, bufs))
        del model
        torch.cuda.empty_cache()
    (l0, lg0, g0, b0), (l1, lg1, g1, b1), (l2, lg2, g2, b2) = outs
    scale = max(1.0, float(lg0.abs().max()))
    gmax = max(float(v.norm()) for v in g0.values())
    names = [n for n, v in g0.items() if float(v.norm()) >= 1e-4 * gmax]

    def dev(lg, g, order):
        rels = [_rel(g[n], g0[n]) for n in names]
        return float((lg - lg0[order]).abs().max()), max(rels), float(np.median(rels))

    p_log, p_max, p_med = dev(lg1, g1, perm)
    c_log, c_max, c_med = dev(lg2, g2, ident)
    print(f"{B} distinct clips - permutation: logits {p_log:.1e}, gradient rel-L2 median {p_med:.1e} / max {p_max:.1e};  "
          f"one-ulp input noise (control): logits {c_log:.1e}, median {c_med:.1e} / max {c_max:.1e}")
    assert abs(l1 - l0) < 5 * abs(l2 - l0) + 2e-6 * max(1.0, abs(l0)), (l0, l1, l2)
    assert p_log < 5 * c_log + 1e-5 * scale, (p_log, c_log)          # every clip keeps ITS logits wherever it sits in the batch
    assert p_max < 5 * c_max + 1e-3, (p_max, c_max)
    assert p_med < 5 * c_med + 1e-4, (p_med, c_med)
    for k, v in b0.items():
        assert _rel(b1[k], v) < 5 * _rel(b2[k], v) + 1e-5, k


def test_mn10_train_step_batch_256_distinct_clips_permutation(golden_dir):
    g = np.load(os.path.join(golden_dir, "mn10_ref.npz"))
    sd = synth.synth_state(synth.mn_shapes(1.0), seed=0)
    for k in g.files:
        if k.startswith("bn/"):
            sd[k[3:]] = torch.from_numpy(g[k])

    def make():
        model = _quiet(mn_mod.get_model, width_mult=1.0)
        model.load_state_dict(sd, strict=True)
        model.train_precision = "auto"                       # the arithmetic bench.py times
        return model.to(DEV).train()

    _permutation_check(make, 256, 1280, seed=101)


def test_mn40_train_step_batch_128_distinct_clips_permutation(mn40_case):
    d = mn40_case

    def make():
        model = _quiet(mn_mod.get_model, width_mult=4.0)
        model.load_state_dict(d["sd"], strict=True)
        model.train_precision = "auto"
        return model.to(DEV).train()

    _permutation_check(make, 128, 5120, seed=202)


def test_dymn20_train_step_batch_128_distinct_clips_permutation(dymn20_case):
    d = dymn20_case
    _permutation_check(lambda: _dymn20(d["sd"], d["temp"]).train(), 128, 2560, seed=303)


# ------------------------------------------------------------------ BASELINE widths against the reference's own outputs
@pytest.mark.parametrize("tag", ["mn40", "dymn20"])
def test_baseline_width_models_match_reference_goldens(tag, golden_dir):
    """mn40 (configs[2]) and dymn20 (configs[3]) on the HIP path against vectors produced by the UNMODIFIED reference at
    those widths (oracle/make_golden.py:golden_widths, 3 s clips): eval logits / features, and one training step at the
    reference's initial DynamicConv temperature 30 - logits, loss, the norm and 8 sampled entries of every gradient."""
    from tests.test_oracle_golden import WIDTHS, width_state
    kind, _ = WIDTHS[tag]
    model, sd, g = width_state(tag, golden_dir)
    model.load_state_dict(sd, strict=True)
    model.to(DEV)
    x = O.mel_forward(synth.parity_clips(96000, seed=45)).unsqueeze(1).to(DEV)

    def set_temp(t):
        for m in model.modules():
            if hasattr(m, "temperature"):
                m.temperature = t
    set_temp(1.0)
    model.eval()
    with torch.no_grad():
        logits, feats = model(x)
    for got, key in ((logits, "logits"), (feats, "features")):
        ref = g[f"{tag}/{key}"]
        assert np.abs(got.cpu().numpy() - ref).max() < 1e-3 * max(1.0, np.abs(ref).max()), key
    set_temp(30.0)
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    if kind == "mn":
        model.train_precision = "fp32"
    tl, _ = model(x)
    loss = F.binary_cross_entropy_with_logits(tl, torch.from_numpy(g[f"{tag}/train_labels"]).to(DEV))
    loss.backward()
    assert abs(loss.item() - float(g[f"{tag}/train_loss"])) < 1e-4
    ref_tl = g[f"{tag}/train_logits"]
    assert np.abs(tl.detach().cpu().numpy() - ref_tl).max() < 1e-3 * max(1.0, np.abs(ref_tl).max())
    gmax = max(float(g[k]) for k in g.files if k.startswith(f"{tag}/gnorm/"))
    rels = []
    for name, p in model.named_parameters():
        ref = float(g[f"{tag}/gnorm/{name}"])
        if ref <= 1e-4 * gmax:
            continue
        rels.append(abs(float(p.grad.norm()) - ref) / ref)
        vals, idx = g[f"{tag}/gval/{name}"], g[f"{tag}/gidx/{name}"]
        got = p.grad.reshape(-1)[torch.from_numpy(idx).to(DEV)].cpu().numpy()
        # single entries scatter more than the norm (the fp32 kink flips of SURVEY 8c): a coarse bound that still catches a
        # transposed / mis-indexed gradient
        assert np.abs(got - vals).max() < 0.25 * max(np.abs(vals).max(), ref / np.sqrt(p.numel())), name
    assert max(rels) < 5e-2 and float(np.median(rels)) < 1e-2, (max(rels), float(np.median(rels)))
    msd = model.state_dict()
    for k in g.files:
        if k.startswith(f"{tag}/bn_after/"):
            assert _rel(msd[k[len(tag) + 10:]], torch.from_numpy(g[k])) < 1e-4, k


# ------------------------------------------------------------------ configs[3] on bf16 activation storage
def test_dymn20_train_step_bf16_storage_tracks_oracle(dymn20_case_t30):
    """dymn20 under train_precision = 'bf16' + act_storage = 'bf16' (the byte contract SURVEY 8(d) quotes for configs[3]; the
    reference's 16-bit surface is Lightning `precision=16`, ex_pl_audioset.py:287-293): bf16 GEMM operands and the wide tensors
    of every dynamic block (z_e, z_d, the DyReLU * CoordAtt output, and the gradients arriving at them) stored in bf16, against
      (a) the oracle's emulation of exactly those roundings (`O.emulate_bf16_pointwise(storage=...)` over `O.dymn_forward`):
          loss / logits agree at the bf16-rounding-boundary level, gradients closer to the emulation than the emulation is to
          the fp32 oracle;
      (b) the fp32 oracle: not further away than the emulated oracle itself is (x1.25 + 1 %) - the criterion of
          test_mn40_train_step_bf16_tracks_oracle.  Temperature 30 (the reference's starting temperature): at T = 1 the
          kernel-attention softmax of this synthetic network turns bf16 noise into O(1) changes of the attention (see
          _ATTENTION_HEAD above), which says nothing about the kernels."""
    from efficientat_amd import ops
    d = dymn20_case_t30
    y = (torch.rand(4, 527, generator=torch.Generator().manual_seed(5)) < 0.01).float()
    keep = (torch.rand(4, 2560, generator=torch.Generator().manual_seed(6)) < 0.8).float()
    # fp32 oracle
    sdf = _grad_state(d["sd"])
    logits_f, _ = d["fwd"](sdf, d["x"], train=True, stats={}, drop_mask=keep)
    loss_f = F.binary_cross_entropy_with_logits(logits_f, y)
    loss_f.backward()
    # which blocks the plan stores in bf16 (the emulation follows the same list)
    blocks, _ = O.block_table(2.0)
    B, _, F0, T0 = d["x"].shape
    f, t = (F0 - 1) // 2 + 1, (T0 - 1) // 2 + 1
    st16 = []
    for i, c in enumerate(blocks):
        ok = ops.dyn_b16_block_ok(B, c["cin"], c["cexp"], c["cout"], f, t, c["k"], c["stride"]) and \
            (c["cexp"] != c["cin"] or (t > 128 and c["k"] == 3 and c["stride"] == 1))
        if ok:
            st16.append(i)
        f, t = ops.conv_out(f, c["k"], c["stride"]), ops.conv_out(t, c["k"], c["stride"])
    assert len(st16) == 15, st16                     # the whole network runs on bf16 storage at the bench geometry
    sde = _grad_state(d["sd"])
    with O.emulate_bf16_pointwise(storage=set(st16)):
        logits_e, _ = d["fwd"](sde, d["x"], train=True, stats={}, drop_mask=keep)
        loss_e = F.binary_cross_entropy_with_logits(logits_e, y)
        loss_e.backward()
    logits_e, logits_f = logits_e.detach(), logits_f.detach()

    model = _dymn20(d["sd"], d["temp"]).train()
    model.train_precision = "bf16"
    model.act_storage = "bf16"
    model._drop_mask_override = keep
    logits, _ = model(d["x"].to(DEV))
    loss = F.binary_cross_entropy_with_logits(logits, y.to(DEV))
    loss.backward()
    logits = logits.detach().cpu()
    gmax = max(float(v.grad.norm()) for v in sdf.values() if getattr(v, "grad", None) is not None)
    names = [n for n, p in model.named_parameters() if float(sdf[n].grad.norm()) >= 1e-4 * gmax]
    for n, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
    gp = dict(model.named_parameters())
    emu_vs_f = np.array([_rel(sde[n].grad, sdf[n].grad) for n in names])
    hip_vs_e = np.array([_rel(gp[n].grad, sde[n].grad) for n in names])
    hip_vs_f = np.array([_rel(gp[n].grad, sdf[n].grad) for n in names])
    scale = float(logits_f.abs().max())
    e_he, e_hf, e_ef = (float((logits - logits_e).abs().max()), float((logits - logits_f).abs().max()),
                        float((logits_e - logits_f).abs().max()))
    print(f"dymn20 bf16 storage: loss hip {loss.item():.6f} / emulated {float(loss_e):.6f} / fp32 {float(loss_f):.6f}; logits max abs "
          f"hip-emu {e_he:.2e}, hip-fp32 {e_hf:.2e}, emu-fp32 {e_ef:.2e} on |logit| <= {scale:.1f}; gradient rel-L2 medians: hip-emu "
          f"{np.median(hip_vs_e):.3f}, hip-fp32 {np.median(hip_vs_f):.3f}, emu-fp32 {np.median(emu_vs_f):.3f}")
    # (a) the same arithmetic
    assert abs(loss.item() - float(loss_e)) < 2e-3 * abs(float(loss_e)), (loss.item(), float(loss_e))
    assert e_he < 2e-2 * scale, (e_he, scale)
    assert float(np.median(hip_vs_e)) < float(np.median(emu_vs_f)), (float(np.median(hip_vs_e)), float(np.median(emu_vs_f)))
    # (b) bf16 noise against the fp32 oracle: not larger than the emulated oracle's own
    assert abs(loss.item() - float(loss_f)) < 2e-2 * abs(float(loss_f))
    assert float(np.median(hip_vs_f)) < 1.25 * float(np.median(emu_vs_f)) + 1e-2
    assert e_hf < 1.5 * e_ef + 1e-2 * scale, (e_hf, e_ef)


# ------------------------------------------------------------------ configs[1] as bench.py times it: the captured forward
@pytest.mark.parametrize("streams", [1, 2])
def test_graphed_forward_reproduces_the_eager_forward_on_every_replay(mn10_b256, streams):
    """`graphs.GraphedForward` (what bench.py's `forward` leg replays: log-mel + mn10 eval forward in one hipGraph, the batch cut
    into `streams` sub-batches on concurrent HIP streams) against the eager forward of the same batch and against the oracle on
    the parity clips - on the first replay, on later replays, and after the input buffer was rewritten (replays must not
    depend on what a previous replay left behind)."""
    from efficientat_amd.graphs import GraphedForward
    d = mn10_b256
    model = _quiet(mn_mod.get_model, width_mult=1.0)
    model.load_state_dict(d["sd"], strict=True)
    model.to(DEV).eval()
    mel = _quiet(AugmentMelSTFT, freqm=0, timem=0).to(DEV).eval()
    wave = d["wave"][:64].clone()
    for i, s in enumerate([0, 1, 31, 32, 63]):
        wave[s] = d["clips"][i]
    wave = wave.to(DEV)
    with torch.no_grad():
        ref_l, ref_f = model(mel(wave).unsqueeze(1))
    fwd = GraphedForward(model, mel, wave, streams=streams)
    for r in range(3):
        lg, ft = fwd()
        torch.cuda.synchronize()
        assert float((lg - ref_l).abs().max()) < 5e-5 and float((ft - ref_f).abs().max()) < 5e-5, (r, float((lg - ref_l).abs().max()))
        assert float((lg[[0, 1, 31, 32, 63]].cpu() - d["ref"]).abs().max()) < 1e-3
    # another batch through the same graph, then the first one again
    w2 = torch.roll(wave, 7, dims=0)
    with torch.no_grad():
        ref2, _ = model(mel(w2).unsqueeze(1))
    lg2, _ = fwd(w2)
    assert float((lg2 - ref2).abs().max()) < 5e-5
    lg3, _ = fwd(wave)
    assert float((lg3 - ref_l).abs().max()) < 5e-5
