"""Generate tests/golden/*.npz by running the UNMODIFIED reference (/root/reference).

Run in the build container only (the reference does not exist on the GPU box):

    python oracle/make_golden.py

The reference's own modules (models/preprocess.py, models/mn/model.py, models/dymn/model.py)
are imported with the three third-party stand-ins of oracle/ref_shims on sys.path and
driven on CPU in fp32 on the seeded inputs / weights of oracle/synth.py.  The stored
vectors pin oracle/eat_oracle.py (tests/test_oracle_golden.py) and, through it, the HIP
path.  TEST INFRASTRUCTURE ONLY.
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path[:0] = [os.path.join(HERE, "ref_shims"), REF, ROOT]
os.chdir(REF)  # helpers/utils.py:38 opens metadata/class_labels_indices.csv relative to CWD

from oracle import synth  # noqa: E402
from models.preprocess import AugmentMelSTFT  # noqa: E402
with contextlib.redirect_stdout(io.StringIO()):
    from models.mn.model import get_model as get_mn  # noqa: E402
    from models.dymn.model import get_model as get_dymn  # noqa: E402
    from models.dymn.model import dymn as dymn_factory  # noqa: E402
    from models.dymn.dy_block import DynamicConv  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
T_EDGE = [0, 1, 2, 3, 499, 996, 997, 998, 999]


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def fmap_summary(fmaps, rng):
    """mean / std / 16 sampled entries per fmap (the full maps are tens of MB)."""
    out = {}
    for i, f in enumerate(fmaps):
        flat = f.reshape(-1)
        idx = rng.integers(0, flat.numel(), 16)
        out[f"fmap{i}_shape"] = np.array(f.shape)
        out[f"fmap{i}_mean"] = np.float64(f.double().mean())
        out[f"fmap{i}_std"] = np.float64(f.double().std())
        out[f"fmap{i}_idx"] = idx
        out[f"fmap{i}_val"] = flat[idx].numpy()
    return out


def grad_summary(model, rng):
    out = {}
    for name, p in model.named_parameters():
        g = p.grad.reshape(-1)
        idx = rng.integers(0, g.numel(), 8)
        out["gnorm/" + name] = np.float64(g.double().norm())
        out["gidx/" + name] = idx
        out["gval/" + name] = g[idx].numpy()
    return out


def golden_mel():
    mel = quiet(AugmentMelSTFT, freqm=0, timem=0)
    mel.eval()
    short = synth.parity_clips(32000, seed=77)
    full = synth.parity_clips(320000, seed=1234)
    with torch.no_grad():
        m_short = mel(short)
        m_full = mel(full)
    res = dict(short=m_short.numpy(), full_edge=m_full[:, :, T_EDGE].numpy(), t_edge=np.array(T_EDGE),
               full_mean=m_full.double().mean(dim=(1, 2)).numpy(),
               full_rowsum=m_full.double().sum(dim=2).numpy())
    # train mode: fmin/fmax jitter + freq/time masking, torch CPU RNG seeded
    mel_t = quiet(AugmentMelSTFT, freqm=48, timem=192)
    mel_t.train()
    torch.manual_seed(2024)
    with torch.no_grad():
        res["train_short_seed2024"] = mel_t(synth.parity_clips(64000, seed=78)).numpy()
    np.savez_compressed(os.path.join(OUT, "mel_ref.npz"), **res)
    print("mel_ref.npz", m_short.shape, m_full.shape)
    return mel


def bn_buffers(model):
    return {k: v.clone() for k, v in model.state_dict().items()
            if k.endswith(("running_mean", "running_var"))}


def calibrate_reference(model, x):
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    for m in bns:
        m.momentum = 1.0
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    with torch.no_grad():
        model(x)
    for m in bns:
        m.momentum = 0.01
        m.num_batches_tracked.zero_()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.2


def golden_model(kind, mel, width, tag, temp_eval=1.0, temp_train=30.0):
    get = get_mn if kind == "mn" else get_dymn
    shapes = (synth.mn_shapes if kind == "mn" else synth.dymn_shapes)(width)
    model = quiet(get, width_mult=width)
    sd = synth.synth_state(shapes, seed=0)
    model.load_state_dict(sd, strict=True)          # pins key names + shapes of oracle/synth.py

    def set_temp(t):
        for m in model.modules():
            if isinstance(m, DynamicConv):
                m.temperature = t

    with torch.no_grad():
        x_cal = mel(synth.calibration_clips()).unsqueeze(1)
        x = mel(synth.parity_clips(320000, seed=1234)).unsqueeze(1)
    set_temp(temp_eval)
    calibrate_reference(model, x_cal)
    res = {"bn/" + k: v.numpy() for k, v in bn_buffers(model).items()}
    res["n_params"] = np.int64(sum(p.numel() for p in model.parameters()))
    res["n_state"] = np.int64(len(model.state_dict()))
    rng = np.random.Generator(np.random.PCG64(5))

    model.eval()
    with torch.no_grad():
        logits, feats = model(x)
        if kind == "mn":
            _, fmaps = model._forward_impl(x, return_fmaps=True)
        else:
            _, fmaps = model(x, return_fmaps=True)
    res.update(eval_logits=logits.numpy(), eval_features=feats.numpy(), temp_eval=np.float64(temp_eval))
    res.update(fmap_summary(fmaps, rng))

    # one train-mode step: BCE-with-logits vs seeded labels, gradients of every parameter
    set_temp(temp_train)
    model.train()
    drop = [m for m in model.modules() if isinstance(m, torch.nn.Dropout)][0]
    rec = {}
    drop.register_forward_pre_hook(lambda m, inp: rec.__setitem__("in", inp[0].detach().clone()))
    drop.register_forward_hook(lambda m, inp, out: rec.__setitem__("out", out.detach().clone()))
    y = torch.from_numpy((np.random.Generator(np.random.PCG64(9)).random((x.shape[0], 527)) < 0.01)
                         .astype(np.float32))
    torch.manual_seed(11)
    logits_t, _ = model(x)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits_t, y)
    loss.backward()
    keep = torch.where(rec["in"] != 0, (rec["out"] != 0).float(), torch.ones_like(rec["in"]))
    res.update(train_logits=logits_t.detach().numpy(), train_loss=np.float64(loss.item()),
               train_labels=y.numpy(), drop_keep=keep.numpy().astype(np.uint8),
               temp_train=np.float64(temp_train))
    res.update(grad_summary(model, rng))
    res.update({"bn_after/" + k: v.numpy() for k, v in bn_buffers(model).items()})
    np.savez_compressed(os.path.join(OUT, f"{tag}_ref.npz"), **res)
    print(tag, "params", int(res["n_params"]), "logits std", float(logits.std()),
          "absmax", float(logits.abs().max()), "train loss", loss.item())


VARIANTS = {   # non-default model variants: get_model kwargs of the reference
    "fc": dict(head_type="fully_convolutional"),
    "fc_s2211": dict(head_type="fully_convolutional", strides=(2, 2, 1, 1)),
    "att": dict(head_type="multihead_attention_pooling", multihead_attention_heads=4),
    # (se over 'f' cannot run in the reference: block_types.py:75 squeezes dim 2 twice, which leaves (B, 1, F, 1) for the
    #  frequency gate and fc1 then sees a last dimension of 1 - RuntimeError; 'c' and 't' work)
    "se_ct_max": dict(se_dims="ct", se_agg="max", input_dim_t=300),
    "se_t_avg": dict(se_dims="t", se_agg="avg", input_dim_t=300),
    "se_ct_min": dict(se_dims="ct", se_agg="min", input_dim_t=300),
    "se_ct_add": dict(se_dims="ct", se_agg="add", input_dim_t=300),
    "se_none": dict(se_dims="none"),
    "dilated_reduced": dict(dilated=True, reduced_tail=True),
}


def golden_variants(mel):
    """Eval logits / features of the reference for every variant on 3 s clips (T = 300 frames), weights from
    synth.synth_state over the reference model's own state_dict shapes, BN statistics calibrated by the reference."""
    res = {}
    with torch.no_grad():
        x_cal = mel(synth.calibration_clips(96000)).unsqueeze(1)
        x = mel(synth.parity_clips(96000, seed=41)).unsqueeze(1)
    for tag, kw in VARIANTS.items():
        model = quiet(get_mn, width_mult=1.0, **kw)
        sd = synth.synth_state(synth.shapes_of(model), seed=3)
        model.load_state_dict(sd, strict=True)
        calibrate_reference(model, x_cal)
        model.eval()
        with torch.no_grad():
            logits, feats = model(x)
        res[f"{tag}/logits"], res[f"{tag}/features"] = logits.numpy(), feats.numpy()
        res[f"{tag}/n_state"] = np.int64(len(model.state_dict()))
        for k, v in bn_buffers(model).items():
            res[f"{tag}/bn/{k}"] = v.numpy()
        print("variant", tag, "logits absmax", float(logits.abs().max()), "std", float(logits.std()))
    np.savez_compressed(os.path.join(OUT, "mn_variants_ref.npz"), **res)


DYMN_VARIANTS = {   # models/dymn/model.py:225-231; ablations of the dynamic block: models/dymn/dy_block.py:269-271
    "replace_se": dict(use_dy_blocks="replace_se"),
    "no_dyrelu": dict(no_dyrelu=True),
    "no_dyconv": dict(no_dyconv=True),
    "no_ca": dict(no_ca=True),
    "static": dict(no_dyrelu=True, no_dyconv=True, no_ca=True),
    # the fully-convolutional head of models/dymn/model.py:119-130 is reachable through the `dymn` factory only
    # (get_model has no head_type argument)
    "fc_head": dict(head_type="fully_convolutional"),
    # dilation 2 in the last three blocks (models/dymn/model.py:212-218,246-250): through the `dymn` factory as well
    "dilated": dict(dilated=True),
}


def golden_dymn_variants(mel):
    """Reference outputs of the DyMN variants on 3 s clips: eval logits / features (DynamicConv temperature 1) and the
    train-mode logits of the same clips (batch-statistics BatchNorm, Dropout off, temperature 30)."""
    res = {}
    with torch.no_grad():
        x_cal = mel(synth.calibration_clips(96000)).unsqueeze(1)
        x = mel(synth.parity_clips(96000, seed=43)).unsqueeze(1)
    for tag, kw in DYMN_VARIANTS.items():
        model = (quiet(dymn_factory, width_mult=1.0, **kw) if ("head_type" in kw or "dilated" in kw)
                 else quiet(get_dymn, width_mult=1.0, **kw))
        sd = synth.synth_state(synth.shapes_of(model), seed=4)
        model.load_state_dict(sd, strict=True)

        def set_temp(t):
            for m in model.modules():
                if isinstance(m, DynamicConv):
                    m.temperature = t

        set_temp(1.0)
        calibrate_reference(model, x_cal)
        for k, v in bn_buffers(model).items():
            res[f"{tag}/bn/{k}"] = v.numpy()
        model.eval()
        with torch.no_grad():
            logits, feats = model(x)
        res[f"{tag}/logits"], res[f"{tag}/features"] = logits.numpy(), feats.numpy()
        res[f"{tag}/n_state"] = np.int64(len(model.state_dict()))
        res[f"{tag}/n_params"] = np.int64(sum(p.numel() for p in model.parameters()))
        res[f"{tag}/keys"] = np.array(list(model.state_dict().keys()))     # synth_state draws in this order
        set_temp(30.0)
        model.train()
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        with torch.no_grad():
            tl, tf = model(x)
        res[f"{tag}/train_logits"], res[f"{tag}/train_features"] = tl.numpy(), tf.numpy()
        print("dymn variant", tag, "logits absmax", float(logits.abs().max()), "std", float(logits.std()),
              "train std", float(tl.std()))
    np.savez_compressed(os.path.join(OUT, "dymn_variants_ref.npz"), **res)


WIDTHS = {"mn40": ("mn", 4.0), "dymn20": ("dymn", 2.0)}      # BASELINE.json configs[2] / configs[3]


def golden_widths(mel):
    """The reference at the widths BASELINE.json names (models/mn/model.py:326-367 with width_mult=4.0,
    models/dymn/model.py:289-361 with width_mult=2.0) on 3 s clips: eval logits / features, and one train-mode step at the
    reference's initial DynamicConv temperature 30 (Dropout off): logits, loss, gradient norm + 8 samples of every
    parameter's gradient, BatchNorm buffers after the step."""
    res = {}
    with torch.no_grad():
        x_cal = mel(synth.calibration_clips(96000)).unsqueeze(1)
        x = mel(synth.parity_clips(96000, seed=45)).unsqueeze(1)
    y = torch.from_numpy((np.random.Generator(np.random.PCG64(10)).random((x.shape[0], 527)) < 0.01).astype(np.float32))
    for tag, (kind, width) in WIDTHS.items():
        model = quiet(get_mn if kind == "mn" else get_dymn, width_mult=width)
        sd = synth.synth_state(synth.shapes_of(model), seed=6)
        model.load_state_dict(sd, strict=True)

        def set_temp(t):
            for m in model.modules():
                if isinstance(m, DynamicConv):
                    m.temperature = t

        set_temp(1.0)
        calibrate_reference(model, x_cal)
        for k, v in bn_buffers(model).items():
            res[f"{tag}/bn/{k}"] = v.numpy()
        res[f"{tag}/keys"] = np.array(list(model.state_dict().keys()))
        res[f"{tag}/n_params"] = np.int64(sum(p.numel() for p in model.parameters()))
        model.eval()
        with torch.no_grad():
            logits, feats = model(x)
        res[f"{tag}/logits"], res[f"{tag}/features"] = logits.numpy(), feats.numpy()
        set_temp(30.0)
        model.train()
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        tl, _ = model(x)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(tl, y)
        loss.backward()
        res[f"{tag}/train_logits"], res[f"{tag}/train_loss"] = tl.detach().numpy(), np.float64(loss.item())
        res[f"{tag}/train_labels"] = y.numpy()
        rng = np.random.Generator(np.random.PCG64(7))
        for k, v in grad_summary(model, rng).items():
            res[f"{tag}/{k}"] = v
        for k, v in bn_buffers(model).items():
            res[f"{tag}/bn_after/{k}"] = v.numpy()
        print("width", tag, "params", int(res[f"{tag}/n_params"]), "logits absmax", float(logits.abs().max()), "train loss", loss.item())
    np.savez_compressed(os.path.join(OUT, "widths_ref.npz"), **res)


if __name__ == "__main__":
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    mel = golden_mel()
    if "--dymn-variants-only" in sys.argv:
        golden_dymn_variants(mel)
        sys.exit(0)
    if "--widths-only" in sys.argv:
        golden_widths(mel)
        sys.exit(0)
    if "--variants-only" not in sys.argv:
        golden_model("mn", mel, 1.0, "mn10")
        golden_model("dymn", mel, 1.0, "dymn10")
    golden_variants(mel)
    golden_dymn_variants(mel)
    golden_widths(mel)
