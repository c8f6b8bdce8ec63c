"""Pins oracle/eat_oracle.py (the CPU restatement) against vectors produced by the
unmodified reference (oracle/make_golden.py -> tests/golden/*.npz).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import eat_oracle as O
from oracle import synth


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_param_counts_match_readme():
    # README.md:94-113 (4.88 M / 68.43 M); DyMN counts as reproduced in SURVEY.md section 4
    assert synth.n_params(synth.mn_shapes(1.0)) == 4876831
    assert synth.n_params(synth.mn_shapes(4.0)) == 68427303
    assert synth.n_params(synth.dymn_shapes(1.0)) == 10548479
    assert synth.n_params(synth.dymn_shapes(2.0)) == 39966143
    assert len(synth.mn_shapes(1.0)) == 312 and len(synth.dymn_shapes(1.0)) == 578


def test_mel_basis_support_and_hf_crosscheck():
    basis = O.kaldi_mel_banks(128, 1024, 32000, 0.0, 15000.0)
    assert basis.shape == (128, 513)
    assert int((basis != 0).sum()) == 948
    assert float(basis[127, 480]) > 0              # the fp32 "bin 480" non-zero (hard parts)
    assert float(basis[:, 512].abs().max()) == 0
    audio_utils = pytest.importorskip("transformers.audio_utils")
    hf = audio_utils.mel_filter_bank(num_frequency_bins=513, num_mel_filters=128, min_frequency=0.0,
                                     max_frequency=15000.0, sampling_rate=32000, norm=None,
                                     mel_scale="kaldi", triangularize_in_mel_space=True)
    hf = torch.from_numpy(np.asarray(hf, dtype=np.float64)).T[:, :513]
    assert float((basis.double() - hf).abs().max()) < 5e-5


def test_mel_matches_reference(golden_dir):
    g = _load(golden_dir, "mel_ref.npz")
    m = O.mel_forward(synth.parity_clips(32000, seed=77))
    assert m.shape == (5, 128, 100)
    assert np.abs(m.numpy() - g["short"]).max() < 2e-5
    full = O.mel_forward(synth.parity_clips(320000, seed=1234))
    assert full.shape == (5, 128, 1000)
    assert np.abs(full[:, :, g["t_edge"]].numpy() - g["full_edge"]).max() < 2e-5
    assert np.abs(full.double().sum(dim=2).numpy() - g["full_rowsum"]).max() < 2e-2


def test_mel_train_mode_matches_reference(golden_dir):
    """Train mode: replay the reference's host RNG draws (preprocess.py:45-46, masking)."""
    g = _load(golden_dir, "mel_ref.npz")
    torch.manual_seed(2024)
    fmin = 0.0 + torch.randint(10, (1,)).item()
    fmax = 15000 + 2000 // 2 - torch.randint(2000, (1,)).item()

    def draw(param, size):
        value = torch.rand(1) * param
        mn = torch.rand(1) * (size - value)
        return int(mn.long()), int(mn.long()) + int(value.long())

    fm = draw(48, 128)
    tm = draw(192, 200)
    m = O.mel_forward(synth.parity_clips(64000, seed=78), fmin=fmin, fmax=fmax, freq_mask=fm, time_mask=tm)
    assert np.abs(m.numpy() - g["train_short_seed2024"]).max() < 2e-5


def _calibrated(kind, g):
    shapes = (synth.mn_shapes if kind == "mn" else synth.dymn_shapes)(1.0)
    sd = synth.synth_state(shapes, seed=0)
    for k in g.files:
        if k.startswith("bn/"):
            sd[k[3:]] = torch.from_numpy(g[k])
    return sd


def _check_model(kind, g, fwd, rel):
    sd = _calibrated(kind, g)
    x = O.mel_forward(synth.parity_clips(320000, seed=1234)).unsqueeze(1)
    with torch.no_grad():
        logits, fmaps = fwd(sd, x, return_fmaps=True)
        _, feats = fwd(sd, x)
    scale = np.abs(g["eval_logits"]).max(axis=1, keepdims=True)
    assert (np.abs(logits.numpy() - g["eval_logits"]) / scale).max() < rel
    assert np.abs(feats.numpy() - g["eval_features"]).max() < rel * max(1.0, np.abs(g["eval_features"]).max())
    assert len(fmaps) == 17
    for i, f in enumerate(fmaps):
        assert tuple(f.shape) == tuple(g[f"fmap{i}_shape"])
        v = f.reshape(-1)[g[f"fmap{i}_idx"]].numpy()
        assert np.abs(v - g[f"fmap{i}_val"]).max() < rel * 10 * max(1.0, float(g[f"fmap{i}_std"]))
        assert abs(float(f.double().std()) - float(g[f"fmap{i}_std"])) < 1e-3 * float(g[f"fmap{i}_std"])
    return sd, x


def _check_train(kind, g, sd, x, fwd, gtol=1e-2):
    for k in sd:
        if not k.endswith(("running_mean", "running_var", "num_batches_tracked", "lambdas", "init_v")):
            sd[k] = sd[k].clone().requires_grad_(True)
    stats = {}
    keep = torch.from_numpy(g["drop_keep"].astype(np.float32))
    logits, _ = fwd(sd, x, train=True, stats=stats, drop_mask=keep)
    y = torch.from_numpy(g["train_labels"])
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, y)
    loss.backward()
    assert abs(loss.item() - float(g["train_loss"])) < 1e-5
    assert np.abs(logits.detach().numpy() - g["train_logits"]).max() < 2e-4 * max(1.0, np.abs(g["train_logits"]).max())
    bad = []
    gmax = max(float(g[k]) for k in g.files if k.startswith("gnorm/"))
    for k in g.files:
        if not k.startswith("gnorm/"):
            continue
        name = k[6:]
        ref = float(g[k])
        got = float(sd[name].grad.double().norm())
        # project-BN biases have an exactly-zero true gradient in train mode: skip tiny norms
        if ref < 1e-5 * gmax:
            continue
        if abs(got - ref) > gtol * ref:
            bad.append((name, ref, got))
    assert not bad, bad[:5]
    for k, v in stats.items():
        assert np.abs(v.numpy() - g["bn_after/" + k]).max() < 1e-4 * max(1.0, np.abs(g["bn_after/" + k]).max())


def test_mn10_oracle_matches_reference(golden_dir):
    g = _load(golden_dir, "mn10_ref.npz")
    assert int(g["n_params"]) == 4876831 and int(g["n_state"]) == 312
    sd, x = _check_model("mn", g, O.mn_forward, 2e-5)
    _check_train("mn", g, sd, x, O.mn_forward)


def test_dymn10_oracle_matches_reference(golden_dir):
    g = _load(golden_dir, "dymn10_ref.npz")
    assert int(g["n_params"]) == 10548479 and int(g["n_state"]) == 578
    te, tt = float(g["temp_eval"]), float(g["temp_train"])
    fwd_e = lambda sd, x, **k: O.dymn_forward(sd, x, temperature=te, **k)
    fwd_t = lambda sd, x, **k: O.dymn_forward(sd, x, temperature=tt, **k)
    sd, x = _check_model("dymn", g, fwd_e, 2e-4)
    # attention-logit grads at T=30 are cancellation-dominated (norm 1e-4): looser bound
    _check_train("dymn", g, sd, x, fwd_t, gtol=3e-2)


# ---------------------------------------------------------------- non-default model variants
VARIANTS = {   # tag -> (product get_model kwargs, oracle mn_forward kwargs); tags = oracle/make_golden.py VARIANTS
    "fc": (dict(head_type="fully_convolutional"), dict(head_type="fully_convolutional")),
    "fc_s2211": (dict(head_type="fully_convolutional", strides=(2, 2, 1, 1)),
                 dict(head_type="fully_convolutional", strides=(2, 2, 1, 1))),
    "att": (dict(head_type="multihead_attention_pooling", multihead_attention_heads=4), dict(head_type="att", num_heads=4)),
    "se_ct_max": (dict(se_dims="ct", se_agg="max", input_dim_t=300), dict(se_dims=(1, 3), se_agg="max")),
    "se_t_avg": (dict(se_dims="t", se_agg="avg", input_dim_t=300), dict(se_dims=(3,), se_agg="avg")),
    "se_ct_min": (dict(se_dims="ct", se_agg="min", input_dim_t=300), dict(se_dims=(1, 3), se_agg="min")),
    "se_ct_add": (dict(se_dims="ct", se_agg="add", input_dim_t=300), dict(se_dims=(1, 3), se_agg="add")),
    "se_none": (dict(se_dims="none"), dict(se_dims=None)),
    "dilated_reduced": (dict(dilated=True, reduced_tail=True), dict(dilated=True, reduced_tail=True)),
}


def variant_state(tag, golden_dir):
    """(product model, seeded state with the reference-calibrated BN buffers, golden arrays) of one variant."""
    import contextlib
    import io
    from efficientat_amd.mn import get_model
    g = np.load(os.path.join(golden_dir, "mn_variants_ref.npz"))
    with contextlib.redirect_stdout(io.StringIO()):
        model = get_model(width_mult=1.0, **VARIANTS[tag][0])
    sd = synth.synth_state(synth.shapes_of(model), seed=3)
    assert len(sd) == int(g[f"{tag}/n_state"])                  # same state_dict layout as the reference's variant
    for k in g.files:
        if k.startswith(f"{tag}/bn/"):
            sd[k[len(tag) + 4:]] = torch.from_numpy(g[k])
    return model, sd, g


@pytest.mark.parametrize("tag", list(VARIANTS))
def test_oracle_variants_match_reference(tag, golden_dir):
    _, sd, g = variant_state(tag, golden_dir)
    x = O.mel_forward(synth.parity_clips(96000, seed=41)).unsqueeze(1)
    with torch.no_grad():
        logits, feats = O.mn_forward(sd, x, **VARIANTS[tag][1])
    ref = g[f"{tag}/logits"]
    assert np.abs(logits.numpy() - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
    assert np.abs(feats.numpy() - g[f"{tag}/features"]).max() < 2e-5 * max(1.0, np.abs(g[f"{tag}/features"]).max())


# ---------------------------------------------------------------- DyMN variants (use_dy_blocks="replace_se")
DYMN_VARIANTS = {"replace_se": (dict(use_dy_blocks="replace_se"), dict(use_dy_blocks="replace_se")),
                 # ablations of the dynamic block (models/dymn/dy_block.py:269-271,291-375)
                 "no_dyrelu": (dict(no_dyrelu=True), dict(no_dyrelu=True)),
                 "no_dyconv": (dict(no_dyconv=True), dict(no_dyconv=True)),
                 "no_ca": (dict(no_ca=True), dict(no_ca=True)),
                 "static": (dict(no_dyrelu=True, no_dyconv=True, no_ca=True),
                            dict(no_dyrelu=True, no_dyconv=True, no_ca=True)),
                 # fully-convolutional head (models/dymn/model.py:119-130): through the `dymn` factory, as in the reference
                 "fc_head": (dict(head_type="fully_convolutional"), dict(head_type="fully_convolutional")),
                 # dilation 2 in the last three (dynamic) blocks, models/dymn/model.py:212-218,246-250; dy_block.py:322-348
                 "dilated": (dict(dilated=True), dict(dilated=True))}


def dymn_variant_state(tag, golden_dir):
    import contextlib
    import io
    from efficientat_amd.dymn import dymn, get_model
    g = np.load(os.path.join(golden_dir, "dymn_variants_ref.npz"))
    kw = DYMN_VARIANTS[tag][0]
    with contextlib.redirect_stdout(io.StringIO()):
        model = dymn(width_mult=1.0, **kw) if ("head_type" in kw or "dilated" in kw) else get_model(width_mult=1.0, **kw)
    shapes = synth.shapes_of(model)
    keys = [str(k) for k in g[f"{tag}/keys"]]                   # the reference's state_dict order (seeded draws follow it)
    assert sorted(keys) == sorted(shapes) and len(keys) == int(g[f"{tag}/n_state"])   # same state_dict layout
    sd = synth.synth_state({k: shapes[k] for k in keys}, seed=4)
    assert sum(p.numel() for p in model.parameters()) == int(g[f"{tag}/n_params"])
    for k in g.files:
        if k.startswith(f"{tag}/bn/"):
            sd[k[len(tag) + 4:]] = torch.from_numpy(g[k])
    return model, sd, g


@pytest.mark.parametrize("tag", list(DYMN_VARIANTS))
def test_oracle_dymn_variants_match_reference(tag, golden_dir):
    """models/dymn/model.py:225-231,102-103: eval (temperature 1) and train-mode (batch statistics, temperature 30)
    logits / features of the unmodified reference."""
    _, sd, g = dymn_variant_state(tag, golden_dir)
    x = O.mel_forward(synth.parity_clips(96000, seed=43)).unsqueeze(1)
    kw = DYMN_VARIANTS[tag][1]
    with torch.no_grad():
        logits, feats = O.dymn_forward(sd, x, temperature=1.0, **kw)
        tl, tf = O.dymn_forward(sd, x, temperature=30.0, train=True, **kw)
    for got, key in ((logits, "logits"), (feats, "features"), (tl, "train_logits"), (tf, "train_features")):
        ref = g[f"{tag}/{key}"]
        assert np.abs(got.numpy() - ref).max() < 5e-5 * max(1.0, np.abs(ref).max()), key


# ---------------------------------------------------------------- BASELINE widths: mn40 (configs[2]), dymn20 (configs[3])
WIDTHS = {"mn40": ("mn", 4.0), "dymn20": ("dymn", 2.0)}


def width_state(tag, golden_dir):
    """(state_dict, golden npz) of the reference-generated fixture of `oracle/make_golden.py:golden_widths`."""
    import contextlib
    import io
    kind, width = WIDTHS[tag]
    g = np.load(os.path.join(golden_dir, "widths_ref.npz"))
    if kind == "mn":
        from efficientat_amd.mn import get_model
    else:
        from efficientat_amd.dymn import get_model
    with contextlib.redirect_stdout(io.StringIO()):
        model = get_model(width_mult=width)
    shapes = synth.shapes_of(model)
    keys = [str(k) for k in g[f"{tag}/keys"]]
    assert sorted(keys) == sorted(shapes)                        # same state_dict layout as the reference at this width
    assert sum(p.numel() for p in model.parameters()) == int(g[f"{tag}/n_params"])
    sd = synth.synth_state({k: shapes[k] for k in keys}, seed=6)
    for k in g.files:
        if k.startswith(f"{tag}/bn/"):
            sd[k[len(tag) + 4:]] = torch.from_numpy(g[k])
    return model, sd, g


@pytest.mark.parametrize("tag", list(WIDTHS))
def test_oracle_at_baseline_widths_matches_reference(tag, golden_dir):
    """models/mn/model.py:326-367 (width_mult=4.0) / models/dymn/model.py:289-361 (width_mult=2.0): eval logits and one
    train-mode step (DynamicConv temperature 30, the reference's initial value) of the unmodified reference."""
    import torch.nn.functional as F
    kind, width = WIDTHS[tag]
    _, sd, g = width_state(tag, golden_dir)
    x = O.mel_forward(synth.parity_clips(96000, seed=45)).unsqueeze(1)
    fwd = (lambda s, xm, **k: O.mn_forward(s, xm, width_mult=width, **k)) if kind == "mn" else \
          (lambda s, xm, temperature=1.0, **k: O.dymn_forward(s, xm, width_mult=width, temperature=temperature, **k))
    with torch.no_grad():
        logits, feats = fwd(sd, x)
    for got, key in ((logits, "logits"), (feats, "features")):
        ref = g[f"{tag}/{key}"]
        assert np.abs(got.numpy() - ref).max() < 1e-4 * max(1.0, np.abs(ref).max()), key
    sdr = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and not k.endswith(
        ("running_mean", "running_var", "lambdas", "init_v")) else v.clone()) for k, v in sd.items()}
    kw = dict(train=True, stats={})
    if kind == "dymn":
        kw["temperature"] = 30.0
    tl, _ = fwd(sdr, x, **kw)
    loss = F.binary_cross_entropy_with_logits(tl, torch.from_numpy(g[f"{tag}/train_labels"]))
    loss.backward()
    assert abs(float(loss) - float(g[f"{tag}/train_loss"])) < 1e-5
    assert np.abs(tl.detach().numpy() - g[f"{tag}/train_logits"]).max() < 1e-4 * max(1.0, np.abs(g[f"{tag}/train_logits"]).max())
    gmax = max(float(g[k]) for k in g.files if k.startswith(f"{tag}/gnorm/"))
    for name, v in sdr.items():
        if getattr(v, "grad", None) is None:
            continue
        ref = float(g[f"{tag}/gnorm/{name}"])
        if ref > 1e-4 * gmax:
            assert abs(float(v.grad.norm()) - ref) < 2e-2 * ref, name
