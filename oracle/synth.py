"""Deterministic synthetic weights / clips for parity tests (TEST INFRASTRUCTURE ONLY).

Weights are drawn from numpy PCG64 streams (stable across platforms) in the
*reference state-dict layout* (SURVEY.md section 8b); ``oracle/make_golden.py`` loads
them into the real reference model with ``strict=True``, which pins names and shapes.
BN running statistics are calibrated on a mixed clip batch so that activations have
O(1) scale ("degenerate-oracle trap").
"""
import math
from collections import OrderedDict

import numpy as np
import torch

from . import eat_oracle as O


# ------------------------------------------------------------------ state layouts
def _cna_shapes(sh, prefix, cin, cout, k, groups):
    sh[prefix + ".0.weight"] = (cout, cin // groups, k, k)
    _bn_shapes(sh, prefix + ".1", cout)


def _bn_shapes(sh, prefix, c):
    sh[prefix + ".weight"] = (c,)
    sh[prefix + ".bias"] = (c,)
    sh[prefix + ".running_mean"] = (c,)
    sh[prefix + ".running_var"] = (c,)
    sh[prefix + ".num_batches_tracked"] = ()


def mn_shapes(width_mult=1.0, num_classes=527):
    blocks, last = O.block_table(width_mult)
    sh = OrderedDict()
    _cna_shapes(sh, "features.0", 1, blocks[0]["cin"], 3, 1)
    for i, c in enumerate(blocks):
        p, j = f"features.{i + 1}.block", 0
        if c["cexp"] != c["cin"]:
            _cna_shapes(sh, f"{p}.{j}", c["cin"], c["cexp"], 1, 1)
            j += 1
        _cna_shapes(sh, f"{p}.{j}", c["cexp"], c["cexp"], c["k"], c["cexp"])
        j += 1
        if c["se"]:
            sq = O.make_divisible(c["cexp"] // 4, 8)
            q = f"{p}.{j}.conc_se_layers.0"
            sh[q + ".fc1.weight"], sh[q + ".fc1.bias"] = (sq, c["cexp"]), (sq,)
            sh[q + ".fc2.weight"], sh[q + ".fc2.bias"] = (c["cexp"], sq), (c["cexp"],)
            j += 1
        _cna_shapes(sh, f"{p}.{j}", c["cexp"], c["cout"], 1, 1)
    clast = blocks[-1]["cout"]
    _cna_shapes(sh, "features.16", clast, 6 * clast, 1, 1)
    sh["classifier.2.weight"], sh["classifier.2.bias"] = (last, 6 * clast), (last,)
    sh["classifier.5.weight"], sh["classifier.5.bias"] = (num_classes, last), (num_classes,)
    return sh


def dymn_shapes(width_mult=1.0, num_classes=527, K=4):
    blocks, last = O.block_table(width_mult)
    sh = OrderedDict()
    _cna_shapes(sh, "in_c", 1, blocks[0]["cin"], 3, 1)
    for i, c in enumerate(blocks):
        p = f"layers.{i}"
        H = O.context_dim(c["cexp"], width_mult)
        convs = []
        if c["cexp"] != c["cin"]:
            convs.append(("exp", c["cin"], c["cexp"], 1, 1))
        convs.append(("depth", c["cexp"], c["cexp"], c["k"], c["cexp"]))
        convs.append(("proj", c["cexp"], c["cout"], 1, 1))
        for name, ci, co, k, g in convs:
            sh[f"{p}.{name}_conv.weight"] = (1, 1, K, co * (ci // g) * k * k)
            sh[f"{p}.{name}_conv.residuals.0.weight"] = (K, H)
            sh[f"{p}.{name}_conv.residuals.0.bias"] = (K,)
            _bn_shapes(sh, f"{p}.{name}_norm", co)
            if name == "depth":
                sh[f"{p}.depth_act.lambdas"] = (4,)
                sh[f"{p}.depth_act.init_v"] = (4,)
                sh[f"{p}.depth_act.coef_net.0.weight"] = (4 * co, H)
                sh[f"{p}.depth_act.coef_net.0.bias"] = (4 * co,)
        sh[f"{p}.context_gen.joint_conv.weight"] = (H, c["cin"], 1, 1)
        _bn_shapes(sh, f"{p}.context_gen.joint_norm", H)
        for n in ("conv_f", "conv_t"):
            sh[f"{p}.context_gen.{n}.weight"] = (c["cexp"], H, 1, 1)
            sh[f"{p}.context_gen.{n}.bias"] = (c["cexp"],)
    clast = blocks[-1]["cout"]
    _cna_shapes(sh, "out_c", clast, 6 * clast, 1, 1)
    sh["classifier.2.weight"], sh["classifier.2.bias"] = (last, 6 * clast), (last,)
    sh["classifier.5.weight"], sh["classifier.5.bias"] = (num_classes, last), (num_classes,)
    return sh


def shapes_of(module):
    """name -> shape of a module's state_dict: lets `synth_state` fill ANY variant of the model (the reference's in
    make_golden.py, the product's in the tests) with the same seeded weights - identical keys and shapes required."""
    return OrderedDict((k, tuple(v.shape)) for k, v in module.state_dict().items())


def n_params(shapes):
    """Learnable parameter count (buffers excluded), to compare with README.md:94-113."""
    skip = ("running_mean", "running_var", "num_batches_tracked", "lambdas", "init_v")
    return sum(int(np.prod(s)) for k, s in shapes.items() if not k.endswith(skip))


# --------------------------------------------------------------------- weights
def synth_state(shapes, seed=0):
    rng = np.random.Generator(np.random.PCG64(seed))
    nrm = lambda shape, std: torch.from_numpy((rng.standard_normal(shape) * std).astype(np.float32))
    sd = OrderedDict()
    for name, shape in shapes.items():
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            sd[name] = torch.zeros((), dtype=torch.int64)
        elif leaf == "running_mean":
            sd[name] = torch.zeros(shape)
        elif leaf == "running_var":
            sd[name] = torch.ones(shape)
        elif leaf == "lambdas":
            sd[name] = torch.tensor([1.0, 1.0, 0.5, 0.5])
        elif leaf == "init_v":
            sd[name] = torch.tensor([1.0, 0.0, 0.0, 0.0])
        elif leaf == "head_weight":                           # attention-pooling head weights (1, heads, 1)
            sd[name] = torch.from_numpy(rng.uniform(0.1, 0.4, shape).astype(np.float32))
        elif len(shape) == 1 and leaf == "weight":            # BN gamma
            sd[name] = torch.from_numpy(rng.uniform(0.5, 1.5, shape).astype(np.float32))
        elif leaf == "bias":
            sd[name] = nrm(shape, 0.1)
        elif len(shape) == 2:                                 # nn.Linear
            std = 1.0 / math.sqrt(shape[1])
            if name == "classifier.5.weight":
                std *= 4.0
            sd[name] = nrm(shape, std)
        elif len(shape) == 4 and shape[0] == 1 and shape[1] == 1:   # DynamicConv bank (1,1,K,N)
            # K correlated kernels (shared base + perturbation), as in a trained DyMN; i.i.d.
            # banks make the per-sample kernel choice chaotic and un-calibratable.
            base = nrm((1, 1, 1, shape[3]), 1.0)
            sd[name] = base + nrm(shape, 0.2)
        elif len(shape) == 4:                                 # conv: kaiming normal, fan_out
            fan_out = shape[0] * shape[2] * shape[3]
            sd[name] = nrm(shape, math.sqrt(2.0 / fan_out))
        else:
            raise ValueError(name)
    return sd


# ----------------------------------------------------------------------- clips
def parity_clips(n_samples=320000, seed=1234, sr=32000):
    """5 clip types x 1 (noise, quiet noise, two-tone, silence+chirp, AM noise+tone)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n_samples) / sr
    dur = n_samples / sr
    clips = [
        np.clip(0.1 * rng.standard_normal(n_samples), -1, 1),
        1e-3 * rng.standard_normal(n_samples),
        0.3 * np.sin(2 * np.pi * 440 * t + rng.uniform(0, 6.28)) + 0.2 * np.sin(2 * np.pi * 5000 * t),
        np.where(t < 0.1 * dur, 0.0, 0.5 * np.sin(2 * np.pi * (100 * t + 0.5 * (12000 / dur) * t * t))),
        (0.5 + 0.5 * np.sin(2 * np.pi * 3 * t)) * 0.05 * rng.standard_normal(n_samples)
        + 0.1 * np.sin(2 * np.pi * 1234.5 * t),
    ]
    return torch.from_numpy(np.stack(clips).astype(np.float32))


def calibration_clips(n_samples=320000):
    """10 clips (two draws of the parity mix) used to calibrate BN running statistics."""
    return torch.cat([parity_clips(n_samples, seed=4321), parity_clips(n_samples, seed=999)])


# ------------------------------------------------------------------ calibration
def calibrate(sd, forward, x_mel):
    """One train-mode oracle pass with momentum 1.0: running stats := batch stats."""
    stats = {"__momentum__": 1.0}
    with torch.no_grad():
        forward(sd, x_mel, train=True, stats=stats)
    stats.pop("__momentum__")
    for k, v in stats.items():
        sd[k] = v
    return sd
