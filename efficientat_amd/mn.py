"""Drop-in for the reference's ``models.mn.model`` (get_model / MN) on the HIP hot path.

Same public API and the same state_dict key layout as the reference
(models/mn/model.py:73-234, 326-367; models/mn/block_types.py:45-181), so released checkpoints
load with ``strict=True``.  The module tree only HOLDS parameters (nn.Conv2d / nn.BatchNorm2d /
nn.Linear leaves under the reference's names); ``MN.forward`` does not call the leaves but runs
a fused launch plan over libeat_hip.so:

    stem conv3x3/s2+BN+hswish  ->  15 x [ pw expand+BN+act -> dw kxk+BN+act (+SE squeeze sums)
    -> SE gate (2 small GEMMs) -> pw project+BN (*SE scale on the input, +residual) ]
    ->  last 1x1 conv+BN+hswish with the global average pool fused  ->  2 Linear layers.

Eval-mode BatchNorm is folded into the conv weights (cached, re-folded when a parameter or
buffer changes).  There is no CPU path: CPU tensors raise.
"""
import os
from functools import partial
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .utils import make_divisible, cnn_out_size, NAME_TO_WIDTH  # noqa: F401

BN_EPS, BN_MOMENTUM = 0.001, 0.01  # models/mn/model.py:114-115
# Blocks whose input has at most this many channels run as one fused kernel (csrc/irb.hip): whole
# block when it has no SE, expand + depthwise when it has.  Measured on MI355X at B=256 (fused vs separate
# kernels): mn10 block 2 0.72 vs 1.13 ms, block 3 0.54 vs 0.65, block 4 0.40 vs 0.51; from C_in = 40 on the
# fused kernel is VALU/MFMA-bound (blocks 5-7: 0.38 vs 0.24, 0.64 vs 0.39 ms) and the separate kernels win.
# 0 disables (A/B switch).
_FUSE_MAX_CIN = 24
# stem + first block as one kernel (csrc/irb.hip, FRONT mode); 0 disables (A/B switch)
_FUSE_FRONT = 1
# arithmetic of the 1x1 convs in eval: fp32 | bf16x3 | bf16 | auto (see _pw_mode)
_PW_MODE = os.environ.get("EAT_PW_MODE", "auto")
model_url = "https://github.com/fschmid56/EfficientAT/releases/download/v0.0.1/"
model_dir = "resources"

# checkpoint file names published by the reference (models/mn/model.py:24-70)
_CKPT = {
    "mn10_im_pytorch": "mn10_im_pytorch.pt",
    **{f"mn{w}_im": f"mn{w}_im.pt" for w in ("01", "02", "04", "05", "10", "20", "30", "40")},
    "mn01_as": "mn01_as_mAP_298.pt", "mn02_as": "mn02_as_mAP_378.pt", "mn04_as": "mn04_as_mAP_432.pt",
    "mn05_as": "mn05_as_mAP_443.pt", "mn10_as": "mn10_as_mAP_471.pt", "mn20_as": "mn20_as_mAP_478.pt",
    "mn30_as": "mn30_as_mAP_482.pt", "mn40_as": "mn40_as_mAP_484.pt", "mn40_as(2)": "mn40_as_mAP_483.pt",
    "mn40_as(3)": "mn40_as_mAP_483(2).pt", "mn40_as_no_im_pre": "mn40_as_no_im_pre_mAP_483.pt",
    "mn40_as_no_im_pre(2)": "mn40_as_no_im_pre_mAP_483(2).pt", "mn40_as_no_im_pre(3)": "mn40_as_no_im_pre_mAP_482.pt",
    "mn40_as_ext": "mn40_as_ext_mAP_487.pt", "mn40_as_ext(2)": "mn40_as_ext_mAP_486.pt",
    "mn40_as_ext(3)": "mn40_as_ext_mAP_485.pt", "mn10_as_hop_5": "mn10_as_hop_5_mAP_475.pt",
    "mn10_as_hop_15": "mn10_as_hop_15_mAP_463.pt", "mn10_as_hop_20": "mn10_as_hop_20_mAP_456.pt",
    "mn10_as_hop_25": "mn10_as_hop_25_mAP_447.pt", "mn10_as_mels_40": "mn10_as_mels_40_mAP_453.pt",
    "mn10_as_mels_64": "mn10_as_mels_64_mAP_461.pt", "mn10_as_mels_256": "mn10_as_mels_256_mAP_474.pt",
    "mn10_as_fc": "mn10_as_fc_mAP_465.pt", "mn10_as_fc_s2221": "mn10_as_fc_s2221_mAP_466.pt",
    "mn10_as_fc_s2211": "mn10_as_fc_s2211_mAP_466.pt",
}
pretrained_models = {k: model_url + v for k, v in _CKPT.items()}


class InvertedResidualConfig:
    """Row of the MobileNetV3 table (models/mn/block_types.py:86-117)."""

    def __init__(self, input_channels, kernel, expanded_channels, out_channels, use_se, activation, stride,
                 dilation, width_mult):
        adj = self.adjust_channels
        self.input_channels = adj(input_channels, width_mult)
        self.kernel = kernel
        self.expanded_channels = adj(expanded_channels, width_mult)
        self.out_channels = adj(out_channels, width_mult)
        self.use_se, self.use_hs = use_se, activation == "HS"
        self.stride, self.dilation = stride, dilation
        self.f_dim = self.t_dim = None

    @staticmethod
    def adjust_channels(channels, width_mult):
        return make_divisible(channels * width_mult, 8)

    def out_size(self, in_size):
        return cnn_out_size(in_size, (self.kernel - 1) // 2 * self.dilation, self.dilation, self.kernel, self.stride)


def _conv_bn_act(cin, cout, k, stride=1, groups=1, act=None):
    """Parameter holder with torchvision-0.14 ConvNormActivation key names (.0 conv, .1 BN, .2 act)."""
    mods = [nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False),
            nn.BatchNorm2d(cout, eps=BN_EPS, momentum=BN_MOMENTUM)]
    if act is not None:
        mods.append(act(inplace=True))
    seq = nn.Sequential(*mods)
    seq.out_channels = cout
    return seq


class SqueezeExcitation(nn.Module):
    """fc1 / ReLU / fc2 / Sigmoid gate over one of the dimensions channel (1), frequency (2), time (3)
    (models/mn/block_types.py:45-83)."""

    def __init__(self, input_dim, squeeze_dim, se_dim=1):
        super().__init__()
        self.fc1 = nn.Linear(input_dim, squeeze_dim)
        self.fc2 = nn.Linear(squeeze_dim, input_dim)
        assert se_dim in [1, 2, 3]
        self.gate_dim = se_dim


class ConcurrentSEBlock(nn.Module):
    """Holder for ``conc_se_layers.{i}`` (models/mn/block_types.py:10-42): one SqueezeExcitation per entry of `se_dims`
    (1 = channels, 2 = frequency, 3 = time), their scaled outputs combined by `se_agg`.  The default (channel SE only)
    runs fused into the depthwise / project kernels; the other configurations run as a separate, unfused eval plan."""

    def __init__(self, c_dim, f_dim, t_dim, se_cnf):
        super().__init__()
        if se_cnf["se_agg"] not in ("max", "avg", "add", "min"):
            raise NotImplementedError(f"SE aggregation operation '{se_cnf['se_agg']}' not implemented")
        dims = [c_dim, f_dim, t_dim]
        if 2 in se_cnf["se_dims"]:
            # the reference cannot run a frequency gate either: block_types.py:75 squeezes dim 2 twice, which leaves
            # (B, 1, F, 1) and fc1 fails on a last dimension of 1 - there is no reference behaviour to match
            raise NotImplementedError("squeeze-excitation over the frequency dimension ('f') fails in the reference "
                                      "(models/mn/block_types.py:75); use se_dims from {'c', 't'}")
        self.se_dims, self.se_agg = list(se_cnf["se_dims"]), se_cnf["se_agg"]
        self.channel_only = self.se_dims == [1]          # max / avg / min of one element is the identity ('add' too)
        self.conc_se_layers = nn.ModuleList(
            [SqueezeExcitation(dims[d - 1], make_divisible(dims[d - 1] // se_cnf["se_r"], 8), d) for d in self.se_dims])


class MultiHeadAttentionPooling(nn.Module):
    """Parameter holder + eval forward of the PSLA multi-head attention pooling head (models/mn/attention_pooling.py:9-56):
    mean over frequency, Linear to (att, val) x heads x classes on the MFMA linear kernel, sigmoid attention normalised
    over time, weighted sum, head-weighted sum.  The (B, heads, T, classes) algebra after the GEMM is a few KB per clip
    and stays in torch ops."""

    def __init__(self, in_dim, out_dim, att_activation="sigmoid", clf_activation="ident", num_heads=4, epsilon=1e-7):
        super().__init__()
        if att_activation != "sigmoid" or clf_activation != "ident":
            raise NotImplementedError("HIP path implements the reference's default activations (sigmoid / ident)")
        self.in_dim, self.out_dim, self.num_heads, self.epsilon = in_dim, out_dim, num_heads, epsilon
        self.subspace_proj = nn.Linear(in_dim, out_dim * 2 * num_heads)
        self.head_weight = nn.Parameter(torch.tensor([1.0 / num_heads] * num_heads).view(1, -1, 1))

    def forward(self, x):
        b, c = x.shape[0], x.shape[1]
        xm = x.mean(dim=2).transpose(1, 2).contiguous()                  # collapse_dim(x, 2) -> (B, T, C)
        n = xm.shape[1]
        if torch.is_grad_enabled() and (xm.requires_grad or self.subspace_proj.weight.requires_grad):
            # training: the (B T, C) x (C, 2 heads classes) GEMM under torch autograd (the class count is not a multiple of 4,
            # which the library's data-gradient GEMM needs; 128 rows per clip)
            p = F.linear(xm.view(b * n, c), self.subspace_proj.weight, self.subspace_proj.bias)
        else:
            p = ops.linear(xm.view(b * n, c), self.subspace_proj.weight, self.subspace_proj.bias, ops.ACT_NONE)
        p = p.view(b, n, 2, self.num_heads, self.out_dim).permute(2, 0, 3, 1, 4)
        att, val = torch.sigmoid(p[0]).clamp(self.epsilon, 1.0 - self.epsilon), p[1]
        att = att / att.sum(dim=2, keepdim=True)
        return ((att * val).sum(dim=2) * self.head_weight).sum(dim=1)


class InvertedResidual(nn.Module):
    """expand 1x1 -> depthwise kxk -> [SE] -> project 1x1 (+ residual); models/mn/block_types.py:120-181."""

    def __init__(self, cnf: InvertedResidualConfig, se_cnf):
        super().__init__()
        if not (1 <= cnf.stride <= 2):
            raise ValueError("illegal stride value")
        self.cnf = cnf
        self.dw_stride = 1 if cnf.dilation > 1 else cnf.stride        # block_types.py:150
        self.use_res_connect = cnf.stride == 1 and cnf.input_channels == cnf.out_channels
        act = nn.Hardswish if cnf.use_hs else nn.ReLU
        layers: List[nn.Module] = []
        self.i_expand = self.i_se = None
        if cnf.expanded_channels != cnf.input_channels:
            self.i_expand = len(layers)
            layers.append(_conv_bn_act(cnf.input_channels, cnf.expanded_channels, 1, act=act))
        self.i_dw = len(layers)
        dw = _conv_bn_act(cnf.expanded_channels, cnf.expanded_channels, cnf.kernel, self.dw_stride,
                          cnf.expanded_channels, act)
        if cnf.dilation > 1:
            dw[0].dilation, dw[0].padding = (cnf.dilation,) * 2, ((cnf.kernel - 1) // 2 * cnf.dilation,) * 2
        layers.append(dw)
        if cnf.use_se and se_cnf["se_dims"] is not None:
            self.i_se = len(layers)
            layers.append(ConcurrentSEBlock(cnf.expanded_channels, cnf.f_dim, cnf.t_dim, se_cnf))
        self.i_proj = len(layers)
        layers.append(_conv_bn_act(cnf.expanded_channels, cnf.out_channels, 1, act=None))
        self.block = nn.Sequential(*layers)
        self.out_channels = cnf.out_channels
        self._is_cn = cnf.stride > 1


def _fold(conv, bn):
    """Eval-mode BN as per-output-channel (scale, bias): y = conv(x)*scale + bias."""
    scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
    return scale, bn.bias - bn.running_mean * scale


def _pw_mode(Co, Ci):
    """Arithmetic of a 1x1 layer in the eval plan: 'fp32' (exact fp32 MFMA 16x16x4), 'bf16x3' (each fp32
    operand split into two bf16, three bf16 MFMAs, fp32 accumulation: ~2^-16 relative error per product) or
    'bf16' (plain bf16 operands).  EAT_PW_MODE=auto (default) uses bf16x3 from C_in >= 40 on: those layers
    are bound by the 157 TFLOP/s fp32 MFMA rate, and the split kernel measured 1.1-2.7x faster on every one
    of them on MI355X (e.g. 112->672 @ 8x63: 154 vs 206 us, 960->160 @ 4x32: 42 vs 114 us at B=256) at a
    logit error of ~1e-4 against the 1e-3 bar; the narrow streaming layers (C_in 16/24) stay exact fp32.
    EAT_PW_MODE=fp32 forces the exact kernel everywhere."""
    if _PW_MODE == "auto":
        return "bf16x3" if Ci >= 40 else "fp32"
    return _PW_MODE


def _fusable(blk, proj):
    """Blocks that run as ONE kernel in eval mode (csrc/irb.hip): the whole SE-less block (proj) or expand + depthwise of a
    block with a channel-only SE; the library says which geometries it has instantiations for (mn10: blocks 2-4)."""
    cnf = blk.cnf
    if not (blk.i_expand is not None and cnf.input_channels <= _FUSE_MAX_CIN and cnf.dilation == 1
            and (blk.i_se is None or blk.block[blk.i_se].channel_only)):
        return False
    if proj and blk.i_se is not None:
        return False
    act = ops.ACT_HSWISH if cnf.use_hs else ops.ACT_RELU
    return ops.block_fused_supported(cnf.input_channels, cnf.expanded_channels, cnf.out_channels, cnf.kernel, cnf.stride,
                                     act, proj)


def _concurrent_se(se_block, y, pool_c):
    """ConcurrentSEBlock.forward for se_dims beyond 'c' (models/mn/block_types.py:36-42,72-83): every gate is two small
    GEMMs on the MFMA linear kernel over the mean of the other two dimensions; the gated copies are combined by se_agg.
    Scales are positive (sigmoid), so max / min over the gated copies is y times the max / min gate where y >= 0 and
    the min / max gate where y < 0 - no stack of full tensors is materialised."""
    B, C, Fq, T = y.shape
    gates = []
    for se in se_block.conc_se_layers:
        d = se.gate_dim
        if d == 1:
            m = pool_c * (1.0 / (Fq * T))
        else:
            m = y.mean(dim=[k for k in (1, 2, 3) if k != d]).contiguous()
        h = ops.linear(m, se.fc1.weight, se.fc1.bias, ops.ACT_RELU)
        g = ops.linear(h, se.fc2.weight, se.fc2.bias, ops.ACT_SIGMOID)
        shape = [B, 1, 1, 1]
        shape[d] = g.shape[1]
        gates.append(g.view(shape))
    if se_block.se_agg == "add":
        return y * sum(gates)
    if se_block.se_agg == "avg":
        return y * (sum(gates) / len(gates))
    hi = lo = gates[0]
    for g in gates[1:]:
        hi, lo = torch.maximum(hi, g), torch.minimum(lo, g)
    if se_block.se_agg == "min":
        hi, lo = lo, hi
    return torch.where(y >= 0, y * hi, y * lo).contiguous()


def _pack_pw(w2d, scale, bias):
    """(packed weights, bias, mode) of one BN-folded 1x1 layer."""
    mode = _pw_mode(*w2d.shape)
    if mode == "fp32":
        return ops.pw_prepack(w2d, scale), bias, mode
    return ops.pw_prepack_bf16(w2d, scale, split=(mode == "bf16x3")), bias, mode


def _pw(x, pack, Co, act, **kw):
    wp, bias, mode = pack
    if mode == "fp32":
        return ops.pw_conv(x, wp, bias, Co, act, **kw)
    return ops.pw_conv_bf16(x, wp, bias, Co, act, mode == "bf16x3", **kw)


def fold_block(blk):
    """BN-folded, MFMA-packed weights of one static InvertedResidual (also the static blocks of a
    `use_dy_blocks="replace_se"` DyMN, models/dymn/model.py:103)."""
    d = {}
    if blk.i_expand is not None:
        cna = blk.block[blk.i_expand]
        s, b = _fold(cna[0], cna[1])
        d["exp"] = _pack_pw(cna[0].weight.flatten(1), s.contiguous(), b.contiguous())
    cna = blk.block[blk.i_dw]
    s, b = _fold(cna[0], cna[1])
    k = blk.cnf.kernel
    d["dw"] = ((cna[0].weight * s.view(-1, 1, 1, 1)).reshape(-1, k * k).contiguous(), b.contiguous())
    cna = blk.block[blk.i_proj]
    s, b = _fold(cna[0], cna[1])
    d["proj"] = _pack_pw(cna[0].weight.flatten(1), s.contiguous(), b.contiguous())
    if _fusable(blk, True) or _fusable(blk, False):
        # the block kernel (csrc/irb.hip) multiplies on the exact fp32 MFMA: its own fp32 packs
        ce = blk.block[blk.i_expand]
        se_, be_ = _fold(ce[0], ce[1])
        d["exp32"] = (ops.pw_prepack(ce[0].weight.flatten(1), se_.contiguous()), be_.contiguous())
        if _fusable(blk, True):
            d["proj32"] = (ops.pw_prepack(cna[0].weight.flatten(1), s.contiguous()), b.contiguous())
    return d


def fold_front(stem, b0):
    """Packed project layer of the first block when stem + block 0 run as one kernel (csrc/irb.hip, FRONT mode), else None."""
    if (_FUSE_FRONT and stem[0].out_channels == 16 and b0.i_expand is None and b0.i_se is None and b0.use_res_connect
            and b0.cnf.kernel == 3 and b0.cnf.stride == 1 and b0.cnf.dilation == 1):
        cna = b0.block[b0.i_proj]
        s, b = _fold(cna[0], cna[1])
        return ops.pw_prepack(cna[0].weight.flatten(1), s.contiguous()), b.contiguous()
    return None


def run_block(blk, w, x, pool=None):
    """Eval forward of one static InvertedResidual (models/mn/block_types.py:138-181) on the folded weights `w`;
    `pool` is the zeroed (B, C_exp) accumulator of the SE squeeze for blocks that have one."""
    cnf = blk.cnf
    act = ops.ACT_HSWISH if cnf.use_hs else ops.ACT_RELU
    inp = x
    scale = None
    # early, bandwidth-bound blocks: the whole block (without SE) or expand + depthwise (with SE) in
    # one kernel, the expanded tensor stays on chip (csrc/irb.hip); late blocks are MFMA-bound
    # and keep the separate kernels
    # (the library's size guard - 32-bit offsets inside a sample - depends on the REAL plane size: asked per call, so that
    #  very long inputs fall back to the separate kernels instead of failing; fold time only knows the channel counts)
    fused = "exp32" in w and ops.block_fused_supported(cnf.input_channels, cnf.expanded_channels, cnf.out_channels, cnf.kernel,
                                                       cnf.stride, act, "proj32" in w, F=x.shape[2], T=x.shape[3])
    if fused and "proj32" in w:
        return ops.mbconv(x, *w["exp32"], *w["dw"], *w["proj32"], cnf.expanded_channels, cnf.out_channels,
                          cnf.kernel, cnf.stride, act, res=inp if blk.use_res_connect else None)
    if fused:
        x = ops.fused_expand_dw(x, *w["exp32"], *w["dw"], cnf.expanded_channels, cnf.kernel, cnf.stride, act, pool)
    elif (blk.i_expand is not None and w["exp"][2] == "bf16x3" and cnf.dilation == 1
          and ops.expand_dw_eligible(x.shape[1], x.shape[2], x.shape[3], cnf.kernel, cnf.stride)):
        # late blocks with small planes: expand + depthwise in one kernel, the expanded tensor stays in LDS
        x = ops.expand_dw_bf16(x, w["exp"][0], w["exp"][1], w["dw"][0], w["dw"][1], cnf.expanded_channels, cnf.kernel,
                               cnf.stride, act, pool)
    else:
        if blk.i_expand is not None:
            x = _pw(x, w["exp"], cnf.expanded_channels, act)
        if cnf.dilation > 1:
            x = ops.dw_conv_dilated(x, w["dw"][0], w["dw"][1], cnf.kernel, blk.dw_stride, cnf.dilation, act, pool)
        else:
            x = ops.dw_conv(x, w["dw"][0], w["dw"][1], cnf.kernel, cnf.stride, act, pool)
    if pool is not None and not blk.block[blk.i_se].channel_only:
        x = _concurrent_se(blk.block[blk.i_se], x, pool)          # SE over f / t (and c), aggregated: unfused plan
    elif pool is not None:
        se = blk.block[blk.i_se].conc_se_layers[0]
        inv_s = 1.0 / (x.shape[2] * x.shape[3])
        h = ops.linear(pool, se.fc1.weight, se.fc1.bias, ops.ACT_RELU, inv_s)
        scale = ops.linear(h, se.fc2.weight, se.fc2.bias, ops.ACT_SIGMOID)
    return _pw(x, w["proj"], cnf.out_channels, ops.ACT_NONE, in_scale=scale, res=inp if blk.use_res_connect else None)


class _FoldCache:
    """Folded / packed weights keyed on the version counters of their source tensors.

    Version counters catch in-place updates made through autograd-visible ops (`load_state_dict`, `mul_`,
    the default optimizers) but NOT fused optimizers (`Adam(fused=True)`), hipGraph replays or our own
    kernels writing BN buffers through raw pointers - so the cache is also dropped on every train()/eval()
    switch (`invalidate`, called from `MN.train` / `DyMN.train`) and by `GraphedTrainStep`."""

    def __init__(self):
        self.key, self.val = None, None

    def invalidate(self):
        self.key, self.val = None, None

    def get(self, tensors, build):
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        if key != self.key:
            with torch.no_grad():
                self.val = build()
            self.key = key
        return self.val


class MN(nn.Module):
    def __init__(self, inverted_residual_setting, last_channel, num_classes=1000, dropout=0.2,
                 in_conv_kernel=3, in_conv_stride=2, in_channels=1, **kwargs):
        super().__init__()
        if not inverted_residual_setting:
            raise ValueError("The inverted_residual_setting should not be empty")
        if not all(isinstance(s, InvertedResidualConfig) for s in inverted_residual_setting):
            raise TypeError("The inverted_residual_setting should be List[InvertedResidualConfig]")
        if (in_conv_kernel, in_conv_stride, in_channels) != (3, 2, 1):
            raise NotImplementedError("HIP stem kernel is 3x3 / stride 2 / 1 input channel")
        se_cnf = kwargs.get("se_conf", dict(se_dims=[1], se_agg="max", se_r=4))
        f_dim, t_dim = kwargs.get("input_dims", (128, 1000))
        f_dim, t_dim = cnn_out_size(f_dim, 1, 1, 3, 2), cnn_out_size(t_dim, 1, 1, 3, 2)
        c0 = inverted_residual_setting[0].input_channels
        layers = [_conv_bn_act(in_channels, c0, 3, 2, act=nn.Hardswish)]
        for cnf in inverted_residual_setting:
            f_dim, t_dim = cnf.out_size(f_dim), cnf.out_size(t_dim)
            cnf.f_dim, cnf.t_dim = f_dim, t_dim
            layers.append(InvertedResidual(cnf, se_cnf))
        c_last = inverted_residual_setting[-1].out_channels
        layers.append(_conv_bn_act(c_last, 6 * c_last, 1, act=nn.Hardswish))
        self.features = nn.Sequential(*layers)
        self.head_type = kwargs.get("head_type", False)
        if self.head_type == "mlp":
            self.classifier = nn.Sequential(
                nn.AdaptiveAvgPool2d(1), nn.Flatten(start_dim=1), nn.Linear(6 * c_last, last_channel),
                nn.Hardswish(inplace=True), nn.Dropout(p=dropout, inplace=True), nn.Linear(last_channel, num_classes))
        elif self.head_type == "fully_convolutional":       # models/mn/model.py:173-185
            self.classifier = nn.Sequential(nn.Conv2d(6 * c_last, num_classes, (1, 1), bias=False),
                                            nn.BatchNorm2d(num_classes), nn.AdaptiveAvgPool2d((1, 1)))
        elif self.head_type == "multihead_attention_pooling":   # models/mn/model.py:170-172
            self.classifier = MultiHeadAttentionPooling(6 * c_last, num_classes,
                                                        num_heads=kwargs.get("multihead_attention_heads") or 4)
        else:
            raise NotImplementedError(f"Head '{self.head_type}' unknown. Must be one of: 'mlp', "
                                      f"'fully_convolutional', 'multihead_attention_pooling'")
        for m in self.modules():  # models/mn/model.py:199-210
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.zeros_(m.bias)
        self._cache = _FoldCache()
        # dilated tails / squeeze-excitation beyond the channel axis train on per-layer autograd Functions
        # (mn_train.forward_train_modular); everything else on ONE autograd Function that hands its gradients to the
        # data-parallel reducer itself (mn_train.py, dp.py)
        self._modular_train = any(b.cnf.dilation > 1 or (b.i_se is not None and not b.block[b.i_se].channel_only)
                                  for b in self.features[1:-1])
        self._monolithic_backward = not self._modular_train
        # arithmetic of the 1x1 forward / data-gradient GEMMs of the train step (ops.precision): "auto" =
        # exact fp32 below C_in 40, split-operand bf16x3 (fp32-class) above; "fp32"; "bf16" = BASELINE config 3
        self.train_precision = os.environ.get("EAT_TRAIN_PRECISION", "auto")
        # storage of the WIDE activation tensors of the train step (mn_train.py): "fp32", or "bf16" = the reference's 16-bit
        # mixed precision (ex_pl_audioset.py:287-293; BASELINE configs[2]) - z_e, z_d, y_d and the gradients arriving at them
        # live in bf16 in HBM; needs train_precision "bf16" (plain bf16 GEMM operands), anything else is a loud error
        self.act_storage = os.environ.get("EAT_ACT_STORAGE", "fp32")

    def train(self, mode: bool = True):
        """nn.Module.train plus dropping the folded eval weights: parameters may have been updated by paths
        that do not bump tensor versions (fused optimizers, graph replays), see _FoldCache."""
        if getattr(self, "_cache", None) is not None:
            self._cache.invalidate()
        return super().train(mode)

    # ------------------------------------------------------------------ folded weights
    def _fold_sources(self):
        mods = list(self.features.modules()) + (list(self.classifier.modules()) if self.head_type == "fully_convolutional" else [])
        return [t for m in mods if isinstance(m, (nn.Conv2d, nn.BatchNorm2d))
                for t in ([m.weight] if isinstance(m, nn.Conv2d) else
                          [m.weight, m.bias, m.running_mean, m.running_var])]

    def _build_folded(self):
        out = {}
        stem = self.features[0]
        s, b = _fold(stem[0], stem[1])
        out["stem"] = ((stem[0].weight * s.view(-1, 1, 1, 1)).reshape(-1, 9).contiguous(), b.contiguous())
        for i, blk in enumerate(self.features[1:-1]):
            out[i] = fold_block(blk)
        front = fold_front(stem, self.features[1])
        if front is not None:
            out["front"] = front
        last = self.features[-1]
        s, b = _fold(last[0], last[1])
        out["last"] = _pack_pw(last[0].weight.flatten(1), s.contiguous(), b.contiguous())
        if self.head_type == "fully_convolutional":
            # eval: mean_s BN(conv1x1(x)) = (scale * W) mean_s(x) + bias - the head collapses onto the pooled features
            s, b = _fold(self.classifier[0], self.classifier[1])
            out["fc_head"] = ((self.classifier[0].weight.flatten(1) * s.view(-1, 1)).contiguous(), b.contiguous())
        return out

    # --------------------------------------------------------------------------- forward
    def _forward_impl(self, x, return_fmaps: bool = False):
        if not x.is_cuda:
            raise ops._lib.EatHipError("MN.forward needs a GPU tensor: efficientat_amd has no CPU path")
        if self.training:
            if self._modular_train:
                from .mn_train import forward_train_modular
                return forward_train_modular(self, x, return_fmaps)
            from .mn_train import forward_train
            return forward_train(self, x, return_fmaps)
        W = self._cache.get(self._fold_sources(), self._build_folded)
        x = x.contiguous().float()
        B = x.shape[0]
        fmaps = []
        blocks = list(self.features[1:-1])
        se_sizes = [b.cnf.expanded_channels for b in blocks if b.i_se is not None]
        c_feat = self.features[-1].out_channels
        # one zeroed arena for every fused-pool accumulator of this forward (SE squeezes + head pool)
        arena = torch.zeros((B * (sum(se_sizes) + c_feat),), device=x.device, dtype=torch.float32)
        off = 0

        def take(c):
            nonlocal off
            t = arena[off:off + B * c].view(B, c)
            off += B * c
            return t

        first = 0
        if "front" in W and not return_fmaps:
            # stem + first block in one kernel (csrc/irb.hip, FRONT mode): the 16 x 64 x 500 stem map never leaves the CU
            c0 = blocks[0].cnf
            x = ops.front(x, *W["stem"], *W[0]["dw"], *W["front"], ops.ACT_HSWISH if c0.use_hs else ops.ACT_RELU)
            first = 1
        else:
            x = ops.stem_conv(x, *W["stem"], ops.ACT_HSWISH)
            if return_fmaps:
                fmaps.append(x)
        for i, blk in enumerate(blocks):
            if i < first:
                continue
            x = run_block(blk, W[i], x, take(blk.cnf.expanded_channels) if blk.i_se is not None else None)
            if return_fmaps:
                fmaps.append(x)
        pooled = take(c_feat)
        S = x.shape[2] * x.shape[3]
        need_map = return_fmaps or self.head_type == "multihead_attention_pooling"
        y = _pw(x, W["last"], c_feat, ops.ACT_HSWISH, pool=pooled, write=need_map)
        if return_fmaps:
            fmaps.append(y)
        if self.head_type == "mlp":
            fc1, fc2 = self.classifier[2], self.classifier[5]
            h = ops.linear(pooled, fc1.weight, fc1.bias, ops.ACT_HSWISH, 1.0 / S)
            logits = ops.linear(h, fc2.weight, fc2.bias, ops.ACT_NONE)
        elif self.head_type == "fully_convolutional":
            logits = ops.linear(pooled, W["fc_head"][0], W["fc_head"][1], ops.ACT_NONE, 1.0 / S)
        else:
            logits = self.classifier(y)
        if return_fmaps:
            return logits, fmaps
        return logits, pooled * (1.0 / S)

    def forward(self, x):
        return self._forward_impl(x)


def _mobilenet_v3_conf(width_mult=1.0, reduced_tail=False, dilated=False, strides=(2, 2, 2, 2), **kwargs):
    """The 15-row MobileNetV3-large table scaled by width_mult (models/mn/model.py:237-271)."""
    div = 2 if reduced_tail else 1
    dil = 2 if dilated else 1
    row = partial(InvertedResidualConfig, width_mult=width_mult)
    c160, c960 = 160 // div, 960 // div
    spec = [
        (16, 3, 16, 16, False, "RE", 1, 1), (16, 3, 64, 24, False, "RE", strides[0], 1),
        (24, 3, 72, 24, False, "RE", 1, 1), (24, 5, 72, 40, True, "RE", strides[1], 1),
        (40, 5, 120, 40, True, "RE", 1, 1), (40, 5, 120, 40, True, "RE", 1, 1),
        (40, 3, 240, 80, False, "HS", strides[2], 1), (80, 3, 200, 80, False, "HS", 1, 1),
        (80, 3, 184, 80, False, "HS", 1, 1), (80, 3, 184, 80, False, "HS", 1, 1),
        (80, 3, 480, 112, True, "HS", 1, 1), (112, 3, 672, 112, True, "HS", 1, 1),
        (112, 5, 672, c160, True, "HS", strides[3], dil), (c160, 5, c960, c160, True, "HS", 1, dil),
        (c160, 5, c960, c160, True, "HS", 1, dil),
    ]
    return [row(*r) for r in spec], InvertedResidualConfig.adjust_channels(1280 // div, width_mult)


def _mobilenet_v3(inverted_residual_setting, last_channel, pretrained_name, **kwargs):
    """Build + optional checkpoint load with class-count surgery (models/mn/model.py:274-313)."""
    model = MN(inverted_residual_setting, last_channel, **kwargs)
    if pretrained_name in pretrained_models:
        from torch.hub import load_state_dict_from_url
        state_dict = load_state_dict_from_url(pretrained_models[pretrained_name], model_dir=model_dir,
                                              map_location="cpu")
        n_ckpt = state_dict["classifier.5.bias"].size(0)
        if kwargs["num_classes"] != n_ckpt:
            print(f"Number of classes defined: {kwargs['num_classes']}, "
                  f"but try to load pre-trained layer with logits: {n_ckpt}\nDropping last layer.")
            del state_dict["classifier.5.weight"], state_dict["classifier.5.bias"]
        try:
            model.load_state_dict(state_dict)
        except RuntimeError as e:
            print(str(e))
            print("Loading weights pre-trained weights in a non-strict manner.")
            model.load_state_dict(state_dict, strict=False)
    elif pretrained_name:
        raise NotImplementedError(f"Model name '{pretrained_name}' unknown.")
    return model


def mobilenet_v3(pretrained_name: Optional[str] = None, **kwargs) -> MN:
    setting, last_channel = _mobilenet_v3_conf(**kwargs)
    return _mobilenet_v3(setting, last_channel, pretrained_name, **kwargs)


def get_model(num_classes: int = 527, pretrained_name: str = None, width_mult: float = 1.0,
              reduced_tail: bool = False, dilated: bool = False, strides=(2, 2, 2, 2),
              head_type: str = "mlp", multihead_attention_heads: int = 4, input_dim_f: int = 128,
              input_dim_t: int = 1000, se_dims: str = "c", se_agg: str = "max", se_r: int = 4):
    """Same signature as models/mn/model.py:326-329."""
    dim_map = {"c": 1, "f": 2, "t": 3}
    assert len(se_dims) <= 3 and all(s in dim_map for s in se_dims) or se_dims == "none"
    se_conf = dict(se_dims=None if se_dims == "none" else [dim_map[s] for s in se_dims], se_agg=se_agg, se_r=se_r)
    m = mobilenet_v3(pretrained_name=pretrained_name, num_classes=num_classes, width_mult=width_mult,
                     reduced_tail=reduced_tail, dilated=dilated, strides=strides, head_type=head_type,
                     multihead_attention_heads=multihead_attention_heads,
                     input_dims=(input_dim_f, input_dim_t), se_conf=se_conf)
    print(m)
    return m
