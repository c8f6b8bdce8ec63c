"""Drop-in for the reference's ``models.dymn.model`` (get_model / DyMN) on the HIP hot path.

Same public API and state_dict key layout as models/dymn/model.py:36-361 and
models/dymn/dy_block.py:44-409 (``in_c``, ``layers.{i}.{exp,depth,proj}_conv.weight (1,1,K,N)`` +
``.residuals.0``, ``.{exp,depth,proj}_norm``, ``.depth_act.{lambdas,init_v,coef_net.0}``,
``.context_gen.{joint_conv,joint_norm,conv_f,conv_t}``, ``out_c``, ``classifier``), so released
checkpoints load with ``strict=True``.  The modules only hold parameters; ``DyMN.forward`` runs
a launch plan over libeat_hip.so:

  per DY_Block:  ctx_pool (row/col means of the block input)
                 -> joint 1x1 conv + BN + hardswish over the (F+T) sequence   [MFMA linear kernel]
                 -> h_c (mean), g_cf / g_ct (two 1x1 convs)                   [MFMA linear kernel]
                 -> expand  : softmax attention over K kernels -> per-sample packed weights
                              (eat_dyn_pw_pack) -> 1x1 MFMA conv with per-sample weights + BN + act
                 -> depth   : per-(b,c) aggregated taps (eat_dyn_aggregate) -> sliding-window
                              depthwise conv + BN with DyReLU-B and CoordAtt fused in the epilogue
                 -> project : per-sample packed weights -> 1x1 MFMA conv + BN (+ residual)

The reference materialises (B*Cout, Cin/g, k, k) weights and runs a grouped conv with groups*B
(dy_block.py:111-127).  Sequence-level glue on (B, L, H)-shaped tensors (mean over L, 3-tap average
pool, softmax over K=4, sigmoid of the DyReLU coefficients) uses torch ops; every pass over a
feature map and every GEMM runs in the library.  Train mode: dymn_train.py.
"""
import os
from functools import partial

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .mn import BN_EPS, BN_MOMENTUM, InvertedResidual, _FoldCache, _conv_bn_act, _fold, fold_block, fold_front, run_block
from .utils import cnn_out_size, make_divisible

model_url = "https://github.com/fschmid56/EfficientAT/releases/download/v0.0.1/"
model_dir = "resources"
_CKPT = {"dymn04_im": "dymn04_im.pt", "dymn10_im": "dymn10_im.pt", "dymn20_im": "dymn20_im.pt",
         "dymn04_as": "dymn04_as.pt", "dymn10_as": "dymn10_as.pt", "dymn20_as": "dymn20_as_mAP_493.pt",
         "dymn20_as(1)": "dymn20_as.pt", "dymn20_as(2)": "dymn20_as_mAP_489.pt", "dymn20_as(3)": "dymn20_as_mAP_490.pt",
         "dymn04_replace_se_as": "dymn04_replace_se_as.pt", "dymn10_replace_se_as": " dymn10_replace_se_as.pt"}
pretrained_models = {k: model_url + v for k, v in _CKPT.items()}


class DynamicInvertedResidualConfig:
    """models/dymn/dy_block.py:11-41."""

    def __init__(self, input_channels, kernel, expanded_channels, out_channels, use_dy_block, activation, stride,
                 dilation, width_mult):
        adj = self.adjust_channels
        self.input_channels = adj(input_channels, width_mult)
        self.kernel = kernel
        self.expanded_channels = adj(expanded_channels, width_mult)
        self.out_channels = adj(out_channels, width_mult)
        self.use_dy_block, self.use_hs, self.use_se = use_dy_block, activation == "HS", False
        self.stride, self.dilation, self.width_mult = stride, dilation, width_mult

    @staticmethod
    def adjust_channels(channels, width_mult):
        return make_divisible(channels * width_mult, 8)

    def out_size(self, in_size):
        return cnn_out_size(in_size, (self.kernel - 1) // 2 * self.dilation, self.dilation, self.kernel, self.stride)


class DynamicConv(nn.Module):
    """Holder of the K-kernel bank and the attention Linear (dy_block.py:44-139)."""

    def __init__(self, in_channels, out_channels, context_dim, kernel_size, stride=1, groups=1, k=4,
                 temp_schedule=(30, 1, 1, 0.05)):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride, self.groups, self.k = stride, groups, k
        self.padding, self.dilation, self.att_groups = (kernel_size - 1) // 2, 1, 1
        self.T_max, self.T_min, self.T0_slope, self.T1_slope = temp_schedule
        self.temperature = self.T_max
        self.residuals = nn.Sequential(nn.Linear(context_dim, k))
        w = torch.randn(k, out_channels, in_channels // groups, kernel_size, kernel_size)
        for i in range(k):
            nn.init.kaiming_normal_(w[i], mode="fan_out")
        self.weight = nn.Parameter(w.view(1, 1, k, -1))
        self.bias = None

    def update_params(self, epoch):
        """Temperature schedule of the kernel attention (dy_block.py:133-139)."""
        t0 = self.T_max - self.T0_slope * epoch
        t1 = 1 + self.T1_slope * (self.T_max - 1) / self.T0_slope - self.T1_slope * epoch
        self.temperature = max(t0, t1, self.T_min)
        print(f"Setting temperature for attention over kernels to {self.temperature}")


class DyReLUB(nn.Module):
    """Holder of the DyReLU-B coefficient net and its buffers (dy_block.py:142-188)."""

    def __init__(self, channels, context_dim, M=2):
        super().__init__()
        if M != 2:
            raise NotImplementedError("HIP path implements DyReLU-B with M=2 linear pieces")
        self.channels, self.M = channels, M
        self.coef_net = nn.Sequential(nn.Linear(context_dim, 2 * M * channels))
        self.sigmoid = nn.Sigmoid()
        self.register_buffer("lambdas", torch.Tensor([1.0] * M + [0.5] * M).float())
        self.register_buffer("init_v", torch.Tensor([1.0] + [0.0] * (2 * M - 1)).float())


class CoordAtt(nn.Module):
    pass


class DynamicWrapper(nn.Module):
    """Holder of a static module in a dynamic slot (dy_block.py:204-211): gives the `.module.` level of the reference's
    state-dict keys for the ablated blocks (`no_dyconv`, `no_dyrelu`, `no_ca`)."""

    def __init__(self, module):
        super().__init__()
        self.module = module


class ContextGen(nn.Module):
    """Parameter holder of the context generator (dy_block.py:214-254)."""

    def __init__(self, context_dim, in_ch, exp_ch, stride=1):
        super().__init__()
        self.joint_conv = nn.Conv2d(in_ch, context_dim, 1, bias=False)
        self.joint_norm = nn.BatchNorm2d(context_dim, eps=BN_EPS, momentum=BN_MOMENTUM)
        self.joint_act = nn.Hardswish(inplace=True)
        self.conv_f = nn.Conv2d(context_dim, exp_ch, 1)
        self.conv_t = nn.Conv2d(context_dim, exp_ch, 1)
        self.stride = stride


class DY_Block(nn.Module):
    def __init__(self, cnf, context_ratio=4, max_context_size=128, min_context_size=32,
                 temp_schedule=(30, 1, 1, 0.05), dyrelu_k=2, dyconv_k=4, no_dyrelu=False, no_dyconv=False,
                 no_ca=False, **kwargs):
        super().__init__()
        if not (1 <= cnf.stride <= 2):
            raise ValueError("illegal stride value")
        self.cnf = cnf
        # a dilated block runs its depthwise conv and its context pooling at stride 1 (dy_block.py:322,385-386)
        self.dw_stride = 1 if cnf.dilation > 1 else cnf.stride
        # ablations of the block (dy_block.py:269-271): static convs / plain activation / no coordinate attention
        self.no_dyrelu, self.no_dyconv, self.no_ca = bool(no_dyrelu), bool(no_dyconv), bool(no_ca)
        self.use_res_connect = cnf.stride == 1 and cnf.input_channels == cnf.out_channels
        self.context_dim = int(np.clip(make_divisible(cnf.expanded_channels // context_ratio, 8),
                                       make_divisible(min_context_size * cnf.width_mult, 8),
                                       make_divisible(max_context_size * cnf.width_mult, 8)))
        H, cin, cexp, cout = self.context_dim, cnf.input_channels, cnf.expanded_channels, cnf.out_channels
        norm = partial(nn.BatchNorm2d, eps=BN_EPS, momentum=BN_MOMENTUM)
        act_layer = nn.Hardswish if cnf.use_hs else nn.ReLU

        def conv(ci, co, k, stride=1, groups=1, dilation=1):
            if no_dyconv:                                            # dy_block.py:291-302,320-332,359-370
                return DynamicWrapper(nn.Conv2d(ci, co, (k, k), (stride, stride), (k - 1) // 2 * dilation,
                                                dilation=(dilation, dilation), groups=groups, bias=False))
            dc = DynamicConv(ci, co, H, k, stride=stride, groups=groups, k=dyconv_k, temp_schedule=temp_schedule)
            dc.dilation, dc.padding = dilation, (k - 1) // 2 * dilation
            return dc

        self.has_expand = cexp != cin
        if self.has_expand:
            self.exp_conv = conv(cin, cexp, 1)
            self.exp_norm = norm(cexp)
            self.exp_act = act_layer(inplace=True)
        else:
            self.exp_conv, self.exp_norm, self.exp_act = nn.Identity(), nn.Identity(), nn.Identity()
        self.depth_conv = conv(cexp, cexp, cnf.kernel, self.dw_stride, cexp, cnf.dilation)
        self.depth_norm = norm(cexp)
        self.depth_act = DynamicWrapper(act_layer(inplace=True)) if no_dyrelu else DyReLUB(cexp, H, M=dyrelu_k)
        self.ca = DynamicWrapper(nn.Identity()) if no_ca else CoordAtt()
        self.proj_conv = conv(cexp, cout, 1)
        self.proj_norm = norm(cout)
        self.context_gen = ContextGen(H, cin, cexp, stride=self.dw_stride)


def _pool3(t, stride):
    """AvgPool(kernel 3, stride, padding 1, count_include_pad) along dim 1 of a (B, L, H) sequence
    (ContextGen.pool_f / pool_t, dy_block.py:227-229), written with pad + strided slices so that it
    stays in the position-major layout (avg_pool1d on the transposed view gave wrong input
    gradients on ROCm for this non-contiguous layout)."""
    L = t.shape[1]
    Lo = (L + 2 - 3) // stride + 1
    tp = F.pad(t, (0, 0, 1, 1))
    hi = stride * (Lo - 1)
    return (tp[:, 0:hi + 1:stride] + tp[:, 1:hi + 2:stride] + tp[:, 2:hi + 3:stride]) / 3.0


def _attention(conv, h_c):
    logits = ops.linear(h_c, conv.residuals[0].weight, conv.residuals[0].bias, ops.ACT_NONE)
    return F.softmax(logits / conv.temperature, dim=-1).contiguous()


class DyMN(nn.Module):
    def __init__(self, inverted_residual_setting, last_channel, num_classes=527, head_type="mlp", block=None,
                 dropout=0.2, in_conv_kernel=3, in_conv_stride=2, in_channels=1, context_ratio=4,
                 max_context_size=128, min_context_size=32, dyrelu_k=2, dyconv_k=4, no_dyrelu=False,
                 no_dyconv=False, no_ca=False, temp_schedule=(30, 1, 1, 0.05), **kwargs):
        super().__init__()
        if not inverted_residual_setting:
            raise ValueError("The inverted_residual_setting should not be empty")
        if not all(isinstance(s, DynamicInvertedResidualConfig) for s in inverted_residual_setting):
            raise TypeError("The inverted_residual_setting should be List[DynamicInvertedResidualConfig]")
        if (in_conv_kernel, in_conv_stride, in_channels) != (3, 2, 1):
            raise NotImplementedError("HIP stem kernel is 3x3 / stride 2 / 1 input channel")
        c0 = inverted_residual_setting[0].input_channels
        self.in_c = _conv_bn_act(in_channels, c0, 3, 2, act=nn.Hardswish)
        self.layers = nn.ModuleList()
        for cnf in inverted_residual_setting:
            if not cnf.use_dy_block:
                # use_dy_blocks="replace_se": a plain SE-less inverted residual (models/dymn/model.py:102-103)
                self.layers.append(InvertedResidual(cnf, None))
                continue
            self.layers.append(DY_Block(cnf, context_ratio=context_ratio, max_context_size=max_context_size,
                                        min_context_size=min_context_size, dyrelu_k=dyrelu_k, dyconv_k=dyconv_k,
                                        no_dyrelu=no_dyrelu, no_dyconv=no_dyconv, no_ca=no_ca,
                                        temp_schedule=temp_schedule))
        c_last = inverted_residual_setting[-1].out_channels
        self.out_c = _conv_bn_act(c_last, 6 * c_last, 1, act=nn.Hardswish)
        self.head_type = head_type
        if head_type == "fully_convolutional":               # models/dymn/model.py:119-130
            self.classifier = nn.Sequential(nn.Conv2d(6 * c_last, num_classes, (1, 1), bias=False),
                                            nn.BatchNorm2d(num_classes), nn.AdaptiveAvgPool2d((1, 1)))
        elif head_type == "mlp":
            self.classifier = nn.Sequential(
                nn.AdaptiveAvgPool2d(1), nn.Flatten(start_dim=1), nn.Linear(6 * c_last, last_channel),
                nn.Hardswish(inplace=True), nn.Dropout(p=dropout, inplace=True), nn.Linear(last_channel, num_classes))
        else:
            raise NotImplementedError(f"Head '{head_type}' unknown. Must be one of: 'mlp', "
                                      f"'fully_convolutional', 'multihead_attention_pooling'")
        for m in self.modules():   # models/dymn/model.py:144-155
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.zeros_(m.bias)
        self._cache = _FoldCache()
        # arithmetic of the 1x1 convs of a train-mode pass (ops.precision; same switch and default as MN)
        self.train_precision = os.environ.get("EAT_TRAIN_PRECISION", "auto")
        # storage of the wide activations of a train-mode pass (same switch as MN): "bf16" = z_e, z_d, the DyReLU * CoordAtt
        # output and the gradients arriving at them live in bf16 in HBM in every fully dynamic block the kernels cover
        # (dymn_train.DyBlockMain; the reference's `precision=16`, ex_pl_audioset.py:287-293); needs train_precision "bf16"
        self.act_storage = os.environ.get("EAT_ACT_STORAGE", "fp32")

    # ----------------------------------------------------------------------- folded BN (eval)
    def _fold_sources(self):
        return [t for m in self.modules() if isinstance(m, nn.BatchNorm2d)
                for t in (m.weight, m.bias, m.running_mean, m.running_var)] + \
               [self.in_c[0].weight, self.out_c[0].weight] + \
               [blk.context_gen.joint_conv.weight for blk in self.layers if isinstance(blk, DY_Block)] + \
               [m.weight for m in self.modules() if isinstance(m, DynamicConv)] + \
               [m.module.weight for m in self.modules() if isinstance(m, DynamicWrapper)
                and isinstance(m.module, nn.Conv2d)] + \
               [m.weight for blk in self.layers if isinstance(blk, InvertedResidual) for m in blk.modules()
                if isinstance(m, nn.Conv2d)]

    def _build_folded(self):
        out = {}
        s, b = _fold(self.in_c[0], self.in_c[1])
        out["stem"] = ((self.in_c[0].weight * s.view(-1, 1, 1, 1)).reshape(-1, 9).contiguous(), b.contiguous())
        for i, blk in enumerate(self.layers):
            if isinstance(blk, InvertedResidual):
                out[i] = fold_block(blk)
                continue
            d = {}
            cg = blk.context_gen
            s, b = _fold(None, cg.joint_norm)
            d["joint"] = ((cg.joint_conv.weight.flatten(1) * s.view(-1, 1)).contiguous(), b.contiguous())
            for name in ("exp", "depth", "proj"):
                bn = getattr(blk, name + "_norm")
                if isinstance(bn, nn.BatchNorm2d):
                    s, b = _fold(None, bn)
                    d[name] = (s.contiguous(), b.contiguous())
                    conv = getattr(blk, name + "_conv")
                    if isinstance(conv, DynamicWrapper):             # `no_dyconv`: static weights, BN scale folded in
                        wt = conv.module.weight
                        if name == "depth":
                            d["depth_taps"] = (wt.flatten(1) * s.view(-1, 1)).contiguous()
                        else:
                            d[name + "_w"] = ops.pw_prepack(wt.flatten(1), s.contiguous())
            out[i] = d
        if isinstance(self.layers[0], InvertedResidual):
            front = fold_front(self.in_c, self.layers[0])
            if front is not None:
                out["front"] = front
        s, b = _fold(self.out_c[0], self.out_c[1])
        out["last"] = (ops.pw_prepack(self.out_c[0].weight.flatten(1), s.contiguous()), b.contiguous())
        return out

    # --------------------------------------------------------------------------------- forward
    def _block_forward(self, blk, w, x):
        cnf = blk.cnf
        B, cin, Fq, T = x.shape
        H, cexp, cout, k, stride = blk.context_dim, cnf.expanded_channels, cnf.out_channels, cnf.kernel, blk.dw_stride
        dil = cnf.dilation
        act = ops.ACT_HSWISH if cnf.use_hs else ops.ACT_RELU
        inp = x
        # ---- context generator (dy_block.py:235-254); the ablated blocks (`no_dyconv`, `no_dyrelu`, `no_ca`) only
        # evaluate the parts of it that something still consumes
        cg = blk.context_gen
        need_hc = not (blk.no_dyconv and blk.no_dyrelu)
        h_c = g_cf = g_ct = None
        if need_hc or not blk.no_ca:
            seq = ops.ctx_pool(x)                                                  # (B, F+T, cin)
            g = ops.linear(seq.view(B * (Fq + T), cin), w["joint"][0], w["joint"][1], ops.ACT_HSWISH).view(B, Fq + T, H)
            if need_hc:
                h_c = g.mean(dim=1)
            if not blk.no_ca:
                h_cf, h_ct = g[:, :Fq], g[:, Fq:]
                if stride > 1:
                    h_cf, h_ct = _pool3(h_cf, stride), _pool3(h_ct, stride)
                Fo, To = h_cf.shape[1], h_ct.shape[1]
                g_cf = ops.linear(h_cf.reshape(B * Fo, H), cg.conv_f.weight.flatten(1), cg.conv_f.bias, ops.ACT_NONE)
                g_ct = ops.linear(h_ct.reshape(B * To, H), cg.conv_t.weight.flatten(1), cg.conv_t.bias, ops.ACT_NONE)
        # ---- expand (dynamic 1x1)
        if blk.has_expand:
            if blk.no_dyconv:
                x = ops.pw_conv(x, w["exp_w"], w["exp"][1], cexp, act)
            else:
                x = self._dyn_pw(blk.exp_conv, w, "exp", x, _attention(blk.exp_conv, h_c), cexp, cin, act)
        # ---- depthwise (dynamic taps) + BN + DyReLU-B + CoordAtt
        if blk.no_dyconv and blk.no_dyrelu and blk.no_ca:
            # nothing dynamic left: the static kernels
            x = (ops.dw_conv(x, w["depth_taps"], w["depth"][1], k, stride, act) if dil == 1 else
                 ops.dw_conv_dilated(x, w["depth_taps"], w["depth"][1], k, stride, dil, act))
        elif dil > 1:
            # dilated dynamic block (models/dymn/model.py:212-218,246-250; dy_block.py:322-348): the per-sample taps make the
            # depthwise conv a conv over B * cexp independent planes - the batch folded into the channel axis of the
            # generic dilated depthwise kernel - followed by the stand-alone BatchNorm + DyReLU-B + CoordAtt kernel
            if blk.no_dyconv:
                taps = (blk.depth_conv.module.weight.flatten(1)).unsqueeze(0).expand(B, -1, -1).reshape(B * cexp, k * k).contiguous()
            else:
                att = _attention(blk.depth_conv, h_c)
                taps = ops.dyn_aggregate(blk.depth_conv.weight.view(blk.depth_conv.k, -1), att)
            z = ops.dw_conv_dyn_dilated(x.contiguous(), taps.view(B, cexp * k * k), k, stride, dil)
            x = ops.dyrelu_ca(z, w["depth"][0], w["depth"][1], act if blk.no_dyrelu else None,
                              None if blk.no_dyrelu else self._dyrelu_coef(blk, h_c, B, cexp), g_cf, g_ct)
        else:
            if blk.no_dyconv:
                taps = w["depth_taps"].unsqueeze(0).expand(B, -1, -1).reshape(B, -1).contiguous()
            else:
                att = _attention(blk.depth_conv, h_c)
                taps = ops.dyn_aggregate(blk.depth_conv.weight.view(blk.depth_conv.k, -1), att, w["depth"][0], k * k)
            coef = None if blk.no_dyrelu else self._dyrelu_coef(blk, h_c, B, cexp)
            if coef is not None and g_cf is not None:
                x = ops.dw_conv_dyn(x, taps, w["depth"][1], coef, g_cf, g_ct, k, stride)
            else:
                x = ops.dw_conv_dyn_act(x, taps, w["depth"][1], act if blk.no_dyrelu else ops.ACT_NONE, coef, g_cf, g_ct,
                                        k, stride)
        # ---- project (dynamic 1x1) + BN (+ residual)
        res = inp if blk.use_res_connect else None
        if blk.no_dyconv:
            return ops.pw_conv(x, w["proj_w"], w["proj"][1], cout, ops.ACT_NONE, res=res)
        return self._dyn_pw(blk.proj_conv, w, "proj", x, _attention(blk.proj_conv, h_c), cout, cexp, ops.ACT_NONE, res=res)

    @staticmethod
    def _dyrelu_coef(blk, h_c, B, cexp):
        """DyReLU-B coefficients (dy_block.py:176-181): (2 sigmoid(coef_net(h_c)) - 1) * lambdas + init_v, (B, cexp, 4)."""
        da = blk.depth_act
        theta = 2.0 * torch.sigmoid(ops.linear(h_c, da.coef_net[0].weight, da.coef_net[0].bias, ops.ACT_NONE)) - 1.0
        return (theta.view(B, cexp, 4) * da.lambdas + da.init_v).contiguous()

    @staticmethod
    def _dyn_pw(conv, w, name, x, att, Co, Ci, act, res=None):
        """Dynamic 1x1 conv + folded BN (models/dymn/dy_block.py:103-131).  Late, small-plane layers (a sample's
        aggregated weight matrix larger than its activations): ONE GEMM over the K-concatenated banks with the attention
        as input scale - no per-sample weights exist (`ops.kcat_eligible`; the packed banks are cached with the folded
        weights).  Elsewhere: aggregate + pack per sample, then the per-sample-weight GEMM."""
        scale, bias = w[name]
        bank = conv.weight.view(conv.k, -1)
        if ops.kcat_eligible(Co, Ci, x.shape[2] * x.shape[3]):
            key = name + "_cat"
            if key not in w:
                w[key] = ops.kcat_pack(bank, Co, Ci, scale)
            return ops.pw_conv_kcat(x, w[key], bias, att, Co, act, res=res)
        wp = ops.dyn_pw_pack(bank, att, Co, Ci, scale)
        return ops.pw_conv_dyn(x, wp, bias, Co, act, res=res)

    def train(self, mode: bool = True):
        """nn.Module.train plus dropping the folded eval weights (see mn._FoldCache)."""
        if getattr(self, "_cache", None) is not None:
            self._cache.invalidate()
        return super().train(mode)

    def _forward_impl(self, x, return_fmaps=False):
        if not x.is_cuda:
            raise ops._lib.EatHipError("DyMN.forward needs a GPU tensor: efficientat_amd has no CPU path")
        if self.training:
            from .dymn_train import forward_train
            return forward_train(self, x, return_fmaps)
        W = self._cache.get(self._fold_sources(), self._build_folded)
        x = x.contiguous().float()
        B = x.shape[0]
        fmaps = []
        first = 0
        if "front" in W and not return_fmaps:
            # static first block (replace_se): stem + block in one kernel, as in MN (csrc/irb.hip, FRONT mode)
            x = ops.front(x, *W["stem"], *W[0]["dw"], *W["front"],
                          ops.ACT_HSWISH if self.layers[0].cnf.use_hs else ops.ACT_RELU)
            first = 1
        else:
            x = ops.stem_conv(x, *W["stem"], ops.ACT_HSWISH)
            fmaps.append(x)
        for i, blk in enumerate(self.layers):
            if i < first:
                continue
            x = run_block(blk, W[i], x) if isinstance(blk, InvertedResidual) else self._block_forward(blk, W[i], x)
            fmaps.append(x)
        c_feat = self.out_c.out_channels
        pooled = torch.zeros((B, c_feat), device=x.device, dtype=torch.float32)
        S = x.shape[2] * x.shape[3]
        y = ops.pw_conv(x, W["last"][0], W["last"][1], c_feat, ops.ACT_HSWISH, pool=pooled, write=return_fmaps)
        if self.head_type == "fully_convolutional":
            # eval: mean_s BN(conv1x1(x)) = (scale * W) mean_s(x) + bias - the head collapses onto the pooled features
            conv, bn = self.classifier[0], self.classifier[1]
            with torch.no_grad():
                sc = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
                wf = (conv.weight.flatten(1) * sc.view(-1, 1)).contiguous()
                bf = (bn.bias - bn.running_mean * sc).contiguous()
            logits = ops.linear(pooled, wf, bf, ops.ACT_NONE, 1.0 / S)
        else:
            fc1, fc2 = self.classifier[2], self.classifier[5]
            h = ops.linear(pooled, fc1.weight, fc1.bias, ops.ACT_HSWISH, 1.0 / S)
            logits = ops.linear(h, fc2.weight, fc2.bias, ops.ACT_NONE)
        if return_fmaps:
            return logits, fmaps + [y]
        return logits, pooled * (1.0 / S)

    def forward(self, x, return_fmaps=False):
        return self._forward_impl(x, return_fmaps)

    def update_params(self, epoch):
        for m in self.modules():
            if isinstance(m, DynamicConv):
                m.update_params(epoch)


def _dymn_conf(width_mult=1.0, reduced_tail=False, dilated=False, strides=(2, 2, 2, 2), use_dy_blocks="all", **kwargs):
    """models/dymn/model.py:209-254."""
    div = 2 if reduced_tail else 1
    dil = 2 if dilated else 1
    if use_dy_blocks == "all":
        dy = [True] * 15
    elif use_dy_blocks == "replace_se":
        dy = [False, False, False, True, True, True, False, False, False, False, True, True, True, True, True]
    else:
        raise NotImplementedError(f"Config use_dy_blocks={use_dy_blocks} not implemented.")
    row = partial(DynamicInvertedResidualConfig, width_mult=width_mult)
    c160, c960 = 160 // div, 960 // div
    spec = [(16, 3, 16, 16, 1, 1), (16, 3, 64, 24, strides[0], 1), (24, 3, 72, 24, 1, 1), (24, 5, 72, 40, strides[1], 1),
            (40, 5, 120, 40, 1, 1), (40, 5, 120, 40, 1, 1), (40, 3, 240, 80, strides[2], 1), (80, 3, 200, 80, 1, 1),
            (80, 3, 184, 80, 1, 1), (80, 3, 184, 80, 1, 1), (80, 3, 480, 112, 1, 1), (112, 3, 672, 112, 1, 1),
            (112, 5, 672, c160, strides[3], dil), (c160, 5, c960, c160, 1, dil), (c160, 5, c960, c160, 1, dil)]
    acts = ["RE"] * 6 + ["HS"] * 9
    setting = [row(ci, k, ce, co, dy[i], acts[i], s, d) for i, (ci, k, ce, co, s, d) in enumerate(spec)]
    return setting, DynamicInvertedResidualConfig.adjust_channels(1280 // div, width_mult)


def _dymn(inverted_residual_setting, last_channel, pretrained_name, **kwargs):
    """Build + optional checkpoint load (models/dymn/model.py:257-281)."""
    model = DyMN(inverted_residual_setting, last_channel, **kwargs)
    if pretrained_name:
        from torch.hub import load_state_dict_from_url
        state_dict = load_state_dict_from_url(pretrained_models.get(pretrained_name), model_dir=model_dir,
                                              map_location="cpu")
        n_ckpt, n_model = state_dict["classifier.5.weight"].shape[0], model.classifier[5].out_features
        if n_ckpt != n_model:
            print(f"The number of classes in the loaded state dict (={n_ckpt}) and the current model (={n_model}) "
                  f"is not the same. Dropping final fully-connected layer and loading weights in non-strict mode!")
            del state_dict["classifier.5.weight"], state_dict["classifier.5.bias"]
            model.load_state_dict(state_dict, strict=False)
        else:
            model.load_state_dict(state_dict)
    return model


def dymn(pretrained_name=None, **kwargs):
    setting, last_channel = _dymn_conf(**kwargs)
    return _dymn(setting, last_channel, pretrained_name, **kwargs)


def get_model(num_classes=527, pretrained_name=None, width_mult=1.0, strides=(2, 2, 2, 2), context_ratio=4,
              max_context_size=128, min_context_size=32, dyrelu_k=2, no_dyrelu=False, dyconv_k=4, no_dyconv=False,
              T_max=30.0, T0_slope=1.0, T1_slope=0.02, T_min=1, pretrain_final_temp=1.0, no_ca=False,
              use_dy_blocks="all"):
    """Same signature as models/dymn/model.py:289-310."""
    if pretrained_name:
        T_max = pretrain_final_temp          # pre-trained on AudioSet -> final temperature of that stage
    m = dymn(num_classes=num_classes, pretrained_name=pretrained_name, block=DY_Block, width_mult=width_mult,
             strides=strides, context_ratio=context_ratio, max_context_size=max_context_size,
             min_context_size=min_context_size, dyrelu_k=dyrelu_k, dyconv_k=dyconv_k, no_dyrelu=no_dyrelu,
             no_dyconv=no_dyconv, no_ca=no_ca, temp_schedule=(T_max, T_min, T0_slope, T1_slope),
             use_dy_blocks=use_dy_blocks)
    print(m)
    return m
