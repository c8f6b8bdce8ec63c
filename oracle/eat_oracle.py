"""Functional fp32 CPU restatement of the reference hot path (TEST INFRASTRUCTURE ONLY).

Everything here works on a plain ``state_dict`` in the reference key layout and on
CPU tensors; there are no nn.Modules and nothing from the product package.

Reference anchors (paths relative to /root/reference):
  * mel front-end ......... models/preprocess.py:40-67
  * kaldi mel banks ....... torchaudio 0.13 compliance.kaldi.get_mel_banks (third party,
                            pinned in requirements.txt:7; restated from the published algorithm)
  * MobileNetV3 config .... models/mn/model.py:237-271, models/mn/block_types.py:86-117
  * MN forward ............ models/mn/model.py:212-231, models/mn/block_types.py:72-83,177-181
  * DyMN config/forward ... models/dymn/model.py:157-183,209-254, models/dymn/dy_block.py:103-131,
                            172-188,195-201,235-254,390-409
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-3  # models/mn/model.py:114-115, models/dymn/dy_block.py:284


# --------------------------------------------------------------------------- mel
def kaldi_mel_banks(n_mels, n_fft, sr, fmin, fmax):
    """(n_mels, n_fft//2+1) fp32 basis, last column zero (preprocess.py:52-55)."""
    nfb = n_fft / 2
    width = sr / n_fft
    lo = 1127.0 * math.log(1.0 + fmin / 700.0)
    hi = 1127.0 * math.log(1.0 + fmax / 700.0)
    d = (hi - lo) / (n_mels + 1)
    b = torch.arange(n_mels).unsqueeze(1)
    left, center, right = lo + b * d, lo + (b + 1.0) * d, lo + (b + 2.0) * d
    mel = (1127.0 * (1.0 + (width * torch.arange(nfb)) / 700.0).log()).unsqueeze(0)
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    bins = torch.max(torch.zeros(1), torch.min(up, down))
    return F.pad(bins, (0, 1))


def mel_forward(x, n_mels=128, sr=32000, win_length=800, hopsize=320, n_fft=1024,
                fmin=0.0, fmax=15000.0, freq_mask=None, time_mask=None, dtype=torch.float32):
    """x (B, L) fp32 -> (B, n_mels, T).  Explicit framing restatement of
    preprocess.py:41-65 (conv1d pre-emphasis, centred reflect-padded STFT with a
    non-periodic hann window zero-padded to n_fft, power, mel matmul, log, normalise).
    ``freq_mask``/``time_mask`` = (start, end) index ranges zeroed after the log
    (train-mode masking, preprocess.py:61-63); the caller draws them."""
    x = x.to(dtype)
    pre = x[:, 1:] - 0.97 * x[:, :-1]                                  # :41
    pad = n_fft // 2
    p = F.pad(pre.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)  # stft(center=True)
    frames = p.unfold(1, n_fft, hopsize)                                # (B, T, n_fft)
    win = torch.hann_window(win_length, periodic=False).to(dtype)   # the fp32 window, as the reference holds it
    lpad = (n_fft - win_length) // 2
    win = F.pad(win, (lpad, n_fft - win_length - lpad))
    spec = torch.fft.rfft(frames * win, dim=-1)                         # (B, T, n_fft/2+1)
    power = (spec.real ** 2 + spec.imag ** 2).transpose(1, 2)           # :44
    basis = kaldi_mel_banks(n_mels, n_fft, sr, fmin, fmax).to(dtype)   # always the fp32-built basis
    mel = torch.matmul(basis, power)                                    # :57
    mel = (mel + 0.00001).log()                                         # :59
    if freq_mask is not None:
        mel[:, freq_mask[0]:freq_mask[1], :] = 0.0
    if time_mask is not None:
        mel[:, :, time_mask[0]:time_mask[1]] = 0.0
    return (mel + 4.5) / 5.0                                            # :65


# ------------------------------------------------------------------- config math
def make_divisible(v, divisor=8):
    """models/mn/utils.py:8-21."""
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


# (in, kernel, expanded, out, use_se, use_hs, stride index or None)   mn/model.py:252-268
_ROWS = [
    (16, 3, 16, 16, False, False, None), (16, 3, 64, 24, False, False, 0), (24, 3, 72, 24, False, False, None),
    (24, 5, 72, 40, True, False, 1), (40, 5, 120, 40, True, False, None), (40, 5, 120, 40, True, False, None),
    (40, 3, 240, 80, False, True, 2), (80, 3, 200, 80, False, True, None), (80, 3, 184, 80, False, True, None),
    (80, 3, 184, 80, False, True, None), (80, 3, 480, 112, True, True, None), (112, 3, 672, 112, True, True, None),
    (112, 5, 672, 160, True, True, 3), (160, 5, 960, 160, True, True, None), (160, 5, 960, 160, True, True, None),
]


def block_table(width_mult=1.0, strides=(2, 2, 2, 2), reduced_tail=False, dilated=False):
    """List of dicts (cin, k, cexp, cout, se, hs, stride, dil) per inverted-residual block; `reduced_tail` halves the
    channels and `dilated` sets dilation 2 in the last three blocks (mn/model.py:244-269)."""
    adj = lambda c: make_divisible(c * width_mult, 8)
    div, dil = (2 if reduced_tail else 1), (2 if dilated else 1)
    out = []
    for i, (cin, k, cexp, cout, se, hs, si) in enumerate(_ROWS):
        tail = i >= 12
        if tail:            # rows 13-15: (112, 5, 672, 160/div), (160/div, 5, 960/div, 160/div) x 2
            cin = cin if i == 12 else cin // div
            cexp = cexp if i == 12 else cexp // div
            cout = cout // div
        out.append(dict(cin=adj(cin), k=k, cexp=adj(cexp), cout=adj(cout), se=se, hs=hs,
                        stride=1 if si is None else strides[si], dil=dil if tail else 1))
    return out, adj(1280 // div)


def _act(x, hs):
    return F.hardswish(x) if hs else F.relu(x)


def _bn(sd, prefix, x, train, stats=None):
    """BatchNorm2d(eps=1e-3, momentum=0.01).  In train mode uses batch statistics and,
    if ``stats`` is a dict, records the updated running buffers there."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if not train:
        return F.batch_norm(x, rm, rv, w, b, False, 0.0, BN_EPS)
    rm2, rv2 = rm.clone(), rv.clone()
    mom = 0.01 if stats is None else stats.get("__momentum__", 0.01)
    y = F.batch_norm(x, rm2, rv2, w, b, True, mom, BN_EPS)
    if stats is not None:
        stats[prefix + ".running_mean"], stats[prefix + ".running_var"] = rm2, rv2
    return y


class _PwBf16(torch.autograd.Function):
    """1x1 conv on bf16-rounded operands with fp32 accumulation - the arithmetic of `train_precision="bf16"`
    (BASELINE configs[2]: bf16 MFMA 1x1 GEMMs, fp32 activations in memory): forward rounds x and W to bf16, the data
    gradient rounds dz and W, the weight gradient rounds dz and x (what autocast does to all three conv GEMMs in the
    reference's bf16 training).  Lets the tests anchor the bf16 path on an ORACLE evaluation of the same arithmetic."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return F.conv2d(x.bfloat16().float(), w.bfloat16().float())

    @staticmethod
    def backward(ctx, dz):
        x, w = ctx.saved_tensors
        dx = F.conv_transpose2d(dz.bfloat16().float(), w.bfloat16().float())
        dw = torch.einsum("bohw,bihw->oi", dz.bfloat16().float(), x.bfloat16().float()).reshape(w.shape)
        return dx, dw


PW_BF16 = False      # set through `emulate_bf16_pointwise()` only
ST_BF16 = False      # ... with storage=True: the wide tensors of the blocks that have an expand conv are STORED in bf16


class _StoreBf16(torch.autograd.Function):
    """A tensor that lives in bf16 in memory: the forward value is rounded once (what a store + load leaves)."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, dy):
        return dy


class _GradBf16(torch.autograd.Function):
    """A point of the graph whose incoming GRADIENT is stored in bf16 (identity in the forward)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        return dy.bfloat16().float()


class emulate_bf16_pointwise:
    """with O.emulate_bf16_pointwise(): every 1x1 convolution of mn_forward runs as `_PwBf16`.
    storage=True additionally emulates `act_storage="bf16"` (efficientat_amd/mn_train.py; the reference's 16-bit mixed
    precision, ex_pl_audioset.py:287-293): in every inverted-residual block the expand output z_e, the depthwise output z_d,
    its activated form y_d and the project conv's output z_p are rounded to bf16 where they are stored, and so are the two wide gradients of the
    backward - the one arriving at the project conv's input and g = dL/d(BN output of the expand conv)."""

    def __init__(self, storage=False):
        # storage: False, True (every block), or - DyMN - the collection of block indices that run on bf16 storage
        self.storage = storage

    def __enter__(self):
        global PW_BF16, ST_BF16
        self.old, PW_BF16 = (PW_BF16, ST_BF16), True
        ST_BF16 = self.storage

    def __exit__(self, *exc):
        global PW_BF16, ST_BF16
        PW_BF16, ST_BF16 = self.old


def _cna(sd, prefix, x, train, stats, k, stride, groups, act, dil=1, store=None):
    """ConvNormActivation (torchvision 0.14): conv(bias=False,pad=(k-1)//2*dilation) + BN + act.
    store (bf16-storage emulation only): "expand" = conv output stored in bf16, gradient w.r.t. the BN output stored in
    bf16; "depthwise" = conv output and activated output stored in bf16; "project" = conv output stored in bf16."""
    if PW_BF16 and k == 1 and groups == 1:
        x = _PwBf16.apply(x, sd[prefix + ".0.weight"])
    else:
        x = F.conv2d(x, sd[prefix + ".0.weight"], None, stride, (k - 1) // 2 * dil, dil, groups)
    if store is not None:
        x = _StoreBf16.apply(x)
    x = _bn(sd, prefix + ".1", x, train, stats)
    if store == "expand":
        x = _GradBf16.apply(x)
    if act == "hs":
        x = F.hardswish(x)
    elif act == "re":
        x = F.relu(x)
    if store == "depthwise":
        x = _StoreBf16.apply(x)
    return x


def _se(sd, prefix, x):
    """SqueezeExcitation over channels, Linear/ReLU/Linear/Sigmoid (block_types.py:72-83)."""
    z = x.mean(dim=(2, 3))
    z = F.relu(F.linear(z, sd[prefix + ".fc1.weight"], sd[prefix + ".fc1.bias"]))
    s = torch.sigmoid(F.linear(z, sd[prefix + ".fc2.weight"], sd[prefix + ".fc2.bias"]))
    return x * s[:, :, None, None]


def _concurrent_se(sd, prefix, x, se_dims, se_agg):
    """ConcurrentSEBlock (block_types.py:10-42) with SqueezeExcitation over dim d in {1: c, 2: f, 3: t} (:66-83): mean over
    the other two dims -> fc1 / ReLU / fc2 / Sigmoid -> gate broadcast along d; gated copies combined by se_agg."""
    outs = []
    for i, d in enumerate(se_dims):
        other = [k for k in (1, 2, 3) if k != d]
        m = x.mean(dim=other, keepdim=True)
        shape = m.shape
        q = f"{prefix}.conc_se_layers.{i}"
        z = F.relu(F.linear(m.squeeze(other[1]).squeeze(other[0]), sd[q + ".fc1.weight"], sd[q + ".fc1.bias"]))
        g = torch.sigmoid(F.linear(z, sd[q + ".fc2.weight"], sd[q + ".fc2.bias"])).view(shape)
        outs.append(g * x)
    st = torch.stack(outs, dim=0)
    return {"max": lambda t: t.max(dim=0)[0], "avg": lambda t: t.mean(dim=0), "add": lambda t: t.sum(dim=0),
            "min": lambda t: t.min(dim=0)[0]}[se_agg](st)


def _inverted_residual(sd, prefix, x, c, train, stats, use_se=True, se_dims=(1,), se_agg="max"):
    """block_types.py:120-181: [expand] -> depthwise -> [SE] -> project (+ residual)."""
    inp, j = x, 0
    a = "hs" if c["hs"] else "re"
    dil = c.get("dil", 1)
    st16 = ST_BF16 and train       # (a block without expand conv: z_d, y_d and the project conv's data gradient only)
    if c["cexp"] != c["cin"]:
        x = _cna(sd, f"{prefix}.block.{j}", x, train, stats, 1, 1, 1, a, store="expand" if st16 else None)
        j += 1
    x = _cna(sd, f"{prefix}.block.{j}", x, train, stats, c["k"], 1 if dil > 1 else c["stride"], c["cexp"], a, dil,
             store="depthwise" if st16 else None)   # :150
    j += 1
    if c["se"] and use_se and se_dims is not None and tuple(se_dims) != (1,):
        x = _concurrent_se(sd, f"{prefix}.block.{j}", x, se_dims, se_agg)
        j += 1
    elif c["se"] and use_se and se_dims is not None:
        x = _se(sd, f"{prefix}.block.{j}.conc_se_layers.0", x)
        j += 1
    if st16:
        x = _GradBf16.apply(x)                              # the project conv's data gradient is stored in bf16
    x = _cna(sd, f"{prefix}.block.{j}", x, train, stats, 1, 1, 1, None, store="project" if st16 else None)
    if c["stride"] == 1 and c["cin"] == c["cout"]:
        x = x + inp
    return x


def _mlp_head(sd, x, train, drop_mask):
    """mn/model.py:186-194 (Dropout replayed from an explicit keep-mask in train mode)."""
    pooled = x.mean(dim=(2, 3))
    h = F.hardswish(F.linear(pooled, sd["classifier.2.weight"], sd["classifier.2.bias"]))
    if train and drop_mask is not None:
        h = h * drop_mask / 0.8
    return F.linear(h, sd["classifier.5.weight"], sd["classifier.5.bias"]), pooled


def _fc_head(sd, x, train, stats):
    """head_type='fully_convolutional' (mn/model.py:173-185): Conv2d 1x1 (no bias) -> BatchNorm2d (default eps 1e-5) ->
    AdaptiveAvgPool2d((1,1)); features = pooled input (mn/model.py:220)."""
    z = F.conv2d(x, sd["classifier.0.weight"])
    w, b, rm, rv = (sd["classifier.1." + k] for k in ("weight", "bias", "running_mean", "running_var"))
    z = F.batch_norm(z, rm.clone(), rv.clone(), w, b, train, 0.1, 1e-5)
    return z.mean(dim=(2, 3)), x.mean(dim=(2, 3))


def _attention_head(sd, x, num_heads=4, eps=1e-7):
    """head_type='multihead_attention_pooling' (mn/attention_pooling.py:37-56)."""
    xm = x.mean(dim=2).transpose(1, 2)                                        # (B, T, C)
    b, n, _ = xm.shape
    p = F.linear(xm, sd["classifier.subspace_proj.weight"], sd["classifier.subspace_proj.bias"])
    out_dim = p.shape[-1] // (2 * num_heads)
    p = p.reshape(b, n, 2, num_heads, out_dim).permute(2, 0, 3, 1, 4)
    att, val = torch.clamp(torch.sigmoid(p[0]), eps, 1.0 - eps), p[1]
    att = att / att.sum(dim=2, keepdim=True)
    out = (att * val).sum(dim=2) * sd["classifier.head_weight"]
    return out.sum(dim=1), x.mean(dim=(2, 3))


def mn_forward(sd, x, width_mult=1.0, strides=(2, 2, 2, 2), train=False, stats=None,
               drop_mask=None, return_fmaps=False, head_type="mlp", se_dims=(1,), se_agg="max",
               reduced_tail=False, dilated=False, num_heads=4):
    """x (B,1,F,T) -> (logits, pooled features) or (logits, fmaps)   (mn/model.py:212-231)."""
    blocks, _ = block_table(width_mult, strides, reduced_tail, dilated)
    fmaps = []
    x = _cna(sd, "features.0", x, train, stats, 3, 2, 1, "hs")
    fmaps.append(x)
    for i, c in enumerate(blocks):
        x = _inverted_residual(sd, f"features.{i + 1}", x, c, train, stats, se_dims=se_dims, se_agg=se_agg)
        fmaps.append(x)
    x = _cna(sd, "features.16", x, train, stats, 1, 1, 1, "hs")
    fmaps.append(x)
    if head_type == "mlp":
        logits, pooled = _mlp_head(sd, x, train, drop_mask)
    elif head_type == "fully_convolutional":
        logits, pooled = _fc_head(sd, x, train, stats)
    else:
        logits, pooled = _attention_head(sd, x, num_heads)
    return (logits, fmaps) if return_fmaps else (logits, pooled)


# --------------------------------------------------------------------------- DyMN
def context_dim(cexp, width_mult, ratio=4, lo=32, hi=128):
    """dy_block.py:278-281."""
    v = make_divisible(cexp // ratio, 8)
    return int(min(max(v, make_divisible(lo * width_mult, 8)), make_divisible(hi * width_mult, 8)))


class _DynPwBf16(torch.autograd.Function):
    """Per-sample 1x1 conv y_b = W_b x_b on bf16-rounded operands with fp32 accumulation: the arithmetic of the dynamic 1x1
    convs under `train_precision="bf16"` with the bf16-storage plan (efficientat_amd/dymn_train.py `_dyn_pw_b16`): the
    aggregated per-sample weights are rounded once (the plain bf16 pack), the data gradient rounds dz, the per-sample weight
    gradient rounds dz and x - the `_PwBf16` rule for weights that differ per sample."""

    @staticmethod
    def forward(ctx, x, w):                       # x (B, Ci, F, T), w (B, Co, Ci)
        ctx.save_for_backward(x, w)
        return torch.einsum("boi,bihw->bohw", w.bfloat16().float(), x.bfloat16().float())

    @staticmethod
    def backward(ctx, dz):
        x, w = ctx.saved_tensors
        d16 = dz.bfloat16().float()
        return (torch.einsum("boi,bohw->bihw", w.bfloat16().float(), d16),
                torch.einsum("bohw,bihw->boi", d16, x.bfloat16().float()))


def _st16(i):
    """bf16-storage emulation of dynamic block i: ST_BF16 is True (every block) or the collection of the block indices that
    run on bf16 storage (the HIP plan keeps fp32 storage for geometries its kernels do not cover)."""
    return ST_BF16 is True or (ST_BF16 is not False and ST_BF16 is not None and i in ST_BF16)


def _dyconv(sd, prefix, x, h_c, cin, cout, k, stride, groups, temperature, dilation=1, pw16=False):
    """DynamicConv (dy_block.py:103-131): per-sample kernel = softmax-weighted sum of K=4; `dilation` with padding
    (k - 1) // 2 * dilation as DY_Block builds the depthwise DynamicConv of a dilated block (dy_block.py:322-348).
    pw16 (emulation only): the 1x1 conv on bf16-rounded operands (`_DynPwBf16`)."""
    b = x.shape[0]
    a = F.softmax(F.linear(h_c, sd[prefix + ".residuals.0.weight"], sd[prefix + ".residuals.0.bias"])
                  / temperature, dim=-1)                                    # (B,K)
    bank = sd[prefix + ".weight"][0, 0]                                      # (K, N)
    if pw16 and k == 1 and groups == 1:
        return _DynPwBf16.apply(x, (a @ bank).reshape(b, cout, cin))
    w = (a @ bank).reshape(b * cout, cin // groups, k, k)
    y = F.conv2d(x.reshape(1, b * cin, *x.shape[2:]), w, None, stride, (k - 1) // 2 * dilation, dilation, groups * b)
    return y.reshape(b, cout, *y.shape[2:])


def _dy_block(sd, prefix, x, c, H, train, stats, temperature, no_dyrelu=False, no_dyconv=False, no_ca=False):
    """DY_Block.forward (dy_block.py:390-409); the ablation flags follow dy_block.py:291-375: `no_dyconv` replaces the
    three DynamicConvs by plain bias-free convolutions (state-dict keys `<conv>.module.weight`, DynamicWrapper
    dy_block.py:204-211), `no_dyrelu` puts the block's plain activation in the place of DyReLU-B, `no_ca` drops the
    coordinate attention.  The context generator runs in every case (dy_block.py:394)."""
    inp = x
    B, C, Fq, T = x.shape
    # bf16 emulation (tests only, `emulate_bf16_pointwise`): pw16 = the 1x1 convs of the block on bf16-rounded operands (the
    # context generator's static convs as `_PwBf16`, the dynamic ones as `_DynPwBf16`); st16 = this block's wide tensors are
    # STORED in bf16 (efficientat_amd/dymn_train.py DyBlockMain with fused == 2): forward z_e, z_d and the DyReLU * CoordAtt
    # output, backward dz_e, g_e = dL/d(exp_norm output), dL/d(depth_norm output) and dL/d(project conv input)
    pw16 = PW_BF16 and train and not (no_dyconv or no_dyrelu or no_ca)     # (ablated blocks run the per-layer fp32-storage plan)
    try:
        st16 = pw16 and _st16(int(prefix.rsplit(".", 1)[1]))
    except ValueError:
        st16 = False
    if PW_BF16 and train:
        pwc = lambda t, w, b=None: _PwBf16.apply(t, w) if b is None else _PwBf16.apply(t, w) + b.view(1, -1, 1, 1)
    else:                                       # (the reference's own call, bias inside the conv: bit-for-bit the pinned oracle)
        pwc = lambda t, w, b=None: F.conv2d(t, w, b)
    rnd = (lambda t: _StoreBf16.apply(t)) if st16 else (lambda t: t)
    grnd = (lambda t: _GradBf16.apply(t)) if st16 else (lambda t: t)
    # ContextGen (dy_block.py:235-254)
    cf, ct = x.mean(dim=3, keepdim=True), x.mean(dim=2, keepdim=True).permute(0, 1, 3, 2)
    g = pwc(torch.cat([cf, ct], dim=2), sd[prefix + ".context_gen.joint_conv.weight"])
    g = F.hardswish(_bn(sd, prefix + ".context_gen.joint_norm", g, train, stats))
    h_cf, h_ct = g[:, :, :Fq], g[:, :, Fq:].permute(0, 1, 3, 2)
    h_c = g.mean(dim=2).reshape(B, H)
    # a dilated block runs its depthwise conv AND its context pooling at stride 1 (dy_block.py:322,385-386); the residual
    # test below still reads the configured stride (dy_block.py:278 `use_res_connect`)
    dil = c.get("dil", 1)
    dw_stride = 1 if dil > 1 else c["stride"]
    if dw_stride > 1:
        h_cf = F.avg_pool2d(h_cf, (3, 1), (dw_stride, 1), (1, 0))
        h_ct = F.avg_pool2d(h_ct, (1, 3), (1, dw_stride), (0, 1))
    g_cf = pwc(h_cf, sd[prefix + ".context_gen.conv_f.weight"], sd[prefix + ".context_gen.conv_f.bias"])
    g_ct = pwc(h_ct, sd[prefix + ".context_gen.conv_t.weight"], sd[prefix + ".context_gen.conv_t.bias"])

    def conv(name, x, cin, cout, k, stride, groups, dilation=1):
        if no_dyconv:
            return F.conv2d(x, sd[f"{prefix}.{name}.module.weight"], None, stride, (k - 1) // 2 * dilation, dilation, groups)
        return _dyconv(sd, f"{prefix}.{name}", x, h_c, cin, cout, k, stride, groups, temperature, dilation, pw16=st16)

    # expand
    if c["cexp"] != c["cin"]:
        x = grnd(rnd(conv("exp_conv", x, c["cin"], c["cexp"], 1, 1, 1)))
        x = _act(grnd(_bn(sd, prefix + ".exp_norm", x, train, stats)), c["hs"])
    # depthwise + DyReLU-B (dy_block.py:172-188) + CoordAtt (195-201)
    x = rnd(conv("depth_conv", x, c["cexp"], c["cexp"], c["k"], dw_stride, c["cexp"], dil))
    x = grnd(_bn(sd, prefix + ".depth_norm", x, train, stats))
    if no_dyrelu:
        x = _act(x, c["hs"])
    else:
        theta = 2 * torch.sigmoid(F.linear(h_c, sd[prefix + ".depth_act.coef_net.0.weight"],
                                           sd[prefix + ".depth_act.coef_net.0.bias"])) - 1
        co = theta.view(B, c["cexp"], 1, 1, 4) * sd[prefix + ".depth_act.lambdas"] + sd[prefix + ".depth_act.init_v"]
        x = torch.maximum(x * co[..., 0] + co[..., 2], x * co[..., 1] + co[..., 3])
    if not no_ca:
        x = x * torch.sigmoid(g_cf) * torch.sigmoid(g_ct)
    x = grnd(rnd(x))
    # project
    x = conv("proj_conv", x, c["cexp"], c["cout"], 1, 1, 1)
    x = _bn(sd, prefix + ".proj_norm", x, train, stats)
    if c["stride"] == 1 and c["cin"] == c["cout"]:
        x = x + inp
    return x


REPLACE_SE_DY = (False, False, False, True, True, True, False, False, False, False, True, True, True, True, True)


def dymn_forward(sd, x, width_mult=1.0, strides=(2, 2, 2, 2), temperature=1.0, train=False,
                 stats=None, drop_mask=None, return_fmaps=False, use_dy_blocks="all", no_dyrelu=False, no_dyconv=False,
                 no_ca=False, head_type="mlp", dilated=False, reduced_tail=False):
    """DyMN (dymn/model.py:157-200); use_dy_blocks "all" or "replace_se" (dymn/model.py:225-231: dynamic blocks only
    where MobileNetV3 has SE, plain SE-less inverted residuals elsewhere, dymn/model.py:102-103); `dilated` / `reduced_tail`:
    the last three blocks with dilation 2 / half the channels (_dymn_conf, dymn/model.py:212-250)."""
    blocks, _ = block_table(width_mult, strides, reduced_tail, dilated)
    dy = (True,) * 15 if use_dy_blocks == "all" else REPLACE_SE_DY
    fmaps = []
    x = _cna(sd, "in_c", x, train, stats, 3, 2, 1, "hs")
    fmaps.append(x)
    for i, c in enumerate(blocks):
        H = context_dim(c["cexp"], width_mult)
        if dy[i]:
            x = _dy_block(sd, f"layers.{i}", x, c, H, train, stats, temperature, no_dyrelu, no_dyconv, no_ca)
        else:
            x = _inverted_residual(sd, f"layers.{i}", x, c, train, stats, use_se=False)
        fmaps.append(x)
    x = _cna(sd, "out_c", x, train, stats, 1, 1, 1, "hs")
    fmaps.append(x)
    if head_type == "fully_convolutional":                 # dymn/model.py:119-130
        logits, pooled = _fc_head(sd, x, train, stats)
    else:
        logits, pooled = _mlp_head(sd, x, train, drop_mask)
    return (logits, fmaps) if return_fmaps else (logits, pooled)


def dyconv_temperature(epoch, T_max=30.0, T_min=1.0, T0_slope=1.0, T1_slope=0.02):
    """dy_block.py:133-139."""
    return max(T_max - T0_slope * epoch, 1 + T1_slope * (T_max - 1) / T0_slope - T1_slope * epoch, T_min)
