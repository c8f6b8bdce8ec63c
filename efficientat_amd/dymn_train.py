"""Train-mode forward / backward of DyMN on the HIP kernels.

Unlike the static MN (one monolithic autograd Function, mn_train.py) the dynamic network is
assembled from per-op `torch.autograd.Function`s, each a thin pair of launch plans over
libeat_hip.so, and autograd routes the gradients through the context path:

    CtxPool, Linear (MFMA), BnAct (batch-stat BatchNorm + activation), StemConv, PwConv,
    DynPwConv / DynDwConv (kernel aggregation + per-sample conv; backward = per-sample weight
    gradients G_b, then dbank = att^T G, datt = G bank^T), DyReluCoordAtt.

Sequence-level glue on (B, L, H)-shaped tensors (BatchNorm of the context sequence, mean over L,
3-tap average pool, softmax over K, sigmoid) stays on torch ops - these tensors are <= 4 % of the
activation volume; every pass over a feature map and every GEMM runs in the library.
Reference semantics: models/dymn/dy_block.py:390-409 (DY_Block.forward), :235-254 (ContextGen),
:103-131 (DynamicConv), :172-188 (DyReLU-B), :195-201 (CoordAtt).
"""
import torch
import torch.nn.functional as F

from . import _lib, ops
from .dymn import _pool3
from .mn_train import _zeros

NONE, RELU, HSWISH = ops.ACT_NONE, ops.ACT_RELU, ops.ACT_HSWISH


def _s():
    return torch.cuda.current_stream().cuda_stream


def _t(x):
    return x.t().contiguous()


def _col_sum(m):
    """Column sums of a (R, C) matrix on our own kernel (torch's multi-block `sum(0)` returned stray non-finite entries
    when replayed inside a captured hipGraph at R >= ~1000 on ROCm 7.0 / torch 2.10)."""
    return ops.col_sum(m.contiguous())


# ------------------------------------------------------------------------------- Functions
class CtxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        return ops.ctx_pool(x)

    @staticmethod
    def backward(ctx, dseq):
        B, C, Fq, T = ctx.shape
        dx = torch.empty(ctx.shape, device=dseq.device, dtype=torch.float32)
        _lib.call("eat_ctx_pool_bwd", dseq.contiguous().data_ptr(), None, dx.data_ptr(), B, C, Fq, T, _s())
        return dx


class Linear(torch.autograd.Function):
    """y = x W^T + b on the MFMA linear kernel (activation applied by the caller through torch)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return ops.linear(x.contiguous(), w.contiguous(), b, NONE)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = ops.linear(dy, _t(w), None, NONE)
        M = dy.shape[0]
        if M >= 4096 and M % 4 == 0 and x.shape[1] % 4 == 0:
            # long reductions (M = B * (F + T) rows of the context sequence): dW = dy^T x is the 1x1 weight-gradient
            # GEMM with one "sample" of M positions - split-K over many blocks instead of 4 waves of the linear kernel
            dw = ops.pw_conv_wgrad(_t(dy).view(1, dy.shape[1], M, 1), _t(x).view(1, x.shape[1], M, 1), exact=False)
        else:
            dw = ops.linear(_t(dy), _t(x), None, NONE)
        db = _col_sum(dy) if ctx.has_bias else None
        return dx, dw, db


class BnAct(torch.autograd.Function):
    """y = act(BatchNorm_batchstats(z)); running buffers of `bn` are updated in forward."""

    @staticmethod
    def forward(ctx, z, gamma, beta, bn, act):
        z = z.contiguous()
        C = z.shape[1]
        st = ops.bn_train_state(z, bn)
        ctx.save_for_backward(z, *st)
        ctx.act = act
        ctx.frozen = getattr(st[2], "_eat_frozen", False)
        return ops.bn_act_fwd(z, st[0], st[1], act)

    @staticmethod
    def backward(ctx, dy):
        z, a, b, mean, invstd = ctx.saved_tensors
        dz, dgam, dbet = ops.bn_act_bwd(dy.contiguous(), z, a, b, mean, invstd, ctx.act, frozen=ctx.frozen)
        return dz, dgam, dbet, None, None


class StemConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        C = w.shape[0]
        return ops.stem_conv(x.contiguous(), w.reshape(C, 9), _zeros.get(C, x.device), NONE)

    @staticmethod
    def backward(ctx, dz):
        x, w = ctx.saved_tensors
        return None, ops.dw_conv_wgrad(dz.contiguous(), x, 3, 2).view_as(w)


class PwConv(torch.autograd.Function):
    """Static 1x1 conv (no bias): forward and data gradient are the same MFMA GEMM."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        Co = w.shape[0]
        return ops.pw_conv(x.contiguous(), ops.pw_prepack(w.flatten(1)), _zeros.get(Co, x.device), Co, NONE)

    @staticmethod
    def backward(ctx, dz):
        x, w = ctx.saved_tensors
        dz = dz.contiguous()
        Ci = x.shape[1]
        dx = ops.pw_conv(dz, ops.pw_prepack(_t(w.flatten(1))), _zeros.get(Ci, x.device), Ci, NONE)
        return dx, ops.pw_conv_wgrad(dz, x, exact=False).view_as(w)


class DwConv(torch.autograd.Function):
    """Static depthwise k x k conv (no bias) of the SE-less blocks of a `use_dy_blocks="replace_se"` network."""

    @staticmethod
    def forward(ctx, x, w, k, stride):
        x = x.contiguous()
        ctx.save_for_backward(x, w)
        ctx.k, ctx.stride = k, stride
        C = x.shape[1]
        return ops.dw_conv(x, w.reshape(C, k * k), _zeros.get(C, x.device), k, stride, NONE)

    @staticmethod
    def backward(ctx, dz):
        x, w = ctx.saved_tensors
        dz = dz.contiguous()
        k, stride = ctx.k, ctx.stride
        dx = ops.dw_conv_dgrad(dz, w.reshape(-1, k * k).contiguous(), x.shape, k, stride)
        return dx, ops.dw_conv_wgrad(dz, x, k, stride).view_as(w), None, None


def _bank_grad(G, att, bank):
    B, K, N = att.shape[0], bank.shape[0], bank.shape[1]
    dbank = torch.empty_like(bank)
    datt = torch.zeros_like(att)
    _lib.call("eat_dyn_bank_grad", G.data_ptr(), att.data_ptr(), bank.data_ptr(), dbank.data_ptr(), datt.data_ptr(),
              B, K, N, _s())
    return dbank, datt


class DynPwConv(torch.autograd.Function):
    """z_b = (sum_k att[b,k] W_k) x_b for a 1x1 DynamicConv; `weight` is the (1,1,K,N) parameter."""

    @staticmethod
    def forward(ctx, x, weight, att, Co):
        x, att = x.contiguous(), att.contiguous()
        K = weight.shape[2]
        Ci = x.shape[1]
        bank = weight.view(K, Co * Ci)
        ctx.save_for_backward(x, weight, att)
        ctx.Co = Co
        if ops.kcat_eligible(Co, Ci, x.shape[2] * x.shape[3]):       # late layers: no per-sample weights (ops.kcat_*)
            return ops.pw_conv_kcat(x, ops.kcat_pack(bank, Co, Ci), _zeros.get(Co, x.device), att, Co, NONE)
        wp = ops.dyn_pw_pack(bank, att, Co, Ci)
        return ops.pw_conv_dyn(x, wp, _zeros.get(Co, x.device), Co, NONE)

    @staticmethod
    def backward(ctx, dz):
        x, weight, att = ctx.saved_tensors
        dz = dz.contiguous()
        B, Ci, Fq, T = x.shape
        Co, K = ctx.Co, weight.shape[2]
        bank = weight.view(K, Co, Ci)
        # data gradient: per-sample W_b^T, packed from the transposed bank
        bank_t = bank.transpose(1, 2).contiguous().view(K, Ci * Co)
        if ops.kcat_eligible(Ci, Co, Fq * T):
            dx = ops.pw_conv_kcat(dz, ops.kcat_pack(bank_t, Ci, Co), _zeros.get(Ci, x.device), att, Ci, NONE)
        else:
            wpt = ops.dyn_pw_pack(bank_t, att, Ci, Co)
            dx = ops.pw_conv_dyn(dz, wpt, _zeros.get(Ci, x.device), Ci, NONE)
        # per-sample weight gradient G_b, then the gradients of the aggregation
        G = torch.zeros((B, Co * Ci), device=x.device, dtype=torch.float32)
        _lib.call("eat_pw_conv_dyn_wgrad", dz.data_ptr(), x.data_ptr(), G.data_ptr(), B, Co, Ci, Fq * T, _s())
        dbank, datt = _bank_grad(G, att, weight.view(K, Co * Ci))
        return dx, dbank.view_as(weight), datt, None


class DynDwConv(torch.autograd.Function):
    """Depthwise DynamicConv: per-(b,c) taps = sum_k att[b,k] w_k[c]."""

    @staticmethod
    def forward(ctx, x, weight, att, k, stride):
        x, att = x.contiguous(), att.contiguous()
        B, C, Fq, T = x.shape
        K = weight.shape[2]
        taps = ops.dyn_aggregate(weight.view(K, C * k * k), att)
        ctx.save_for_backward(x, weight, att, taps)
        ctx.k, ctx.stride = k, stride
        Fo, To = ops.conv_out(Fq, k, stride), ops.conv_out(T, k, stride)
        y = torch.empty((B, C, Fo, To), device=x.device, dtype=torch.float32)
        _lib.call("eat_dw_conv_dyn_fwd", x.data_ptr(), taps.data_ptr(), _zeros.get(C, x.device).data_ptr(), None, None,
                  None, y.data_ptr(), B, C, Fq, T, Fo, To, k, stride, _s())
        return y

    @staticmethod
    def backward(ctx, dz):
        x, weight, att, taps = ctx.saved_tensors
        dz = dz.contiguous()
        B, C, Fq, T = x.shape
        k, stride, K = ctx.k, ctx.stride, weight.shape[2]
        Fo, To = dz.shape[2], dz.shape[3]
        dx = torch.empty_like(x)
        _lib.call("eat_dw_conv_dyn_dgrad", dz.data_ptr(), taps.data_ptr(), None, dx.data_ptr(), B, C, Fq, T, Fo, To, k,
                  stride, _s())
        G = torch.zeros((B, C * k * k), device=x.device, dtype=torch.float32)
        _lib.call("eat_dw_conv_dyn_wgrad", dz.data_ptr(), x.data_ptr(), G.data_ptr(), B, C, Fq, T, Fo, To, k, stride, _s())
        dbank, datt = _bank_grad(G, att, weight.view(K, C * k * k))
        return dx, dbank.view_as(weight), datt, None, None


class DyReluCoordAtt(torch.autograd.Function):
    """out = max(a1 v + b1, a2 v + b2) * sigmoid(g_cf) * sigmoid(g_ct); coef (B,C,4), gates (B,L,C)."""

    @staticmethod
    def forward(ctx, v, coef, g_cf, g_ct):
        v, coef, g_cf, g_ct = v.contiguous(), coef.contiguous(), g_cf.contiguous(), g_ct.contiguous()
        B, C, Fo, To = v.shape
        ctx.save_for_backward(v, coef, g_cf, g_ct)
        out = torch.empty_like(v)
        _lib.call("eat_dyrelu_ca_fwd", v.data_ptr(), None, None, coef.data_ptr(), g_cf.data_ptr(), g_ct.data_ptr(),
                  out.data_ptr(), B, C, Fo, To, _s())
        return out

    @staticmethod
    def backward(ctx, dout):
        v, coef, g_cf, g_ct = ctx.saved_tensors
        B, C, Fo, To = v.shape
        dv, dcoef = torch.empty_like(v), torch.empty_like(coef)
        dgf, dgt = torch.empty_like(g_cf), torch.empty_like(g_ct)
        _lib.call("eat_dyrelu_ca_bwd", dout.contiguous().data_ptr(), v.data_ptr(), None, None, coef.data_ptr(),
                  g_cf.data_ptr(), g_ct.data_ptr(), dv.data_ptr(), dcoef.data_ptr(), dgf.data_ptr(), dgt.data_ptr(), B, C,
                  Fo, To, _s())
        return dv, dcoef, dgf, dgt


# ----------------------------------------------------------------------------------- plan
def _attention(conv, h_c):
    logits = Linear.apply(h_c, conv.residuals[0].weight, conv.residuals[0].bias)
    return F.softmax(logits / conv.temperature, dim=-1)


def _block_train(blk, x):
    cnf = blk.cnf
    B, cin, Fq, T = x.shape
    H, cexp, cout, k, stride = blk.context_dim, cnf.expanded_channels, cnf.out_channels, cnf.kernel, cnf.stride
    act = HSWISH if cnf.use_hs else RELU
    inp = x
    cg = blk.context_gen
    L = Fq + T
    seq = CtxPool.apply(x)                                                            # (B, L, cin)
    gj = Linear.apply(seq.view(B * L, cin), cg.joint_conv.weight.flatten(1), None)     # (B*L, H)
    # BatchNorm of the context sequence over (B, L) per channel (joint_norm), batch statistics
    bn = cg.joint_norm
    if bn.training:
        gj = F.batch_norm(gj, bn.running_mean, bn.running_var, bn.weight, bn.bias, True,
                          bn.momentum if bn.momentum is not None else 1.0 / (int(bn.num_batches_tracked) + 1), bn.eps)
        bn.num_batches_tracked += 1
    else:                                    # frozen BatchNorm inside a train-mode pass: running statistics
        gj = F.batch_norm(gj, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
    g = F.hardswish(gj).view(B, L, H)
    h_c = g.mean(dim=1)
    h_cf, h_ct = g[:, :Fq], g[:, Fq:]
    if stride > 1:
        h_cf, h_ct = _pool3(h_cf, stride), _pool3(h_ct, stride)
    Fo, To = h_cf.shape[1], h_ct.shape[1]
    g_cf = Linear.apply(h_cf.reshape(B * Fo, H), cg.conv_f.weight.flatten(1), cg.conv_f.bias).view(B, Fo, cexp)
    g_ct = Linear.apply(h_ct.reshape(B * To, H), cg.conv_t.weight.flatten(1), cg.conv_t.bias).view(B, To, cexp)
    # The ablated blocks (dy_block.py:269-271) keep the whole context generator above - the reference evaluates it
    # (and updates joint_norm's running statistics) whatever consumes it - and swap the consumers: static convs on the MN
    # Functions, the plain activation in BnAct, DyReLU-B / CoordAtt neutralised by constant operands (a1 = a2 = 1:
    # max(v, v) = v;  gates = +40: sigmoid = 1 and its derivative 0 in fp32).
    no_dyconv, no_dyrelu, no_ca = blk.no_dyconv, blk.no_dyrelu, blk.no_ca
    if blk.has_expand:
        z = PwConv.apply(x, blk.exp_conv.module.weight) if no_dyconv else \
            DynPwConv.apply(x, blk.exp_conv.weight, _attention(blk.exp_conv, h_c), cexp)
        x = BnAct.apply(z, blk.exp_norm.weight, blk.exp_norm.bias, blk.exp_norm, act)
    z = DwConv.apply(x, blk.depth_conv.module.weight, k, stride) if no_dyconv else \
        DynDwConv.apply(x, blk.depth_conv.weight, _attention(blk.depth_conv, h_c), k, stride)
    v = BnAct.apply(z, blk.depth_norm.weight, blk.depth_norm.bias, blk.depth_norm, act if no_dyrelu else NONE)
    if no_dyrelu and no_ca:
        x = v
    else:
        if no_dyrelu:
            coef = torch.tensor([1.0, 1.0, 0.0, 0.0], device=x.device).expand(B, cexp, 4)
        else:
            da = blk.depth_act
            theta = 2.0 * torch.sigmoid(Linear.apply(h_c, da.coef_net[0].weight, da.coef_net[0].bias)) - 1.0
            coef = theta.view(B, cexp, 4) * da.lambdas + da.init_v
        if no_ca:
            g_cf, g_ct = torch.full_like(g_cf, 40.0), torch.full_like(g_ct, 40.0)
        x = DyReluCoordAtt.apply(v, coef, g_cf, g_ct)
    z = PwConv.apply(x, blk.proj_conv.module.weight) if no_dyconv else \
        DynPwConv.apply(x, blk.proj_conv.weight, _attention(blk.proj_conv, h_c), cout)
    x = BnAct.apply(z, blk.proj_norm.weight, blk.proj_norm.bias, blk.proj_norm, NONE)
    return x + inp if blk.use_res_connect else x


def _static_block_train(blk, x):
    """SE-less InvertedResidual in train mode (models/mn/block_types.py:138-181 with se_cnf None)."""
    cnf = blk.cnf
    act = HSWISH if cnf.use_hs else RELU
    if cnf.dilation > 1:
        raise NotImplementedError("training a dilated block is not on the HIP path")
    inp = x
    if blk.i_expand is not None:
        conv, bn = blk.block[blk.i_expand][0], blk.block[blk.i_expand][1]
        x = BnAct.apply(PwConv.apply(x, conv.weight), bn.weight, bn.bias, bn, act)
    conv, bn = blk.block[blk.i_dw][0], blk.block[blk.i_dw][1]
    x = BnAct.apply(DwConv.apply(x, conv.weight, cnf.kernel, cnf.stride), bn.weight, bn.bias, bn, act)
    conv, bn = blk.block[blk.i_proj][0], blk.block[blk.i_proj][1]
    x = BnAct.apply(PwConv.apply(x, conv.weight), bn.weight, bn.bias, bn, NONE)
    return x + inp if blk.use_res_connect else x


def forward_train(model, x, return_fmaps=False):
    """Train-mode `(logits, embedding)` - or `(logits, fmaps)`, models/dymn/model.py:157-195 - of DyMN with autograd
    support (models/dymn/model.py:185-200).  The fully-convolutional head (:119-130) runs as torch ops on the last 4 x 32
    map (its class count, 527, is not a multiple of 4, which the library's data-gradient GEMM needs)."""
    x = x.contiguous().float()
    z = StemConv.apply(x, model.in_c[0].weight)
    x = BnAct.apply(z, model.in_c[1].weight, model.in_c[1].bias, model.in_c[1], HSWISH)
    fmaps = [x]
    for blk in model.layers:
        x = _block_train(blk, x) if hasattr(blk, "context_gen") else _static_block_train(blk, x)
        fmaps.append(x)
    z = PwConv.apply(x, model.out_c[0].weight)
    x = BnAct.apply(z, model.out_c[1].weight, model.out_c[1].bias, model.out_c[1], HSWISH)
    fmaps.append(x)
    feat = x.mean(dim=(2, 3))
    if model.head_type == "fully_convolutional":
        logits = model.classifier(x).reshape(x.shape[0], -1)
        return (logits, fmaps) if return_fmaps else (logits, feat)
    fc1, fc2, drop = model.classifier[2], model.classifier[5], model.classifier[4]
    h = F.hardswish(Linear.apply(feat, fc1.weight, fc1.bias))
    override = getattr(model, "_drop_mask_override", None)
    if not drop.training:
        pass                                  # nn.Dropout switched to eval() inside model.train()
    elif override is not None:
        h = h * (override.to(h.device).float() / (1.0 - drop.p))
    elif drop.p > 0:
        h = F.dropout(h, drop.p, True)
    logits = Linear.apply(h, fc2.weight, fc2.bias)
    return (logits, fmaps) if return_fmaps else (logits, feat)
