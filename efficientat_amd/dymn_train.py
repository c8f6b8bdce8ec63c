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
import os as _os

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


class _in_precision:
    """Re-enter, in a backward, the 1x1-conv arithmetic (`ops.precision`) the forward of the same Function ran under:
    autograd calls backward() outside `forward_train`'s context manager."""

    def __init__(self, ctx):
        self.p = ops.precision(ctx.prec)

    def __enter__(self):
        self.p.__enter__()

    def __exit__(self, *exc):
        self.p.__exit__(*exc)


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


# Hand-over of a block's input gradient from the main path's backward to the context path's (below).  RESTRICTION: it
# needs the WHOLE graph to run backward (`loss.backward()` / `torch.autograd.grad` w.r.t. the parameters or the input): a
# partial backward that stops between the two Functions would drop the main path's dx.  The hand-over dicts of a forward
# live on the model (`_eat_handovers`, one list per model - no process-global state); the NEXT forward of that model finds
# an uncollected dx and fails loudly, naming the cause.


_handovers = []           # the current forward's list (= model._eat_handovers of the model being run)


class CtxPoolCm(torch.autograd.Function):
    """ContextGen's pools with the sequence channel-major, batch folded into the positions: -> (1, C, B*(F+T), 1)."""

    @staticmethod
    def forward(ctx, x, hand_over=None):
        ctx.shape, ctx.hand_over = x.shape, hand_over
        return ops.ctx_pool_cm(x.contiguous())

    @staticmethod
    def backward(ctx, dseq):
        # hand_over: the block's main path left its own input gradient here instead of returning it to autograd (it runs
        # first - everything the context path's backward consumes comes out of it): the two contributions to dx are summed
        # inside this kernel instead of by a separate pass over the block input
        add = ctx.hand_over.pop("dx", None) if ctx.hand_over is not None else None
        return ops.ctx_pool_cm_bwd(dseq.contiguous(), ctx.shape, add=add), None


class CtxSplit(torch.autograd.Function):
    """g (1, H, B*(F+T), 1) -> h_cf (1, H, B*Fo, 1), h_ct (1, H, B*To, 1) [AvgPool(3, stride, 1) for stride 2], h_c (B, H)."""

    @staticmethod
    def forward(ctx, g, B, F_, T, stride):
        ctx.geo = (g.shape[1], B, F_, T, stride)
        return ops.ctx_split(g.contiguous(), B, F_, T, stride)

    @staticmethod
    def backward(ctx, dhcf, dhct, dhc):
        H, B, F_, T, stride = ctx.geo
        Fo, To = (F_ - 1) // stride + 1, (T - 1) // stride + 1
        if dhcf is None:
            dhcf = torch.zeros((1, H, B * Fo, 1), device=dhc.device)
        if dhct is None:
            dhct = torch.zeros((1, H, B * To, 1), device=dhc.device)
        dg = ops.ctx_split_bwd(dhcf.contiguous(), dhct.contiguous(), None if dhc is None else dhc.contiguous(), H, B, F_, T,
                               stride)
        return dg, None, None, None, None


class Linear(torch.autograd.Function):
    """y = x W^T + b on the MFMA linear kernel (activation applied by the caller through torch)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        ctx.prec = ops.precision.mode
        return ops.linear(x.contiguous(), w.contiguous(), b, NONE)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = ops.linear(dy, _t(w), None, NONE) if ctx.needs_input_grad[0] else None
        M = dy.shape[0]
        if M >= 4096 and M % 4 == 0 and x.shape[1] % 4 == 0:
            # long reductions (M = B * (F + T) rows of the context sequence): dW = dy^T x is the 1x1 weight-gradient
            # GEMM with one "sample" of M positions - split-K over many blocks instead of 4 waves of the linear kernel
            with _in_precision(ctx):
                dw = ops.pw_conv_wgrad(_t(dy).view(1, dy.shape[1], M, 1), _t(x).view(1, x.shape[1], M, 1),
                                       exact=None)
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
        dw = ops.dw_conv_wgrad(dz.contiguous(), x, 3, 2).view_as(w)
        return None, dw


# All weight packs of the network's STATIC 1x1 convs (the context generators' joint_conv / conv_f / conv_t, out_c, the convs of
# SE-less static blocks) - forward and data-gradient forms - from ONE launch per step (ops.PrepackPlan, as the MN plan does):
# before, every PwConv packed its matrix when it ran, 92 launches of ~7 us per dymn20 step.  `_PLAN` = (plan, run id) of the
# pass being built; a Function keeps it in its ctx and its backward uses the plan's views only while no later forward has
# re-packed them (two forwards before one backward).
_PLAN = (None, 0)


def _static_prepack_plan(model):
    plan = getattr(model, "_eat_prepack_plan", None)
    if plan is not None and not plan.stale():
        return plan
    if torch.cuda.is_current_stream_capturing() or _os.environ.get("EAT_DYMN_PLAN", "1") == "0":
        return None
    entries, seen = [], set()
    for m in model.modules():
        if isinstance(m, torch.nn.Conv2d) and m.kernel_size == (1, 1) and m.groups == 1 and m.weight.is_cuda \
                and m.weight.dtype == torch.float32 and m.weight.data_ptr() not in seen:
            seen.add(m.weight.data_ptr())
            entries += [((m.weight.data_ptr(), False), m.weight, False), ((m.weight.data_ptr(), True), m.weight, True)]
    try:
        plan = ops.PrepackPlan(entries) if entries else None
    except _lib.EatHipError:                              # channel counts the one-launch pack does not take: per-matrix packs
        plan = None
    model._eat_prepack_plan = plan
    return plan


def _pack(plan_run, w, trans):
    plan, run = plan_run
    if plan is not None and plan.runs == run:
        v = plan.views.get((w.data_ptr(), trans))
        if v is not None:
            return v
    return ops.pw_prepack(w.flatten(1), trans=trans)


class PwConv(torch.autograd.Function):
    """Static 1x1 conv (no bias): forward and data gradient are the same MFMA GEMM."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        ctx.prec = ops.precision.mode
        ctx.plan = _PLAN
        Co = w.shape[0]
        return ops.pw_conv(x.contiguous(), _pack(_PLAN, w, False), _zeros.get(Co, x.device), Co, NONE)

    @staticmethod
    def backward(ctx, dz):
        x, w = ctx.saved_tensors
        dz = dz.contiguous()
        Ci = x.shape[1]
        with _in_precision(ctx):
            dx = ops.pw_conv(dz, _pack(ctx.plan, w, True), _zeros.get(Ci, x.device), Ci, NONE)
            return dx, ops.pw_conv_wgrad(dz, x, exact=None).view_as(w)


class PwConvB(torch.autograd.Function):
    """Static 1x1 conv WITH bias (conv_f / conv_t of the context generator on the channel-major sequence)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.prec = ops.precision.mode
        ctx.plan = _PLAN
        Co = w.shape[0]
        return ops.pw_conv(x.contiguous(), _pack(_PLAN, w, False), b, Co, NONE)

    @staticmethod
    def backward(ctx, dz):
        x, w = ctx.saved_tensors
        dz = dz.contiguous()
        Ci, Co = x.shape[1], w.shape[0]
        with _in_precision(ctx):
            dx = ops.pw_conv(dz, _pack(ctx.plan, w, True), _zeros.get(Ci, x.device), Ci, NONE)
            dw = ops.pw_conv_wgrad(dz, x, exact=None).view_as(w)
        db = ops.bn_stats(dz)[:Co].float()                   # per-channel sums over (batch, positions), fp64 accumulation
        return dx, dw, db


class DwConv(torch.autograd.Function):
    """Static depthwise k x k conv (no bias): the SE-less blocks of a `use_dy_blocks="replace_se"` network and the modular
    MN train path (mn_train.forward_train_modular); dilation > 1 (models/mn/model.py:244-269) runs the generic dilated kernels."""

    @staticmethod
    def forward(ctx, x, w, k, stride, dilation=1):
        x = x.contiguous()
        ctx.save_for_backward(x, w)
        ctx.k, ctx.stride, ctx.dilation = k, stride, dilation
        C = x.shape[1]
        if dilation > 1:
            return ops.dw_conv_dilated(x, w.reshape(C, k * k), _zeros.get(C, x.device), k, stride, dilation, NONE)
        return ops.dw_conv(x, w.reshape(C, k * k), _zeros.get(C, x.device), k, stride, NONE)

    @staticmethod
    def backward(ctx, dz):
        x, w = ctx.saved_tensors
        dz = dz.contiguous()
        k, stride, dil = ctx.k, ctx.stride, ctx.dilation
        w2 = w.reshape(-1, k * k).contiguous()
        if dil > 1:
            return (ops.dw_conv_dilated_dgrad(dz, w2, x.shape, k, stride, dil),
                    ops.dw_conv_dilated_wgrad(dz, x, k, stride, dil).view_as(w), None, None, None)
        dx = ops.dw_conv_dgrad(dz, w2, x.shape, k, stride)
        return dx, ops.dw_conv_wgrad(dz, x, k, stride).view_as(w), None, None, None


def _bank_grad(G, att, bank, datt=None):
    """datt: zeroed (B, K) memory to accumulate into (a slice of the block's (n_att, B, K) tensor), or None."""
    B, K, N = att.shape[0], bank.shape[0], bank.shape[1]
    dbank = torch.empty_like(bank)
    if datt is None:
        datt = ops.zero_arena.zeros(tuple(att.shape), torch.float32, att.device)
    _lib.call("eat_dyn_bank_grad", G.data_ptr(), att.data_ptr(), bank.data_ptr(), dbank.data_ptr(), datt.data_ptr(),
              B, K, N, _s())
    return dbank, datt


def _dyn_pw(x, bank, att, transposed, res=None, stats_bn=None):
    """z_b = W_b x_b (+ res), W_b = sum_k att[b,k] bank[k] (Co, Ci) - or, `transposed`, W_b^T (the data gradient) - of a
    dynamic 1x1 conv (models/dymn/dy_block.py:103-131), in the arithmetic of the active `ops.precision`:
    late small-plane layers as ONE GEMM over the K-concatenated banks (no per-sample weights, ops.kcat_*), elsewhere
    aggregate + pack per sample (bf16 hi / lo fragments from C_in >= 40 on under 'auto', fp32 fragments otherwise)."""
    K, Co, Ci = bank.shape
    if transposed:
        Co, Ci = Ci, Co
    bank2 = bank.reshape(K, Co * Ci)                 # transposed: rows of W_k^T (the packs read it in place)
    S = x.shape[2] * x.shape[3]
    zero = _zeros.get(Co, x.device)
    tr = transposed and Co % 4 == 0
    if transposed and not tr:
        bank2 = bank.transpose(1, 2).contiguous().view(K, Co * Ci)
    # (measured, dymn20 at B = 128: with per-sample weights as bf16 hi / lo fragments the aggregate-and-pack form beats the
    #  K-concat GEMM - 4x the MFMA work - on every layer, 54.4 vs 55.3 ms per step; K-concat stays for geometries the
    #  bf16 pack does not take)
    bf16 = ops.dyn_bf16_eligible(Co, Ci, S)
    if stats_bn is not None:
        # forward conv followed by a BatchNorm: -> (z, BatchNorm state); the batch statistics leave the conv's epilogue
        if stats_bn.training and _EPI_STATS and (bf16 or not ops.kcat_eligible(Co, Ci, S)):
            wp = ops.dyn_pw_pack_bf16(bank2, att, Co, Ci) if bf16 else ops.dyn_pw_pack(bank2, att, Co, Ci)
            z, parts = ops.pw_conv_stats(x, wp, Co, per_sample=True)
            if z is not None:
                return z, ops.bn_state_from_partials(parts, stats_bn, z.numel() // Co)
        z = _dyn_pw(x, bank, att, transposed)
        return z, ops.bn_train_state(z, stats_bn)
    if bf16:
        return ops.pw_conv_dyn_bf16(x, ops.dyn_pw_pack_bf16(bank2, att, Co, Ci, trans=tr), zero, Co, NONE, res=res)
    if ops.kcat_eligible(Co, Ci, S):
        if tr:
            bank2 = bank.transpose(1, 2).contiguous().view(K, Co * Ci)
        return ops.pw_conv_kcat(x, ops.kcat_pack(bank2, Co, Ci), zero, att, Co, NONE, res=res)
    return ops.pw_conv_dyn(x, ops.dyn_pw_pack(bank2, att, Co, Ci, trans=tr), zero, Co, NONE, res=res)


def _dyn_pw_b16(x, bank, att, transposed, res=None, stats_bn=None):
    """`_dyn_pw` of the bf16-storage plan: exactly one of x / z is the wide bf16 tensor (x fp32 -> z bf16: expand conv, project
    data gradient; x bf16 -> z fp32 (+ res): project conv, expand data gradient), per-sample weights as plain bf16 fragments,
    the batch statistics of z AS STORED from the conv's epilogue (stats_bn)."""
    K, Co, Ci = bank.shape
    if transposed:
        Co, Ci = Ci, Co
    wp = ops.dyn_pw_pack_b16(bank.reshape(K, Co * Ci), att, Co, Ci, trans=transposed)
    if stats_bn is None:
        return ops.pw_conv_dyn_b16(x, wp, Co, NONE, res=res)
    if stats_bn.training:
        z, parts = ops.pw_conv_dyn_b16(x, wp, Co, NONE, stats=True)
        return z, ops.bn_state_from_partials(parts, stats_bn, z.numel() // Co)
    return ops.pw_conv_dyn_b16(x, wp, Co, NONE), ops.bn_frozen_state(stats_bn)


def _dyn_pw_wgrad(dz, x, bank, att, datt=None):
    """-> (dbank (K, Co*Ci), datt (B, K)): per-sample weight gradients G_b = dz_b x_b^T, then dbank = att^T G,
    datt = G bank^T."""
    K, Co, Ci = bank.shape
    B, S = x.shape[0], x.shape[2] * x.shape[3]
    if dz.dtype == torch.bfloat16 or x.dtype == torch.bfloat16:            # bf16-storage plan: one wide operand
        return _bank_grad(ops.pw_conv_dyn_wgrad_b16(dz, x), att, bank.reshape(K, Co * Ci), datt)
    if _WIDE_PS_WGRAD and S % 4 == 0 and Ci % 4 == 0:
        # fp32 storage: the same wide-tile producer / consumer kernel with one k-slice (or a few) per sample, split-operand
        # products - measured against the per-(tile, sample) kernels below (us): 1344 x 224 @ 504 298 -> 186-class, 960 x 160 195 -> 74-class
        G = ops.pw_conv_dyn_wgrad_b16(dz, x)
        if G is not None:
            return _bank_grad(G, att, bank.reshape(K, Co * Ci), datt)
    if ops.dyn_wgrad_needs_zero(Co, Ci, S):
        G = ops.zero_arena.zeros((B, Co * Ci), torch.float32, x.device)
    else:                                       # bf16x3 kernel in per-sample mode: plain stores
        G = torch.empty((B, Co * Ci), device=x.device, dtype=torch.float32)
    _lib.call("eat_pw_conv_dyn_wgrad", dz.data_ptr(), x.data_ptr(), G.data_ptr(), B, Co, Ci, S, _s())
    return _bank_grad(G, att, bank.reshape(K, Co * Ci), datt)


class DynPwConv(torch.autograd.Function):
    """z_b = (sum_k att[b,k] W_k) x_b for a 1x1 DynamicConv; `weight` is the (1,1,K,N) parameter."""

    @staticmethod
    def forward(ctx, x, weight, att, Co):
        x, att = x.contiguous(), att.contiguous()
        K = weight.shape[2]
        Ci = x.shape[1]
        ctx.save_for_backward(x, weight, att)
        ctx.Co = Co
        ctx.prec = ops.precision.mode
        return _dyn_pw(x, weight.view(K, Co, Ci), att, False)

    @staticmethod
    def backward(ctx, dz):
        x, weight, att = ctx.saved_tensors
        dz = dz.contiguous()
        Co, K = ctx.Co, weight.shape[2]
        bank = weight.view(K, Co, x.shape[1])
        with _in_precision(ctx):
            dx = _dyn_pw(dz, bank, att, True)
            dbank, datt = _dyn_pw_wgrad(dz, x, bank, att)
        return dx, dbank.view_as(weight), datt, None


class DynDwConv(torch.autograd.Function):
    """Depthwise DynamicConv: per-(b,c) taps = sum_k att[b,k] w_k[c]."""

    @staticmethod
    def forward(ctx, x, weight, att, k, stride, dilation=1):
        x, att = x.contiguous(), att.contiguous()
        B, C, Fq, T = x.shape
        K = weight.shape[2]
        taps = ops.dyn_aggregate(weight.view(K, C * k * k), att)
        ctx.save_for_backward(x, weight, att, taps)
        ctx.k, ctx.stride, ctx.dilation = k, stride, dilation
        if dilation > 1:
            # dilated dynamic block (models/dymn/model.py:212-218; dy_block.py:322-348): per-sample taps = a depthwise conv over
            # B * C independent planes - the batch folded into the channel axis of the generic dilated kernels
            return ops.dw_conv_dyn_dilated(x, taps, k, stride, dilation)
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
        if ctx.dilation > 1:
            dx, G = ops.dw_conv_dyn_dilated_bwd(dz, x, taps, k, stride, ctx.dilation)
            dbank, datt = _bank_grad(G, att, weight.view(K, C * k * k))
            return dx, dbank.view_as(weight), datt, None, None, None
        dx = torch.empty_like(x)
        _lib.call("eat_dw_conv_dyn_dgrad", dz.data_ptr(), taps.data_ptr(), None, dx.data_ptr(), B, C, Fq, T, Fo, To, k,
                  stride, _s())
        G = torch.zeros((B, C * k * k), device=x.device, dtype=torch.float32)
        _lib.call("eat_dw_conv_dyn_wgrad", dz.data_ptr(), x.data_ptr(), G.data_ptr(), B, C, Fq, T, Fo, To, k, stride, _s())
        dbank, datt = _bank_grad(G, att, weight.view(K, C * k * k))
        return dx, dbank.view_as(weight), datt, None, None, None


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


def _context(blk, x):
    """ContextGen (models/dymn/dy_block.py:235-254) in train mode -> (h_c (B,H), g_cf (B,Fo,cexp), g_ct (B,To,cexp))."""
    cnf = blk.cnf
    B, cin, Fq, T = x.shape
    H, cexp, stride = blk.context_dim, cnf.expanded_channels, blk.dw_stride
    cg = blk.context_gen
    L = Fq + T
    seq = CtxPool.apply(x)                                                            # (B, L, cin)
    gj = Linear.apply(seq.view(B * L, cin), cg.joint_conv.weight.flatten(1), None)     # (B*L, H)
    # BatchNorm of the context sequence over (B, L) per channel (joint_norm), batch statistics
    bn = cg.joint_norm
    if bn.training:
        gj = F.batch_norm(gj, bn.running_mean, bn.running_var, bn.weight, bn.bias, True,
                          bn.momentum if bn.momentum is not None else 1.0 / (int(bn.num_batches_tracked) + 1), bn.eps)
        ops.bn_counters.bump(bn)
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
    return h_c, g_cf, g_ct


def _context_cm(blk, x, hand_over=None):
    """The same with the sequence channel-major and the batch folded into the position axis (csrc/dymn.hip, round 4):
    -> (h_c (B, H), g_cf (1, cexp, B*Fo, 1) = [c][b][f], g_ct (1, cexp, B*To, 1)); every conv / BatchNorm of the context
    generator runs on the kernels (and autograd Functions) of the feature maps, no transposed copies."""
    B, cin, Fq, T = x.shape
    cg = blk.context_gen
    seq = CtxPoolCm.apply(x, hand_over)                                               # (1, cin, B*L, 1)
    gj = PwConv.apply(seq, cg.joint_conv.weight)
    g = BnAct.apply(gj, cg.joint_norm.weight, cg.joint_norm.bias, cg.joint_norm, HSWISH)
    h_cf, h_ct, h_c = CtxSplit.apply(g, B, Fq, T, blk.cnf.stride)
    g_cf = PwConvB.apply(h_cf, cg.conv_f.weight, cg.conv_f.bias)
    g_ct = PwConvB.apply(h_ct, cg.conv_t.weight, cg.conv_t.bias)
    return h_c, g_cf, g_ct


def _block_train(blk, x):
    # (a dilated block - models/dymn/model.py:212-218 - takes the per-layer Functions: its depthwise conv runs on the generic
    #  dilated kernels with the batch folded into the channel axis, DynDwConv / DwConv)
    if not (blk.no_dyconv or blk.no_dyrelu or blk.no_ca) and _FUSED_BLOCK and blk.cnf.dilation == 1:
        return _block_train_fused(blk, x)
    cnf = blk.cnf
    B, cin, Fq, T = x.shape
    H, cexp, cout, k, stride = blk.context_dim, cnf.expanded_channels, cnf.out_channels, cnf.kernel, blk.dw_stride
    dil = cnf.dilation
    act = HSWISH if cnf.use_hs else RELU
    inp = x
    h_c, g_cf, g_ct = _context(blk, x)
    # The ablated blocks (dy_block.py:269-271) keep the whole context generator above - the reference evaluates it
    # (and updates joint_norm's running statistics) whatever consumes it - and swap the consumers: static convs on the MN
    # Functions, the plain activation in BnAct, DyReLU-B / CoordAtt neutralised by constant operands (a1 = a2 = 1:
    # max(v, v) = v;  gates = +40: sigmoid = 1 and its derivative 0 in fp32).
    no_dyconv, no_dyrelu, no_ca = blk.no_dyconv, blk.no_dyrelu, blk.no_ca
    if blk.has_expand:
        z = PwConv.apply(x, blk.exp_conv.module.weight) if no_dyconv else \
            DynPwConv.apply(x, blk.exp_conv.weight, _attention(blk.exp_conv, h_c), cexp)
        x = BnAct.apply(z, blk.exp_norm.weight, blk.exp_norm.bias, blk.exp_norm, act)
    z = DwConv.apply(x, blk.depth_conv.module.weight, k, stride, dil) if no_dyconv else \
        DynDwConv.apply(x, blk.depth_conv.weight, _attention(blk.depth_conv, h_c), k, stride, dil)
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


# ------------------------------------------------------------------ the fully dynamic block as ONE autograd Function
import os as _os

_FUSED_BLOCK = True       # the dynamic block as one autograd Function (round 4; ablated blocks keep the per-layer Functions)
_EPI_STATS = True         # BatchNorm statistics in the dynamic 1x1 convs' epilogue
_FUSED_DW = _os.environ.get("EAT_DYMN_FUSED_DW", "1") != "0"      # A/B: the round-4 depthwise / DyReLU kernels of the block
_WIDE_PS_WGRAD = _os.environ.get("EAT_DYMN_WIDE_WGRAD", "1") != "0"     # A/B: fp32 per-sample weight gradients on the wide-tile kernel
_STORE16 = False          # set by forward_train for the pass: model.act_storage == "bf16" (the bf16-storage plan of the blocks)


class _Ones:
    def __init__(self):
        self.buf = None

    def get(self, n, device):
        if self.buf is None or self.buf.numel() < n or self.buf.device != device:
            self.buf = torch.ones((max(n, 4096),), device=device, dtype=torch.float32)
        return self.buf[:n]


_ones = _Ones()


class _HcHeads(torch.autograd.Function):
    """The four Linear layers that read h_c (the kernel attentions of the three dynamic convs, dy_block.py:106-109, and
    DyReLU-B's coefficient net, :176-181) as ONE GEMM over the row-concatenated weights, with their pointwise tails:
    -> att (3 or 2, B, K) = softmax(logits / T), coef (B, cexp, 4) = (2 sigmoid(.) - 1) * lambdas + init_v."""

    @staticmethod
    def forward(ctx, h_c, lambdas, init_v, temps, cexp, *wb):
        ws, bs = wb[0::2], wb[1::2]
        W = torch.cat(ws, 0)
        bias = torch.cat(bs, 0)
        h_c = h_c.contiguous()
        y = ops.linear(h_c, W, bias, NONE)                                    # (B, n_att * K + 4 cexp)
        B = h_c.shape[0]
        n_att = len(ws) - 1
        K = ws[0].shape[0]
        # softmax(logits / T) per DynamicConv, sigmoid + affine of the DyReLU coefficients: ONE launch (csrc/dymn.hip)
        att = torch.empty((n_att, B, K), device=y.device, dtype=torch.float32)
        sg = torch.empty((B, 4 * cexp), device=y.device, dtype=torch.float32)
        coef = torch.empty((B, cexp, 4), device=y.device, dtype=torch.float32)
        it = [1.0 / t for t in temps] + [1.0] * (3 - n_att)
        lam, iv = lambdas.contiguous().float(), init_v.contiguous().float()
        _lib.call("eat_dyn_heads_fwd", y.data_ptr(), B, n_att, K, cexp, it[0], it[1], it[2], lam.data_ptr(), iv.data_ptr(),
                  att.data_ptr(), sg.data_ptr(), coef.data_ptr(), _s())
        ctx.save_for_backward(h_c, W, att, sg, lam)
        ctx.sizes, ctx.inv_t, ctx.cexp = [w.shape[0] for w in ws], it, cexp
        return att, coef

    @staticmethod
    def backward(ctx, datt, dcoef):
        h_c, W, att, sg, lam = ctx.saved_tensors
        n_att, B, K = att.shape
        it = ctx.inv_t
        dy = torch.empty((B, n_att * K + 4 * ctx.cexp), device=att.device, dtype=torch.float32)
        _lib.call("eat_dyn_heads_bwd", datt.contiguous().data_ptr(), dcoef.contiguous().data_ptr(), att.data_ptr(), sg.data_ptr(),
                  lam.data_ptr(), B, n_att, K, ctx.cexp, it[0], it[1], it[2], dy.data_ptr(), _s())
        dh = ops.linear(dy, _t(W), None, NONE)
        dW = ops.linear(_t(dy), _t(h_c), None, NONE)
        db = _col_sum(dy)
        outs = []
        o = 0
        for n in ctx.sizes:
            outs += [dW[o:o + n], db[o:o + n]]
            o += n
        return (dh, None, None, None, None, *outs)


class DyBlockMain(torch.autograd.Function):
    """Feature-map path of a fully dynamic DY_Block (models/dymn/dy_block.py:390-409) as one Function: dynamic expand
    1x1 -> BN -> act -> dynamic depthwise -> BN -> DyReLU-B * CoordAtt -> dynamic project 1x1 -> BN (+ residual).
    The context generator (CtxPool / Linear / _HcHeads) stays on autograd and hands in att_*, coef and the gates.
    Passes this form removes against the per-op Functions: the normalised depthwise output v is never written (DyReLU /
    CoordAtt and their backward evaluate the BatchNorm affine on load), the residual is added by the last BatchNorm
    pass and enters the expand data-gradient GEMM as its `res` operand."""

    @staticmethod
    def forward(ctx, blk, fused, hand_over, x, att, coef, g_cf, g_ct, w_e, w_d, w_p, ge, be, gd, bd, gp, bp):
        cnf = blk.cnf
        ctx.hand_over = hand_over
        x = x.contiguous()
        B, cin, Fq, T = x.shape
        cexp, cout, k, stride = cnf.expanded_channels, cnf.out_channels, cnf.kernel, cnf.stride
        act = HSWISH if cnf.use_hs else RELU
        K = w_d.shape[2]
        has_e = w_e is not None
        att = att.contiguous()
        att_e, att_d, att_p = (att[0], att[1], att[2]) if has_e else (None, att[0], att[1])
        coef, g_cf, g_ct = coef.contiguous(), g_cf.contiguous(), g_ct.contiguous()
        Fo, To = ops.conv_out(Fq, k, stride), ops.conv_out(T, k, stride)
        # fused: round-4 kernels (per-plane taps in the register-resident depthwise kernels, one-wave-per-plane DyReLU on
        # channel-major gates); otherwise the separate passes on position-major gates
        # fused == 2: the bf16-storage plan (model.act_storage = "bf16"): z_e, z_d, x2 (and dx2, dv, g_e, dz_e in the backward)
        # are bf16 in HBM, plain bf16 GEMM operands, statistics of the values as stored (include/eat_hip.h, "bf16 activation
        # storage for the DyMN blocks"); block input / output, context path and per-sample weight gradients stay fp32
        b16 = fused == 2
        sv = {"fused": fused, "b16": b16}
        taps = ops.dyn_aggregate(w_d.view(K, cexp * k * k), att_d)
        if has_e and b16:
            z_e, st_e = _dyn_pw_b16(x, w_e.view(K, cexp, cin), att_e, False, stats_bn=blk.exp_norm)
            sv.update(z_e=z_e, st_e=st_e)
        elif has_e:
            z_e, st_e = _dyn_pw(x, w_e.view(K, cexp, cin), att_e, False, stats_bn=blk.exp_norm)
            sv.update(z_e=z_e, st_e=st_e)
        if fused:
            # expand BatchNorm + activation on load, depth_norm's statistics in the epilogue: y_e is never written
            z_d, parts = ops.dw_conv_dyn_stats(z_e if has_e else x, taps, k, stride,
                                               tf=(st_e[0], st_e[1], act) if has_e else None, out_b16=b16)
            st_d = ops.bn_state_from_partials(parts, blk.depth_norm, B * Fo * To) if blk.depth_norm.training else \
                ops.bn_frozen_state(blk.depth_norm)
            x2 = ops.dyrelu_ca_fwd2(z_d, st_d[0], st_d[1], coef, g_cf, g_ct)
        else:
            y_e = ops.bn_act_fwd(z_e, st_e[0], st_e[1], act) if has_e else x
            z_d = torch.empty((B, cexp, Fo, To), device=x.device, dtype=torch.float32)
            _lib.call("eat_dw_conv_dyn_fwd", y_e.data_ptr(), taps.data_ptr(), _zeros.get(cexp, x.device).data_ptr(), None,
                      None, None, z_d.data_ptr(), B, cexp, Fq, T, Fo, To, k, stride, _s())
            st_d = ops.bn_train_state(z_d, blk.depth_norm)
            x2 = torch.empty_like(z_d)
            _lib.call("eat_dyrelu_ca_fwd", z_d.data_ptr(), st_d[0].data_ptr(), st_d[1].data_ptr(), coef.data_ptr(),
                      g_cf.data_ptr(), g_ct.data_ptr(), x2.data_ptr(), B, cexp, Fo, To, _s())
            sv.update(y_e=y_e if has_e else None)
        z_p, st_p = (_dyn_pw_b16 if b16 else _dyn_pw)(x2, w_p.view(K, cout, cexp), att_p, False, stats_bn=blk.proj_norm)
        out = ops.bn_act_fwd(z_p, st_p[0], st_p[1], NONE, res=x if blk.use_res_connect else None)
        sv.update(x=x, taps=taps, z_d=z_d, st_d=st_d, x2=x2, z_p=z_p, st_p=st_p, att=att, coef=coef, g_cf=g_cf, g_ct=g_ct,
                  w_e=w_e, w_d=w_d, w_p=w_p)
        ctx.sv, ctx.blk, ctx.act, ctx.prec = sv, blk, act, ops.precision.mode
        return out

    @staticmethod
    def backward(ctx, dout):
        sv, blk, act = ctx.sv, ctx.blk, ctx.act
        ctx.sv = None
        cnf = blk.cnf
        x = sv["x"]
        B, cin, Fq, T = x.shape
        cexp, cout, k, stride = cnf.expanded_channels, cnf.out_channels, cnf.kernel, cnf.stride
        w_e, w_d, w_p = sv["w_e"], sv["w_d"], sv["w_p"]
        K = w_d.shape[2]
        has_e = w_e is not None
        att = sv["att"]
        att_e, att_d, att_p = (att[0], att[1], att[2]) if has_e else (None, att[0], att[1])
        dout = dout.contiguous()
        z_d, st_d = sv["z_d"], sv["st_d"]
        Fo, To = z_d.shape[2], z_d.shape[3]
        res = dout if blk.use_res_connect else None
        dge = dbe = None
        with _in_precision(ctx):
            # project: BN backward, data gradient, per-sample weight gradients
            dz_p, dgp, dbp = ops.bn_act_bwd(dout, sv["z_p"], *sv["st_p"], NONE)
            bank_p = w_p.view(K, cout, cexp)
            dyn_pw = _dyn_pw_b16 if sv["b16"] else _dyn_pw
            dx2 = dyn_pw(dz_p, bank_p, att_p, True)
            # the attention gradients of the block's dynamic convs accumulate into ONE zeroed (n_att, B, K) tensor (was: three
            # tensors + torch.stack)
            datt = ops.zero_arena.zeros(tuple(att.shape), torch.float32, att.device)
            dbank_p, _ = _dyn_pw_wgrad(dz_p, sv["x2"], bank_p, att_p, datt[-1])
            if sv["fused"]:
                # DyReLU-B * CoordAtt on the BatchNorm affine of z_d; the sums of depth_norm's backward leave its epilogue
                dv, dcoef, dgf, dgt, bnpart = ops.dyrelu_ca_bwd2(dx2, z_d, st_d[0], st_d[1], sv["coef"], sv["g_cf"], sv["g_ct"])
                sums_d, dgd, dbd = ops.bn_bwd_combine_partials(bnpart, bnpart.view(-1)[1:], 2, B, cexp, 1, st_d[2], st_d[3])
                # merged depthwise backward: depth_norm's backward on load, tap gradients per plane, g_e = dy_e act'(.)
                if has_e:
                    st_e = sv["st_e"]
                    g_e, G, parts = ops.dw_conv_dyn_bwd_bn_g(dv, z_d, st_d, NONE, sums_d, sv["taps"], sv["z_e"], st_e[0],
                                                             st_e[1], act, k, stride)
                    sums_e, dge, dbe = ops.bn_bwd_combine_partials(parts[0], parts[1], 1, B, cexp, parts[2], st_e[2], st_e[3])
                    dz_e = ops.bn_bwd_apply(g_e, sv["z_e"], *st_e, sums_e)
                else:
                    one, zero = _ones.get(cexp, x.device), _zeros.get(cexp, x.device)
                    dx, G, _ = ops.dw_conv_dyn_bwd_bn_g(dv, z_d, st_d, NONE, sums_d, sv["taps"], x, one, zero, NONE, k, stride,
                                                        res=res, want_sums=False)
            else:
                dv, dcoef = torch.empty_like(z_d), torch.empty_like(sv["coef"])
                dgf, dgt = torch.empty_like(sv["g_cf"]), torch.empty_like(sv["g_ct"])
                _lib.call("eat_dyrelu_ca_bwd", dx2.data_ptr(), z_d.data_ptr(), st_d[0].data_ptr(), st_d[1].data_ptr(),
                          sv["coef"].data_ptr(), sv["g_cf"].data_ptr(), sv["g_ct"].data_ptr(), dv.data_ptr(),
                          dcoef.data_ptr(), dgf.data_ptr(), dgt.data_ptr(), B, cexp, Fo, To, _s())
                dz_d, dgd, dbd = ops.bn_act_bwd(dv, z_d, *st_d, NONE)
                y_e = sv["y_e"] if has_e else x
                dy_e = torch.empty_like(y_e)
                _lib.call("eat_dw_conv_dyn_dgrad", dz_d.data_ptr(), sv["taps"].data_ptr(),
                          None if (has_e or res is None) else res.data_ptr(), dy_e.data_ptr(), B, cexp, Fq, T, Fo, To, k,
                          stride, _s())
                G = ops.zero_arena.zeros((B, cexp * k * k), torch.float32, x.device)
                _lib.call("eat_dw_conv_dyn_wgrad", dz_d.data_ptr(), y_e.data_ptr(), G.data_ptr(), B, cexp, Fq, T, Fo, To, k,
                          stride, _s())
                if has_e:
                    dz_e, dge, dbe = ops.bn_act_bwd(dy_e, sv["z_e"], *sv["st_e"], act)
                else:
                    dx = dy_e                        # (the skip connection's gradient entered the data gradient as `res`)
            dbank_d, _ = _bank_grad(G, att_d, w_d.view(K, cexp * k * k), datt[-2])
            if has_e:
                bank_e = w_e.view(K, cexp, cin)
                dx = dyn_pw(dz_e, bank_e, att_e, True, res=res)
                dbank_e, _ = _dyn_pw_wgrad(dz_e, x, bank_e, att_e, datt[0])
                dwe = dbank_e.view_as(w_e)
            else:
                dwe = None
        if ctx.hand_over is not None and ctx.needs_input_grad[3]:
            ctx.hand_over["dx"] = dx                  # collected (and added to the pools' gradient) by CtxPoolCm.backward
            dx = None
        return (None, None, None, dx, datt, dcoef, dgf, dgt, dwe, dbank_d.view_as(w_d), dbank_p.view_as(w_p), dge, dbe, dgd,
                dbd, dgp, dbp)


def _block_train_fused(blk, x):
    cnf = blk.cnf
    B, _, Fq, T = x.shape
    k, stride, cexp = cnf.kernel, cnf.stride, cnf.expanded_channels
    Fo, To = ops.conv_out(Fq, k, stride), ops.conv_out(T, k, stride)
    fused = _FUSED_DW and To <= 512 and ops.dw_bwd_merged_ok((B, cexp, Fo, To), (B, cexp, Fq, T), k, stride)
    if fused and _STORE16 and blk.depth_norm.training and ops.dyn_b16_block_ok(B, x.shape[1], cexp, cnf.out_channels, Fq, T, k, stride) \
            and (blk.has_expand or (T > 128 and k == 3 and stride == 1)):
        fused = 2                  # bf16 storage of this block's wide tensors (DyBlockMain); other blocks keep fp32 storage
    # (x needs a gradient: then the context path's backward always runs after the main path's and collects its dx)
    hand_over = {} if (fused and x.requires_grad) else None
    if hand_over is not None:
        _handovers.append(hand_over)
    h_c, g_cf, g_ct = _context_cm(blk, x, hand_over) if fused else _context(blk, x)
    convs = ([blk.exp_conv] if blk.has_expand else []) + [blk.depth_conv, blk.proj_conv]
    da = blk.depth_act
    wb = []
    for cv in convs:
        wb += [cv.residuals[0].weight, cv.residuals[0].bias]
    wb += [da.coef_net[0].weight, da.coef_net[0].bias]
    temps = tuple(float(cv.temperature) for cv in convs)
    att, coef = _HcHeads.apply(h_c, da.lambdas, da.init_v, temps, cnf.expanded_channels, *wb)
    e = blk.has_expand
    return DyBlockMain.apply(blk, fused, hand_over, x, att, coef, g_cf, g_ct, blk.exp_conv.weight if e else None, blk.depth_conv.weight,
                             blk.proj_conv.weight, blk.exp_norm.weight if e else None, blk.exp_norm.bias if e else None,
                             blk.depth_norm.weight, blk.depth_norm.bias, blk.proj_norm.weight, blk.proj_norm.bias)


def _static_block_train(blk, x):
    """SE-less InvertedResidual in train mode (models/mn/block_types.py:138-181 with se_cnf None)."""
    cnf = blk.cnf
    act = HSWISH if cnf.use_hs else RELU
    dw_stride = 1 if cnf.dilation > 1 else cnf.stride          # models/mn/block_types.py:150
    inp = x
    if blk.i_expand is not None:
        conv, bn = blk.block[blk.i_expand][0], blk.block[blk.i_expand][1]
        x = BnAct.apply(PwConv.apply(x, conv.weight), bn.weight, bn.bias, bn, act)
    conv, bn = blk.block[blk.i_dw][0], blk.block[blk.i_dw][1]
    x = BnAct.apply(DwConv.apply(x, conv.weight, cnf.kernel, dw_stride, cnf.dilation), bn.weight, bn.bias, bn, act)
    conv, bn = blk.block[blk.i_proj][0], blk.block[blk.i_proj][1]
    x = BnAct.apply(PwConv.apply(x, conv.weight), bn.weight, bn.bias, bn, NONE)
    return x + inp if blk.use_res_connect else x


def forward_train(model, x, return_fmaps=False):
    """Train-mode `(logits, embedding)` - or `(logits, fmaps)`, models/dymn/model.py:157-195 - of DyMN with autograd
    support (models/dymn/model.py:185-200).  The fully-convolutional head (:119-130) runs as torch ops on the last 4 x 32
    map (its class count, 527, is not a multiple of 4, which the library's data-gradient GEMM needs)."""
    global _handovers
    prev = getattr(model, "_eat_handovers", None) or []
    dropped = any("dx" in h for h in prev)
    model._eat_handovers = _handovers = []         # the hand-over dicts of THIS forward (filled by _block_train_fused)
    if dropped:
        raise _lib.EatHipError("DyMN: the previous backward of this model ran only part of the graph - a dynamic block's input "
                               "gradient was handed to its context path, whose backward never ran (torch.autograd.grad over "
                               "a subset of the graph?).  The fused DyMN blocks need the whole backward; EAT_DYMN_FUSED_DW=0 "
                               "selects the per-layer Functions, which have no such restriction.")
    global _STORE16
    prec = getattr(model, "train_precision", "fp32")
    st = getattr(model, "act_storage", "fp32")
    if st not in ("fp32", "bf16"):
        raise _lib.EatHipError(f"act_storage must be 'fp32' or 'bf16' (got {st!r})")
    if st == "bf16" and prec != "bf16":
        raise _lib.EatHipError("act_storage='bf16' needs train_precision='bf16' (plain bf16 GEMM operands): the split-operand "
                               "modes keep fp32-class products, which a bf16-stored operand cannot feed")
    ops.zero_arena.begin("dymn_step")          # one zero-filled arena per step (forward + the backward autograd runs later)
    global _PLAN
    _STORE16 = st == "bf16"
    try:
        with ops.precision(prec), ops.bn_counters:
            plan = _static_prepack_plan(model)
            if plan is not None:
                plan.run()
                _PLAN = (plan, plan.runs)
            out = _forward_train(model, x, return_fmaps)
    finally:
        _STORE16 = False
        _PLAN = (None, 0)
    if out[0].requires_grad:
        # the arena closes when THIS backward pass is over, whichever node runs last (a frozen stem never runs its backward;
        # ADVICE r5): the first gradient to arrive queues an engine callback, which fires after the last node of the pass
        def _arm(grad):
            torch.autograd.Variable._execution_engine.queue_callback(lambda: ops.zero_arena.end("dymn_step"))
            return grad
        out[0].register_hook(_arm)
    else:
        ops.zero_arena.end("dymn_step")
    return out


def _forward_train(model, x, return_fmaps):
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
