"""bf16 ACTIVATION STORAGE for the DyMN blocks (BASELINE configs[3] on the byte contract SURVEY 8(d) quotes for it; the
reference's 16-bit surface is Lightning `precision=16`, ex_pl_audioset.py:287-293, over models/dymn/dy_block.py:390-409).
Every `_b16` entry point against its fp32 twin fed the SAME (bf16-representable) operands - storage must change nothing but
the rounding of what is stored - and one DY_Block / the dymn20 step against the oracle's emulation of the plan."""
import contextlib
import io

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import eat_oracle as O
from oracle import synth

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("no GPU", allow_module_level=True)

from efficientat_amd import ops  # noqa: E402
from efficientat_amd.dymn import get_model  # noqa: E402

DEV = torch.device("cuda:0")
BF = torch.bfloat16


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _rel(got, ref):
    got, ref = got.detach().cpu().double().reshape(-1), ref.detach().cpu().double().reshape(-1)
    return float((got - ref).norm() / max(1e-30, float(ref.norm())))


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _r16(t):
    """fp32 tensor whose values are bf16-representable."""
    return t.to(BF).float()


def _same_after_rounding(got16, ref32, frac=2e-3, acc_noise=0.0):
    """`got16` (bf16, from a kernel that computed in fp32 and rounded on store) against an fp32 evaluation of the same
    arithmetic: equal after rounding, except where fp32 summation-order noise sits on a rounding boundary (one bf16 ulp).
    acc_noise: accumulation noise of a long fp32 reduction relative to the largest output (a GEMM's cancelling sums land a
    few of their own - tiny - ulps away)."""
    ref16 = ref32.to(BF)
    diff = (got16.float() - ref16.float()).abs()
    ulp = ref16.float().abs() * 2.0 ** -7 + 1e-30 + acc_noise * float(ref32.abs().max())
    assert float((diff / ulp).max()) <= 1.0 + 1e-3, float((diff / ulp).max())
    assert float((diff > 0).float().mean()) < frac, float((diff > 0).float().mean())


@pytest.mark.parametrize("B,Ci,Co,Fq,T", [(3, 48, 144, 8, 63), (5, 160, 96, 4, 32), (2, 40, 20, 16, 125), (2, 16, 64, 32, 250),
                                          (3, 320, 1920, 4, 32)])
def test_pw_conv_dyn_b16_both_directions(B, Ci, Co, Fq, T):
    """Per-sample-weight 1x1 conv on plain bf16 operands: fp32 x -> bf16 z (+ statistics of the stored z), bf16 x -> fp32 z
    (+ residual, + statistics), and the transposed pack (the data gradient), vs an fp64 evaluation over the rounded operands."""
    K = 4
    x, bank = _rand(B, Ci, Fq, T, seed=1), _rand(K, Co * Ci, seed=2, scale=Ci ** -0.5)
    att = torch.softmax(_rand(B, K, seed=3), dim=-1)
    W = (att @ bank).view(B, Co, Ci)                                           # fp32 aggregation, then ONE rounding
    Wd = ops.dyn_aggregate(bank.to(DEV), att.to(DEV)).view(B, Co, Ci).to(BF).double().cpu()
    wp = ops.dyn_pw_pack_b16(bank.to(DEV), att.to(DEV), Co, Ci)
    # fp32 in -> bf16 out, statistics of the stored values
    ref = torch.einsum("boi,bist->bost", Wd, x.to(BF).double())
    z, parts = ops.pw_conv_dyn_b16(x.to(DEV), wp, Co, ops.ACT_NONE, stats=True)
    assert z.dtype == BF
    _same_after_rounding(z.cpu(), ref.float(), frac=2e-2, acc_noise=2e-6)
    part, tiles, _ = parts
    p = part.view(tiles, 2, Co).double().sum(0).cpu()
    zs = z.double().cpu()
    assert _rel(p[0], zs.sum(dim=(0, 2, 3))) < 1e-5 and _rel(p[1], (zs * zs).sum(dim=(0, 2, 3))) < 1e-5
    # bf16 in -> fp32 out (+ residual)
    x16 = x.to(BF)
    res = _rand(B, Co, Fq, T, seed=6)
    y = ops.pw_conv_dyn_b16(x16.to(DEV), wp, Co, ops.ACT_NONE, res=res.to(DEV))
    assert y.dtype == torch.float32
    assert _rel(y, ref + res.double()) < 2e-6
    y2, parts2 = ops.pw_conv_dyn_b16(x16.to(DEV), wp, Co, ops.ACT_NONE, stats=True)
    p2 = parts2[0].view(parts2[1], 2, Co).double().sum(0).cpu()
    assert _rel(p2[0], ref.sum(dim=(0, 2, 3))) < 1e-4 and _rel(p2[1], (ref * ref).sum(dim=(0, 2, 3))) < 1e-5
    # the data-gradient pack from the bank of the transposed matrices
    if Co % 4 == 0:
        wpt = ops.dyn_pw_pack_b16(bank.to(DEV), att.to(DEV), Ci, Co, trans=True)      # packs W_b^T (Ci x Co)
        dz = _rand(B, Co, Fq, T, seed=7)
        dx = ops.pw_conv_dyn_b16(dz.to(DEV), wpt, Ci, ops.ACT_NONE)
        ref_dx = torch.einsum("boi,bost->bist", Wd, dz.to(BF).double())
        _same_after_rounding(dx.cpu(), ref_dx.float(), frac=2e-2, acc_noise=2e-6)
    del W


@pytest.mark.parametrize("B,Co,Ci,Fq,T,wide_x", [(3, 24, 144, 8, 63, True), (3, 144, 24, 8, 63, False), (2, 320, 1920, 4, 32, True),
                                                 (2, 1920, 320, 4, 32, False), (2, 64, 16, 32, 250, False), (5, 40, 240, 16, 125, True)])
def test_pw_conv_dyn_wgrad_b16(B, Co, Ci, Fq, T, wide_x):
    """Per-sample weight gradients G_b = dz_b x_b^T with one bf16-stored operand (plain bf16 products, fp32 accumulation)."""
    dz, x = _rand(B, Co, Fq, T, seed=1), _rand(B, Ci, Fq, T, seed=2)
    ref = torch.einsum("bost,bist->boi", dz.to(BF).double(), x.to(BF).double()).reshape(B, Co * Ci)
    G = ops.pw_conv_dyn_wgrad_b16(dz.to(DEV), x.to(BF).to(DEV)) if wide_x else ops.pw_conv_dyn_wgrad_b16(dz.to(BF).to(DEV), x.to(DEV))
    assert G.shape == (B, Co * Ci) and torch.isfinite(G).all()
    assert _rel(G, ref) < 3e-6


@pytest.mark.parametrize("B,C,Fq,T,k,s,x16", [(2, 8, 64, 500, 3, 1, False), (2, 8, 64, 500, 3, 2, True), (3, 12, 32, 250, 5, 2, True),
                                              (3, 16, 32, 250, 3, 1, True), (3, 24, 16, 125, 5, 1, True), (3, 24, 16, 125, 3, 2, True),
                                              (4, 40, 8, 63, 3, 1, True), (4, 40, 8, 63, 5, 2, True), (5, 48, 4, 32, 5, 1, True)])
def test_dynamic_depthwise_train_kernels_b16(B, C, Fq, T, k, s, x16):
    """Dynamic depthwise conv (+ expand BatchNorm / activation on load, statistics) and its merged backward on bf16 storage
    against the fp32 kernels fed the same bf16-representable tensors: outputs equal after rounding, statistics those of the
    stored values, tap gradients / partial sums to fp32 round-off."""
    act = ops.ACT_HSWISH
    x = _r16(_rand(B, C, Fq, T, seed=1))
    taps = _rand(B, C * k * k, seed=2, scale=0.3)
    a, b = torch.rand(C, generator=torch.Generator().manual_seed(3)) + 0.5, _rand(C, seed=4, scale=0.2)
    tf = (a.to(DEV), b.to(DEV), act) if x16 else None
    xd32 = x.to(DEV)
    xd = x.to(BF).to(DEV) if x16 else xd32
    y32, _ = ops.dw_conv_dyn_stats(xd32, taps.to(DEV), k, s, tf=tf)
    y16, parts = ops.dw_conv_dyn_stats(xd, taps.to(DEV), k, s, tf=tf, out_b16=True)
    assert y16.dtype == BF
    _same_after_rounding(y16.cpu(), y32.cpu())
    part, outer, inner = parts
    Fo, To = y16.shape[2], y16.shape[3]
    ps = part[:B * 2 * C * inner].view(B, 2, C, inner).double().sum(dim=(0, 3)).cpu()      # layout [b][2][C][inner]
    ys = y16.double().cpu()
    assert _rel(ps[0], ys.sum(dim=(0, 2, 3))) < 1e-5 and _rel(ps[1], (ys * ys).sum(dim=(0, 2, 3))) < 1e-5
    # merged backward: BatchNorm backward of THIS conv's output on load, tap gradients per plane, g = dx * act'(.)
    z = y16                                                                  # the stored conv output
    n = B * Fo * To
    zf = z.float()
    mean = zf.mean(dim=(0, 2, 3))
    var = zf.var(dim=(0, 2, 3), unbiased=False)
    invstd = torch.rsqrt(var + 1e-3)
    gam = torch.rand(C, device=DEV) + 0.5
    st = ((gam * invstd).contiguous(), (0.1 - mean * gam * invstd).contiguous(), mean.contiguous(), invstd.contiguous())
    dv = _r16(_rand(B, C, Fo, To, seed=5)).to(DEV)
    xhat = (zf - mean[None, :, None, None]) * invstd[None, :, None, None]
    sums = torch.cat([dv.double().sum(dim=(0, 2, 3)), (dv.double() * xhat.double()).sum(dim=(0, 2, 3))]).contiguous()
    one, zero = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    ia, ib, iact = (tf[0], tf[1], act) if x16 else (one, zero, ops.ACT_NONE)
    res = None if x16 else _rand(B, C, Fq, T, seed=8).to(DEV)
    g32, dw32, p32 = ops.dw_conv_dyn_bwd_bn_g(dv, zf.contiguous(), st, ops.ACT_NONE, sums, taps.to(DEV), xd32, ia, ib, iact, k, s, res=res)
    g16, dw16, p16 = ops.dw_conv_dyn_bwd_bn_g(dv.to(BF), z, st, ops.ACT_NONE, sums, taps.to(DEV), xd, ia, ib, iact, k, s, res=res)
    assert g16.dtype == (BF if x16 else torch.float32)
    if x16:
        _same_after_rounding(g16.cpu(), g32.cpu())
    else:
        assert _rel(g16, g32) < 1e-6
    assert _rel(dw16, dw32) < 1e-5
    # partial sums: of g as stored (before the residual), and of g * x
    gs = (g16.double() - (res.double() if res is not None else 0.0)) if not x16 else g16.double()
    s0 = p16[0][:B * C * p16[2]].view(B, C, p16[2]).double().sum(dim=(0, 2)).cpu()
    s1 = p16[1][:B * C * p16[2]].view(B, C, p16[2]).double().sum(dim=(0, 2)).cpu()
    assert _rel(s0, gs.sum(dim=(0, 2, 3)).cpu()) < 1e-4
    assert _rel(s1, (gs * x.to(DEV).double()).sum(dim=(0, 2, 3)).cpu()) < 1e-4
    (n, p32)


@pytest.mark.parametrize("B,C,Fo,To", [(3, 24, 8, 63), (3, 7, 4, 32), (2, 5, 16, 125), (2, 6, 32, 250), (1, 3, 64, 500)])
def test_dyrelu_coordatt_b16(B, C, Fo, To):
    """DyReLU-B * CoordAtt on the BatchNorm affine of a bf16-stored z, forward and backward, against the fp32 kernels."""
    z = _r16(_rand(B, C, Fo, To, seed=1)).to(DEV)
    a, b = (torch.rand(C, generator=torch.Generator().manual_seed(2)) + 0.5).to(DEV), _rand(C, seed=3, scale=0.2).to(DEV)
    coef = (_rand(B, C, 4, seed=4, scale=0.5) + torch.tensor([1.0, 0.5, 0.0, 0.0])).to(DEV)
    gf, gt = _rand(C, B, Fo, seed=5).to(DEV), _rand(C, B, To, seed=6).to(DEV)
    o32 = ops.dyrelu_ca_fwd2(z, a, b, coef, gf, gt)
    o16 = ops.dyrelu_ca_fwd2(z.to(BF), a, b, coef, gf, gt)
    assert o16.dtype == BF
    _same_after_rounding(o16.cpu(), o32.cpu(), frac=1e-6)
    dout = _r16(_rand(B, C, Fo, To, seed=7)).to(DEV)
    dv32, dc32, dgf32, dgt32, bn32 = ops.dyrelu_ca_bwd2(dout, z, a, b, coef, gf, gt)
    dv16, dc16, dgf16, dgt16, bn16 = ops.dyrelu_ca_bwd2(dout.to(BF), z.to(BF), a, b, coef, gf, gt)
    assert dv16.dtype == BF
    _same_after_rounding(dv16.cpu(), dv32.cpu(), frac=1e-6)
    assert _rel(dc16, dc32) < 1e-6 and _rel(dgf16, dgf32) < 1e-6 and _rel(dgt16, dgt32) < 1e-6
    d = dv16.double()
    assert _rel(bn16[..., 0], d.sum(dim=(2, 3))) < 1e-5 and _rel(bn16[..., 1], (d * z.double()).sum(dim=(2, 3))) < 1e-4
    bn32


def test_bn_bwd_apply_b16():
    B, C, S = 5, 24, 504
    g, z = _r16(_rand(B, C, 8, 63, seed=1)).to(DEV), _r16(_rand(B, C, 8, 63, seed=2)).to(DEV)
    mean, var = z.mean(dim=(0, 2, 3)), z.var(dim=(0, 2, 3), unbiased=False)
    invstd = torch.rsqrt(var + 1e-3)
    a = ((torch.rand(C, device=DEV) + 0.5) * invstd).contiguous()
    b = (0.1 - mean * a).contiguous()
    xhat = (z - mean[None, :, None, None]) * invstd[None, :, None, None]
    sums = torch.cat([g.double().sum(dim=(0, 2, 3)), (g.double() * xhat.double()).sum(dim=(0, 2, 3))]).contiguous()
    d32 = ops.bn_bwd_apply(g.clone(), z, a, b, mean.contiguous(), invstd.contiguous(), sums)
    g16 = g.to(BF)
    d16 = ops.bn_bwd_apply(g16, z.to(BF), a, b, mean.contiguous(), invstd.contiguous(), sums)
    assert d16.dtype == BF and d16.data_ptr() == g16.data_ptr()                   # in place
    _same_after_rounding(d16.cpu(), d32.cpu(), frac=1e-6)
    (S,)


# ------------------------------------------------------------------ one block, then the network
@pytest.mark.parametrize("i,Fq,T", [(0, 64, 500), (1, 64, 500), (3, 32, 250), (5, 16, 125), (12, 8, 63), (13, 4, 32)])
def test_dy_block_train_bf16_storage_tracks_the_emulated_oracle(i, Fq, T):
    """One DY_Block under train_precision = 'bf16' + act_storage = 'bf16' against torch-CPU autograd over the oracle block
    with the SAME roundings emulated (`O.emulate_bf16_pointwise(storage=...)`: bf16-rounded GEMM operands, the wide tensors
    and their gradients rounded where the plan stores them).  Bars: what separates two evaluations of the same bf16
    arithmetic - values within fp32 round-off of a bf16 rounding boundary round the other way (test_gpu_configs.py,
    mn40 bf16) - i.e. the bf16 grain, not the fp32 one."""
    from efficientat_amd import dymn_train as DT
    sd = synth.synth_state(synth.dymn_shapes(1.0), seed=0)
    model = _quiet(get_model, width_mult=1.0)
    model.load_state_dict(sd)
    model.to(DEV).train()
    blocks, _ = O.block_table(1.0)
    c, B, temp = blocks[i], 3, 30.0
    H = O.context_dim(c["cexp"], 1.0)
    x = _rand(B, c["cin"], Fq, T, seed=i)
    skip = ("running_mean", "running_var", "num_batches_tracked", "lambdas", "init_v")
    sdr = {k: (v.clone().requires_grad_(True) if not k.endswith(skip) else v.clone())
           for k, v in sd.items() if k.startswith(f"layers.{i}.")}
    xr = x.clone().requires_grad_(True)
    with O.emulate_bf16_pointwise(storage=True):
        out_ref = O._dy_block(sdr, f"layers.{i}", xr, c, H, True, {}, temp)
    dout = _rand(*out_ref.shape, seed=99)
    out_ref.backward(dout)
    # the fp32 oracle, for scale: how far the bf16 plan is from fp32 on this block
    sdf = {k: (v.clone().requires_grad_(True) if not k.endswith(skip) else v.clone())
           for k, v in sd.items() if k.startswith(f"layers.{i}.")}
    xf = x.clone().requires_grad_(True)
    out_f = O._dy_block(sdf, f"layers.{i}", xf, c, H, True, {}, temp)
    out_f.backward(dout)
    blk = model.layers[i]
    for m in blk.modules():
        if hasattr(m, "temperature"):
            m.temperature = temp
    xd = x.to(DEV).requires_grad_(True)
    seen = []
    orig = DT.DyBlockMain.apply
    DT._STORE16 = True
    try:
        with ops.precision("bf16"):
            ops.zero_arena.begin("dymn_step")
            out = DT._block_train(blk, xd)
            out.backward(dout.to(DEV))
            ops.zero_arena.end("dymn_step")
    finally:
        DT._STORE16 = False
    (orig, seen)
    emu_vs_f = _rel(out_ref, out_f)
    assert _rel(out, out_ref) < 0.5 * emu_vs_f + 1e-4, (_rel(out, out_ref), emu_vs_f)
    assert _rel(xd.grad, xr.grad) < max(2e-2, 0.75 * _rel(xr.grad, xf.grad)), (_rel(xd.grad, xr.grad), _rel(xr.grad, xf.grad))
    gmax = max(float(v.grad.norm()) for v in sdr.values() if getattr(v, "grad", None) is not None)
    worst = []
    for n, p in blk.named_parameters():
        ref = sdr[f"layers.{i}.{n}"].grad
        if float(ref.norm()) > 1e-3 * gmax:
            r, e = _rel(p.grad, ref), _rel(ref, sdf[f"layers.{i}.{n}"].grad)
            worst.append((round(r, 4), round(e, 4), n))
            assert r < max(3e-2, 0.75 * e), (n, r, e)
    print(f"block {i}: out vs emulated oracle {_rel(out, out_ref):.2e} (emulated vs fp32 oracle {emu_vs_f:.2e}); worst gradients "
          f"(hip vs emulated, emulated vs fp32): {sorted(worst, reverse=True)[:4]}")
