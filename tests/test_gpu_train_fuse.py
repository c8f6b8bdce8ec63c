"""GPU tests of the round-3 training kernels (csrc/train_fuse.hip, the training epilogues of csrc/dw_plane.hip):
BatchNorm statistics as per-wave partials of the depthwise conv, the depthwise data gradient with the activation
derivative + sum in its epilogue, the Gram-matrix BatchNorm state of the expand conv and its backward without dz.
Reference = torch CPU (fp64 where cancellation matters) of the same op compositions the reference model runs
(models/mn/block_types.py:138-171 under nn.BatchNorm2d(eps=1e-3, momentum=0.01), train mode)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("no GPU", allow_module_level=True)

from efficientat_amd import _lib, ops  # noqa: E402

DEV = torch.device("cuda:0")
ACTS = [lambda t: t, F.relu, F.hardswish]


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _rel(got, ref):
    got = got.detach().cpu().double().reshape(-1)
    ref = ref.detach().cpu().double().reshape(-1)
    return float((got - ref).norm() / max(1e-30, float(ref.norm())))


# plane kernels (8x63, 16x125, 4x32), tile kernels (T > 128), row-ring fallback (odd geometries), two-planes-per-wave
GEOMS = [(3, 64, 64, 500, 3, 2, 1), (2, 16, 64, 500, 3, 1, 1), (3, 72, 32, 250, 5, 2, 1), (2, 24, 32, 250, 3, 1, 1),
         (5, 120, 16, 125, 5, 1, 2), (3, 240, 16, 125, 3, 2, 2), (4, 200, 8, 63, 3, 1, 2), (3, 672, 8, 63, 5, 2, 2),
         (5, 96, 4, 32, 5, 1, 2), (2, 40, 9, 21, 3, 1, 1), (3, 9, 33, 71, 3, 2, 2), (2, 8, 64, 200, 5, 1, 1),
         (2, 3, 1, 2, 3, 2, 0), (70, 24, 8, 63, 3, 1, 2)]


@pytest.mark.parametrize("B,C,F_,T,k,s,act", GEOMS)
def test_dw_conv_stats_and_finalize(B, C, F_, T, k, s, act):
    x = _rand(B, C, F_, T, seed=1, scale=1.5) + _rand(1, C, 1, 1, seed=2)
    w = _rand(C, 1, k, k, seed=3, scale=0.3)
    ia, ib = torch.rand(C, generator=torch.Generator().manual_seed(4)) + 0.5, _rand(C, seed=5, scale=0.3)
    gamma, beta = torch.rand(C, generator=torch.Generator().manual_seed(6)) + 0.5, _rand(C, seed=7, scale=0.3)
    for tf in (False, True):
        xin = ACTS[act](x * ia[None, :, None, None] + ib[None, :, None, None]) if tf else x
        y_ref = F.conv2d(xin.double(), w.double(), None, s, (k - 1) // 2, 1, C)
        bn_ref = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.01).double().train()
        with torch.no_grad():
            bn_ref.weight.copy_(gamma)
            bn_ref.bias.copy_(beta)
        out_ref = bn_ref(y_ref)
        bn = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.01).to(DEV).train()
        with torch.no_grad():
            bn.weight.copy_(gamma)
            bn.bias.copy_(beta)
        y, parts = ops.dw_conv_stats(x.to(DEV), w.reshape(C, k * k).contiguous().to(DEV), k, s,
                                     tf=(ia.to(DEV), ib.to(DEV), act) if tf else None)
        assert _rel(y, y_ref) < 3e-6, tf
        n = y.numel() // C
        a, b, mean, invstd = ops.bn_state_from_partials(parts, bn, n)
        got = y.cpu().double() * a.cpu().double()[None, :, None, None] + b.cpu().double()[None, :, None, None]
        assert _rel(got, out_ref) < 1e-5, tf
        assert _rel(mean, y_ref.mean((0, 2, 3))) < 1e-5
        assert _rel(bn.running_mean, bn_ref.running_mean) < 1e-5 and _rel(bn.running_var, bn_ref.running_var) < 1e-5
        assert int(bn.num_batches_tracked) == 1
        # the stand-alone producer of the same partials (the fallback of geometries without a fused kernel)
        bn2 = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.01).to(DEV).train()
        with torch.no_grad():
            bn2.weight.copy_(gamma)
            bn2.bias.copy_(beta)
        st2 = ops.bn_state_from_partials(ops.bn_stats_partial(y), bn2, n)
        assert _rel(st2[0], a) < 1e-5 and _rel(st2[1], b) < 1e-5


@pytest.mark.parametrize("B,C,F_,T,k,s,act", GEOMS)
def test_dw_conv_dgrad_with_activation_epilogue(B, C, F_, T, k, s, act):
    p = (k - 1) // 2
    Fo, To = (F_ + 2 * p - k) // s + 1, (T + 2 * p - k) // s + 1
    x = _rand(B, C, F_, T, seed=1).double().requires_grad_(True)
    w = _rand(C, 1, k, k, seed=2, scale=0.3)
    dz = _rand(B, C, Fo, To, seed=3)
    F.conv2d(x, w.double(), None, s, p, 1, C).backward(dz.double())
    dx_ref = x.grad
    gz = _rand(B, C, F_, T, seed=4, scale=2.5)           # spans the Hardswish kinks at +-3
    ga, gb = torch.rand(C, generator=torch.Generator().manual_seed(5)) + 0.5, _rand(C, seed=6, scale=0.3)
    u = (gz.double() * ga.double()[None, :, None, None] + gb.double()[None, :, None, None]).requires_grad_(True)
    dact, = torch.autograd.grad(ACTS[act](u).sum(), u) if act else (torch.ones_like(u),)
    g_ref = dx_ref * dact
    g, (gpart, outer, inner) = ops.dw_conv_dgrad_g(dz.to(DEV), w.reshape(C, k * k).contiguous().to(DEV), (B, C, F_, T), k, s,
                                                   gz.to(DEV), ga.to(DEV), gb.to(DEV), act)
    assert _rel(g, g_ref) < 3e-6
    assert outer == B and inner >= 1
    sums = gpart[:B * C * inner].view(B, C, inner).sum(2)
    assert float((sums.cpu().double() - g_ref.sum((2, 3))).abs().max()) < 1e-4 * max(1.0, float(g_ref.sum((2, 3)).abs().max()))
    # stand-alone form
    dx = ops.dw_conv_dgrad(dz.to(DEV), w.reshape(C, k * k).contiguous().to(DEV), (B, C, F_, T), k, s)
    g2, (gp2, _, _) = ops.act_grad_sum(dx, gz.to(DEV), ga.to(DEV), gb.to(DEV), act)
    assert _rel(g2, g_ref) < 3e-6
    assert float((gp2.view(B, C).cpu().double() - g_ref.sum((2, 3))).abs().max()) < 1e-4 * max(1.0, float(g_ref.sum((2, 3)).abs().max()))


@pytest.mark.parametrize("B,Ci,Co,F_,T,mode,per_sample,tf", [(3, 64, 16, 64, 500, "fp32", False, True), (5, 240, 40, 16, 125, "auto", False, True),
                                                             (7, 672, 112, 8, 63, "auto", False, False), (9, 960, 160, 4, 32, "auto", False, False),
                                                             (4, 96, 24, 32, 250, "bf16", False, False), (3, 32, 128, 16, 125, "fp32", True, False),
                                                             (5, 160, 400, 8, 63, "auto", True, False), (3, 1344, 224, 4, 32, "auto", True, False),
                                                             (2, 48, 20, 5, 20, "auto", False, False)])
def test_pw_conv_with_statistics_epilogue(B, Ci, Co, F_, T, mode, per_sample, tf):
    """Train-mode 1x1 conv z = W x (static or per-sample weights, optional on-load BatchNorm + activation + SE scale of the
    input) with the batch statistics of z in the conv's epilogue: z against the plain conv (identical arithmetic), the
    BatchNorm state against fp64 statistics of z (models/mn/block_types.py:167-171, models/dymn/dy_block.py:313-316)."""
    x = _rand(B, Ci, F_, T, seed=1) + 0.3 * _rand(1, Ci, 1, 1, seed=2)
    xd = x.to(DEV)
    S = F_ * T
    bn = torch.nn.BatchNorm2d(Co, eps=1e-3, momentum=0.01).to(DEV).train()
    with ops.precision(mode):
        if per_sample:
            K = 4
            bank = _rand(K, Co * Ci, seed=3, scale=Ci ** -0.5).to(DEV)
            att = torch.softmax(_rand(B, K, seed=4), dim=-1).to(DEV)
            wp = ops.dyn_pw_pack_bf16(bank, att, Co, Ci) if ops.dyn_bf16_eligible(Co, Ci, S) else ops.dyn_pw_pack(bank, att, Co, Ci)
            z, parts = ops.pw_conv_stats(xd, wp, Co, per_sample=True)
            zero = torch.zeros(Co, device=DEV)
            z_plain = ops.pw_conv_dyn_bf16(xd, wp, zero, Co, ops.ACT_NONE) if wp.dtype == torch.bfloat16 else \
                ops.pw_conv_dyn(xd, wp, zero, Co, ops.ACT_NONE)
        else:
            W = _rand(Co, Ci, seed=3, scale=Ci ** -0.5).to(DEV)
            wp = ops.pw_prepack(W)
            tfs = (torch.rand(Ci, device=DEV) + 0.5, torch.randn(Ci, device=DEV) * 0.2, 2) if tf else None
            sc = torch.rand(B, Ci, device=DEV) if tf else None
            z, parts = ops.pw_conv_stats(xd, wp, Co, tf=tfs, in_scale=sc)
            zero = torch.zeros(Co, device=DEV)
            z_plain = ops.pw_conv_tf(xd, tfs, wp, zero, Co, ops.ACT_NONE, in_scale=sc) if tf else \
                ops.pw_conv(xd, wp, zero, Co, ops.ACT_NONE)
    if S % 4 != 0:
        assert z is None and parts is None                   # no epilogue there: the caller runs the separate pass
        return
    assert torch.equal(z, z_plain)
    st = ops.bn_state_from_partials(parts, bn, B * S)
    zd = z.cpu().double()
    mu, var = zd.mean((0, 2, 3)), zd.var((0, 2, 3), unbiased=False)
    assert _rel(st[2], mu) < 1e-5 and _rel(st[3], (var + 1e-3).rsqrt()) < 1e-5
    assert _rel(bn.running_var, 0.99 + 0.01 * var * (B * S) / (B * S - 1)) < 1e-5


@pytest.mark.parametrize("B,Ci,Co,F_,T,act,mode", [(3, 16, 64, 32, 250, 1, "fp32"), (2, 24, 72, 16, 125, 1, "fp32"),
                                                    (5, 80, 240, 8, 63, 2, "auto"), (37, 80, 184, 8, 63, 2, "auto"),
                                                    (2, 16, 16, 64, 500, 1, "auto"), (3, 40, 120, 16, 125, 0, "auto")])
def test_data_gradient_conv_with_batchnorm_backward_sums_in_its_epilogue(B, Ci, Co, F_, T, act, mode):
    """eat_pw_conv_gstats_fwd: y = W^T dz_p as the plain conv gives it, and the channel sums of the depthwise BatchNorm's
    backward over (y, z_d) as the separate reduce pass (eat_bn_act_bwd_reduce) gives them - incl. ragged channel tiles,
    tiles that straddle samples and both arithmetic modes of the data-gradient GEMM."""
    dz = _rand(B, Ci, F_, T, seed=1).to(DEV)
    W = _rand(Ci, Co, seed=2, scale=Ci ** -0.5).to(DEV)                 # project weight (Ci = its out_channels)
    z_d = (_rand(B, Co, F_, T, seed=3) * 1.5 + _rand(1, Co, 1, 1, seed=4)).to(DEV)
    a = (torch.rand(Co, generator=torch.Generator().manual_seed(5)) + 0.5).to(DEV)
    b = _rand(Co, seed=6, scale=0.5).to(DEV)
    mean = _rand(Co, seed=7, scale=0.3).to(DEV)
    invstd = (torch.rand(Co, generator=torch.Generator().manual_seed(8)) + 0.5).to(DEV)
    with ops.precision(mode):
        wpt = ops.pw_prepack(W, trans=True)
        y_ref = ops.pw_conv(dz, wpt, torch.zeros(Co, device=DEV), Co, ops.ACT_NONE)
        y, sums = ops.pw_conv_gstats(dz, wpt, Co, z_d, (a, b, mean, invstd), act)
    assert y is not None and torch.equal(y, y_ref)
    ref, _, _ = ops.bn_act_bwd_sums(y_ref, z_d, a, b, mean, invstd, act)
    # fp64 reference of the same sums from the same y
    u = a.double()[None, :, None, None] * z_d.double() + b.double()[None, :, None, None]
    d = {0: torch.ones_like(u), 1: (u > 0).double(), 2: ((u >= -3) & (u <= 3)).double() * (u / 3 + 0.5) + (u > 3).double()}[act]
    g = y_ref.double() * d
    s0 = g.sum((0, 2, 3))
    s1 = invstd.double() * (g * (z_d.double() - mean.double()[None, :, None, None])).sum((0, 2, 3))
    scale0, scale1 = float(g.abs().sum((0, 2, 3)).max()), float((g * z_d.double()).abs().sum((0, 2, 3)).max())
    assert float((sums[:Co].cpu() - s0.cpu()).abs().max()) < 2e-6 * scale0
    assert float((sums[Co:].cpu() - s1.cpu()).abs().max()) < 2e-6 * scale1 * float(invstd.max())
    assert float((ref[:Co].cpu() - s0.cpu()).abs().max()) < 2e-6 * scale0          # (the pass it replaces, same bar)


@pytest.mark.parametrize("shift", [10.0, 300.0])
def test_batchnorm_backward_sums_epilogue_with_large_channel_means(shift):
    """the epilogue's fp32 tile partials hold sum g (z_d - c), c = -b / a (the zero of the BatchNorm's
    pre-activation, within a few sigma of the channel mean), not sum g z_d - a channel with |mean| = 300 sigma keeps its
    digits: the centred sum is held to 1e-5 of sum |g (z_d - mean)| (the uncentred form is off by ~|mean| / sigma x fp32 eps)."""
    B, Ci, Co, F_, T = 4, 24, 72, 16, 125
    dz = _rand(B, Ci, F_, T, seed=1).to(DEV)
    W = _rand(Ci, Co, seed=2, scale=Ci ** -0.5).to(DEV)
    sigma = (torch.rand(Co, generator=torch.Generator().manual_seed(3)) + 0.5)
    mu = _rand(Co, seed=4) * shift * sigma
    z_d = (_rand(B, Co, F_, T, seed=5) * sigma[None, :, None, None] + mu[None, :, None, None]).to(DEV)
    mean = z_d.double().mean((0, 2, 3)).float()
    invstd = (z_d.double().var((0, 2, 3), unbiased=False) + 1e-3).rsqrt().float()
    gamma = (torch.rand(Co, generator=torch.Generator().manual_seed(6)) + 0.5).to(DEV)
    beta = _rand(Co, seed=7, scale=0.3).to(DEV)
    a = gamma * invstd
    b = beta - mean * a
    with ops.precision("fp32"):
        wpt = ops.pw_prepack(W, trans=True)
        y, sums = ops.pw_conv_gstats(dz, wpt, Co, z_d, (a, b, mean, invstd), ops.ACT_RELU)
    u = a.double()[None, :, None, None] * z_d.double() + b.double()[None, :, None, None]
    g = y.double() * (u > 0).double()
    zc = z_d.double() - mean.double()[None, :, None, None]
    s1 = invstd.double() * (g * zc).sum((0, 2, 3))
    scale = invstd.double() * (g * zc).abs().sum((0, 2, 3))
    err = float(((sums[Co:] - s1).abs() / scale.clamp_min(1e-30)).max())
    assert err < 1e-5, err
    assert float((sums[:Co] - g.sum((0, 2, 3))).abs().max()) < 2e-6 * float(g.abs().sum((0, 2, 3)).max())


@pytest.mark.parametrize("outer,C,inner", [(32000, 16, 1), (8000, 24, 1), (256, 16, 16), (256, 64, 16), (5000, 130, 1),
                                           (300, 7, 9), (2047, 1, 1), (2048, 1, 1), (100, 40, 1)])
def test_bn_finalize_from_many_partial_rows(outer, C, inner):
    """BatchNorm state from producer partials [outer][2][C][inner]: with many rows (one per 256-column tile of a 1x1 conv: 32 000
    at B = 256 on the first project conv) the rows are first summed in groups with coalesced reads (round 5: 92 -> ~10 us),
    otherwise by one block per channel; both against fp64 sums of the same partials."""
    g = torch.Generator().manual_seed(outer + C)
    n = 977.0 * outer * inner
    part = torch.randn(outer, 2, C, inner, generator=g)
    part[:, 0] = part[:, 0] * 3.0 + 40.0 * torch.randn(1, C, 1, generator=g)         # sums of z
    part[:, 1] = part[:, 1].abs() * 50.0 + (part[:, 0] ** 2) / 977.0 * 1.5           # sums of z^2 (>= mean^2 n)
    bn = torch.nn.BatchNorm2d(C).to(DEV).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
    s1, s2 = part[:, 0].double().sum((0, 2)), part[:, 1].double().sum((0, 2))
    mu = s1 / n
    var = (s2 / n - mu * mu).clamp_min(0.0)
    uses_groups = int(_lib.lib().eat_bn_finalize_ws_doubles(outer, C, inner)) > 0
    assert uses_groups == (outer * inner >= 2048)
    a, b, mean, invstd = ops.bn_state_from_partials((part.to(DEV).reshape(-1), outer, inner), bn, n)
    is_ref = (var + bn.eps).rsqrt()
    assert _rel(mean, mu) < 1e-6 and _rel(invstd, is_ref) < 1e-6
    assert _rel(a, bn.weight.detach().cpu().double() * is_ref) < 1e-6
    assert _rel(bn.running_mean, 0.1 * mu) < 1e-5


@pytest.mark.parametrize("shift", [10.0, 30.0])
@pytest.mark.parametrize("B,Ci,Co,F_,T", [(64, 960, 160, 4, 32), (16, 240, 40, 16, 125), (4, 64, 16, 64, 500)])
def test_pw_conv_statistics_epilogue_with_large_channel_means(B, Ci, Co, F_, T, shift):
    """the epilogue sums z and z^2 of a 256-column tile in fp32 and only the tiles in fp64; with |mean| = `shift`
    standard deviations the variance is a difference of (shift^2 + 1)-times larger sums.  Each tile's fp32 sum carries
    ~2^-24 sqrt(256) relative round-off, independent from tile to tile, so the variance error is
    ~1e-6 (shift^2 + 1) / sqrt(tiles) - measured here against fp64 statistics of the same z: <= 2e-4 relative on invstd at
    30 sigma with 32 tiles, far inside what eps = 1e-3 and the 1e-3 logit budget absorb (the expand convs, whose statistics
    come from the Gram matrix, were the case that needed centring: test_expand_conv_bn_act_via_gram_matrix)."""
    S = F_ * T
    W = _rand(Co, Ci, seed=3, scale=Ci ** -0.5)
    x = _rand(B, Ci, F_, T, seed=1)
    # per-output-channel offset of `shift` standard deviations of z = W x (std of z ~ 1): add it through one input channel
    z0 = torch.einsum("oi,bifs->bofs", W.double(), x.double())
    sd = z0.std((0, 2, 3))
    bias_like = (shift * sd).float()                                  # realised as an extra input channel of ones
    Wb = torch.cat([W, bias_like[:, None], torch.zeros(Co, 3)], dim=1)          # Ci + 4 channels (Ci % 4 stays 0)
    xb = torch.cat([x, torch.ones(B, 1, F_, T), torch.zeros(B, 3, F_, T)], dim=1)
    bn = torch.nn.BatchNorm2d(Co, eps=1e-3, momentum=0.01).to(DEV).train()
    with ops.precision("auto"):
        z, parts = ops.pw_conv_stats(xb.to(DEV), ops.pw_prepack(Wb.to(DEV)), Co)
    st = ops.bn_state_from_partials(parts, bn, B * S)
    zd = z.cpu().double()
    mu, var = zd.mean((0, 2, 3)), zd.var((0, 2, 3), unbiased=False)
    ratio = float((mu.abs() / var.sqrt()).median())
    assert abs(ratio - shift) < 0.2 * shift, ratio                    # the construction really put the mean `shift` sigmas out
    e_mu, e_is = _rel(st[2], mu), _rel(st[3], (var + 1e-3).rsqrt())
    print(f"pw stats epilogue |mean|/std = {ratio:.1f}, {B * S // 256} tiles: rel err mean {e_mu:.1e}, invstd {e_is:.1e}")
    assert e_mu < 1e-5 and e_is < 2e-4 * max(1.0, shift / 30.0), (e_mu, e_is)


@pytest.mark.parametrize("shift", [0.5, 10.0, 30.0])
@pytest.mark.parametrize("B,Ci,Co,F_,T,act,exact", [(3, 16, 64, 64, 500, 1, True), (3, 16, 64, 64, 500, 1, False),
                                                     (4, 40, 120, 16, 125, 1, False), (5, 112, 672, 8, 63, 2, False),
                                                     (6, 160, 960, 4, 32, 2, True), (2, 8, 24, 9, 21, 2, True)])
def test_expand_conv_bn_act_via_gram_matrix(B, Ci, Co, F_, T, act, exact, shift):
    """conv1x1 -> BatchNorm(train) -> act: forward state from the Gram matrix of the input, backward without dz
    (dW, dgamma, dbeta, dx) against fp64 autograd of the op sequence of models/mn/block_types.py:138-147.
    shift: channel means of the input in units of its standard deviation - with |mean| = 10 ... 30 std the plain Gram
    matrix loses 2 - 3 digits of the variance to cancellation, the centred one (x - mean on load) does not; two input
    channels additionally have a narrow spread (std 0.02) around their mean."""
    x = (_rand(B, Ci, F_, T, seed=1) + shift * _rand(1, Ci, 1, 1, seed=2))
    if shift > 1.0:
        x[:, :2] = 0.02 * x[:, :2] + 0.98 * shift * _rand(1, Ci, 1, 1, seed=2)[:, :2]
    W = _rand(Co, Ci, seed=3, scale=Ci ** -0.5)
    gamma, beta = torch.rand(Co, generator=torch.Generator().manual_seed(4)) + 0.5, _rand(Co, seed=5, scale=0.3)
    dy = _rand(B, Co, F_, T, seed=6)
    res = _rand(B, Ci, F_, T, seed=7)
    xr, Wr = x.double().requires_grad_(True), W.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    bn_ref = torch.nn.BatchNorm2d(Co, eps=1e-3, momentum=0.01).double().train()
    z_ref = F.conv2d(xr, Wr[:, :, None, None])
    u_ref = F.batch_norm(z_ref, bn_ref.running_mean, bn_ref.running_var, gr, br, True, 0.01, 1e-3)
    # the activation derivative is discontinuous at the kinks: fp32 round-off of z flips it for a handful of the 10^6
    # elements ("gradient-parity budget"); take those positions out of the incoming gradient on both sides
    near = (u_ref.detach().abs() < 1e-3) if act == 1 else ((u_ref.detach().abs() - 3.0).abs() < 1e-3)
    dy = dy * (~near).float()
    y_ref = ACTS[act](u_ref)
    (y_ref * dy.double()).sum().backward()

    xd, Wd = x.to(DEV), W.to(DEV)
    bn = torch.nn.BatchNorm2d(Co, eps=1e-3, momentum=0.01).to(DEV).train()
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    n = B * F_ * T
    sx = xd.double().sum((0, 2, 3)).float().contiguous()
    G0 = ops.gram(xd, exact=exact)                                        # plain Gram matrix (kept entry point)
    assert _rel(G0, torch.einsum("bift,bjft->ij", x.double(), x.double())) < (2e-6 if exact else 2e-5)
    G = ops.gram(xd, exact=exact, sx=sx)                                  # centred: what the train plan uses
    assert torch.equal(G, ops.gram(xd, exact=exact, sx=sx))               # bit-reproducible (one slot per block, fixed order)
    xc = x.double() - (sx.cpu().double() / n).view(1, Ci, 1, 1)
    assert _rel(G, torch.einsum("bift,bjft->ij", xc, xc)) < (2e-6 if exact else 2e-5)
    Tm = ops.linear(Wd, G, None, ops.ACT_NONE)
    a, b, mean, invstd = ops.gram_bn_state(Tm, Wd, sx, bn, n, centered=True)
    assert _rel(mean, z_ref.mean((0, 2, 3))) < 1e-5
    assert _rel(invstd, (z_ref.var((0, 2, 3), unbiased=False) + 1e-3).rsqrt()) < 2e-5
    assert _rel(bn.running_mean, bn_ref.running_mean) < 1e-5 and _rel(bn.running_var, bn_ref.running_var) < 2e-5
    # the one-launch form (W G formed inside, in fp64)
    bn2 = torch.nn.BatchNorm2d(Co, eps=1e-3, momentum=0.01).to(DEV).train()
    with torch.no_grad():
        bn2.weight.copy_(gamma)
        bn2.bias.copy_(beta)
    Tm2, st2 = ops.gram_bn_state_g(G, Wd, sx, bn2, n, centered=True)
    assert _rel(Tm2, W.double() @ G.cpu().double()) < 1e-6
    assert _rel(st2[2], z_ref.mean((0, 2, 3))) < 1e-5 and _rel(st2[3], (z_ref.var((0, 2, 3), unbiased=False) + 1e-3).rsqrt()) < 2e-5
    assert _rel(st2[0], a) < 1e-5 and _rel(st2[1], b) < 1e-4 and _rel(bn2.running_var, bn_ref.running_var) < 2e-5
    with ops.precision("fp32" if exact else "auto"):
        z = ops.pw_conv(xd, ops.pw_prepack(Wd), torch.zeros(Co, device=DEV), Co, ops.ACT_NONE)
        g, gparts = ops.act_grad_sum(dy.to(DEV), z, a, b, act)
        Gx = ops.pw_conv_wgrad(g, xd, exact=exact)
        dW, dgam, dbet, WaT, M, c0 = ops.expand_bwd_coef(Wd, Gx, Tm, sx, gparts, a, mean, invstd, n, centered=True)
        t = ops.pw_conv(xd, ops.pw_prepack(M), c0, Ci, ops.ACT_NONE, res=res.to(DEV))
        dx = ops.pw_conv(g, ops.pw_prepack(WaT), torch.zeros(Ci, device=DEV), Ci, ops.ACT_NONE, res=t)
    # (Gx - m1 sx^T is formed in fp64 from the fp32 sum Gx = sum g x^T: dW keeps a FIRST-order sensitivity mu/sigma - the
    #  variance and T = W Gc, where it was second order, are centred at accumulation time)
    tol = (2e-5 if exact else 1e-4) * max(1.0, shift / 5.0)
    assert _rel(dW, Wr.grad) < tol, _rel(dW, Wr.grad)
    assert _rel(dgam, gr.grad) < tol and _rel(dbet, br.grad) < tol
    assert _rel(dx, xr.grad + res.double()) < tol, _rel(dx, xr.grad + res.double())
    # frozen statistics (the layer in eval mode inside model.train()): dz = a g, no batch-mean terms
    dWf, dgf, dbf, WaTf, Mf, _ = ops.expand_bwd_coef(Wd, Gx, Gx, mean, gparts, a, mean, invstd, n, frozen=True)
    assert Mf is None
    gd = g.cpu().double()
    assert _rel(dWf, torch.einsum("bcft,bkft->ck", gd * a.cpu().double()[None, :, None, None], x.double())) < tol
    assert _rel(dbf, gd.sum((0, 2, 3))) < tol


@pytest.mark.parametrize("B,C,F_,T,act", [(4, 72, 16, 125, 1), (5, 120, 16, 125, 1), (6, 480, 8, 63, 2), (70, 96, 4, 32, 2),
                                         (3, 24, 9, 21, 2)])
def test_se_block_bn_backward_in_one_pass(B, C, F_, T, act):
    """Gate gradient + BatchNorm-backward sums of a squeeze-excitation block from ONE pass over (d, z):
    against the two-pass composition (eat_plane_dot, eat_bn_act_bwd_reduce + apply) and torch autograd."""
    z = _rand(B, C, F_, T, seed=1, scale=2.0) + _rand(1, C, 1, 1, seed=2)
    gamma, beta = torch.rand(C, generator=torch.Generator().manual_seed(3)) + 0.5, _rand(C, seed=4, scale=0.3)
    d = _rand(B, C, F_, T, seed=6)
    gs = torch.rand(B, C, generator=torch.Generator().manual_seed(7)) + 0.5
    ga = _rand(B, C, seed=8, scale=0.1)
    bn = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.01).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    zd, dd = z.to(DEV), d.to(DEV)
    a, b, mean, invstd = ops.bn_finalize(ops.bn_stats(zd), bn, B * F_ * T)
    P = ops.se_bn_bwd_partials(dd, zd, a, b, mean, act)
    y = ops.bn_act_fwd(zd, a, b, act)
    assert _rel(P[0], ops.plane_dot(dd, y)) < 1e-5
    dz_ref, dg_ref, db_ref = ops.bn_act_bwd(dd, zd, a, b, mean, invstd, act, gscale=gs.to(DEV), gadd=ga.to(DEV))
    dz, dg, db = ops.bn_act_bwd_se(dd, zd, a, b, mean, invstd, act, P, gs.to(DEV), ga.to(DEV))
    assert _rel(dz, dz_ref) < 1e-5 and _rel(dg, dg_ref) < 1e-5 and _rel(db, db_ref) < 1e-5
    # and against autograd (fp64)
    zr = z.double().requires_grad_(True)
    y_ref = ACTS[act](F.batch_norm(zr, None, None, gamma.double(), beta.double(), True, 0.0, 1e-3))
    (y_ref * (d.double() * gs.double()[:, :, None, None] + ga.double()[:, :, None, None])).sum().backward()
    assert _rel(dz, zr.grad) < 5e-5


@pytest.mark.parametrize("B,Ci,Co,F_,T,act,se", [(3, 64, 24, 32, 250, 1, False), (4, 72, 40, 16, 125, 1, True), (5, 240, 80, 8, 63, 2, False),
                                                 (6, 672, 160, 4, 32, 2, True), (2, 16, 16, 64, 500, 1, False), (3, 200, 80, 8, 63, 2, False),
                                                 (70, 120, 40, 4, 32, 1, True)])
@pytest.mark.parametrize("mode", ["fp32", "auto", "bf16"])
def test_project_conv_with_bn_act_on_load(B, Ci, Co, F_, T, act, se, mode):
    """1x1 conv (and its weight gradient) whose input act(a z + b) * s is evaluated on load == the explicit composition
    bn_act_fwd -> pw_conv (models/mn/block_types.py:150-171 in train mode), on the three arithmetic paths."""
    z = _rand(B, Ci, F_, T, seed=1, scale=2.0) + _rand(1, Ci, 1, 1, seed=2)
    a, b = torch.rand(Ci, generator=torch.Generator().manual_seed(3)) + 0.5, _rand(Ci, seed=4, scale=0.3)
    W = _rand(Co, Ci, seed=5, scale=Ci ** -0.5)
    sc = (torch.rand(B, Ci, generator=torch.Generator().manual_seed(6)) + 0.25) if se else None
    dz = _rand(B, Co, F_, T, seed=7)
    zd, ad, bd, Wd = z.to(DEV), a.to(DEV), b.to(DEV), W.to(DEV)
    scd = sc.to(DEV) if se else None
    zero = torch.zeros(Co, device=DEV)
    with ops.precision(mode):
        wp = ops.pw_prepack(Wd)
        y_act = ops.bn_act_fwd(zd, ad, bd, act)
        ref = ops.pw_conv(y_act, wp, zero, Co, ops.ACT_NONE, in_scale=scd)
        got = ops.pw_conv_tf(zd, (ad, bd, act), wp, zero, Co, ops.ACT_NONE, in_scale=scd)
        dW_ref = ops.pw_conv_wgrad(dz.to(DEV), y_act, x_scale=scd)
        dW = ops.pw_conv_wgrad(dz.to(DEV), zd, x_scale=scd, tf=(ad, bd, act))
    # same arithmetic on both sides (the transform is the same fp32 fma + activation): differences are atomics order
    # (bf16 operands: the two activation formulas differ in the last fp32 bit, which flips a bf16 rounding now and then)
    assert _rel(got, ref) < (2e-6 if mode != "bf16" else 1e-4), _rel(got, ref)
    assert _rel(dW, dW_ref) < (2e-5 if mode != "bf16" else 1e-4), _rel(dW, dW_ref)
    if mode == "fp32":
        x64 = ACTS[act](z.double() * a.double()[None, :, None, None] + b.double()[None, :, None, None])
        if se:
            x64 = x64 * sc.double()[:, :, None, None]
        assert _rel(got, F.conv2d(x64, W.double()[:, :, None, None])) < 1e-5
        assert _rel(dW, torch.einsum("bofs,bifs->oi", dz.double(), x64)) < 2e-5


@pytest.mark.parametrize("B,C,F_,T,k,s,act", [g for g in GEOMS if g[6] != 0] + [(9, 40, 16, 125, 5, 1, 2), (3, 8, 64, 500, 5, 2, 1)])
def test_dw_conv_backward_merged_kernel(B, C, F_, T, k, s, act):
    """eat_dw_conv_bwd_g: weight gradient w.r.t. act(a x + b), data gradient times act'(a x + b) and its sums from one
    pass, against fp64 autograd of F.conv2d over the activated map (models/mn/block_types.py:138-162 in train mode)."""
    p = (k - 1) // 2
    Fo, To = (F_ + 2 * p - k) // s + 1, (T + 2 * p - k) // s + 1
    x = _rand(B, C, F_, T, seed=1, scale=2.5)                              # pre-BN expand output (spans the kinks)
    ia, ib = torch.rand(C, generator=torch.Generator().manual_seed(5)) + 0.5, _rand(C, seed=6, scale=0.3)
    w = _rand(C, 1, k, k, seed=2, scale=0.3)
    dz = _rand(B, C, Fo, To, seed=3)
    u = (x.double() * ia.double()[None, :, None, None] + ib.double()[None, :, None, None]).requires_grad_(True)
    wr = w.double().requires_grad_(True)
    y = ACTS[act](u)
    F.conv2d(y, wr, None, s, p, 1, C).backward(dz.double())
    g_ref = u.grad                                                          # = dgrad(dz) * act'(u)
    dw_ref = wr.grad.reshape(C, k * k)
    g, (gpart, outer, inner), dw = ops.dw_conv_bwd_g(dz.to(DEV), w.reshape(C, k * k).contiguous().to(DEV), x.to(DEV),
                                                     ia.to(DEV), ib.to(DEV), act, k, s)
    assert _rel(g, g_ref) < 3e-6
    assert _rel(dw, dw_ref) < 2e-5
    sums = gpart[:B * C * inner].view(B, C, inner).sum(2)
    ref_s = g_ref.sum((2, 3))
    assert float((sums.cpu().double() - ref_s).abs().max()) < 1e-4 * max(1.0, float(ref_s.abs().max()))


@pytest.mark.parametrize("B,C1,C2,Co,F_,T", [(3, 64, 16, 16, 64, 500), (4, 72, 24, 24, 32, 250), (5, 120, 40, 40, 16, 125),
                                             (6, 672, 112, 112, 8, 63), (70, 960, 160, 160, 4, 32), (2, 24, 8, 8, 9, 20)])
@pytest.mark.parametrize("mode", ["fp32", "auto", "bf16"])
def test_two_source_pointwise_conv(B, C1, C2, Co, F_, T, mode):
    """eat_pw_conv_cat_fwd: W [x1 ; x2] + bias + res == the 1x1 conv over the materialised concatenation."""
    x1, x2 = _rand(B, C1, F_, T, seed=1), _rand(B, C2, F_, T, seed=2)
    W = _rand(Co, C1 + C2, seed=3, scale=(C1 + C2) ** -0.5)
    bias, res = _rand(Co, seed=4, scale=0.3), _rand(B, Co, F_, T, seed=5)
    x1d, x2d, Wd, bd, rd = x1.to(DEV), x2.to(DEV), W.to(DEV), bias.to(DEV), res.to(DEV)
    with ops.precision(mode):
        wp = ops.pw_prepack(Wd)
        ref = ops.pw_conv(torch.cat([x1d, x2d], 1).contiguous(), wp, bd, Co, ops.ACT_NONE, res=rd)
        got = ops.pw_conv_cat(x1d, x2d, wp, bd, Co, ops.ACT_NONE, res=rd)
    assert _rel(got, ref) < 1e-6, _rel(got, ref)
    if mode == "fp32":
        ref64 = F.conv2d(torch.cat([x1, x2], 1).double(), W.double()[:, :, None, None], bias.double()) + res.double()
        assert _rel(got, ref64) < 1e-5


@pytest.mark.parametrize("B,Co,Ci,F_,T,se,tf_act", [(70, 24, 72, 32, 250, True, 1), (70, 72, 24, 32, 250, False, None),
                                                    (280, 40, 120, 16, 125, True, 2), (280, 120, 40, 16, 125, False, None),
                                                    (300, 240, 40, 16, 125, False, None), (70, 24, 64, 32, 250, False, 1)])
def test_thin_weight_gradient_streaming_kernel(B, Co, Ci, F_, T, se, tf_act):
    """1x1 weight gradients of the early / middle layers (few rows, >= 512 k positions) run on the LDS-free streaming kernel
    in groups of row tiles: against fp64 einsum, with the SE scale and the on-load BatchNorm + activation of the x operand."""
    dz = _rand(B, Co, F_, T, seed=1)
    x = _rand(B, Ci, F_, T, seed=2, scale=1.5)
    sc = (torch.rand(B, Ci, generator=torch.Generator().manual_seed(3)) + 0.25) if se else None
    a, b = torch.rand(Ci, generator=torch.Generator().manual_seed(4)) + 0.5, _rand(Ci, seed=5, scale=0.3)
    x64 = x.double()
    if tf_act is not None:
        x64 = ACTS[tf_act](x64 * a.double()[None, :, None, None] + b.double()[None, :, None, None])
    if se:
        x64 = x64 * sc.double()[:, :, None, None]
    ref = torch.einsum("bofs,bifs->oi", dz.double(), x64)
    tf = (a.to(DEV), b.to(DEV), tf_act) if tf_act is not None else None
    dW = ops.pw_conv_wgrad(dz.to(DEV), x.to(DEV), x_scale=sc.to(DEV) if se else None, exact=False, tf=tf)
    assert _rel(dW, ref) < 3e-5, _rel(dW, ref)


@pytest.mark.parametrize("B,C,F_,T", [(3, 16, 128, 1000), (2, 64, 128, 1000), (5, 16, 40, 301), (2, 8, 64, 99), (1, 24, 7, 5)])
def test_stem_without_its_preactivation_tensor(B, C, F_, T):
    """Conv2d(1, C, 3, stride 2, padding 1) -> BatchNorm2d(train) -> Hardswish (models/mn/model.py:124-133): statistics from
    the Gram matrix of the log-mel's 3x3 patches, BN + act in the conv epilogue, backward (dW, dgamma, dbeta) in one pass
    over the incoming gradient - against fp64 autograd of the same op sequence."""
    x = _rand(B, 1, F_, T, seed=1) * 0.6 + 0.1
    W = _rand(C, 1, 3, 3, seed=2, scale=1.0 / 3)
    gamma, beta = torch.rand(C, generator=torch.Generator().manual_seed(4)) + 0.5, _rand(C, seed=5, scale=0.5)
    Wr = W.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    bn_ref = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.01).double().train()
    z_ref = F.conv2d(x.double(), Wr, stride=2, padding=1)
    u_ref = F.batch_norm(z_ref, bn_ref.running_mean, bn_ref.running_var, gr, br, True, 0.01, 1e-3)
    dy = _rand(*z_ref.shape, seed=6)
    near = (u_ref.detach().abs() - 3.0).abs() < 1e-3            # Hardswish' jumps at +-3: round-off of u flips a few elements
    dy = dy * (~near).float()
    y_ref = F.hardswish(u_ref)
    (y_ref * dy.double()).sum().backward()

    xd, Wd = x.to(DEV), W.reshape(C, 9).to(DEV)
    bn = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.01).to(DEV).train()
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    n = z_ref.numel() // C
    Tm, sp = ops.stem_gram(xd, Wd)
    Tm2, sp2 = ops.stem_gram(xd, Wd)
    assert torch.equal(Tm, Tm2) and torch.equal(sp, sp2)                      # fixed reduction order
    patches = F.unfold(x.double(), 3, padding=1, stride=2)                    # (B, 9, L)
    G9 = torch.einsum("bil,bjl->ij", patches, patches)
    assert _rel(Tm, W.reshape(C, 9).double() @ G9) < 1e-5 and _rel(sp, patches.sum((0, 2))) < 1e-5
    a, b, mean, invstd = ops.gram_bn_state(Tm, Wd, sp, bn, n)
    assert _rel(mean, z_ref.mean((0, 2, 3))) < 1e-5
    assert _rel(invstd, (z_ref.var((0, 2, 3), unbiased=False) + 1e-3).rsqrt()) < 2e-5
    assert _rel(bn.running_mean, bn_ref.running_mean) < 1e-5 and _rel(bn.running_var, bn_ref.running_var) < 2e-5
    y = ops.stem_conv(xd, Wd * a.unsqueeze(1), b, ops.ACT_HSWISH)
    assert _rel(y, y_ref) < 1e-5
    dy2 = _rand(*dy.shape, seed=9)
    Gx, gparts = ops.stem_bwd((dy - dy2).to(DEV), xd, Wd, a, b, ops.ACT_HSWISH, dy2=dy2.to(DEV))    # two summands, added on load
    dW, dgam, dbet = ops.expand_bwd_coef(Wd, Gx, Tm, sp, gparts, a, mean, invstd, n, need_dx=False)[:3]
    assert _rel(dW, Wr.grad) < 2e-5, _rel(dW, Wr.grad)
    assert _rel(dgam, gr.grad) < 2e-5 and _rel(dbet, br.grad) < 2e-5


@pytest.mark.parametrize("B,C,F_,T,k,s,act", [(3, 64, 64, 500, 3, 2, 1), (2, 16, 64, 500, 3, 1, 1), (3, 72, 32, 250, 5, 2, 1),
                                               (2, 24, 32, 250, 3, 1, 2), (2, 8, 64, 200, 5, 1, 1), (3, 9, 33, 171, 3, 2, 2),
                                               # small planes: whole rows per lane group, 2 / 4 samples per wave (odd batches:
                                               # lane groups without a sample), one plane per wave at 16 x 125
                                               (5, 120, 16, 125, 5, 1, 2), (3, 240, 16, 125, 3, 2, 2), (5, 200, 8, 63, 3, 1, 2),
                                               (3, 672, 8, 63, 5, 2, 2), (5, 96, 4, 32, 5, 1, 2), (7, 12, 4, 32, 3, 2, 1),
                                               (70, 24, 8, 63, 3, 1, 2), (3, 40, 9, 21, 3, 1, 1), (2, 3, 1, 2, 3, 2, 1),
                                               (9, 5, 17, 64, 5, 1, 1), (6, 7, 20, 33, 3, 2, 2)])
@pytest.mark.parametrize("variant", ["plain", "se", "frozen", "no_expand"])
def test_dw_conv_backward_with_its_batchnorm_backward_on_load(B, C, F_, T, k, s, act, variant):
    """eat_dw_conv_bwd_bn_g: act(BN_train(dwconv(act(a x + b)))) backward from (dy, z) without writing dz - against fp64
    autograd of the op sequence (models/mn/block_types.py:138-162 + the SE gate's per-plane scale / add of :72-83);
    `no_expand`: the conv input is the block input itself (first block)."""
    p = (k - 1) // 2
    x = _rand(B, C, F_, T, seed=1, scale=2.5)
    ia, ib = torch.rand(C, generator=torch.Generator().manual_seed(5)) + 0.5, _rand(C, seed=6, scale=0.3)
    in_act = act
    if variant == "no_expand":
        ia, ib, in_act = torch.ones(C), torch.zeros(C), 0
    w = _rand(C, 1, k, k, seed=2, scale=0.3)
    gamma, beta = torch.rand(C, generator=torch.Generator().manual_seed(7)) + 0.5, _rand(C, seed=8, scale=0.3)
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y_in = ACTS[in_act](xr * ia.double()[None, :, None, None] + ib.double()[None, :, None, None])
    z_ref = F.conv2d(y_in, wr, None, s, p, 1, C)
    rm, rv = torch.zeros(C).double() + 0.1, torch.ones(C).double() * 1.3
    train = variant != "frozen"
    u_ref = F.batch_norm(z_ref, rm.clone(), rv.clone(), gr, br, train, 0.01, 1e-3)
    dy = _rand(*z_ref.shape, seed=3)
    near = (u_ref.detach().abs() < 2e-3) if act == 1 else ((u_ref.detach().abs() - 3.0).abs() < 2e-3)
    dy = dy * (~near).float()
    gs = ga = None
    if variant == "se":
        gs = torch.rand(B, C, generator=torch.Generator().manual_seed(9)) + 0.2
        ga = _rand(B, C, seed=10, scale=0.05)
        d_eff = dy.double() * gs.double()[:, :, None, None] + ga.double()[:, :, None, None]
    else:
        d_eff = dy.double()
    (ACTS[act](u_ref) * d_eff).sum().backward()

    # forward state on the device (the statistics kernels have their own tests)
    zd = z_ref.detach().float().to(DEV)
    if train:
        mean = z_ref.detach().mean((0, 2, 3))
        invstd = (z_ref.detach().var((0, 2, 3), unbiased=False) + 1e-3).rsqrt()
    else:
        mean, invstd = rm, (rv + 1e-3).rsqrt()
    a = (gamma.double() * invstd).float().to(DEV)
    b = (beta.double() - mean * gamma.double() * invstd).float().to(DEV)
    mean_d, invstd_d = mean.float().to(DEV), invstd.float().to(DEV)
    if not train:
        mean_d._eat_frozen = True
    st = (a, b, mean_d, invstd_d)
    dyd = dy.to(DEV)
    gsd, gad = (None, None) if gs is None else (gs.to(DEV), ga.to(DEV))
    sums, dgam, dbet = ops.bn_act_bwd_sums(dyd, zd, *st, act, gscale=gsd, gadd=gad)
    g, gparts, dw = ops.dw_conv_bwd_bn_g(dyd, zd, st, act, sums, w.reshape(C, k * k).contiguous().to(DEV), x.to(DEV), ia.to(DEV),
                                         ib.to(DEV), in_act, k, s, gscale=gsd, gadd=gad, want_gsum=variant != "no_expand")
    g_ref = xr.grad / ia.double()[None, :, None, None]                     # gradient w.r.t. u = a x + b
    assert _rel(dgam, gr.grad) < 2e-5 and _rel(dbet, br.grad) < 2e-5
    assert _rel(g, g_ref) < 1e-5, _rel(g, g_ref)
    assert _rel(dw, wr.grad.reshape(C, k * k)) < 5e-5, _rel(dw, wr.grad.reshape(C, k * k))
    if gparts is not None:
        gpart, outer, inner = gparts
        sums_g = gpart[:B * C * inner].view(B, C, inner).sum(2)
        ref_s = g_ref.sum((2, 3))
        assert float((sums_g.cpu().double() - ref_s).abs().max()) < 2e-4 * max(1.0, float(ref_s.abs().max()))


@pytest.mark.parametrize("B,C,Cr,S", [(256, 72, 24, 2000), (37, 120, 32, 504), (5, 960, 240, 128), (3, 33, 7, 10), (64, 672, 168, 504)])
def test_se_gate_mlp_backward_in_two_launches(B, C, Cr, S):
    """eat_se_mlp_bwd against fp64 autograd of scale = sigmoid(fc2(relu(fc1(mean y)))), loss = sum_b,c ds * scale
    (models/mn/block_types.py:72-83; ds = the gate's incoming gradient)."""
    pool = _rand(B, C, seed=1) * S * 0.5
    W1, b1 = _rand(Cr, C, seed=2, scale=C ** -0.5), _rand(Cr, seed=3, scale=0.1)
    W2, b2 = _rand(C, Cr, seed=4, scale=Cr ** -0.5), _rand(C, seed=5, scale=0.1)
    ds = _rand(B, C, seed=6)
    pr = pool.double().requires_grad_(True)
    W1r, b1r, W2r, b2r = (t.double().requires_grad_(True) for t in (W1, b1, W2, b2))
    h_ref = F.relu(F.linear(pr / S, W1r, b1r))
    s_ref = torch.sigmoid(F.linear(h_ref, W2r, b2r))
    (s_ref * ds.double()).sum().backward()
    h = ops.linear(pool.to(DEV), W1.to(DEV), b1.to(DEV), ops.ACT_RELU, 1.0 / S)
    scale = ops.linear(h, W2.to(DEV), b2.to(DEV), ops.ACT_SIGMOID)
    dW1, db1, dW2, db2, gadd = ops.se_mlp_bwd(ds.to(DEV), scale, h, pool.to(DEV), W1.to(DEV), W2.to(DEV), S)
    for got, ref in ((dW1, W1r.grad), (db1, b1r.grad), (dW2, W2r.grad), (db2, b2r.grad), (gadd, pr.grad)):
        assert _rel(got, ref) < 2e-5, _rel(got, ref)


@pytest.mark.parametrize("B,C,H,N,drop", [(256, 960, 1280, 527, True), (8, 3840, 5120, 527, True), (37, 96, 130, 10, False),
                                           (3, 33, 7, 5, True)])
def test_classifier_head_backward_in_two_launches(B, C, H, N, drop):
    """eat_mlp_head_bwd against fp64 autograd of Linear -> Hardswish -> Dropout (replayed keep mask) -> Linear
    (models/mn/model.py:186-194)."""
    feat = _rand(B, C, seed=1)
    W1, b1 = _rand(H, C, seed=2, scale=C ** -0.5), _rand(H, seed=3, scale=0.1)
    W2, b2 = _rand(N, H, seed=4, scale=H ** -0.5), _rand(N, seed=5, scale=0.1)
    mask = ((torch.rand(B, H, generator=torch.Generator().manual_seed(6)) < 0.8).float() / 0.8) if drop else None
    dl = _rand(B, N, seed=7)
    fr, W1r, b1r, W2r, b2r = (t.double().requires_grad_(True) for t in (feat, W1, b1, W2, b2))
    u_ref = F.linear(fr, W1r, b1r)
    near = (u_ref.detach().abs() - 3.0).abs() < 1e-4                     # Hardswish' jumps at +-3
    h2_ref = F.hardswish(u_ref) * (mask.double() if drop else 1.0)
    out = F.linear(h2_ref, W2r, b2r)
    (out * dl.double()).sum().backward()
    u = u_ref.detach().float().to(DEV)
    h2 = h2_ref.detach().float().to(DEV)
    dW1, db1, dW2, db2, dfeat = ops.mlp_head_bwd(dl.to(DEV), h2, u, mask.to(DEV) if drop else None, feat.to(DEV), W1.to(DEV),
                                                 W2.to(DEV))
    tol = 1e-3 if bool(near.any()) else 2e-5
    assert _rel(dW2, W2r.grad) < 2e-5 and _rel(db2, b2r.grad) < 2e-5
    assert _rel(dW1, W1r.grad) < tol and _rel(db1, b1r.grad) < tol and _rel(dfeat, fr.grad) < tol


@pytest.mark.parametrize("mode", ["fp32", "auto", "bf16x3", "bf16"])
def test_all_weight_packs_from_one_launch(mode):
    """eat_pw_prepack_multi: every pack bit-identical to the single-matrix entry points, normal and transposed."""
    shapes = [(64, 16), (24, 72), (960, 160), (160, 960), (527, 1280), (40, 120), (8, 8)]
    ws = [(_rand(co, ci, 1, 1, seed=10 + i) * 0.3).to(DEV) for i, (co, ci) in enumerate(shapes)]
    entries = []
    for i, w in enumerate(ws):
        entries += [((i, False), w, False), ((i, True), w, True)]
    with ops.precision(mode):
        entries = [e for e in entries if ((e[1].shape[0] if e[2] else e[1].shape[1]) % 4 == 0) or mode in ("bf16", "bf16x3")]
        plan = ops.PrepackPlan(entries)
        plan.run()
        assert not plan.stale()
        for key, w, trans in entries:
            ref = ops.pw_prepack(w.flatten(1), trans=trans)
            got = plan.get(key)
            assert got.dtype == ref.dtype and got.shape == ref.shape
            assert getattr(got, "_eat_split", False) == getattr(ref, "_eat_split", False)
            assert torch.equal(got.view(torch.int16 if got.dtype == torch.bfloat16 else torch.int32),
                               ref.view(torch.int16 if ref.dtype == torch.bfloat16 else torch.int32)), (key, mode)
    with ops.precision("fp32" if mode != "fp32" else "auto"):
        assert plan.stale()



# (round 6) operands of the expand data gradient dx = [WaT | M] [g ; x] + c0 written straight as the GEMM's weight pack
@pytest.mark.parametrize("Co,Ci", [(64, 16), (72, 24), (120, 40), (240, 40), (184, 80), (672, 112), (960, 160), (40, 24), (88, 24)])
@pytest.mark.parametrize("mode", ["fp32", "auto", "bf16"])
def test_expand_backward_operand_pack_from_one_launch(Co, Ci, mode):
    """eat_expand_bwd_wcat: Wcat = [(diag(a) W)^T | -W^T diag(e2) W] in MFMA-fragment order + c0 = W^T e1, against fp64 through
    the two-source GEMM that consumes the pack (every pack kind; ragged row tiles Ci = 24 / 40, k padding; every element of the
    pack written - it starts as NaN)."""
    B, F_, T = 2, 8, 24
    W = _rand(Co, Ci, seed=1, scale=Ci ** -0.5).to(DEV)
    a = (torch.rand(Co, generator=torch.Generator().manual_seed(2)) + 0.5).to(DEV)
    e2 = _rand(Co, seed=3, scale=0.3).to(DEV)
    e1 = _rand(Co, seed=4, scale=0.3).to(DEV)
    g = _rand(B, Co, F_, T, seed=5).to(DEV)
    x = _rand(B, Ci, F_, T, seed=6).to(DEV)
    with ops.precision(mode):
        kind = ops.cat_pack_kind(Co + Ci)
        assert kind == {"fp32": 0, "bf16": 1}.get(mode, 2 if Co + Ci >= 40 else 0)
        nel = int(_lib.lib().eat_expand_bwd_wcat_elems(Co, Ci, kind))
        wcat = torch.full((nel,), float("nan"), device=DEV, dtype=torch.float32 if kind == 0 else torch.bfloat16)
        if kind == 2:
            wcat._eat_split = True
        c0 = torch.full((Ci,), float("nan"), device=DEV)
        _lib.call("eat_expand_bwd_wcat", W.data_ptr(), a.data_ptr(), e2.data_ptr(), e1.data_ptr(), Co, Ci, kind,
                  wcat.data_ptr(), c0.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert not bool(torch.isnan(wcat.float()).any()) and not bool(torch.isnan(c0).any())
        y = ops.pw_conv_cat(g, x, wcat, c0, Ci, ops.ACT_NONE)
        # the pack the previous five-launch path produced from the same operands (M in fp32 by torch): same layout, same values
        Wd = W.double()
        M = -(Wd.t() * e2.double()[None, :]) @ Wd
        ref_pack = ops.pw_prepack(torch.cat([(a.double()[:, None] * Wd).t(), M], dim=1).float().contiguous())
    assert ref_pack.shape == wcat.shape and ref_pack.dtype == wcat.dtype
    tol_pack = {0: 2e-6, 1: 1.6e-2, 2: 1.6e-2}[kind]                    # (kinds 1, 2: per bf16 element; hi + lo checked via y)
    assert float((wcat.float() - ref_pack.float()).abs().max()) <= tol_pack * float(ref_pack.float().abs().max())
    c0_ref = Wd.t() @ e1.double()
    assert _rel(c0, c0_ref) < 2e-6
    y_ref = (torch.einsum("ic,bcft->bift", (a.double()[:, None] * Wd).t(), g.double()) + torch.einsum("ij,bjft->bift", M, x.double())
             + c0_ref[None, :, None, None])
    assert _rel(y, y_ref) < {0: 5e-6, 1: 1.5e-2, 2: 3e-5}[kind]
