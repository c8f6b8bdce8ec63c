"""Python-side launchers for the C ABI (include/eat_hip.h): argument checks, output
allocation through PyTorch's caching allocator, launch on torch's current HIP stream.
PyTorch is plumbing here (device memory + streams); all arithmetic is in libeat_hip.so."""
import os

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_HSWISH, ACT_SIGMOID = 0, 1, 2, 3


_PRIMARY = ("x", "dz", "wave", "z", "dy")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(t, name):
    if not t.is_cuda:
        raise _lib.EatHipError(f"{name} must live on the GPU: efficientat_amd has no CPU path "
                               f"(got device {t.device})")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise _lib.EatHipError(f"{name} must be contiguous float32 (got {t.dtype}, contiguous={t.is_contiguous()})")
    if name in _PRIMARY and t.device.index != torch.cuda.current_device():   # checked on the main operand only
        # kernels are launched on the CURRENT device's stream: one process per GPU (torch.cuda.set_device), or
        # wrap the call in `with torch.cuda.device(t.device):`
        raise _lib.EatHipError(f"{name} lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}")
    return t.data_ptr()


def _opt(t, name):
    return None if t is None else _dev(t, name)


def _dev16(t, name):
    """Device pointer of a contiguous bfloat16 activation tensor (the wide tensors of the bf16-storage plan)."""
    if not t.is_cuda:
        raise _lib.EatHipError(f"{name} must live on the GPU: efficientat_amd has no CPU path (got device {t.device})")
    if t.dtype != torch.bfloat16 or not t.is_contiguous():
        raise _lib.EatHipError(f"{name} must be contiguous bfloat16 (got {t.dtype}, contiguous={t.is_contiguous()})")
    if t.device.index != torch.cuda.current_device():
        raise _lib.EatHipError(f"{name} lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}")
    return t.data_ptr()


def _is16(t):
    return t is not None and t.dtype == torch.bfloat16

_KCAT = True   # K-concat form of the late dynamic 1x1 convs (eval; `kcat_eligible`)


def conv_out(n, k, stride):
    """floor((n + 2p - (k-1) - 1)/s + 1), p=(k-1)//2   (models/mn/utils.py:24-26, dilation 1)."""
    p = (k - 1) // 2
    return (n + 2 * p - (k - 1) - 1) // stride + 1


def mel_fwd(wave, window, twiddle, band_w2, band_start, band_cnt, n_fft, hop, n_mels, fmask=(0, 0), tmask=(0, 0), out=None):
    B, L = wave.shape
    T = 1 + (L - 1) // hop
    if band_w2.shape[1:] != (n_mels, 2) or band_start.numel() != n_mels or band_cnt.numel() != n_mels:
        raise _lib.EatHipError("mel_fwd: band table must be (pairs, n_mels, 2) with n_mels starts / counts")
    if out is None:
        out = torch.empty((B, n_mels, T), device=wave.device, dtype=torch.float32)
    elif out.numel() != B * n_mels * T or out.dtype != torch.float32 or not out.is_contiguous():
        raise _lib.EatHipError(f"mel_fwd: out must be a contiguous fp32 buffer of {B} x {n_mels} x {T} elements")
    _lib.call("eat_mel_fwd", _dev(wave, "wave"), B, L, _dev(window, "window"), window.numel(), n_fft, hop,
              _dev(twiddle, "twiddle"), _dev(band_w2, "band_w2"), band_start.data_ptr(), band_cnt.data_ptr(), n_mels,
              band_w2.shape[0], out.data_ptr(), T, fmask[0], fmask[1], tmask[0], tmask[1], _stream())
    return out


def stem_conv(x, w, bias, act):
    B, _, F, T = x.shape
    C = w.shape[0]
    Fo, To = conv_out(F, 3, 2), conv_out(T, 3, 2)
    y = torch.empty((B, C, Fo, To), device=x.device, dtype=torch.float32)
    _lib.call("eat_stem_conv_fwd", _dev(x, "x"), _dev(w, "w"), _dev(bias, "bias"), y.data_ptr(), B, C, F, T,
              Fo, To, act, _stream())
    return y


def dw_conv(x, w, bias, k, stride, act, pool=None):
    B, C, F, T = x.shape
    Fo, To = conv_out(F, k, stride), conv_out(T, k, stride)
    y = torch.empty((B, C, Fo, To), device=x.device, dtype=torch.float32)
    _lib.call("eat_dw_conv_fwd", _dev(x, "x"), _dev(w, "w"), _dev(bias, "bias"), y.data_ptr(),
              _opt(pool, "pool"), B, C, F, T, Fo, To, k, stride, act, _stream())
    return y


def dw_conv_dilated(x, w, bias, k, stride, dilation, act, pool=None):
    B, C, F, T = x.shape
    pad = (k - 1) // 2 * dilation
    Fo = (F + 2 * pad - dilation * (k - 1) - 1) // stride + 1
    To = (T + 2 * pad - dilation * (k - 1) - 1) // stride + 1
    y = torch.empty((B, C, Fo, To), device=x.device, dtype=torch.float32)
    _lib.call("eat_dw_conv_dilated_fwd", _dev(x, "x"), _dev(w, "w"), _dev(bias, "bias"), y.data_ptr(), _opt(pool, "pool"),
              B, C, F, T, Fo, To, k, stride, dilation, act, _stream())
    return y


def dilated_out(F, T, k, stride, dilation):
    pad = (k - 1) // 2 * dilation
    return (F + 2 * pad - dilation * (k - 1) - 1) // stride + 1, (T + 2 * pad - dilation * (k - 1) - 1) // stride + 1


def _plane_chunks(B, C):
    """Batch ranges whose (samples x channels) planes fit the y dimension of a launch grid (65535)."""
    step = max(1, 65535 // C)
    return [(i, min(B, i + step)) for i in range(0, B, step)]


def dw_conv_dyn_dilated(x, taps, k, stride, dilation):
    """Dilated depthwise conv with PER-SAMPLE taps (B, C*k*k) - the depthwise DynamicConv of a dilated DY_Block
    (models/dymn/dy_block.py:103-131,322-348): a depthwise conv over B * C independent planes, i.e. the generic dilated
    kernels with the batch folded into the channel axis (in chunks of <= 65535 planes)."""
    B, C, F, T = x.shape
    Fo, To = dilated_out(F, T, k, stride, dilation)
    y = torch.empty((B, C, Fo, To), device=x.device, dtype=torch.float32)
    zb = _zero_bias(min(B, 65535 // C + 1) * C, x.device)
    for b0, b1 in _plane_chunks(B, C):
        n = (b1 - b0) * C
        _lib.call("eat_dw_conv_dilated_fwd", _dev(x, "x") + 4 * b0 * C * F * T, _dev(taps, "taps") + 4 * b0 * C * k * k,
                  zb.data_ptr(), y.data_ptr() + 4 * b0 * C * Fo * To, None, 1, n, F, T, Fo, To, k, stride, dilation, ACT_NONE,
                  _stream())
    return y


def dw_conv_dyn_dilated_bwd(dz, x, taps, k, stride, dilation):
    """-> (dx, G (B, C*k*k) per-plane tap gradients) of `dw_conv_dyn_dilated`."""
    B, C, F, T = x.shape
    Fo, To = dz.shape[2], dz.shape[3]
    dx = torch.empty_like(x)
    G = torch.empty((B, C * k * k), device=x.device, dtype=torch.float32)
    for b0, b1 in _plane_chunks(B, C):
        n = (b1 - b0) * C
        _lib.call("eat_dw_conv_dilated_dgrad", _dev(dz, "dz") + 4 * b0 * C * Fo * To, _dev(taps, "taps") + 4 * b0 * C * k * k,
                  dx.data_ptr() + 4 * b0 * C * F * T, 1, n, F, T, Fo, To, k, stride, dilation, _stream())
        _lib.call("eat_dw_conv_dilated_wgrad", _dev(dz, "dz") + 4 * b0 * C * Fo * To, _dev(x, "x") + 4 * b0 * C * F * T,
                  G.data_ptr() + 4 * b0 * C * k * k, 1, n, F, T, Fo, To, k, stride, dilation, _stream())
    return dx, G


def dyrelu_ca(z, a, b, act, coef, gate_f, gate_t):
    """BatchNorm affine (a, b per channel) + DyReLU-B (coef (B, C, 4); None: the plain activation `act`) + CoordAtt (position-
    major PRE-sigmoid gates (B*Fo, C) / (B*To, C); None: no attention) of a (B, C, Fo, To) tensor - the stand-alone kernel
    of the dilated dynamic block (eat_dyrelu_ca_fwd; the ablations are expressed through constant operands: a1 = a2 = 1 is
    the identity, gates of +40 a sigmoid of exactly 1 in fp32)."""
    B, C, Fo, To = z.shape
    if coef is None:
        # plain activation in DyReLU's place: BatchNorm + act through the generic pass, attention (if any) on its output
        y = bn_act_fwd(z, a, b, act)
        if gate_f is None:
            return y
        z, a, b = y, None, None
        coef = torch.tensor([1.0, 1.0, 0.0, 0.0], device=z.device).expand(B, C, 4).contiguous()
    if gate_f is None:
        gate_f = torch.full((B * Fo, C), 40.0, device=z.device)
        gate_t = torch.full((B * To, C), 40.0, device=z.device)
    out = torch.empty_like(z)
    _lib.call("eat_dyrelu_ca_fwd", _dev(z, "z"), _opt(a, "a"), _opt(b, "b"), _dev(coef.contiguous(), "coef"),
              _dev(gate_f.contiguous(), "gate_f"), _dev(gate_t.contiguous(), "gate_t"), out.data_ptr(), B, C, Fo, To, _stream())
    return out


def dw_conv_dilated_dgrad(dz, w, x_shape, k, stride, dilation):
    """Data gradient of the dilated depthwise conv (training of `dilated=True` networks; generic kernel)."""
    B, C, F, T = x_shape
    dx = torch.empty((B, C, F, T), device=dz.device, dtype=torch.float32)
    _lib.call("eat_dw_conv_dilated_dgrad", _dev(dz, "dz"), _dev(w, "w"), dx.data_ptr(), B, C, F, T, dz.shape[2], dz.shape[3],
              k, stride, dilation, _stream())
    return dx


def dw_conv_dilated_wgrad(dz, x, k, stride, dilation):
    B, C, F, T = x.shape
    dw = torch.empty((C, k * k), device=dz.device, dtype=torch.float32)
    _lib.call("eat_dw_conv_dilated_wgrad", _dev(dz, "dz"), _dev(x, "x"), dw.data_ptr(), B, C, F, T, dz.shape[2], dz.shape[3],
              k, stride, dilation, _stream())
    return dw


def dw_conv_tf(x, in_a, in_b, in_act, w, bias, k, stride):
    """Depthwise conv of act_in(in_a[c] * x + in_b[c]) (evaluated on load), no output activation (train mode)."""
    B, C, F, T = x.shape
    Fo, To = conv_out(F, k, stride), conv_out(T, k, stride)
    y = torch.empty((B, C, Fo, To), device=x.device, dtype=torch.float32)
    _lib.call("eat_dw_conv_fwd_tf", _dev(x, "x"), _dev(in_a, "in_a"), _dev(in_b, "in_b"), in_act, _dev(w, "w"),
              _dev(bias, "bias"), y.data_ptr(), B, C, F, T, Fo, To, k, stride, _stream())
    return y


def pw_prepack(w2d, row_scale=None, trans=False):
    """trans: pack w2d^T (w2d is the stored (Ci, Co) matrix) - the data-gradient GEMM without a transposed copy."""
    Co, Ci = (w2d.shape[1], w2d.shape[0]) if trans else w2d.shape
    wp = torch.empty(((Ci // 4) * ((Co + 15) // 16) * 64,), device=w2d.device, dtype=torch.float32)
    _lib.call("eat_pw_prepack_t" if trans else "eat_pw_prepack", _dev(w2d, "w"), _opt(row_scale, "row_scale"),
              wp.data_ptr(), Co, Ci, _stream())
    return wp


def pw_conv(x, wp, bias, Co, act, in_scale=None, res=None, pool=None, write=True):
    B, Ci, F, T = x.shape
    y = torch.empty((B, Co, F, T), device=x.device, dtype=torch.float32) if write else None
    _lib.call("eat_pw_conv_fwd", _dev(x, "x"), _dev(wp, "wp"), _dev(bias, "bias"), _opt(in_scale, "in_scale"),
              _opt(res, "res"), None if y is None else y.data_ptr(), _opt(pool, "pool"), B, Ci, Co, F * T, act,
              _stream())
    return y


def linear(x, w, bias, act, x_scale=1.0):
    B, K = x.shape
    N = w.shape[0]
    y = torch.empty((B, N), device=x.device, dtype=torch.float32)
    _lib.call("eat_linear_fwd", _dev(x, "x"), _dev(w, "w"), _opt(bias, "bias"), y.data_ptr(), B, K, N,
              float(x_scale), act, _stream())
    return y


# ------------------------------------------------------------------ training-step launchers
class _ZeroArena:
    """One zero-filled allocation per pass instead of ~90 `torch.zeros` launches: the kernels that accumulate with atomics
    (weight gradients, fp64 BatchNorm sums) carve their zeroed outputs out of it.  `with zero_arena.scope("bwd"):` opens
    a pass; its size is learnt from the first pass of that name (which falls back to individual `torch.zeros`)."""

    def __init__(self):
        self.need, self.buf, self.off, self.req, self.name = {}, None, 0, 0, None

    def scope(self, name):
        arena = self

        class _Scope:
            def __enter__(self):
                self.outer = (arena.name, arena.buf, arena.off, arena.req)
                arena.name, arena.buf, arena.off, arena.req = name, None, 0, 0

            def __exit__(self, *exc):
                arena.need[name] = max(arena.need.get(name, 0), arena.req)
                arena.name, arena.buf, arena.off, arena.req = self.outer

        return _Scope()

    def begin(self, name):
        """Open a pass that stays open until the next `begin` of the same name (DyMN: the backward runs inside autograd,
        after `forward_train` has returned, so no `with` block can span the step)."""
        if self.name == name:
            self.need[name] = max(self.need.get(name, 0), self.req)
        self.name, self.buf, self.off, self.req = name, None, 0, 0

    def end(self, name):
        """Close a pass opened with `begin` (DyMN: called when the backward reaches the stem, and by GraphedTrainStep after
        its capture): later un-scoped callers get their own `torch.zeros` again instead of slices of the step's buffer -
        which a captured graph re-zeroes on every replay."""
        if self.name == name:
            self.need[name] = max(self.need.get(name, 0), self.req)
            self.name, self.buf, self.off, self.req = None, None, 0, 0

    def zeros(self, shape, dtype, device):
        if self.name is None:
            return torch.zeros(shape, device=device, dtype=dtype)
        numel = 1
        for d in shape:
            numel *= d
        nbytes = (numel * torch.empty((), dtype=dtype).element_size() + 255) // 256 * 256
        self.req += nbytes
        need = self.need.get(self.name, 0)
        if self.buf is None and need:
            self.buf = torch.zeros((need,), device=device, dtype=torch.uint8)
        if self.buf is None or self.off + nbytes > self.buf.numel() or self.buf.device != device:
            return torch.zeros(shape, device=device, dtype=dtype)
        v = self.buf[self.off:self.off + nbytes].view(dtype)[:numel].view(shape)
        self.off += nbytes
        return v


zero_arena = _ZeroArena()


class _DeferredCounters:
    """`bn.num_batches_tracked += 1` of every BatchNorm layer of a forward pass as ONE multi-tensor launch."""

    def __init__(self):
        self.depth, self.pending = 0, []

    def __enter__(self):
        self.depth += 1
        return self

    def __exit__(self, *exc):
        self.depth -= 1
        if self.depth == 0 and self.pending:
            torch._foreach_add_(self.pending, 1)
            self.pending = []

    def bump(self, bn):
        if self.depth:
            self.pending.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked += 1


bn_counters = _DeferredCounters()


def bn_stats(z):
    B, C = z.shape[0], z.shape[1]
    S = z.numel() // (B * C)
    sums = zero_arena.zeros((2 * C,), torch.float64, z.device)
    _lib.call("eat_bn_stats", _dev(z, "z"), B, C, S, sums.data_ptr(), _stream())
    return sums


def bn_finalize(sums, bn, n, momentum=None):
    """-> (a, b, mean, invstd); updates bn.running_mean / running_var in place (momentum rule)."""
    C = bn.num_features
    out = torch.empty((4, C), device=sums.device, dtype=torch.float32)
    mom = momentum if momentum is not None else (bn.momentum if bn.momentum is not None else 0.0)
    _lib.call("eat_bn_finalize", sums.data_ptr(), _dev(bn.weight, "gamma"), _dev(bn.bias, "beta"),
              bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(mom), float(bn.eps), float(n), C,
              out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), _stream())
    return out[0], out[1], out[2], out[3]


def bn_frozen_state(bn):
    """(a, b, mean, invstd) of a BatchNorm layer in eval mode inside a train-mode pass (running statistics, no buffer
    update); flagged so that the backward treats the layer as a fixed affine map."""
    with torch.no_grad():
        invstd = torch.rsqrt(bn.running_var + bn.eps)
        a = (bn.weight * invstd).contiguous()
        st = (a, (bn.bias - bn.running_mean * a).contiguous(), bn.running_mean.clone(), invstd.contiguous())
    st[2]._eat_frozen = True
    return st


def bn_train_state(z, bn):
    """(a, b, mean, invstd) of one BatchNorm layer inside a train-mode pass, honouring the layer's OWN mode:
    * bn.training: batch statistics; running buffers updated with `momentum` (None = cumulative moving average
      with factor 1 / num_batches_tracked, torch semantics), num_batches_tracked += 1;
    * bn.eval() inside model.train() (the freeze-BN fine-tuning recipe): running statistics, no buffer update; the
      returned state is flagged so that `bn_act_bwd` treats the layer as a fixed affine map."""
    C = z.shape[1]
    if not bn.training:
        return bn_frozen_state(bn)
    mom = bn.momentum
    if mom is None:
        mom = 1.0 / (int(bn.num_batches_tracked) + 1)          # host read: this mode is not graph-capturable
    st = bn_finalize(bn_stats(z), bn, z.numel() // C, momentum=mom)
    bn_counters.bump(bn)
    return st


def bn_act_fwd(z, a, b, act, res=None, pool=None, write=True, y_f32=False, copy16=False):
    """y = act(a[c] z + b[c]) [+ res]; pool (B, C): plane sums of y.  A bf16 z (bf16-storage plan): y is bf16 too - or, with
    y_f32 (the project conv's BatchNorm: z_p stored in bf16, the block output fp32), fp32 with the optional fp32 residual;
    copy16 (with y_f32): -> (y, bf16 rounding of y), both written by the one pass."""
    B, C = z.shape[0], z.shape[1]
    S = z.numel() // (B * C)
    if _is16(z):                                   # bf16 storage (BASELINE configs[2]): pool = sums of the values as stored
        y16 = not y_f32
        if res is not None and y16:
            raise _lib.EatHipError("bn_act_fwd: a residual is added to an fp32 output only (y_f32=True)")
        y = torch.empty(z.shape, device=z.device, dtype=torch.bfloat16 if y16 else torch.float32) if write else None
        yc = torch.empty(z.shape, device=z.device, dtype=torch.bfloat16) if copy16 and y_f32 and write else None
        _lib.call("eat_bn_act_fwd_b16", _dev16(z, "z"), a.data_ptr(), b.data_ptr(), _opt(res, "res"),
                  None if y is None else y.data_ptr(), 1 if y16 else 0, None if yc is None else yc.data_ptr(),
                  _opt(pool, "pool"), B, C, S, act, _stream())
        return (y, yc) if copy16 else y
    y = torch.empty_like(z) if write else None
    _lib.call("eat_bn_act_fwd", _dev(z, "z"), a.data_ptr(), b.data_ptr(), _opt(res, "res"),
              None if y is None else y.data_ptr(), _opt(pool, "pool"), B, C, S, act, _stream())
    return y


def bn_act_bwd(dy, z, a, b, mean, invstd, act, gscale=None, gadd=None, frozen=None, sums=None, copy16=False):
    """-> (dz, dgamma, dbeta) for y = act(BN_batch(z)); incoming grad = dy*gscale[b,c] + gadd[b,c].
    frozen (default: the flag `bn_train_state` left on `mean`): the layer normalised with its running statistics,
    i.e. dz = a * g without the batch-mean terms (dgamma / dbeta are the same reductions)."""
    if frozen is None:
        frozen = getattr(mean, "_eat_frozen", False)
    B, C = z.shape[0], z.shape[1]
    S = z.numel() // (B * C)
    own = sums is None            # sums: a zeroed (2C,) float64 slice of the caller's arena (one fp32 conversion per pass)
    if own:
        sums = zero_arena.zeros((2 * C,), torch.float64, z.device)
    asums = torch.zeros_like(sums) if frozen else sums
    if _is16(z):                  # bf16-storage plan: the project conv's z_p is bf16, its gradient and dz are fp32 tensors
        tail = (a.data_ptr(), b.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _opt(gscale, "gscale"), _opt(gadd, "gadd"))
        _lib.call("eat_bn_act_bwd_reduce_b16", _dev(dy, "dy"), 0, _dev16(z, "z"), *tail, B, C, S, act, sums.data_ptr(), _stream())
        dz = torch.empty(z.shape, device=z.device, dtype=torch.float32)
        dzc = torch.empty(z.shape, device=z.device, dtype=torch.bfloat16) if copy16 else None
        _lib.call("eat_bn_act_bwd_apply_b16", _dev(dy, "dy"), _dev16(z, "z"), *tail, asums.data_ptr(), dz.data_ptr(),
                  None if dzc is None else dzc.data_ptr(), B, C, S, act, _stream())
        if copy16:                 # (dz, its bf16 rounding): bf16-storage plan, z = the project conv's z_p
            dz = (dz, dzc)
        if not own:
            return dz, None, None
        sf = sums.float()
        return dz, sf[C:], sf[:C]
    args = (_dev(dy, "dy"), _dev(z, "z"), a.data_ptr(), b.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
            _opt(gscale, "gscale"), _opt(gadd, "gadd"))
    _lib.call("eat_bn_act_bwd_reduce", *args, B, C, S, act, sums.data_ptr(), _stream())
    dz = torch.empty_like(z)
    _lib.call("eat_bn_act_bwd_apply", *args, asums.data_ptr(), dz.data_ptr(), B, C, S, act, _stream())
    if not own:
        return dz, None, None
    sf = sums.float()
    return dz, sf[C:], sf[:C]


def se_bn_bwd_partials(d, z, a, b, mean, act):
    """One pass over (d, z) of a squeeze-excitation block -> P (5, B, C): P[0] = d s (the gate gradient, = plane_dot of d
    with act(a z + b)), P[1..4] the plane sums `bn_act_bwd_se` combines once gadd is known (csrc/train_fuse.hip)."""
    B, C = z.shape[0], z.shape[1]
    P = torch.empty((5, B, C), device=z.device, dtype=torch.float32)
    if _is16(z):
        _lib.call("eat_se_bn_bwd_partials_b16", _dev16(d, "dy"), _dev16(z, "z"), a.data_ptr(), b.data_ptr(), mean.data_ptr(),
                  P.data_ptr(), B, C, z.numel() // (B * C), act, _stream())
        return P
    _lib.call("eat_se_bn_bwd_partials", _dev(d, "dy"), _dev(z, "z"), a.data_ptr(), b.data_ptr(), mean.data_ptr(),
              P.data_ptr(), B, C, z.numel() // (B * C), act, _stream())
    return P


def bn_act_bwd_se(dy, z, a, b, mean, invstd, act, P, gscale, gadd, sums=None):
    """bn_act_bwd for a squeeze-excitation block whose plane sums P were taken by `se_bn_bwd_partials`: no reduce pass."""
    B, C = z.shape[0], z.shape[1]
    S = z.numel() // (B * C)
    own = sums is None
    if own:
        sums = torch.empty((2 * C,), device=z.device, dtype=torch.float64)
    _lib.call("eat_se_bn_bwd_combine", P.data_ptr(), _dev(gscale, "gscale"), _dev(gadd, "gadd"), invstd.data_ptr(), B, C,
              sums.data_ptr(), _stream())
    frozen = getattr(mean, "_eat_frozen", False)
    dz = torch.empty_like(z)
    asums = torch.zeros_like(sums) if frozen else sums
    _lib.call("eat_bn_act_bwd_apply", _dev(dy, "dy"), _dev(z, "z"), a.data_ptr(), b.data_ptr(), mean.data_ptr(),
              invstd.data_ptr(), gscale.data_ptr(), gadd.data_ptr(), asums.data_ptr(), dz.data_ptr(), B, C, S, act,
              _stream())
    if not own:
        return dz, None, None
    sf = sums.float()
    return dz, sf[C:], sf[:C]


def plane_dot(u, v, a=None, b=None, act=ACT_NONE):
    B, C = u.shape[0], u.shape[1]
    S = u.numel() // (B * C)
    out = torch.empty((B, C), device=u.device, dtype=torch.float32)
    _lib.call("eat_plane_dot", _dev(u, "u"), _dev(v, "v"), None if a is None else a.data_ptr(),
              None if b is None else b.data_ptr(), out.data_ptr(), B, C, S, act, _stream())
    return out


def dw_conv_dgrad(dz, w, x_shape, k, stride, res=None):
    B, C, F, T = x_shape
    Fo, To = dz.shape[2], dz.shape[3]
    dx = torch.empty(x_shape, device=dz.device, dtype=torch.float32)
    _lib.call("eat_dw_conv_dgrad", _dev(dz, "dz"), _dev(w, "w"), _opt(res, "res"), dx.data_ptr(), B, C, F, T, Fo, To,
              k, stride, _stream())
    return dx


def dw_conv_wgrad(dz, x, k, stride):
    B, C, Fo, To = dz.shape
    XC, F, T = x.shape[1], x.shape[2], x.shape[3]
    dw = zero_arena.zeros((C, k * k), torch.float32, dz.device)
    _lib.call("eat_dw_conv_wgrad", _dev(dz, "dz"), _dev(x, "x"), dw.data_ptr(), B, C, XC, F, T, Fo, To, k, stride,
              _stream())
    return dw


def dw_conv_wgrad_tf(dz, x, in_a, in_b, in_act, k, stride):
    """Depthwise weight gradient whose x operand is act_in(in_a[c] * x + in_b[c]), evaluated on load."""
    B, C, Fo, To = dz.shape
    F, T = x.shape[2], x.shape[3]
    dw = zero_arena.zeros((C, k * k), torch.float32, dz.device)
    _lib.call("eat_dw_conv_wgrad_tf", _dev(dz, "dz"), _dev(x, "x"), _dev(in_a, "in_a"), _dev(in_b, "in_b"), in_act,
              dw.data_ptr(), B, C, F, T, Fo, To, k, stride, _stream())
    return dw


def pw_tf_eligible(Ci, S):
    """Geometry of the on-load BatchNorm + activation of the 1x1 kernels (eat_pw_conv_tf_fwd / eat_pw_conv_wgrad_tf)."""
    return Ci % 8 == 0 and S % 4 == 0


def pw_conv_tf(x, tf, wp, bias, Co, act, in_scale=None, res=None):
    """1x1 conv of act_in(tf_a[k] x + tf_b[k]) [* in_scale] evaluated on load; tf = (a, b, act_in); wp from `pw_prepack`."""
    B, Ci, F, T = x.shape
    wmode = 0 if wp.dtype == torch.float32 else (2 if getattr(wp, "_eat_split", False) else 1)
    y = torch.empty((B, Co, F, T), device=x.device, dtype=torch.float32)
    _lib.call("eat_pw_conv_tf_fwd", _dev(x, "x"), tf[0].data_ptr(), tf[1].data_ptr(), tf[2], wp.data_ptr(), wmode,
              _dev(bias, "bias"), _opt(in_scale, "in_scale"), _opt(res, "res"), y.data_ptr(), B, Ci, Co, F * T, act,
              _stream())
    return y


def pw_conv_cat(x1, x2, wp, bias, Co, act, res=None):
    """1x1 conv over the concatenated channels of x1 (B,C1,F,T) and x2 (B,C2,F,T) without materialising the concatenation;
    wp = pw_prepack of the (Co, C1 + C2) matrix."""
    B, C1, F, T = x1.shape
    C2 = x2.shape[1]
    wmode = 0 if wp.dtype == torch.float32 else (2 if getattr(wp, "_eat_split", False) else 1)
    y = torch.empty((B, Co, F, T), device=x1.device, dtype=torch.float32)
    _lib.call("eat_pw_conv_cat_fwd", _dev(x1, "x"), C1, _dev(x2, "x2"), C2, wp.data_ptr(), wmode, _dev(bias, "bias"),
              _opt(res, "res"), y.data_ptr(), B, Co, F * T, act, _stream())
    return y


def gram(x, exact=False, sx=None, plain_bf16=False):
    """G (C, C) = sum_{b,s} x x^T, bit-reproducible from run to run: every block of the weight-gradient kernel adds into
    its own zeroed copy and the copies are summed in a fixed order (the BatchNorm statistics of the expand conv follow
    from G - csrc/train_fuse.hip - so its round-off decides on which side of a ReLU / Hardswish kink activations fall).
    sx (C,) = sum_{b,s} x: the CENTRED matrix Gc = sum (x - m)(x - m)^T, m = sx / n, instead (`centered=True` in
    `gram_bn_state*` / `expand_bwd_coef`): the variance w^T Gc w / n is then not a difference of two large sums.
    plain_bf16: single bf16 products (the bf16-storage plan: the conv output these statistics describe is itself stored
    with 2^-9 relative rounding; the centring happens BEFORE the rounding, the accumulation stays fp32)."""
    B, C = x.shape[0], x.shape[1]
    S = x.numel() // (B * C)
    mode = 1 if exact else (2 if plain_bf16 else 0)
    # (64 < C <= 160: the wide-tile kernel STORES its per-slice copies - no zero fill needed for them)
    stored = _lib.lib().eat_pw_wgrad_kernel_kind(B, C, C, S, mode, 1, 0, 0) == 3
    if sx is not None:
        slots = int(_lib.lib().eat_pw_wgrad_slots(B, C, C, S, mode, 1))
        G = zero_arena.zeros((C, C), torch.float32, x.device)
        n_ws = 2 * C + slots * C * C
        ws = torch.empty((n_ws,), dtype=torch.float32, device=x.device) if stored else zero_arena.zeros((n_ws,), torch.float32, x.device)
        _lib.call("eat_gram_centered", _dev(x, "x"), _dev(sx, "sx"), 1.0 / (B * S), G.data_ptr(), ws.data_ptr(), slots, B, C,
                  S, mode, _stream())
        return G
    slots = int(_lib.lib().eat_pw_wgrad_slots(B, C, C, S, mode, 1))
    G = zero_arena.zeros((C, C), torch.float32, x.device)
    ws = torch.empty((slots, C, C), dtype=torch.float32, device=x.device) if stored else zero_arena.zeros((slots, C, C), torch.float32, x.device)
    _lib.call("eat_pw_conv_wgrad_ws", _dev(x, "x"), _dev(x, "x"), None, G.data_ptr(), ws.data_ptr(), slots, B, C, C, S, mode,
              _stream())
    return G


def pw_conv_wgrad(dz, x, x_scale=None, exact=None, tf=None, out=None):
    """dW (Co, Ci) = sum_b dz[b] (Co,S) . (x[b] * x_scale[b])^T.  exact=True: fp32 MFMA kernel; False: split-operand
    bf16x3 kernel (fp32-class); None: follow the active `precision` context ('fp32' -> exact, 'bf16' -> plain bf16
    operands with fp32 accumulation, as autocast does to the conv weight gradient; otherwise bf16x3)."""
    mode = 1 if exact else 0
    if exact is None:
        mode = {"fp32": 1, "bf16": 2}.get(precision.mode, 0)
    B, Co = dz.shape[0], dz.shape[1]
    Ci = x.shape[1]
    S = dz.numel() // (B * Co)
    # out: ZERO-FILLED (Co, Ci) memory to accumulate into (dp.GradReducer.alloc: the gradient is produced inside its bucket)
    dW = out if out is not None else zero_arena.zeros((Co, Ci), torch.float32, dz.device)
    if tf is not None:
        ws = zero_arena.zeros((8, Co, Ci), torch.float32, dz.device) if (Co <= 64 and Ci <= 64 and S % 4 == 0 and mode != 1) else None
        _lib.call("eat_pw_conv_wgrad_tf", _dev(dz, "dz"), _dev(x, "x"), tf[0].data_ptr(), tf[1].data_ptr(), tf[2],
                  _opt(x_scale, "x_scale"), dW.data_ptr(), None if ws is None else ws.data_ptr(), 0 if ws is None else 8,
                  B, Co, Ci, S, mode, _stream())
        return dW
    if Co <= 64 and Ci <= 64 and S % 4 == 0 and mode != 1:
        # small matrices, long reductions: the streaming kernel spreads its atomics over 8 copies of dW (csrc/train.hip)
        ws = zero_arena.zeros((8, Co, Ci), torch.float32, dz.device)
        _lib.call("eat_pw_conv_wgrad_ws", _dev(dz, "dz"), _dev(x, "x"), _opt(x_scale, "x_scale"), dW.data_ptr(),
                  ws.data_ptr(), 8, B, Co, Ci, S, mode, _stream())
        return dW
    h = _lib.lib()
    if h.eat_pw_wgrad_kernel_kind(B, Co, Ci, S, mode, 1 if dz.data_ptr() == x.data_ptr() else 0, 0 if x_scale is None else 1, 0) == 3:
        # late-layer shapes: the wide-tile kernel stores one copy of dW per k-slice (no atomics, bit-reproducible); the copies
        # need no zero fill
        n = int(h.eat_pw_wgrad_slots(B, Co, Ci, S, mode, 0))
        ws = torch.empty((n, Co, Ci), device=dz.device, dtype=torch.float32)
        _lib.call("eat_pw_conv_wgrad_ws", _dev(dz, "dz"), _dev(x, "x"), _opt(x_scale, "x_scale"), dW.data_ptr(),
                  ws.data_ptr(), n, B, Co, Ci, S, mode, _stream())
        return dW
    _lib.call("eat_pw_conv_wgrad", _dev(dz, "dz"), _dev(x, "x"), _opt(x_scale, "x_scale"), dW.data_ptr(), B, Co, Ci,
              S, mode, _stream())
    return dW


# ------------------------------------------------------------------ round-3 training launchers (csrc/train_fuse.hip)
import ctypes as _ct


def dw_partials_inner(F, T, Fo, To, k, stride, dgrad):
    """Upper bound of the partial slots per plane the fused training epilogues write (host helper)."""
    return int(_lib.lib().eat_dw_partials_inner(F, T, Fo, To, k, stride, 1 if dgrad else 0))


def dw_conv_stats(x, w, k, stride, tf=None, out_b16=False):
    """Train-mode depthwise conv + the partial sums of its output for the BatchNorm that follows.
    tf = (in_a, in_b, in_act): the conv input is act(in_a[c] x + in_b[c]) evaluated on load.
    -> (y, (part, outer, inner)) for `bn_state_from_partials`.  A bf16 x - or out_b16 with an fp32 x - selects the
    bf16-storage kernels: y bf16, statistics of the stored values."""
    B, C, F, T = x.shape
    Fo, To = conv_out(F, k, stride), conv_out(T, k, stride)
    cap = dw_partials_inner(F, T, Fo, To, k, stride, False)
    x16 = _is16(x)
    y = torch.empty((B, C, Fo, To), device=x.device, dtype=torch.bfloat16 if (x16 or out_b16) else torch.float32)
    part = torch.empty((B * 2 * C * cap,), device=x.device, dtype=torch.float32)
    inner = _ct.c_int(0)
    a, b, act = tf if tf is not None else (None, None, 0)
    if x16 or out_b16:                             # bf16 storage: z_e (or the fp32 stem output) in, z_d out, statistics of the stored z_d
        _lib.call("eat_dw_conv_fwd_stats_b16", _dev16(x, "x") if x16 else _dev(x, "x"), 1 if x16 else 0, _opt(a, "in_a"),
                  _opt(b, "in_b"), act, _dev(w, "w"),
                  y.data_ptr(), part.data_ptr(), cap, _ct.addressof(inner), B, C, F, T, Fo, To, k, stride, _stream())
        return y, (part, B, inner.value)
    _lib.call("eat_dw_conv_fwd_stats", _dev(x, "x"), _opt(a, "in_a"), _opt(b, "in_b"), act, _dev(w, "w"), y.data_ptr(),
              part.data_ptr(), cap, _ct.addressof(inner), B, C, F, T, Fo, To, k, stride, _stream())
    return y, (part, B, inner.value)


def bn_stats_partial(z):
    B, C = z.shape[0], z.shape[1]
    part = torch.empty((B * 2 * C,), device=z.device, dtype=torch.float32)
    _lib.call("eat_bn_stats_partial", _dev(z, "z"), B, C, z.numel() // (B * C), part.data_ptr(), _stream())
    return part, B, 1


def _bn_momentum(bn):
    mom = bn.momentum
    if mom is None:
        mom = 1.0 / (int(bn.num_batches_tracked) + 1)          # host read: this mode is not graph-capturable
    return float(mom)


def bn_state_from_partials(parts, bn, n):
    """(a, b, mean, invstd) of a TRAINING BatchNorm layer from the producer's partial sums; updates the running buffers
    and num_batches_tracked as `bn_train_state` does."""
    part, outer, inner = parts
    C = bn.num_features
    out = torch.empty((4, C), device=part.device, dtype=torch.float32)
    nws = int(_lib.lib().eat_bn_finalize_ws_doubles(outer, C, inner))      # > 0: many partial rows, summed in groups first
    ws = torch.empty((nws,), device=part.device, dtype=torch.float64) if nws else None
    _lib.call("eat_bn_finalize_partials", part.data_ptr(), outer, C, inner, _dev(bn.weight, "gamma"), _dev(bn.bias, "beta"),
              bn.running_mean.data_ptr(), bn.running_var.data_ptr(), _bn_momentum(bn), float(bn.eps), float(n),
              out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(),
              None if ws is None else ws.data_ptr(), _stream())
    bn_counters.bump(bn)
    return out[0], out[1], out[2], out[3]


def gram_bn_state(Tm, W, sx, bn, n, centered=False):
    """(a, b, mean, invstd) of the TRAINING BatchNorm after the 1x1 conv z = W x from Tm = W G, sx (Gram matrix / sum of
    the conv input): see csrc/train_fuse.hip."""
    Co, Ci = W.shape
    out = torch.empty((4, Co), device=W.device, dtype=torch.float32)
    _lib.call("eat_gram_bn_finalize", _dev(Tm, "Tm"), _dev(W, "W"), _dev(sx, "sx"), Co, Ci, _dev(bn.weight, "gamma"),
              _dev(bn.bias, "beta"), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), _bn_momentum(bn),
              float(bn.eps), float(n), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(),
              1 if centered else 0, _stream())
    bn_counters.bump(bn)
    return out[0], out[1], out[2], out[3]


def gram_bn_state_g(G, W, sx, bn, n, centered=False):
    """`gram_bn_state` straight from the Gram matrix G: -> (Tm = W G, (a, b, mean, invstd)); one launch, fp64 inside."""
    Co, Ci = W.shape
    out = torch.empty((4 * Co + Co * Ci,), device=W.device, dtype=torch.float32)
    st, Tm = out[:4 * Co].view(4, Co), out[4 * Co:].view(Co, Ci)
    _lib.call("eat_gram_bn_finalize_g", _dev(G, "G"), _dev(W, "W"), _dev(sx, "sx"), Co, Ci, _dev(bn.weight, "gamma"),
              _dev(bn.bias, "beta"), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), _bn_momentum(bn),
              float(bn.eps), float(n), Tm.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(),
              st[3].data_ptr(), 1 if centered else 0, _stream())
    bn_counters.bump(bn)
    return Tm, (st[0], st[1], st[2], st[3])


def act_grad_sum(dy, z, a, b, act, inplace=False):
    """g = dy * act'(a[c] z + b[c]) and its per-plane sums -> (g, (gpart, B, 1))."""
    B, C = z.shape[0], z.shape[1]
    g = dy if inplace else torch.empty_like(dy)
    gpart = torch.empty((B * C,), device=z.device, dtype=torch.float32)
    _lib.call("eat_act_grad_sum", _dev(dy, "dy"), _dev(z, "z"), a.data_ptr(), b.data_ptr(), act, g.data_ptr(),
              gpart.data_ptr(), B, C, z.numel() // (B * C), _stream())
    return g, (gpart, B, 1)


def dw_conv_dgrad_g(dz, w, x_shape, k, stride, gz, ga, gb, gact):
    """Depthwise data gradient times act'(ga[c] gz + gb[c]) + its partial sums -> (g, (gpart, B, inner))."""
    B, C, F, T = x_shape
    Fo, To = dz.shape[2], dz.shape[3]
    cap = dw_partials_inner(F, T, Fo, To, k, stride, True)
    g = torch.empty(x_shape, device=dz.device, dtype=torch.float32)
    gpart = torch.empty((B * C * cap,), device=dz.device, dtype=torch.float32)
    inner = _ct.c_int(0)
    _lib.call("eat_dw_conv_dgrad_g", _dev(dz, "dz"), _dev(w, "w"), _dev(gz, "gz"), ga.data_ptr(), gb.data_ptr(), gact,
              g.data_ptr(), gpart.data_ptr(), cap, _ct.addressof(inner), B, C, F, T, Fo, To, k, stride, _stream())
    return g, (gpart, B, inner.value)


def dw_conv_bwd_g(dz, w, x, in_a, in_b, in_act, k, stride):
    """Merged depthwise backward: -> (g, (gpart, B, inner), dw) with g = dgrad(dz) * act'(in_a x + in_b) and dw the weight
    gradient w.r.t. the conv input act(in_a x + in_b); dz and x are read once (csrc/dw_plane.hip: dw_bwd_tile_kernel)."""
    B, C, F, T = x.shape
    Fo, To = dz.shape[2], dz.shape[3]
    h = _lib.lib()
    cap = max(int(h.eat_dw_bwd_partials_inner(F, T, Fo, To, k, stride)), dw_partials_inner(F, T, Fo, To, k, stride, True))
    g = torch.empty((B, C, F, T), device=dz.device, dtype=torch.float32)
    gpart = torch.empty((B * C * cap,), device=dz.device, dtype=torch.float32)
    dw = zero_arena.zeros((C, k * k), torch.float32, dz.device)
    inner = _ct.c_int(0)
    _lib.call("eat_dw_conv_bwd_g", _dev(dz, "dz"), _dev(x, "x"), in_a.data_ptr(), in_b.data_ptr(), in_act, _dev(w, "w"),
              g.data_ptr(), dw.data_ptr(), gpart.data_ptr(), cap, _ct.addressof(inner), B, C, F, T, Fo, To, k, stride,
              _stream())
    return g, (gpart, B, inner.value), dw


def dw_bwd_merged_ok(dy_shape, x_shape, k, stride):
    """True where the merged depthwise backward kernel runs (large planes): `dw_conv_bwd_bn_g` is available there."""
    B, C, F, T = x_shape
    return bool(_lib.lib().eat_dw_bwd_merged_ok(B, C, F, T, dy_shape[2], dy_shape[3], k, stride))


def bn_act_bwd_sums(dy, z, a, b, mean, invstd, act, gscale=None, gadd=None, se_P=None, sums=None):
    """The reduce half of `bn_act_bwd` / `bn_act_bwd_se`: -> (sums (2C,) float64, dgamma, dbeta); dz is left to the consumer
    (`dw_conv_bwd_bn_g` evaluates it on load)."""
    B, C = z.shape[0], z.shape[1]
    S = z.numel() // (B * C)
    own = sums is None
    if se_P is not None:
        if own:
            sums = torch.empty((2 * C,), device=z.device, dtype=torch.float64)
        _lib.call("eat_se_bn_bwd_combine", se_P.data_ptr(), _dev(gscale, "gscale"), _dev(gadd, "gadd"), invstd.data_ptr(), B,
                  C, sums.data_ptr(), _stream())
    else:
        if own:
            sums = zero_arena.zeros((2 * C,), torch.float64, z.device)
        b16 = _is16(z)
        tail = (a.data_ptr(), b.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _opt(gscale, "gscale"), _opt(gadd, "gadd"), B, C, S,
                act, sums.data_ptr(), _stream())
        if b16:
            _lib.call("eat_bn_act_bwd_reduce_b16", _dev16(dy, "dy"), 1, _dev16(z, "z"), *tail)
        else:
            _lib.call("eat_bn_act_bwd_reduce", _dev(dy, "dy"), _dev(z, "z"), *tail)
    if not own:
        return sums, None, None
    sf = sums.float()
    return sums, sf[C:], sf[:C]


def dw_conv_bwd_bn_g(dy, z, st, act, sums, w, x, in_a, in_b, in_act, k, stride, gscale=None, gadd=None, want_gsum=True):
    """Merged depthwise backward with the BatchNorm + activation backward of the conv's own output evaluated on load from
    (dy, z): -> (g, (gpart, B, inner) or None, dw).  st = the forward's (a, b, mean, invstd); only where `dw_bwd_merged_ok`."""
    B, C, F, T = x.shape
    Fo, To = z.shape[2], z.shape[3]
    h = _lib.lib()
    cap = int(h.eat_dw_bwd_partials_inner(F, T, Fo, To, k, stride))
    b16 = _is16(z)                                 # bf16 storage: dy and z bf16; x and g bf16 (or both fp32: the first block)
    g = torch.empty((B, C, F, T), device=z.device, dtype=x.dtype if b16 else torch.float32)
    gpart = torch.empty((B * C * cap,), device=z.device, dtype=torch.float32) if want_gsum else None
    dw = zero_arena.zeros((C, k * k), torch.float32, z.device)
    inner = _ct.c_int(0)
    frozen = 1 if getattr(st[2], "_eat_frozen", False) else 0
    if b16:
        _lib.call("eat_dw_conv_bwd_bn_g_b16", _dev16(dy, "dy"), _dev16(z, "z"), st[0].data_ptr(), st[1].data_ptr(),
                  st[2].data_ptr(), st[3].data_ptr(), _opt(gscale, "gscale"), _opt(gadd, "gadd"), sums.data_ptr(), act, frozen,
                  _dev16(x, "x") if _is16(x) else _dev(x, "x"), 1 if _is16(x) else 0, in_a.data_ptr(), in_b.data_ptr(), in_act,
                  _dev(w, "w"), g.data_ptr(), dw.data_ptr(),
                  None if gpart is None else gpart.data_ptr(), cap, _ct.addressof(inner), B, C, F, T, Fo, To, k, stride,
                  _stream())
        return g, ((gpart, B, inner.value) if want_gsum else None), dw
    _lib.call("eat_dw_conv_bwd_bn_g", _dev(dy, "dy"), _dev(z, "z"), st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(),
              st[3].data_ptr(), _opt(gscale, "gscale"), _opt(gadd, "gadd"), sums.data_ptr(), act, frozen, _dev(x, "x"),
              in_a.data_ptr(), in_b.data_ptr(), in_act, _dev(w, "w"), g.data_ptr(), dw.data_ptr(),
              None if gpart is None else gpart.data_ptr(), cap, _ct.addressof(inner), B, C, F, T, Fo, To, k, stride,
              _stream())
    return g, ((gpart, B, inner.value) if want_gsum else None), dw


def se_mlp_bwd(ds, scale, h, pool, W1, W2, S, dW1_out=None, dW2_out=None):
    """Backward of the SE gate MLP in two launches -> (dW1, db1, dW2, db2, gadd); see csrc/se_train.hip."""
    B, C = ds.shape
    Cr = h.shape[1]
    dev = ds.device
    n_dh = int(_lib.lib().eat_se_mlp_dh_floats(B, C, Cr))            # dh + its k slices where the contraction is split
    buf = torch.empty((2 * C * Cr + C + Cr + n_dh + B * C,), device=dev, dtype=torch.float32)
    o = 0
    dW1 = buf[o:o + Cr * C].view(Cr, C); o += Cr * C
    dW2 = buf[o:o + C * Cr].view(C, Cr); o += C * Cr
    dW1 = dW1_out if dW1_out is not None else dW1            # (written, not accumulated: memory of the gradient buckets)
    dW2 = dW2_out if dW2_out is not None else dW2
    db1 = buf[o:o + Cr]; o += Cr
    db2 = buf[o:o + C]; o += C
    dh = buf[o:o + n_dh]; o += n_dh
    gadd = buf[o:o + B * C].view(B, C)
    _lib.call("eat_se_mlp_bwd", _dev(ds, "ds"), _dev(scale, "scale"), _dev(h, "h"), _dev(pool, "pool"), _dev(W1, "W1"),
              _dev(W2, "W2"), 1.0 / S, dW1.data_ptr(), db1.data_ptr(), dW2.data_ptr(), db2.data_ptr(), dh.data_ptr(),
              gadd.data_ptr(), B, C, Cr, _stream())
    return dW1, db1, dW2, db2, gadd


def mlp_head_bwd(dlogits, h2, u, drop_mask, feat, W1, W2, dW1_out=None, dW2_out=None):
    """Backward of the classifier head Linear -> Hardswish -> Dropout -> Linear in two launches (csrc/se_train.hip):
    -> (dW1, db1, dW2, db2, dfeat)."""
    B, N = dlogits.shape
    H, C = W1.shape
    dev = dlogits.device
    n_df = int(_lib.lib().eat_mlp_head_dfeat_floats(B, C, H))       # dfeat + its k slices where the contraction is split
    buf = torch.empty((H * C + H + N * H + N + B * H + n_df,), device=dev, dtype=torch.float32)
    o = 0
    dW1 = buf[o:o + H * C].view(H, C); o += H * C
    db1 = buf[o:o + H]; o += H
    dW2 = buf[o:o + N * H].view(N, H); o += N * H
    dW1 = dW1_out if dW1_out is not None else dW1            # (written, not accumulated: memory of the gradient buckets)
    dW2 = dW2_out if dW2_out is not None else dW2
    db2 = buf[o:o + N]; o += N
    du = buf[o:o + B * H]; o += B * H
    dfeat = buf[o:o + B * C].view(B, C)
    _lib.call("eat_mlp_head_bwd", _dev(dlogits, "dy"), _dev(h2, "h2"), _dev(u, "u"), _opt(drop_mask, "drop_mask"),
              _dev(feat, "feat"), _dev(W1, "W1"), _dev(W2, "W2"), dW1.data_ptr(), db1.data_ptr(), dW2.data_ptr(),
              db2.data_ptr(), du.data_ptr(), dfeat.data_ptr(), B, C, H, N, _stream())
    return dW1, db1, dW2, db2, dfeat


def stem_gram(x, W):
    """(Tm = W G9 (C, 9), sp (9)) of the stem conv's 3x3 / stride-2 patches of the log-mel x (B, 1, F, T): feeds
    `gram_bn_state` (csrc/stem_train.hip).  Bit-reproducible."""
    B, _, F, T = x.shape
    C = W.shape[0]
    Fo = (F - 1) // 2 + 1
    nblk = int(_lib.lib().eat_stem_gram_blocks(B, Fo))
    part = torch.empty((nblk, 54), device=x.device, dtype=torch.float32)
    out = torch.empty((C * 9 + 9,), device=x.device, dtype=torch.float32)
    Tm, sp = out[:C * 9].view(C, 9), out[C * 9:]
    _lib.call("eat_stem_gram", _dev(x, "x"), _dev(W, "W"), part.data_ptr(), Tm.data_ptr(), sp.data_ptr(), B, C, F, T,
              _stream())
    return Tm, sp


def stem_bwd(dy, x, W, a, b, act, dy2=None):
    """-> (Gx (C, 9) = sum g p^T, (s1, 1, 1)) with g = dy * act'(a (W p) + b): the stem's weight-gradient pass with the
    BatchNorm + activation backward folded in (`expand_bwd_coef` finishes)."""
    B, _, F, T = x.shape
    C = W.shape[0]
    buf = torch.empty((C * 10,), device=x.device, dtype=torch.float32)
    gx, s1 = buf[:C * 9].view(C, 9), buf[C * 9:]
    nblk = int(_lib.lib().eat_stem_bwd_blocks(B, (F - 1) // 2 + 1))
    part = torch.empty((nblk, C, 10), device=x.device, dtype=torch.float32)
    _lib.call("eat_stem_bwd", _dev(dy, "dy"), _opt(dy2, "dy2"), _dev(x, "x"), _dev(W, "W"), a.data_ptr(), b.data_ptr(), act,
              part.data_ptr(), gx.data_ptr(), s1.data_ptr(), B, C, F, T, _stream())
    return gx, (s1, 1, 1)


def expand_bwd_coef(W, Gx, Tm, sx, gparts, a, mean, invstd, n, frozen=False, need_dx=True, centered=False, dW_out=None):
    """-> (dW, dgamma, dbeta, WaT (Ci,Co), M (Ci,Ci), c0 (Ci)): backward of conv1x1 -> BN(train) -> act without dz."""
    gpart, outer, inner = gparts
    Co, Ci = W.shape
    dev = W.device
    dW = dW_out if dW_out is not None else torch.empty((Co, Ci), device=dev, dtype=torch.float32)   # (written, not accumulated)
    vec = torch.empty((2, Co), device=dev, dtype=torch.float32)           # dgamma, dbeta
    tr = torch.empty((3 * Ci + 1, Co), device=dev, dtype=torch.float32)   # WaT, WT, [W2T ; e1]: M and c0 from ONE GEMM
    w2e = tr[2 * Ci:]
    tr = (tr[:Ci], tr[Ci:2 * Ci], w2e[:Ci])
    _lib.call("eat_expand_bwd_coef", _dev(W, "W"), _dev(Gx, "Gx"), _dev(Tm, "Tm"), _dev(sx, "sx"), gpart.data_ptr(),
              outer, inner, Co, Ci, a.data_ptr(), mean.data_ptr(), invstd.data_ptr(), float(n), 1 if frozen else 0,
              dW.data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(), tr[0].data_ptr(), tr[1].data_ptr(), tr[2].data_ptr(),
              w2e[Ci].data_ptr(), 1 if centered else 0, None, _stream())
    if frozen or not need_dx:
        return dW, vec[0], vec[1], tr[0], None, None
    if precision.mode != "fp32" and Ci % 4 == 0 and Co % 4 == 0 and W.is_contiguous():
        # [W2T ; e1] . W as the 1x1 conv of the "image" W (1, Co, Ci, 1) with the (Ci + 1, Co) matrix on the split-operand
        # bf16x3 kernel (fp32-class products; round 5: the K-split `linear` kernel ran these C_in x C_in x C_exp products at
        # 15 TFLOP/s - 14 launches of up to 213 us in the mn40 step)
        wp3 = pw_prepack_bf16(w2e, None, split=True)
        Mc = pw_conv_bf16(W.view(1, Co, Ci, 1), wp3, _zero_bias(Ci + 1, dev), Ci + 1, ACT_NONE, split=True).view(Ci + 1, Ci)
    else:
        Mc = linear(w2e, tr[1], None, ACT_NONE)                            # [W2T ; e1] . WT^T -> (Ci + 1, Ci)
    return dW, vec[0], vec[1], tr[0], Mc[:Ci], Mc[Ci]


def cat_pack_kind(K):
    """The pack `pw_prepack` would choose for a matrix with K reduction channels under the active precision
    (0 fp32 fragments, 1 bf16, 2 bf16 hi + lo)."""
    m = precision.mode
    if m == "bf16":
        return 1
    if m == "bf16x3" or (m == "auto" and K >= 40 and K % 4 == 0):
        return 2
    return 0


def expand_bwd_coef_cat(W, Gx, Tm, sx, gparts, a, mean, invstd, n, centered=False, dW_out=None):
    """`expand_bwd_coef` for the two-source data-gradient GEMM: -> (dW, dgamma, dbeta, wcat, c0) with wcat = the weight pack
    of [WaT | M] (Ci x (Co + Ci)) and c0 (Ci) written by ONE launch (csrc/train_fuse.hip: eat_expand_bwd_wcat) - no
    transposes, no separate M GEMM, no torch.cat, no prepack launches.  Training BatchNorm (not frozen) only."""
    gpart, outer, inner = gparts
    Co, Ci = W.shape
    dev = W.device
    dW = dW_out if dW_out is not None else torch.empty((Co, Ci), device=dev, dtype=torch.float32)
    vec = torch.empty((4, Co), device=dev, dtype=torch.float32)           # dgamma, dbeta, e1, e2
    _lib.call("eat_expand_bwd_coef", _dev(W, "W"), _dev(Gx, "Gx"), _dev(Tm, "Tm"), _dev(sx, "sx"), gpart.data_ptr(),
              outer, inner, Co, Ci, a.data_ptr(), mean.data_ptr(), invstd.data_ptr(), float(n), 0,
              dW.data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(), None, None, None, vec[2].data_ptr(),
              1 if centered else 0, vec[3].data_ptr(), _stream())
    kind = cat_pack_kind(Co + Ci)
    nel = int(_lib.lib().eat_expand_bwd_wcat_elems(Co, Ci, kind))
    wcat = torch.empty((nel,), device=dev, dtype=torch.float32 if kind == 0 else torch.bfloat16)
    if kind == 2:
        wcat._eat_split = True
    c0 = torch.empty((Ci,), device=dev, dtype=torch.float32)
    _lib.call("eat_expand_bwd_wcat", _dev(W, "W"), a.data_ptr(), vec[3].data_ptr(), vec[2].data_ptr(), Co, Ci, kind,
              wcat.data_ptr(), c0.data_ptr(), _stream())
    return dW, vec[0], vec[1], wcat, c0


# ------------------------------------------------------------------------- DyMN launchers
def ctx_pool(x):
    B, C, F, T = x.shape
    seq = torch.empty((B, F + T, C), device=x.device, dtype=torch.float32)
    _lib.call("eat_ctx_pool", _dev(x, "x"), seq.data_ptr(), B, C, F, T, _stream())
    return seq


def dyn_aggregate(bank, att, gscale=None, group=1):
    K, N = bank.shape
    B = att.shape[0]
    out = torch.empty((B, N), device=bank.device, dtype=torch.float32)
    _lib.call("eat_dyn_aggregate", _dev(bank, "bank"), _dev(att, "att"), _opt(gscale, "gscale"), out.data_ptr(), B, K,
              N, group, _stream())
    return out


def dyn_pw_pack(bank, att, Co, Ci, row_scale=None, trans=False):
    """trans: `bank` (K, Ci*Co) stores the TRANSPOSED matrices (the data-gradient pack without a transposed bank copy)."""
    K, B = bank.shape[0], att.shape[0]
    wp = torch.empty((B, (Ci // 4) * ((Co + 15) // 16) * 64), device=bank.device, dtype=torch.float32)
    _lib.call("eat_dyn_pw_pack_t" if trans else "eat_dyn_pw_pack", _dev(bank, "bank"), _dev(att, "att"),
              _opt(row_scale, "row_scale"), wp.data_ptr(), B, K, Co, Ci, _stream())
    return wp


def pw_conv_dyn(x, wp_b, bias, Co, act, res=None):
    B, Ci, F, T = x.shape
    y = torch.empty((B, Co, F, T), device=x.device, dtype=torch.float32)
    _lib.call("eat_pw_conv_dyn_fwd", _dev(x, "x"), _dev(wp_b, "wp_b"), _dev(bias, "bias"), _opt(res, "res"),
              y.data_ptr(), B, Ci, Co, F * T, act, _stream())
    return y


def dyn_bf16_eligible(Co, Ci, S):
    """Per-sample-weight dynamic 1x1 conv on split bf16 operands (bf16x3) - the rule `pw_prepack` applies to the static
    convs ('auto': from C_in = 40 on; 'bf16x3': everywhere; never under 'fp32' / plain 'bf16')."""
    m = precision.mode
    return S % 4 == 0 and Ci % 4 == 0 and (m == "bf16x3" or (m == "auto" and Ci >= 40))


def dyn_wgrad_needs_zero(Co, Ci, S):
    """True where `eat_pw_conv_dyn_wgrad` accumulates its per-sample output with atomics (the exact-fp32 tile kernel); the
    bf16x3 kernel covers a sample's whole reduction in one block and stores."""
    return bool(_lib.lib().eat_pw_dyn_wgrad_accumulates(Co, Ci, S))


def dyn_pw_pack_bf16(bank, att, Co, Ci, trans=False):
    """Aggregated per-sample weights sum_k att[b,k] bank[k] as bf16 hi / lo MFMA fragments (the per-sample form of
    `pw_prepack_bf16(split=True)`) -> (B, KK*MT*2*512) bfloat16.  trans: as in `dyn_pw_pack`."""
    K, B = bank.shape[0], att.shape[0]
    n = ((Ci + 31) // 32) * ((Co + 15) // 16) * 2 * 512
    wp = torch.empty((B, n), device=bank.device, dtype=torch.bfloat16)
    _lib.call("eat_dyn_pw_pack_bf16_t" if trans else "eat_dyn_pw_pack_bf16", _dev(bank, "bank"), _dev(att, "att"),
              wp.data_ptr(), B, K, Co, Ci, _stream())
    return wp


def pw_conv_dyn_bf16(x, wp_b, bias, Co, act, res=None):
    """Per-sample-weight 1x1 conv on the bf16x3 kernel (csrc/conv_pw_bf16.hip, tiles inside one sample)."""
    B, Ci, F, T = x.shape
    y = torch.empty((B, Co, F, T), device=x.device, dtype=torch.float32)
    _lib.call("eat_pw_conv_dyn_bf16_fwd", _dev(x, "x"), wp_b.data_ptr(), _dev(bias, "bias"), _opt(res, "res"),
              y.data_ptr(), B, Ci, Co, F * T, act, _stream())
    return y


def kcat_eligible(Co, Ci, S):
    """K-concat form of a dynamic 1x1 conv (no per-sample weights) pays where a sample's aggregated weight matrix is
    larger than its activations: the late, small-plane layers.  (Ci % 32: a k-chunk must not straddle two banks.)"""
    # (the K-concat kernel computes on split bf16x3 operands: under precision('fp32') the per-sample path with the exact
    #  fp32 MFMA is kept, so that `train_precision='fp32'` means what it says for every layer)
    return _KCAT and precision.mode != "fp32" and Ci % 32 == 0 and S % 4 == 0 and Co * Ci > (Ci + Co) * S


def kcat_pack(bank, Co, Ci, row_scale=None):
    """bank (K, Co*Ci) -> packed bf16 hi/lo fragments of [W_0 | ... | W_{K-1}]  (Co x K*Ci).
    (Not cached across training steps: fused optimizers and hipGraph replays update the bank without moving its version
    counter, so a version-keyed cache would serve stale weights; the eval plan caches its packs with the folded weights.)"""
    K = bank.shape[0]
    wcat = bank.view(K, Co, Ci).permute(1, 0, 2).reshape(Co, K * Ci).contiguous()
    return pw_prepack_bf16(wcat, row_scale, split=True)


def pw_conv_kcat(x, wp_cat, bias, att, Co, act, res=None):
    """Dynamic 1x1 conv as one GEMM over the concatenated banks; att (B, K) is the attention of the sample."""
    B, Ci, F, T = x.shape
    K = att.shape[1]
    scale = att.repeat_interleave(Ci, dim=1).contiguous()                  # (B, K*Ci)
    y = torch.empty((B, Co, F, T), device=x.device, dtype=torch.float32)
    _lib.call("eat_pw_conv_kcat_fwd", _dev(x, "x"), wp_cat.data_ptr(), _dev(bias, "bias"), _dev(scale, "att_scale"),
              _opt(res, "res"), y.data_ptr(), B, Ci, K, Co, F * T, act, _stream())
    return y


def dw_conv_dyn(x, w_bc, bias, coef, gate_f, gate_t, k, stride):
    B, C, F, T = x.shape
    Fo, To = conv_out(F, k, stride), conv_out(T, k, stride)
    y = torch.empty((B, C, Fo, To), device=x.device, dtype=torch.float32)
    _lib.call("eat_dw_conv_dyn_fwd", _dev(x, "x"), _dev(w_bc, "w_bc"), _dev(bias, "bias"), _dev(coef, "coef"),
              _dev(gate_f, "gate_f"), _dev(gate_t, "gate_t"), y.data_ptr(), B, C, F, T, Fo, To, k, stride, _stream())
    return y


def dw_conv_dyn_act(x, w_bc, bias, act, coef, gate_f, gate_t, k, stride):
    """dw_conv_dyn for the block's ablations: plain activation `act` instead of / before DyReLU-B (coef None), no
    coordinate attention (gates None)."""
    B, C, F, T = x.shape
    Fo, To = conv_out(F, k, stride), conv_out(T, k, stride)
    y = torch.empty((B, C, Fo, To), device=x.device, dtype=torch.float32)
    _lib.call("eat_dw_conv_dyn_act_fwd", _dev(x, "x"), _dev(w_bc, "w_bc"), _dev(bias, "bias"), act, _opt(coef, "coef"),
              _opt(gate_f, "gate_f"), _opt(gate_t, "gate_t"), y.data_ptr(), B, C, F, T, Fo, To, k, stride, _stream())
    return y


def pw_conv_stats(x, wp, Co, per_sample=False, tf=None, in_scale=None):
    """Train-mode 1x1 conv z = W x with the batch statistics of z in the conv's epilogue: -> (z, (part, outer, inner)) for
    `bn_state_from_partials`, or (z, None) where the epilogue does not exist (the caller runs `bn_stats`).  wp: any pack of
    `pw_prepack` / `dyn_pw_pack*` (per_sample=True: one pack per sample); tf = (a, b, act): input transform on load."""
    B, Ci, F, T = x.shape
    S = F * T
    wmode = 0 if wp.dtype == torch.float32 else (2 if (per_sample or getattr(wp, "_eat_split", False)) else 1)   # (per-sample bf16 packs are hi / lo)
    tiles = int(_lib.lib().eat_pw_conv_stat_tiles(B, S, 1 if per_sample else 0))
    y = torch.empty((B, Co, F, T), device=x.device, dtype=torch.float32)
    part = torch.empty((tiles * 2 * Co,), device=x.device, dtype=torch.float32)
    a, b, act = tf if tf is not None else (None, None, 0)
    rc = _lib.call_rc("eat_pw_conv_stats_fwd", _dev(x, "x"), wp.data_ptr(), wmode, 1 if per_sample else 0, _opt(a, "tf_a"),
                      _opt(b, "tf_b"), act, _opt(in_scale, "in_scale"), _zero_bias(Co, x.device).data_ptr(), y.data_ptr(),
                      part.data_ptr(), B, Ci, Co, S, _stream())
    if rc == 1:
        return None, None
    return y, (part, tiles, 1)


def pw_conv_gstats(x, wp, Co, z, st, act, sums=None):
    """Project conv's data-gradient GEMM y = W^T x (wp: transposed pack, fp32 or split bf16) with the BatchNorm + activation
    backward statistics of the depthwise output `z` (st = its (a, b, mean, invstd), act its activation) in the epilogue:
    -> (y, sums (2C,) float64) - what `bn_act_bwd_sums(y, z, *st, act)` returns, without the pass over (y, z) - or (None, None)
    where the epilogue does not exist (S % 4 != 0, plain-bf16 / fp32-free packs: the caller runs the separate pass)."""
    B, Ci, F, T = x.shape
    S = F * T
    if wp.dtype == torch.float32:
        wmode = 0
    elif getattr(wp, "_eat_split", False):
        wmode = 2
    else:
        return None, None
    if S % 4 != 0 or z.dtype != torch.float32:
        return None, None
    a, b, mean, invstd = st
    tiles = int(_lib.lib().eat_pw_conv_stat_tiles(B, S, 0))
    y = torch.empty((B, Co, F, T), device=x.device, dtype=torch.float32)
    part = torch.empty((tiles * 2 * Co,), device=x.device, dtype=torch.float32)
    rc = _lib.call_rc("eat_pw_conv_gstats_fwd", _dev(x, "x"), wp.data_ptr(), wmode, _zero_bias(Co, x.device).data_ptr(),
                      y.data_ptr(), _dev(z, "z"), a.data_ptr(), b.data_ptr(), act, part.data_ptr(), B, Ci, Co, S, _stream())
    if rc == 1:
        return None, None
    if sums is None:
        sums = torch.empty((2 * Co,), device=x.device, dtype=torch.float64)
    nws = int(_lib.lib().eat_bn_bwd_sums_ws_doubles(tiles, Co))
    ws = torch.empty((nws,), device=x.device, dtype=torch.float64) if nws else None
    _lib.call("eat_bn_bwd_sums_from_tiles", part.data_ptr(), tiles, Co, mean.data_ptr(), invstd.data_ptr(), a.data_ptr(),
              b.data_ptr(), None if ws is None else ws.data_ptr(), sums.data_ptr(), _stream())
    return y, sums


_zb = {}


def _zero_bias(n, device):
    buf = _zb.get(device)
    if buf is None or buf.numel() < n:
        buf = _zb[device] = torch.zeros((max(n, 4096),), device=device, dtype=torch.float32)
    return buf[:n]


# ---- round 4: fused training passes of the dynamic block (csrc/dymn.hip, csrc/dw_plane.hip)
def dw_conv_dyn_stats(x, w_bc, k, stride, tf=None, out_b16=False):
    """`dw_conv_stats` with per-(b,c) taps w_bc (B, C*k*k): -> (y, (part, outer, inner)).  A bf16 x - or out_b16 with an fp32 x
    (the block without expand conv) - selects the bf16-storage kernels: y bf16, statistics of the stored values."""
    B, C, F, T = x.shape
    Fo, To = conv_out(F, k, stride), conv_out(T, k, stride)
    cap = dw_partials_inner(F, T, Fo, To, k, stride, False)
    x16 = _is16(x)
    y = torch.empty((B, C, Fo, To), device=x.device, dtype=torch.bfloat16 if (x16 or out_b16) else torch.float32)
    part = torch.empty((B * 2 * C * cap,), device=x.device, dtype=torch.float32)
    inner = _ct.c_int(0)
    a, b, act = tf if tf is not None else (None, None, 0)
    if x16 or out_b16:
        _lib.call("eat_dw_conv_dyn_fwd_stats_b16", _dev16(x, "x") if x16 else _dev(x, "x"), 1 if x16 else 0, _opt(a, "in_a"),
                  _opt(b, "in_b"), act, _dev(w_bc, "w_bc"), y.data_ptr(), part.data_ptr(), cap, _ct.addressof(inner), B, C, F, T,
                  Fo, To, k, stride, _stream())
        return y, (part, B, inner.value)
    _lib.call("eat_dw_conv_dyn_fwd_stats", _dev(x, "x"), _opt(a, "in_a"), _opt(b, "in_b"), act, _dev(w_bc, "w_bc"),
              y.data_ptr(), part.data_ptr(), cap, _ct.addressof(inner), B, C, F, T, Fo, To, k, stride, _stream())
    return y, (part, B, inner.value)


def ctx_pool_cm(x):
    """ContextGen's two average pools, channel-major: x (B, C, F, T) -> seq (1, C, B*(F+T), 1) = [c][b][l]."""
    B, C, F, T = x.shape
    seq = torch.empty((1, C, B * (F + T), 1), device=x.device, dtype=torch.float32)
    _lib.call("eat_ctx_pool_cm", _dev(x, "x"), seq.data_ptr(), B, C, F, T, _stream())
    return seq


def ctx_pool_cm_bwd(dseq, shape, add=None):
    B, C, F, T = shape
    dx = torch.empty(shape, device=dseq.device, dtype=torch.float32)
    _lib.call("eat_ctx_pool_cm_bwd", _dev(dseq, "dseq"), _opt(add, "add"), dx.data_ptr(), B, C, F, T, _stream())
    return dx


def ctx_split(g, B, F, T, stride):
    """g (1, H, B*(F+T), 1) -> (h_cf (1, H, B*Fo, 1), h_ct (1, H, B*To, 1), h_c (B, H))."""
    H = g.shape[1]
    Fo, To = (F - 1) // stride + 1, (T - 1) // stride + 1
    hcf = torch.empty((1, H, B * Fo, 1), device=g.device, dtype=torch.float32)
    hct = torch.empty((1, H, B * To, 1), device=g.device, dtype=torch.float32)
    hc = torch.empty((B, H), device=g.device, dtype=torch.float32)
    _lib.call("eat_ctx_split", _dev(g, "g"), hcf.data_ptr(), hct.data_ptr(), hc.data_ptr(), H, B, F, T, stride, _stream())
    return hcf, hct, hc


def ctx_split_bwd(dhcf, dhct, dhc, H, B, F, T, stride):
    dg = torch.empty((1, H, B * (F + T), 1), device=dhcf.device, dtype=torch.float32)
    _lib.call("eat_ctx_split_bwd", _dev(dhcf, "dhcf"), _dev(dhct, "dhct"), _opt(dhc, "dhc"), dg.data_ptr(), H, B, F, T, stride,
              _stream())
    return dg


def dyrelu_ca_fwd2(z, a, b, coef, gate_f, gate_t):
    """gate_f (C, B, Fo) / gate_t (C, B, To): channel-major pre-sigmoid gates (any shape with that memory layout)."""
    B, C, Fo, To = z.shape
    out = torch.empty_like(z)
    z16 = _is16(z)                                  # bf16 storage: z and out (rounded on store) are bf16
    _lib.call("eat_dyrelu_ca_fwd2_b16" if z16 else "eat_dyrelu_ca_fwd2", _dev16(z, "z") if z16 else _dev(z, "z"), _opt(a, "a"),
              _opt(b, "b"), _dev(coef, "coef"), _dev(gate_f, "gate_f"), _dev(gate_t, "gate_t"), out.data_ptr(), B, C, Fo, To,
              _stream())
    return out


def dyrelu_ca_bwd2(dout, z, a, b, coef, gate_f, gate_t, want_bn=True):
    """-> (dv, dcoef (B,C,4), dgate_f, dgate_t (layouts of the gates), bnpart (B,C,2) or None)."""
    B, C, Fo, To = z.shape
    dv = torch.empty_like(z)
    buf = torch.empty((B * C * (4 + (2 if want_bn else 0)),), device=z.device, dtype=torch.float32)
    dcoef = buf[:B * C * 4].view(B, C, 4)
    bnpart = buf[B * C * 4:].view(B, C, 2) if want_bn else None
    dgf, dgt = torch.empty_like(gate_f), torch.empty_like(gate_t)
    z16 = _is16(z)                                  # bf16 storage: dout, z and dv are bf16 (bnpart: sums of dv as stored)
    if z16 != _is16(dout):
        raise _lib.EatHipError("dyrelu_ca_bwd2: dout and z share one storage type")
    _lib.call("eat_dyrelu_ca_bwd2_b16" if z16 else "eat_dyrelu_ca_bwd2", _dev16(dout, "dout") if z16 else _dev(dout, "dout"),
              _dev16(z, "z") if z16 else _dev(z, "z"), _opt(a, "a"), _opt(b, "b"), _dev(coef, "coef"),
              _dev(gate_f, "gate_f"), _dev(gate_t, "gate_t"), dv.data_ptr(), dcoef.data_ptr(), dgf.data_ptr(), dgt.data_ptr(),
              None if bnpart is None else bnpart.data_ptr(), B, C, Fo, To, _stream())
    return dv, dcoef, dgf, dgt, bnpart


def bn_bwd_combine_partials(p0, p1, stride_e, B, C, inner, mean, invstd):
    """Channel sums of a BatchNorm backward from per-plane partials -> (sums (2C,) float64, dgamma, dbeta)."""
    sums = torch.empty((2 * C,), device=mean.device, dtype=torch.float64)
    vec = torch.empty((2, C), device=mean.device, dtype=torch.float32)
    _lib.call("eat_bn_bwd_combine_partials", p0.data_ptr(), p1.data_ptr(), stride_e, B, C, inner, mean.data_ptr(),
              invstd.data_ptr(), sums.data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(), _stream())
    return sums, vec[0], vec[1]


def dw_conv_dyn_bwd_bn_g(dy, z, st, bn_act, sums, w_bc, x, in_a, in_b, in_act, k, stride, res=None, want_sums=True):
    """Merged backward of a dynamic depthwise conv with its own BatchNorm backward on load (`dw_conv_bwd_bn_g` with
    per-plane taps): -> (g, dw_bc (B, C*k*k), (gpart, gzpart, inner) or None)."""
    B, C, F, T = x.shape
    Fo, To = z.shape[2], z.shape[3]
    cap = int(_lib.lib().eat_dw_bwd_partials_inner(F, T, Fo, To, k, stride))
    g = torch.empty((B, C, F, T), device=z.device, dtype=x.dtype)
    parts = torch.empty((2, B * C * cap), device=z.device, dtype=torch.float32) if want_sums else None
    dw = zero_arena.zeros((B, C * k * k), torch.float32, z.device)
    inner = _ct.c_int(0)
    frozen = 1 if getattr(st[2], "_eat_frozen", False) else 0
    if _is16(z):                                    # bf16 storage: dy, z bf16; x and g bf16, or fp32 (block without expand conv)
        x16 = _is16(x)
        _lib.call("eat_dw_conv_dyn_bwd_bn_g_b16", _dev16(dy, "dy"), _dev16(z, "z"), st[0].data_ptr(), st[1].data_ptr(),
                  st[2].data_ptr(), st[3].data_ptr(), sums.data_ptr(), bn_act, frozen, _dev16(x, "x") if x16 else _dev(x, "x"),
                  1 if x16 else 0, in_a.data_ptr(), in_b.data_ptr(), in_act, _dev(w_bc, "w_bc"), _opt(res, "res"), g.data_ptr(),
                  dw.data_ptr(), None if parts is None else parts[0].data_ptr(), None if parts is None else parts[1].data_ptr(),
                  cap, _ct.addressof(inner), B, C, F, T, Fo, To, k, stride, _stream())
        return g, dw, ((parts[0], parts[1], inner.value) if want_sums else None)
    _lib.call("eat_dw_conv_dyn_bwd_bn_g", _dev(dy, "dy"), _dev(z, "z"), st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(),
              st[3].data_ptr(), sums.data_ptr(), bn_act, frozen, _dev(x, "x"), in_a.data_ptr(), in_b.data_ptr(), in_act,
              _dev(w_bc, "w_bc"), _opt(res, "res"), g.data_ptr(), dw.data_ptr(),
              None if parts is None else parts[0].data_ptr(), None if parts is None else parts[1].data_ptr(), cap,
              _ct.addressof(inner), B, C, F, T, Fo, To, k, stride, _stream())
    return g, dw, ((parts[0], parts[1], inner.value) if want_sums else None)


def bn_bwd_apply(g, z, a, b, mean, invstd, sums, inplace=True):
    """dz = a (g - m1 - xhat m2) from the channel sums (g already carries the activation derivative); frozen statistics:
    dz = a g.  In place by default."""
    B, C = z.shape[0], z.shape[1]
    S = z.numel() // (B * C)
    dz = g if inplace else torch.empty_like(g)
    asums = torch.zeros_like(sums) if getattr(mean, "_eat_frozen", False) else sums
    if _is16(g):                                    # bf16 storage: g_e, z_e and dz_e are all wide tensors
        _lib.call("eat_bn_bwd_apply_b16", _dev16(g, "g"), _dev16(z, "z"), a.data_ptr(), b.data_ptr(), mean.data_ptr(),
                  invstd.data_ptr(), asums.data_ptr(), dz.data_ptr(), B, C, S, ACT_NONE, _stream())
        return dz
    _lib.call("eat_bn_act_bwd_apply", _dev(g, "g"), _dev(z, "z"), a.data_ptr(), b.data_ptr(), mean.data_ptr(),
              invstd.data_ptr(), None, None, asums.data_ptr(), dz.data_ptr(), B, C, S, ACT_NONE, _stream())
    return dz


def block_fused_supported(Cin, Cexp, Cout, k, stride, act, proj, F=64, T=500):
    """True where the register-resident block kernel (csrc/irb.hip) has an instantiation: `mbconv` (proj) / `fused_expand_dw`."""
    return bool(_lib.lib().eat_block_fused_supported(Cin, Cexp, Cout, F, T, k, stride, act, 1 if proj else 0))


def fused_expand_dw(x, wp_e, bias_e, w_d, bias_d, Cexp, k, stride, act, pool=None):
    """expand 1x1 + BN + act -> depthwise k x k + BN + act in one kernel (eval; early blocks)."""
    B, Cin, F, T = x.shape
    Fo, To = conv_out(F, k, stride), conv_out(T, k, stride)
    y = torch.empty((B, Cexp, Fo, To), device=x.device, dtype=torch.float32)
    _lib.call("eat_fused_expand_dw_fwd", _dev(x, "x"), _dev(wp_e, "wp_e"), _dev(bias_e, "bias_e"), _dev(w_d, "w_d"),
              _dev(bias_d, "bias_d"), y.data_ptr(), _opt(pool, "pool"), B, Cin, Cexp, F, T, Fo, To, k, stride, act,
              _stream())
    return y


def front(x, w_s, bias_s, w_d, bias_d, wp_p, bias_p, act):
    """Stem conv + BN + Hardswish and the first (expand-less, SE-less) block in one kernel (eval)."""
    B, _, F, T = x.shape
    C = w_s.shape[0]
    Fo, To = conv_out(F, 3, 2), conv_out(T, 3, 2)
    y = torch.empty((B, C, Fo, To), device=x.device, dtype=torch.float32)
    _lib.call("eat_front_fwd", _dev(x, "x"), _dev(w_s, "w_s"), _dev(bias_s, "bias_s"), _dev(w_d, "w_d"),
              _dev(bias_d, "bias_d"), _dev(wp_p, "wp_p"), _dev(bias_p, "bias_p"), y.data_ptr(), B, C, F, T, Fo, To, act,
              _stream())
    return y


def mbconv(x, wp_e, bias_e, w_d, bias_d, wp_p, bias_p, Cexp, Cout, k, stride, act, res=None):
    """Whole inverted-residual block without SE in one kernel (eval; early blocks): expand + depthwise +
    project (+ residual); wp_e / wp_p are fp32 `pw_prepack` buffers."""
    B, Cin, F, T = x.shape
    Fo, To = conv_out(F, k, stride), conv_out(T, k, stride)
    y = torch.empty((B, Cout, Fo, To), device=x.device, dtype=torch.float32)
    _lib.call("eat_mbconv_fwd", _dev(x, "x"), _dev(wp_e, "wp_e"), _dev(bias_e, "bias_e"), _dev(w_d, "w_d"),
              _dev(bias_d, "bias_d"), _dev(wp_p, "wp_p"), _dev(bias_p, "bias_p"), _opt(res, "res"), y.data_ptr(), B, Cin,
              Cexp, Cout, F, T, Fo, To, k, stride, act, _stream())
    return y


def pw_prepack_bf16(w2d, row_scale=None, split=True, trans=False):
    Co, Ci = (w2d.shape[1], w2d.shape[0]) if trans else w2d.shape
    n = ((Ci + 31) // 32) * ((Co + 15) // 16) * (2 if split else 1) * 512
    wp = torch.empty((n,), device=w2d.device, dtype=torch.bfloat16)
    _lib.call("eat_pw_prepack_bf16_t" if trans else "eat_pw_prepack_bf16", _dev(w2d, "w"), _opt(row_scale, "row_scale"),
              wp.data_ptr(), Co, Ci, 1 if split else 0, _stream())
    return wp


_FUSE_EXPAND_DW = True   # expand + depthwise of the 8 x 63-class blocks as one kernel (csrc/expand_dw.hip)


def expand_dw_eligible(Ci, F, T, k, stride):
    """Geometry of eat_expand_dw_bf16_fwd (csrc/expand_dw.hip): small planes, 3x3 / stride 1, C_in <= 128."""
    S = F * T
    return (_FUSE_EXPAND_DW and k == 3 and stride == 1 and 1 <= T <= 64 and 1 <= F <= 8 and S % 4 == 0 and S <= 512
            and Ci % 4 == 0 and 4 <= Ci <= 128)


def expand_dw_bf16(x, wp16_e, bias_e, w_d, bias_d, Ce, k, stride, act, pool=None):
    """expand 1x1 (bf16x3) + BN + act -> depthwise 3x3 + BN + act (+ SE squeeze sums) in one kernel; the expanded tensor
    never reaches HBM (eval; late blocks with small planes)."""
    B, Ci, F, T = x.shape
    y = torch.empty((B, Ce, F, T), device=x.device, dtype=torch.float32)
    _lib.call("eat_expand_dw_bf16_fwd", _dev(x, "x"), wp16_e.data_ptr(), _dev(bias_e, "bias_e"), _dev(w_d, "w_d"),
              _dev(bias_d, "bias_d"), y.data_ptr(), _opt(pool, "pool"), B, Ci, Ce, F, T, k, stride, act, _stream())
    return y


def pw_stream_mode(mode=-1):
    """Variant of the bf16 1x1 kernels (eat_pw_stream_mode): bit 0 expand-shaped layers on the x-resident kernel,
    bit 1 project-shaped layers on the K-streaming kernel; returns the previous mode, a negative argument only queries."""
    return _lib.lib().eat_pw_stream_mode(int(mode))


def pw_conv_bf16(x, wp16, bias, Co, act, split=True, in_scale=None, res=None, pool=None, write=True):
    """1x1 conv on the bf16 matrix cores (split=True: bf16x3, fp32-class accuracy; False: plain bf16)."""
    B, Ci, F, T = x.shape
    y = torch.empty((B, Co, F, T), device=x.device, dtype=torch.float32) if write else None
    _lib.call("eat_pw_conv_bf16_fwd", _dev(x, "x"), wp16.data_ptr(), _dev(bias, "bias"), _opt(in_scale, "in_scale"),
              _opt(res, "res"), None if y is None else y.data_ptr(), _opt(pool, "pool"), B, Ci, Co, F * T, act,
              1 if split else 0, _stream())
    return y


# ------------------------------------------------------------------ bf16 activation storage (BASELINE configs[2])
def b16_block_ok(B, C_exp, F, T, k, stride):
    """True where the bf16-storage kernels cover an inverted-residual block whose depthwise conv maps (F, T) planes of
    C_exp channels (csrc/dw_plane.hip geometries, even plane sizes, 16-byte row pieces for the 1x1 kernels)."""
    Fo, To = conv_out(F, k, stride), conv_out(T, k, stride)
    return (bool(_lib.lib().eat_dw_conv_b16_ok(B, C_exp, F, T, Fo, To, k, stride)) and (F * T) % 8 == 0
            and (Fo * To) % 8 == 0 and C_exp % 8 == 0)


def pw_conv_b16(x, wp, bias, Co, act, tf=None, in_scale=None, res=None, x2=None, stats=False, out_b16=False, gstat=None):
    """1x1 conv of the bf16-storage plan: exactly one of input / output is the wide bf16 tensor (`eat_pw_conv_b16_fwd`).
    x fp32 -> y bf16 (expand conv, project data gradient); x bf16 -> y fp32 (project conv with tf = (a, b, act) / in_scale /
    stats=True -> (y, parts); two-source data-gradient GEMM with x2 fp32 + res).  wp: plain bf16 pack (precision 'bf16')."""
    if wp.dtype != torch.bfloat16 or getattr(wp, "_eat_split", False):
        raise _lib.EatHipError("pw_conv_b16: needs a plain bf16 weight pack (precision('bf16'))")
    B, C1, F, T = x.shape
    S = F * T
    x16 = _is16(x)
    Ci = C1 + (x2.shape[1] if x2 is not None else 0)
    y16 = out_b16 or not x16                       # (fp32 in: always a bf16 output; bf16 in: fp32, or bf16 for z_p)
    y = torch.empty((B, Co, F, T), device=x.device, dtype=torch.bfloat16 if y16 else torch.float32)
    part = None
    if stats or gstat is not None:
        tiles = int(_lib.lib().eat_pw_conv_stat_tiles(B, S, 0))
        part = torch.empty((tiles * 2 * Co,), device=x.device, dtype=torch.float32)
    a, b, tact = tf if tf is not None else (None, None, 0)
    # gstat = (z_d bf16, (a, b, mean, invstd), act, sums or None): bf16 -> bf16 project data gradient with the depthwise
    # BatchNorm's backward sums in the epilogue -> (y, sums (2 Co,) float64), see pw_conv_gstats
    gz, gst, gact, gsums = gstat if gstat is not None else (None, (None, None, None, None), 0, None)
    _lib.call("eat_pw_conv_b16_fwd", _dev16(x, "x") if x16 else _dev(x, "x"), 1 if x16 else 0, _opt(x2, "x2"), C1,
              wp.data_ptr(), _dev(bias, "bias"), _opt(a, "tf_a"), _opt(b, "tf_b"), tact, _opt(in_scale, "in_scale"),
              _opt(res, "res"), y.data_ptr(), 1 if y16 else 0, None if part is None else part.data_ptr(),
              None if gz is None else _dev16(gz, "gz"), _opt(gst[0], "g_a"), _opt(gst[1], "g_b"), gact, B, Ci, Co, S, act,
              _stream())
    if gstat is not None:
        sums = gsums if gsums is not None else torch.empty((2 * Co,), device=x.device, dtype=torch.float64)
        nws = int(_lib.lib().eat_bn_bwd_sums_ws_doubles(tiles, Co))
        ws = torch.empty((nws,), device=x.device, dtype=torch.float64) if nws else None
        _lib.call("eat_bn_bwd_sums_from_tiles", part.data_ptr(), tiles, Co, gst[2].data_ptr(), gst[3].data_ptr(),
                  gst[0].data_ptr(), gst[1].data_ptr(), None if ws is None else ws.data_ptr(), sums.data_ptr(), _stream())
        return y, sums
    return (y, (part, tiles, 1)) if stats else y


def cast_b16(x):
    """bf16 copy of a contiguous fp32 tensor (numel % 8 == 0; `eat_cast_b16`)."""
    y = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    _lib.call("eat_cast_b16", _dev(x, "x"), y.data_ptr(), x.numel(), _stream())
    return y


def pw_conv_wgrad_b16(dz, x, x_scale=None, tf=None, out=None):
    """dW (Co, Ci) = sum_b dz[b] . (act(tf_a x + tf_b) * x_scale)[b]^T with exactly one bf16 (wide) operand; plain bf16
    products, fp32 accumulation (`eat_pw_conv_wgrad_b16`)."""
    B, Co = dz.shape[0], dz.shape[1]
    Ci = x.shape[1]
    S = dz.numel() // (B * Co)
    d16, x16 = _is16(dz), _is16(x)
    n = int(_lib.lib().eat_pw_wgrad_b16_slots(B, Co, Ci, S, 1 if x16 else 0))
    dW = out if out is not None else zero_arena.zeros((Co, Ci), torch.float32, dz.device)
    ws = torch.empty((n, Co, Ci), device=dz.device, dtype=torch.float32)
    a, b, tact = tf if tf is not None else (None, None, 0)
    _lib.call("eat_pw_conv_wgrad_b16", _dev16(dz, "dz") if d16 else _dev(dz, "dz"), 1 if d16 else 0,
              _dev16(x, "x") if x16 else _dev(x, "x"), 1 if x16 else 0, _opt(a, "tf_a"), _opt(b, "tf_b"), tact,
              _opt(x_scale, "x_scale"), dW.data_ptr(), ws.data_ptr(), n, B, Co, Ci, S, _stream())
    return dW


# ---- bf16 activation storage of the DyMN blocks (dymn_train.py; include/eat_hip.h "bf16 activation storage for the DyMN blocks")
def dyn_b16_block_ok(B, cin, cexp, cout, F, T, k, stride):
    """True where the bf16-storage kernels cover a fully dynamic DY_Block: depthwise geometry (`b16_block_ok`), 16-byte row
    pieces and whole 4-channel groups for the per-sample 1x1 kernels."""
    return b16_block_ok(B, cexp, F, T, k, stride) and cin % 4 == 0 and cout % 4 == 0 and cexp % 8 == 0


def dyn_pw_pack_b16(bank, att, Co, Ci, trans=False):
    """Aggregated per-sample weights sum_k att[b,k] bank[k] as PLAIN bf16 MFMA fragments (`eat_dyn_pw_pack_b16`) ->
    (B, KK*MT*512) bfloat16.  trans: `bank` (K, Ci*Co) holds the transposed matrices (the data-gradient pack)."""
    K, B = bank.shape[0], att.shape[0]
    n = ((Ci + 31) // 32) * ((Co + 15) // 16) * 512
    wp = torch.empty((B, n), device=bank.device, dtype=torch.bfloat16)
    _lib.call("eat_dyn_pw_pack_b16", _dev(bank, "bank"), _dev(att, "att"), wp.data_ptr(), B, K, Co, Ci, 1 if trans else 0,
              _stream())
    return wp


def pw_conv_dyn_b16(x, wp_b, Co, act, res=None, stats=False):
    """Per-sample-weight 1x1 conv of the bf16-storage plan (`eat_pw_conv_dyn_b16_fwd`): x fp32 -> y bf16, or x bf16 -> y fp32
    (+ res).  stats: -> (y, (part, tiles, 1)) with the batch statistics of y as stored."""
    B, Ci, F, T = x.shape
    S = F * T
    x16 = _is16(x)
    y = torch.empty((B, Co, F, T), device=x.device, dtype=torch.float32 if x16 else torch.bfloat16)
    part = None
    if stats:
        tiles = int(_lib.lib().eat_pw_conv_stat_tiles(B, S, 1))
        part = torch.empty((tiles * 2 * Co,), device=x.device, dtype=torch.float32)
    _lib.call("eat_pw_conv_dyn_b16_fwd", _dev16(x, "x") if x16 else _dev(x, "x"), 1 if x16 else 0, wp_b.data_ptr(),
              _zero_bias(Co, x.device).data_ptr(), _opt(res, "res"), y.data_ptr(), 0 if x16 else 1,
              None if part is None else part.data_ptr(), B, Ci, Co, S, act, _stream())
    return (y, (part, tiles, 1)) if stats else y


def pw_conv_dyn_wgrad_b16(dz, x):
    """Per-sample weight gradients G (B, Co*Ci) = dz[b] x[b]^T on the wide-tile kernel (`eat_pw_conv_dyn_wgrad_b16`): one bf16
    (wide) operand - plain bf16 products - or both fp32 - split-operand bf16x3 products; every element is stored.  None where
    the kernel does not take an all-fp32 shape."""
    B, Co = dz.shape[0], dz.shape[1]
    Ci = x.shape[1]
    S = dz.numel() // (B * Co)
    d16, x16 = _is16(dz), _is16(x)
    ns = int(_lib.lib().eat_pw_dyn_wgrad_b16_slices(B, Co, Ci, S, 1 if x16 else (0 if d16 else 2)))
    if ns < 1:
        return None                                 # (both fp32 and a shape the wide-tile kernel does not take: the caller's fallback)
    buf = torch.empty((ns, B, Co * Ci), device=dz.device, dtype=torch.float32)     # copy 0 = the result, the rest k-slice workspace
    _lib.call("eat_pw_conv_dyn_wgrad_b16", _dev16(dz, "dz") if d16 else _dev(dz, "dz"), 1 if d16 else 0,
              _dev16(x, "x") if x16 else _dev(x, "x"), 1 if x16 else 0, buf.data_ptr(), ns, B, Co, Ci, S, _stream())
    return buf[0]


# ------------------------------------------------------------------ precision switch (training plans)
class precision:
    """Context manager selecting the arithmetic of the 1x1 convs issued through pw_prepack/pw_conv:
    'fp32' (exact fp32 MFMA), 'auto' (fp32 below C_in = 40, split-operand bf16x3 - fp32-class accuracy at a
    multiple of the fp32 MFMA rate - from there on; see mn._pw_mode), 'bf16x3' (split everywhere) or 'bf16'
    (plain bf16 operands on v_mfma_f32_16x16x32_bf16, fp32 accumulation and fp32 activations in memory -
    BASELINE config 3)."""
    mode = "fp32"

    def __init__(self, mode):
        if mode not in ("fp32", "auto", "bf16x3", "bf16"):
            raise ValueError(f"unknown precision {mode!r}")
        self.new = mode

    def __enter__(self):
        self.old, precision.mode = precision.mode, self.new

    def __exit__(self, *exc):
        precision.mode = self.old


_pw_prepack_fp32, _pw_conv_fp32 = pw_prepack, pw_conv


def pw_prepack(w2d, row_scale=None, trans=False):  # noqa: F811
    m = precision.mode
    ci = w2d.shape[0] if trans else w2d.shape[1]
    if m == "bf16":
        return pw_prepack_bf16(w2d, row_scale, split=False, trans=trans)
    if m == "bf16x3" or (m == "auto" and ci >= 40 and ci % 4 == 0):
        wp = pw_prepack_bf16(w2d, row_scale, split=True, trans=trans)
        wp._eat_split = True                 # both bf16 packs share the dtype: mark the hi/lo one
        return wp
    return _pw_prepack_fp32(w2d, row_scale, trans=trans)


def pw_conv(x, wp, bias, Co, act, in_scale=None, res=None, pool=None, write=True):  # noqa: F811
    if wp.dtype == torch.bfloat16:
        return pw_conv_bf16(x, wp, bias, Co, act, getattr(wp, "_eat_split", False), in_scale=in_scale, res=res,
                            pool=pool, write=write)
    return _pw_conv_fp32(x, wp, bias, Co, act, in_scale=in_scale, res=res, pool=pool, write=write)


class PrepackPlan:
    """The weight packs of a model's 1x1 convs for one pass, produced by ONE launch (`eat_pw_prepack_multi`) instead of one
    per matrix.  entries: [(key, w2d, trans)] with w2d a contiguous (rows, cols) view of a parameter; the arithmetic
    (fp32 / bf16 / bf16x3 pack) follows `precision.mode` at construction exactly as `pw_prepack` decides it.  The packs live
    in a buffer owned by the plan and are refreshed by `run()`; `get(key)` hands out views (valid until the next run)."""

    def __init__(self, entries):
        import numpy as np
        self.mode = precision.mode
        dev = entries[0][1].device
        recs, views, off, max_threads = [], {}, 0, 1
        for key, w2d, trans in entries:      # w2d: the parameter itself (Co, Ci[, 1, 1]), contiguous
            if not w2d.is_contiguous():
                raise _lib.EatHipError("PrepackPlan: weights must be contiguous")
            rows, cols = w2d.shape[0], w2d.numel() // w2d.shape[0]
            Co, Ci = (cols, rows) if trans else (rows, cols)
            m = self.mode
            if m == "bf16":
                kind = 1
            elif m == "bf16x3" or (m == "auto" and Ci >= 40 and Ci % 4 == 0):
                kind = 2
            else:
                kind = 0
            mt = (Co + 15) // 16
            if kind == 0:
                if Ci % 4:
                    raise _lib.EatHipError(f"PrepackPlan: Ci={Ci} must be a multiple of 4")
                nbytes, threads = (Ci // 4) * mt * 64 * 4, (Ci // 4) * mt * 64
            else:
                nbytes, threads = ((Ci + 31) // 32) * mt * kind * 512 * 2, ((Ci + 31) // 32) * mt * 64
            max_threads = max(max_threads, threads)
            recs.append((w2d, off, Co, Ci, kind, 1 if trans else 0, nbytes, key))
            off += (nbytes + 255) // 256 * 256
        self.buf = torch.empty((off,), device=dev, dtype=torch.uint8)
        table = np.zeros((len(recs),), dtype=[("w", "<u8"), ("wp", "<u8"), ("Co", "<i4"), ("Ci", "<i4"), ("kind", "<i4"), ("trans", "<i4")])
        self.ptrs = []
        for i, (w2d, o, Co, Ci, kind, trans, nbytes, key) in enumerate(recs):
            table[i] = (w2d.data_ptr(), self.buf.data_ptr() + o, Co, Ci, kind, trans)
            self.ptrs.append(w2d.data_ptr())
            v = self.buf[o:o + nbytes].view(torch.float32 if kind == 0 else torch.bfloat16)
            if kind == 2:
                v._eat_split = True
            views[key] = v
        self.keep = [r[0] for r in recs]
        self.views, self.n, self.max_threads = views, len(recs), max_threads
        self.table = torch.from_numpy(table.view(np.uint8).copy()).to(dev)

    def stale(self):
        """Parameters were re-allocated (model.to(), a new state dict with fresh storage) or the arithmetic changed."""
        return self.mode != precision.mode or any(t.data_ptr() != p for t, p in zip(self.keep, self.ptrs))

    runs = 0          # number of `run()` calls: a backward that finds it changed knows its views were re-packed since

    def run(self):
        self.runs += 1
        _lib.call("eat_pw_prepack_multi", self.table.data_ptr(), self.n, self.max_threads, _stream())

    def get(self, key):
        return self.views[key]


# ------------------------------------------------------------------ training-loop glue (ex_audioset.py:142-194)
def col_sum(m):
    """Column sums of a contiguous (R, C) matrix (bias gradients of the DyMN context path)."""
    R, C = m.shape
    out = torch.empty((C,), device=m.device, dtype=torch.float32)
    _lib.call("eat_col_sum", _dev(m, "m"), out.data_ptr(), R, C, _stream())
    return out


def mixup_fwd(x, perm, lam):
    """x[b] * lam[b] + x[perm[b]] * (1 - lam[b]) over the flattened per-sample axis; perm int32 (B), lam fp32 (B)."""
    B = x.shape[0]
    out = torch.empty_like(x)
    _lib.call("eat_mixup_fwd", _dev(x, "x"), perm.data_ptr(), lam.data_ptr(), out.data_ptr(), B, x.numel() // B, _stream())
    return out


def wave_i16_to_f32(src, out=None, scale=1.0 / 32768.0):
    """int16 PCM (any shape, contiguous, on the GPU) -> fp32 waveform `src * scale` (f2: 16-bit transport over PCIe)."""
    if not src.is_cuda or src.dtype != torch.int16 or not src.is_contiguous():
        raise _lib.EatHipError(f"wave_i16_to_f32: src must be a contiguous int16 GPU tensor (got {src.dtype} on {src.device})")
    if out is None:
        out = torch.empty(src.shape, device=src.device, dtype=torch.float32)
    elif out.numel() != src.numel():
        raise _lib.EatHipError("wave_i16_to_f32: out has a different number of samples")
    _lib.call("eat_wave_i16_to_f32", src.data_ptr(), _dev(out, "out"), src.numel(), float(scale), _stream())
    return out


def kd_loss_fwd_bwd(logits, y, perm, lam, teacher, teacher_idx, kd_lambda, sums):
    """-> dlogits; accumulates (loss, label part, distillation part) into `sums` (3,) on the device."""
    B, C = logits.shape
    dlogits = torch.empty_like(logits)
    _lib.call("eat_kd_loss_fwd_bwd", _dev(logits, "logits"), _dev(y, "y"), None if perm is None else perm.data_ptr(),
              None if lam is None else lam.data_ptr(), None if teacher is None else _dev(teacher, "teacher"),
              None if teacher_idx is None else teacher_idx.data_ptr(), 0 if teacher is None else teacher.shape[0],
              float(kd_lambda), B, C, sums.data_ptr(), dlogits.data_ptr(), _stream())
    return dlogits
