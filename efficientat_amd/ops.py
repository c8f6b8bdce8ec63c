"""Python-side launchers for the C ABI (include/eat_hip.h): argument checks, output
allocation through PyTorch's caching allocator, launch on torch's current HIP stream.
PyTorch is plumbing here (device memory + streams); all arithmetic is in libeat_hip.so."""
import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_HSWISH, ACT_SIGMOID = 0, 1, 2, 3


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(t, name):
    if not t.is_cuda:
        raise _lib.EatHipError(f"{name} must live on the GPU: efficientat_amd has no CPU path "
                               f"(got device {t.device})")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise _lib.EatHipError(f"{name} must be contiguous float32 (got {t.dtype}, contiguous={t.is_contiguous()})")
    return t.data_ptr()


def _opt(t, name):
    return None if t is None else _dev(t, name)


def conv_out(n, k, stride):
    """floor((n + 2p - (k-1) - 1)/s + 1), p=(k-1)//2   (models/mn/utils.py:24-26, dilation 1)."""
    p = (k - 1) // 2
    return (n + 2 * p - (k - 1) - 1) // stride + 1


def mel_fwd(wave, window, twiddle, band_w, band_start, n_fft, hop, n_mels, fmask=(0, 0), tmask=(0, 0)):
    B, L = wave.shape
    T = 1 + (L - 1) // hop
    out = torch.empty((B, n_mels, T), device=wave.device, dtype=torch.float32)
    _lib.call("eat_mel_fwd", _dev(wave, "wave"), B, L, _dev(window, "window"), window.numel(), n_fft, hop,
              _dev(twiddle, "twiddle"), _dev(band_w, "band_w"), band_start.data_ptr(), n_mels,
              band_w.shape[1], out.data_ptr(), T, fmask[0], fmask[1], tmask[0], tmask[1], _stream())
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


def pw_prepack(w2d, row_scale=None):
    Co, Ci = w2d.shape
    wp = torch.empty(((Ci // 4) * ((Co + 15) // 16) * 64,), device=w2d.device, dtype=torch.float32)
    _lib.call("eat_pw_prepack", _dev(w2d, "w"), _opt(row_scale, "row_scale"), wp.data_ptr(), Co, Ci, _stream())
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
