"""Drop-in for the reference's ``models.preprocess.AugmentMelSTFT`` (models/preprocess.py:6-67)
backed by the fused HIP kernel ``eat_mel_fwd``.

Same constructor signature, buffers (``window`` and ``preemphasis_coefficient``, both
non-persistent so the state_dict stays empty), host RNG draw sequence and output
``(B, n_mels, T)``.  The kaldi mel basis is built on the host with the reference's exact fp32
op order (so the set of non-zero (mel, bin) pairs is identical, incl. the fp32-only entry at
bin 480 / filter 127) and shipped to the kernel as a banded table of 8-byte weight pairs.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import ops


def kaldi_mel_basis(n_mels, n_fft, sr, fmin, fmax):
    """torchaudio 0.13 ``compliance.kaldi.get_mel_banks(n_mels, n_fft, sr, fmin, fmax, 100, -500, 1.0)``
    restated op-for-op in fp32 torch CPU arithmetic (call site models/preprocess.py:52-53)."""
    nyquist = 0.5 * sr
    if fmax <= 0.0:
        fmax += nyquist
    if not (0.0 <= fmin < nyquist and 0.0 < fmax <= nyquist and fmin < fmax):
        raise ValueError(f"Bad values in options: low-freq {fmin} and high-freq {fmax} vs. nyquist {nyquist}")
    bin_width = sr / n_fft
    m_lo = 1127.0 * math.log(1.0 + fmin / 700.0)
    m_hi = 1127.0 * math.log(1.0 + fmax / 700.0)
    delta = (m_hi - m_lo) / (n_mels + 1)
    idx = torch.arange(n_mels).unsqueeze(1)
    left = m_lo + idx * delta
    center = m_lo + (idx + 1.0) * delta
    right = m_lo + (idx + 2.0) * delta
    mel = (1127.0 * (1.0 + (bin_width * torch.arange(n_fft / 2)) / 700.0).log()).unsqueeze(0)
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    return torch.max(torch.zeros(1), torch.min(up, down))          # (n_mels, n_fft/2)


def band_table(basis, pairs=None):
    """Dense (n_mels, n_fft/2) basis -> the banded table the kernel walks in 8-byte pairs:
         band_w2    (P, n_mels, 2) fp32   pair j of row m = basis[m, start[m] + 2j : start[m] + 2j + 2]
         band_start (n_mels) int32        EVEN first bin of each band, start[m] + 2 P <= n_fft/2
         band_cnt   (n_mels) int32        pairs that cover row m's non-zeros (0 for an empty row)
    Every non-zero of row m lies in [start[m], start[m] + 2 cnt[m]); pairs beyond cnt[m] are zero.
    pairs: a fixed P (>= the widest band) - tables of different (fmin, fmax) then share one shape, which is what lets a
    captured hipGraph take the basis as an input buffer (train_loop.GraphedKDTrainer)."""
    nz = basis != 0
    n_mels, nb = basis.shape
    assert nb % 2 == 0
    cols = torch.arange(nb)
    first = torch.where(nz, cols, nb).min(dim=1).values
    last = torch.where(nz, cols, -1).max(dim=1).values
    empty = last < 0
    first = torch.where(empty, torch.zeros_like(first), first)
    last = torch.where(empty, torch.zeros_like(last), last)
    start = first - first % 2
    widest = int(max(1, ((last - start + 2) // 2).max().item()))
    if pairs is None:
        pairs = widest
    elif pairs < widest:
        raise ValueError(f"band_table: the widest band needs {widest} pairs, the fixed table holds {pairs}")
    start = torch.minimum(start, torch.full_like(start, nb - 2 * pairs)).clamp_(min=0)
    cnt = torch.where(empty, torch.zeros_like(last), (last - start + 2) // 2)
    gather = start.unsqueeze(1) + torch.arange(2 * pairs).unsqueeze(0)
    w2 = basis.gather(1, gather).reshape(n_mels, pairs, 2).permute(1, 0, 2).contiguous()
    return w2, start.to(torch.int32).contiguous(), cnt.to(torch.int32).contiguous()


def fft_twiddles(n_fft):
    j = np.arange(n_fft, dtype=np.float64)
    ang = -2.0 * np.pi * j / n_fft
    return torch.from_numpy(np.stack([np.cos(ang), np.sin(ang)], axis=1).astype(np.float32))


class AugmentMelSTFT(nn.Module):
    def __init__(self, n_mels=128, sr=32000, win_length=800, hopsize=320, n_fft=1024, freqm=48, timem=192,
                 fmin=0.0, fmax=None, fmin_aug_range=10, fmax_aug_range=2000):
        super().__init__()
        self.win_length, self.n_mels, self.n_fft, self.sr = win_length, n_mels, n_fft, sr
        self.fmin = fmin
        if fmax is None:
            fmax = sr // 2 - fmax_aug_range // 2
            print(f"Warning: FMAX is None setting to {fmax} ")
        self.fmax = fmax
        self.hopsize = hopsize
        self.register_buffer("window", torch.hann_window(win_length, periodic=False), persistent=False)
        assert fmin_aug_range >= 1, f"fmin_aug_range={fmin_aug_range} should be >=1; 1 means no augmentation"
        assert fmax_aug_range >= 1, f"fmax_aug_range={fmax_aug_range} should be >=1; 1 means no augmentation"
        self.fmin_aug_range, self.fmax_aug_range = fmin_aug_range, fmax_aug_range
        self.register_buffer("preemphasis_coefficient", torch.as_tensor([[[-.97, 1]]]), persistent=False)
        self.freqm, self.timem = freqm, timem          # mask parameters (0 = off), fused into the kernel
        self._twiddle = None
        self._tables = {}                              # (fmin, fmax, device) -> (band_w2, band_start, band_cnt)
        self._static = None                            # graph mode: fixed-shape device tables + pinned staging ring

    # -- host helpers -------------------------------------------------------------------
    def _device_tables(self, fmin, fmax, device):
        key = (float(fmin), float(fmax), str(device))
        hit = self._tables.get(key)
        if hit is None:
            hit = tuple(t.to(device, non_blocking=True)
                        for t in band_table(kaldi_mel_basis(self.n_mels, self.n_fft, self.sr, fmin, fmax)))
            if len(self._tables) > 64:
                self._tables.clear()
            self._tables[key] = hit
        if self._twiddle is None or self._twiddle.device != device:
            self._twiddle = fft_twiddles(self.n_fft).to(device)
        return hit

    @staticmethod
    def _draw_mask(param, size):
        """torchaudio 0.13 mask_along_axis on a 3-D (B,F,T) input: one mask for the batch."""
        value = torch.rand(1) * param
        min_value = torch.rand(1) * (size - value)
        start = int(min_value.long())
        return start, start + int(value.long())

    def draw(self, n_samples):
        """The host RNG draws of one forward call, in the reference's order (preprocess.py:45-46, 61-63; also drawn in
        eval, where they are ignored) -> (fmin, fmax, fmask, tmask)."""
        fmin = self.fmin + torch.randint(self.fmin_aug_range, (1,)).item()
        fmax = self.fmax + self.fmax_aug_range // 2 - torch.randint(self.fmax_aug_range, (1,)).item()
        if not self.training:
            fmin, fmax = self.fmin, self.fmax
        T = 1 + (n_samples - 1) // self.hopsize
        fmask = tmask = (0, 0)
        if self.training:
            if self.freqm:
                fmask = self._draw_mask(self.freqm, self.n_mels)
            if self.timem:
                tmask = self._draw_mask(self.timem, T)
        return fmin, fmax, fmask, tmask

    def forward(self, x, out=None):
        fmin, fmax, fmask, tmask = self.draw(x.shape[1])
        x = x.contiguous().float()
        band_w2, band_start, band_cnt = self._device_tables(fmin, fmax, x.device)
        return ops.mel_fwd(x, self.window, self._twiddle, band_w2, band_start, band_cnt, self.n_fft, self.hopsize,
                           self.n_mels, fmask, tmask, out=out)

    # -- graph mode: the mel basis as an INPUT BUFFER of a captured step (train_loop.GraphedKDTrainer) -------------------
    def max_band_pairs(self):
        """Pairs of the widest band over every (fmin, fmax) the train-mode jitter can draw: the band of the top filter
        grows with fmax and (slightly) with falling fmin, so the corners of the draw space bound the band WIDTH; the even alignment of the band
        start adds up to one pair at some interior fmax (brute force over the default space: 14 pairs at (0, 15688) against
        13 at the corners): + 2 pairs of slack.  `band_table(..., pairs=P)` raises if a table ever needed more."""
        hi = self.fmax + self.fmax_aug_range // 2
        w = 1
        for fmin in (self.fmin, self.fmin + self.fmin_aug_range - 1):
            for fmax in (hi, hi - self.fmax_aug_range + 1, self.fmax):
                w = max(w, band_table(kaldi_mel_basis(self.n_mels, self.n_fft, self.sr, fmin, fmax))[0].shape[0])
        return w + 2

    def static_tables(self, device, ring=4):
        """Fixed-shape device tables (band_w2 (P, n_mels, 2), band_start, band_cnt) whose CONTENTS `stage_tables` replaces
        per step, with a ring of pinned host copies so that a step's upload never waits for the previous one."""
        if self._static is None or self._static["dev"][0].device != torch.device(device):
            P = self.max_band_pairs()
            dev = (torch.zeros((P, self.n_mels, 2), device=device), torch.zeros(self.n_mels, dtype=torch.int32, device=device),
                   torch.zeros(self.n_mels, dtype=torch.int32, device=device))
            host = [tuple(torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in dev) for _ in range(ring)]
            self._static = {"dev": dev, "host": host, "ev": [None] * ring, "i": 0, "P": P, "cache": {}}
            self._device_tables(self.fmin, self.fmax, torch.device(device))       # twiddles
        return self._static["dev"]

    def stage_tables(self, fmin, fmax, stream=None):
        """Upload the band table of (fmin, fmax) into the static device tables (asynchronous H2D from pinned memory on
        `stream`, default: the current stream - i.e. ordered before a graph replay issued on it afterwards)."""
        st = self._static
        key = (float(fmin), float(fmax))
        tab = st["cache"].get(key)
        if tab is None:
            tab = band_table(kaldi_mel_basis(self.n_mels, self.n_fft, self.sr, fmin, fmax), pairs=st["P"])
            if len(st["cache"]) > 256:
                st["cache"].clear()
            st["cache"][key] = tab
        i = st["i"]
        st["i"] = (i + 1) % len(st["host"])
        if st["ev"][i] is not None:
            st["ev"][i].synchronize()                     # (ring of 4: that upload finished steps ago)
        for h, t, d in zip(st["host"][i], tab, st["dev"]):
            h.copy_(t)
            d.copy_(h, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(stream if stream is not None else torch.cuda.current_stream())
        st["ev"][i] = ev

    def forward_static(self, x, out=None, fmask=(0, 0), tmask=(0, 0)):
        """The kernel launch alone, reading the static tables (capturable: no host draw, no allocation besides `out`)."""
        band_w2, band_start, band_cnt = self._static["dev"]
        return ops.mel_fwd(x, self.window, self._twiddle, band_w2, band_start, band_cnt, self.n_fft, self.hopsize,
                           self.n_mels, fmask, tmask, out=out)
