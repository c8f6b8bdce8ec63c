"""Restatement of torchaudio 0.13 ``compliance.kaldi.get_mel_banks`` (published
algorithm; the upstream source is not in this container).  Called by the
reference at models/preprocess.py:52-53."""
import math
import torch


def get_mel_banks(num_bins, window_length_padded, sample_freq, low_freq, high_freq,
                  vtln_low, vtln_high, vtln_warp_factor):
    assert num_bins > 3 and window_length_padded % 2 == 0
    num_fft_bins = window_length_padded / 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    assert 0.0 <= low_freq < nyquist and 0.0 < high_freq <= nyquist and low_freq < high_freq
    fft_bin_width = sample_freq / window_length_padded
    mel_low = 1127.0 * math.log(1.0 + low_freq / 700.0)
    mel_high = 1127.0 * math.log(1.0 + high_freq / 700.0)
    delta = (mel_high - mel_low) / (num_bins + 1)
    assert vtln_warp_factor == 1.0, "VTLN warping is not used by the reference hot path"
    b = torch.arange(num_bins).unsqueeze(1)
    left = mel_low + b * delta
    center = mel_low + (b + 1.0) * delta
    right = mel_low + (b + 2.0) * delta
    center_freqs = 700.0 * ((center / 1127.0).exp() - 1.0)
    mel = (1127.0 * (1.0 + (fft_bin_width * torch.arange(num_fft_bins)) / 700.0).log()).unsqueeze(0)
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    bins = torch.max(torch.zeros(1), torch.min(up, down))
    return bins, center_freqs
