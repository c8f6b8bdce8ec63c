"""Host-side audio loading for the inference surfaces (`inference.py:45`, `windowed_inference.py`): the reference calls
`librosa.core.load(path, sr=32000, mono=True)` (decode, down-mix, resample to 32 kHz, float32 in [-1, 1]).  librosa is
not available offline; this is the same contract on scipy: PCM / float WAV decode, channel mean, polyphase resampling
(`scipy.signal.resample_poly`, the rational up/down factors of the two rates)."""
import math

import numpy as np


def load_audio(path, sr=32000, mono=True):
    """-> (waveform float32 (n,) [or (channels, n) if not mono], sr)."""
    from scipy.io import wavfile
    from scipy.signal import resample_poly
    src_sr, data = wavfile.read(path)
    if data.dtype == np.uint8:
        x = (data.astype(np.float32) - 128.0) / 128.0
    elif np.issubdtype(data.dtype, np.integer):
        x = data.astype(np.float32) / float(2 ** (8 * data.dtype.itemsize - 1))
    else:
        x = data.astype(np.float32)
    if x.ndim == 2:
        x = x.mean(axis=1) if mono else x.T
    if sr is not None and src_sr != sr:
        g = math.gcd(int(sr), int(src_sr))
        x = resample_poly(x, sr // g, src_sr // g, axis=-1).astype(np.float32)
    return np.ascontiguousarray(x, dtype=np.float32), (sr if sr is not None else src_sr)
