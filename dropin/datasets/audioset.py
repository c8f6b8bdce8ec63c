"""`datasets.audioset` of the reference (datasets/audioset.py): same public surface, two back-ends.

The reference reads mp3 bytes from three AudioSet HDF5 files (decode with `av`, pad / truncate to 10 s, optional gain /
roll / waveform mix-up) and hands the training loop tuples
    (waveform float32 (1, clip_samples), audio_name str, target float32 (527,)[, index int])
(datasets/audioset.py:94-103,138-161).  Back-end selection, at import like the reference's `dataset_dir` assertion:

  * REAL: `dataset_dir` (below; or the environment variable EAT_AUDIOSET_DIR) names a directory holding
    balanced_train_segments_mp3.hdf / unbalanced_train_segments_mp3.hdf / eval_segments_mp3.hdf -> the HDF5 + mp3 reader
    of `_hdf5_reader.py` (h5py rows `audio_name`, `mp3`, bit-packed `target`; PyAV decode; needs h5py and av - QUARANTINED:
    never executed in the images this package was built in, it announces that on import; tests/test_host_cpu.py holds a
    round-trip test that runs wherever both libraries exist);
  * SYNTHETIC: only on the explicit opt-in EAT_SYNTH_AUDIOSET=1 (the benchmarks, the GPU tests and the staged runs of
    the reference's own scripts set it): clips of the same layout generated on the fly - deterministic per index (numpy
    PCG64 seeded with the index), band-limited noise + tones + silence so that every mel band is exercised;
  * neither: AssertionError, as in the reference - nobody trains or reports mAP on noise by accident.

Same public functions, argument names and defaults as the reference: `get_test_set`, `get_training_set`,
`get_full_training_set`, `get_base_*`, `get_ft_weighted_sampler`, `get_ft_cls_balanced_sample_weights`, `MixupDataset`,
`AddIndexDataset`, `pad_or_truncate`, `pydub_augment`, `decode_mp3`.

Synthetic sizes: EAT_SYNTH_AUDIOSET_TRAIN (default 2048 balanced + 6144 "unbalanced") / EAT_SYNTH_AUDIOSET_TEST (default
1054) clips.
"""
import io
import os

import numpy as np
import torch
from torch.utils.data import ConcatDataset, Dataset as TorchDataset, WeightedRandomSampler

dataset_dir = os.environ.get("EAT_AUDIOSET_DIR")          # or assign the AudioSet location here, as in the reference
_HDF_NAMES = {"balanced_train_hdf5": "balanced_train_segments_mp3.hdf",
              "unbalanced_train_hdf5": "unbalanced_train_segments_mp3.hdf", "eval_hdf5": "eval_segments_mp3.hdf"}
SYNTHETIC = os.environ.get("EAT_SYNTH_AUDIOSET", "0") == "1"
if not SYNTHETIC:
    assert dataset_dir is not None, (
        "Specify the AudioSet location (datasets.audioset.dataset_dir or EAT_AUDIOSET_DIR: the directory with the three "
        "*_segments_mp3.hdf files of https://github.com/kkoutini/PaSST/tree/main/audioset), or opt in to the SYNTHETIC "
        "stand-in explicitly with EAT_SYNTH_AUDIOSET=1")
    _missing = [f for f in _HDF_NAMES.values() if not os.path.isfile(os.path.join(dataset_dir, f))]
    assert not _missing, f"AudioSet files not found under {dataset_dir}: {_missing} (EAT_SYNTH_AUDIOSET=1 selects the synthetic stand-in)"
    dataset_config = {k: os.path.join(dataset_dir, f) for k, f in _HDF_NAMES.items()}
    dataset_config["num_of_classes"] = 527
else:
    dataset_dir = "synthetic"
    _N_TRAIN = int(os.environ.get("EAT_SYNTH_AUDIOSET_TRAIN", "2048"))
    dataset_config = {
        "balanced_train_hdf5": ("balanced", _N_TRAIN, 0),
        "unbalanced_train_hdf5": ("unbalanced", 3 * _N_TRAIN, 1 << 20),
        "eval_hdf5": ("eval", int(os.environ.get("EAT_SYNTH_AUDIOSET_TEST", "1054")), 527 << 12),
        "num_of_classes": 527,
    }


def _reader():
    """The quarantined HDF5 + mp3 reader module (loaded on first use; also when this file was loaded outside its package)."""
    global _READER
    if _READER is None:
        import importlib.util
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_hdf5_reader.py")
        spec = importlib.util.spec_from_file_location("eat_dropin_hdf5_reader", path)
        _READER = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_READER)
        _READER.pad_or_truncate, _READER.pydub_augment = pad_or_truncate, pydub_augment
    return _READER


_READER = None


def decode_mp3(mp3_arr):
    """uint8 array holding one mp3 file -> float32 waveform (datasets/audioset.py:32-47; PyAV - see _hdf5_reader.py)."""
    return _reader().decode_mp3(mp3_arr)


def pad_or_truncate(x, audio_length):
    """Pad all audio to specific length (datasets/audioset.py:50-55)."""
    if len(x) <= audio_length:
        return np.concatenate((x, np.zeros(audio_length - len(x), dtype=np.float32)), axis=0)
    return x[0:audio_length]


def pydub_augment(waveform, gain_augment=0):
    """Random gain of +-gain_augment dB, drawn from torch's RNG like the reference (datasets/audioset.py:58-63)."""
    if gain_augment:
        gain = torch.randint(gain_augment * 2, (1,)).item() - gain_augment
        waveform = waveform * (10 ** (gain / 20))
    return waveform


def synth_name(global_index):
    return "syn%07d" % global_index


def _synth_target(rng, classes_num, g):
    y = (rng.random(classes_num) < 2.7 / classes_num).astype(np.float32)
    y[g % classes_num] = 1.0        # every clip carries a label, and >= classes_num consecutive clips cover every class
    return y                        # (sklearn's per-class ROC / AP of `_test` need a positive and a negative per class)


class _SyntheticAudioSet(TorchDataset):
    """Same constructor and item layout as the reference class (datasets/audioset.py:106-177); `hdf5_file` is one of the
    synthetic `dataset_config` entries (name, length, index offset) instead of a path."""

    def __init__(self, hdf5_file, sample_rate=32000, resample_rate=32000, classes_num=527, clip_length=10, in_mem=False,
                 gain_augment=0):
        self.name, self.length, self.offset = hdf5_file
        self.sample_rate, self.resample_rate = sample_rate, resample_rate
        self.clip_length = clip_length * sample_rate
        self.classes_num, self.gain_augment = classes_num, gain_augment
        print(f"Dataset from synthetic:{self.name} with length {self.length}.")

    def __len__(self):
        return self.length

    def targets(self):
        return np.stack([_synth_target(np.random.Generator(np.random.PCG64(self.offset + i)), self.classes_num, self.offset + i)
                         for i in range(self.length)])

    def __getitem__(self, index):
        g = self.offset + int(index)
        rng = np.random.Generator(np.random.PCG64(g))
        target = _synth_target(rng, self.classes_num, g)
        n = self.clip_length - int(rng.integers(0, self.sample_rate // 2)) * (g % 3 == 0)   # some clips are shorter: padded
        t = np.arange(n, dtype=np.float32) / self.sample_rate
        wave = 0.05 * rng.standard_normal(n).astype(np.float32)
        for _ in range(1 + g % 3):
            f0, amp = float(rng.uniform(60.0, 9000.0)), float(rng.uniform(0.02, 0.3))
            wave += amp * np.sin(2 * np.pi * f0 * t + float(rng.uniform(0, 6.28))).astype(np.float32)
        if g % 5 == 0:
            wave[: n // 8] = 0.0                              # leading silence: log floor of the mel front-end
        wave = pydub_augment(wave, self.gain_augment)
        wave = pad_or_truncate(np.clip(wave, -1.0, 1.0).astype(np.float32), self.clip_length)
        return self.resample(wave).reshape(1, -1), synth_name(g), target

    def resample(self, waveform):
        if self.resample_rate == 32000:
            return waveform
        if self.resample_rate == 16000:
            return waveform[0::2]
        if self.resample_rate == 8000:
            return waveform[0::4]
        raise Exception("Incorrect sample rate!")


if SYNTHETIC:
    AudioSetDataset = _SyntheticAudioSet
else:
    # the reference's HDF5 + mp3 reader lives in its own module: it needs h5py and PyAV, neither of which exists in any image
    # this package was built or tested in - importing it says so loudly (see its header) instead of pretending coverage
    AudioSetDataset = _reader().Hdf5AudioSet


class MixupDataset(TorchDataset):
    """Waveform mix-up with probability `rate`, lambda ~ max(Beta, 1 - Beta) (datasets/audioset.py:66-91)."""

    def __init__(self, dataset, beta=2, rate=0.5):
        self.beta, self.rate, self.dataset = beta, rate, dataset
        print(f"Mixing up waveforms from dataset of len {len(dataset)}")

    def __getitem__(self, index):
        if torch.rand(1) < self.rate:
            x1, f1, y1 = self.dataset[index]
            x2, _, y2 = self.dataset[torch.randint(len(self.dataset), (1,)).item()]
            lam = np.random.beta(self.beta, self.beta)
            lam = max(lam, 1.0 - lam)
            x = (x1 - x1.mean()) * lam + (x2 - x2.mean()) * (1.0 - lam)
            return x - x.mean(), f1, y1 * lam + y2 * (1.0 - lam)
        return self.dataset[index]

    def __len__(self):
        return len(self.dataset)


class AddIndexDataset(TorchDataset):
    def __init__(self, ds):
        self.ds = ds

    def __getitem__(self, index):
        x, f, y = self.ds[index]
        return x, f, y, index

    def __len__(self):
        return len(self.ds)


class _RollDataset(TorchDataset):
    """datasets.helpers.audiodatasets.PreprocessDataset(ds, get_roll_func()): random circular shift along time."""

    def __init__(self, ds, shift_range=10000):
        self.ds, self.shift_range = ds, shift_range

    def __getitem__(self, index):
        x, f, y = self.ds[index]
        sf = int(np.random.randint(-self.shift_range, self.shift_range))
        return np.roll(x, sf, axis=1), f, y

    def __len__(self):
        return len(self.ds)


def get_ft_cls_balanced_sample_weights(sample_weight_offset=100, sample_weight_sum=True):
    """Class-balancing sample weights over [balanced, unbalanced] in that order (datasets/audioset.py:185-214)."""
    all_y = torch.as_tensor(np.concatenate([get_base_training_set().targets(), get_unbalanced_training_set().targets()]))
    per_class = sample_weight_offset + all_y.long().sum(0).float().reshape(1, -1)
    if sample_weight_offset > 0:
        print(f"Warning: sample_weight_offset={sample_weight_offset} minnow={per_class.min()}")
    all_weight = all_y * (1000.0 / per_class)
    return all_weight.sum(dim=1) if sample_weight_sum else all_weight.max(dim=1)[0]


def get_ft_weighted_sampler(epoch_len=100000, sampler_replace=False):
    w = get_ft_cls_balanced_sample_weights()
    return WeightedRandomSampler(w, num_samples=min(epoch_len, len(w)) if not sampler_replace else epoch_len,
                                 replacement=sampler_replace)


def get_base_training_set(resample_rate=32000, gain_augment=0):
    return AudioSetDataset(dataset_config["balanced_train_hdf5"], resample_rate=resample_rate, gain_augment=gain_augment)


def get_unbalanced_training_set(resample_rate=32000, gain_augment=0):
    return AudioSetDataset(dataset_config["unbalanced_train_hdf5"], resample_rate=resample_rate, gain_augment=gain_augment)


def get_base_full_training_set(resample_rate=32000, gain_augment=0):
    return ConcatDataset([get_base_training_set(resample_rate=resample_rate, gain_augment=gain_augment),
                          get_unbalanced_training_set(resample_rate=resample_rate, gain_augment=gain_augment)])


def get_base_test_set(resample_rate=32000):
    return AudioSetDataset(dataset_config["eval_hdf5"], resample_rate=resample_rate)


def _wrap(ds, add_index, roll, wavmix):
    if roll:
        ds = _RollDataset(ds)
    if wavmix:
        ds = MixupDataset(ds)
    if add_index:
        ds = AddIndexDataset(ds)
    return ds


def get_training_set(add_index=True, roll=False, wavmix=False, gain_augment=0, resample_rate=32000):
    return _wrap(get_base_training_set(resample_rate=resample_rate, gain_augment=gain_augment), add_index, roll, wavmix)


def get_full_training_set(add_index=True, roll=False, wavmix=False, gain_augment=0, resample_rate=32000):
    return _wrap(get_base_full_training_set(resample_rate=resample_rate, gain_augment=gain_augment), add_index, roll, wavmix)


def get_test_set(resample_rate=32000):
    return get_base_test_set(resample_rate=resample_rate)
