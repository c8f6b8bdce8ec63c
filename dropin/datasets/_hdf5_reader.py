"""QUARANTINED: the reference's AudioSet reader (datasets/audioset.py:32-47, 106-177) - HDF5 rows + mp3 decode.

STATUS: this module has NEVER been executed in the environments efficientat_amd was built and tested in: neither `h5py` nor
`av` (PyAV) is installed there and no AudioSet file exists (dataset I/O is outside the hot path).
It is a restatement of the reference's reader kept so that `datasets.audioset` has the same public surface; importing it
prints this status once.  `tests/test_host_cpu.py::test_hdf5_reader_round_trip` builds a 3-clip HDF5 + mp3 file and reads it
back - it runs (instead of skipping) on any machine that has both libraries; until it has passed somewhere, treat the file
as unverified.  Nothing in the benchmarks, the GPU tests or the training programs imports it unless EAT_AUDIOSET_DIR points at
real files."""
import io
import sys

import numpy as np
from torch.utils.data import Dataset as TorchDataset

print("[datasets.audioset] using the HDF5 + mp3 AudioSet reader (dropin/datasets/_hdf5_reader.py): UNVERIFIED code path - "
      "never executed where this package was built (no h5py / PyAV there); see the module header", file=sys.stderr)


def decode_mp3(mp3_arr):
    """uint8 array holding one mp3 file -> float32 waveform (PyAV; datasets/audioset.py:32-47)."""
    import av
    container = av.open(io.BytesIO(mp3_arr.tobytes()))
    stream = next(s for s in container.streams if s.type == "audio")
    chunks = [frame.to_ndarray().reshape(-1) for packet in container.demux(stream) for frame in packet.decode()]
    waveform = np.concatenate(chunks)
    if waveform.dtype != np.float32:
        raise RuntimeError("Unexpected wave type")
    return waveform



class Hdf5AudioSet(TorchDataset):
    """The reference's reader (datasets/audioset.py:106-177): one HDF5 file with the rows `audio_name` (bytes), `mp3`
    (variable-length uint8) and `target` (527 labels packed into 66 bytes); the file handle is opened lazily so that
    every DataLoader worker gets its own."""

    def __init__(self, hdf5_file, sample_rate=32000, resample_rate=32000, classes_num=527, clip_length=10, in_mem=False,
                 gain_augment=0):
        import h5py
        self.sample_rate, self.resample_rate = sample_rate, resample_rate
        self.hdf5_file = hdf5_file
        if in_mem:
            print("\nPreloading in memory\n")
            with open(hdf5_file, "rb") as f:
                self.hdf5_file = io.BytesIO(f.read())
        with h5py.File(hdf5_file, "r") as f:
            self.length = len(f["audio_name"])
        print(f"Dataset from {hdf5_file} with length {self.length}.")
        self.dataset_file = None
        self.clip_length = clip_length * sample_rate
        self.classes_num, self.gain_augment = classes_num, gain_augment

    def __len__(self):
        return self.length

    def __del__(self):
        if getattr(self, "dataset_file", None) is not None:
            self.dataset_file.close()
            self.dataset_file = None

    def _file(self):
        if self.dataset_file is None:
            import h5py
            self.dataset_file = h5py.File(self.hdf5_file, "r")
        return self.dataset_file

    def targets(self):
        """(length, classes_num) float32 label matrix (for the class-balancing sampler)."""
        return np.unpackbits(self._file()["target"][:], axis=-1, count=self.classes_num).astype(np.float32)

    def __getitem__(self, index):
        f = self._file()
        # stored names look like "Y<youtube id>.mp3": back to the official file name
        audio_name = f["audio_name"][index].decode().replace(".mp3", "").split("Y", 1)[1]
        # (pad_or_truncate / pydub_augment: injected by datasets.audioset._reader - the helpers live there)
        waveform = pad_or_truncate(pydub_augment(decode_mp3(f["mp3"][index]), self.gain_augment), self.clip_length)
        target = np.unpackbits(f["target"][index], axis=-1, count=self.classes_num).astype(np.float32)
        return self.resample(waveform).reshape(1, -1), audio_name, target

    def resample(self, waveform):
        if self.resample_rate == 32000:
            return waveform
        if self.resample_rate == 16000:
            return waveform[0::2]
        if self.resample_rate == 8000:
            return waveform[0::4]
        raise Exception("Incorrect sample rate!")


