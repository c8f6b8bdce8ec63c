"""Host -> HBM staging of the training / evaluation batches.

The reference's loops move every batch with a blocking `x.to(device)` on the compute stream right before `mel(x)`
(ex_audioset.py:140-141, 303-304): 1.28 MB per clip from pageable DataLoader memory, i.e. a pageable -> pinned bounce and
a serialised copy per step.  At the rates of the HIP path (7 k clips/s training, 65 k clips/s forward on one MI355X) the
copy is the step: a PCIe Gen5 x16 link carries ~40-45 k clips/s of fp32 waveforms at best.

`DevicePrefetcher` wraps any iterable of reference-style batches - tuples / lists whose tensor entries are
`(waveform (B, 1, L) float32, names, target (B, 527) float32[, index (B,) int64])` as produced by
`DataLoader(datasets.audioset.get_training_set(...))` - and keeps `depth` batches in flight:

  * each slot owns PINNED host buffers (allocated once, reused) into which the DataLoader's tensors are copied - the one
    host memcpy that replaces the driver's hidden bounce buffer;
  * the H2D copies run on a dedicated HIP stream, overlapped with the compute stream's kernels of the previous step;
  * `__next__` makes the compute stream wait on the slot's copy event (no host sync) and hands out device tensors; the
    slot is recycled only after the consumer's stream has passed a "released" event (recorded when the consumer asks for
    the next batch), so nothing is overwritten under kernels already enqueued - but the tensors of a batch are only
    valid UNTIL THE NEXT BATCH IS REQUESTED (clone what must outlive the step);
  * non-tensor entries (file names) pass through untouched.

16-bit transport (`Int16Waveform`, `DevicePrefetcher(transport="int16")`): the waveforms cross PCIe as int16 PCM (0.64 MB per
clip instead of 1.28) and become fp32 again on the device (`ops.wave_i16_to_f32`, one streaming kernel; `GraphedKDTrainer`
converts straight into its captured input buffer).  AudioSet's mp3 sources decode to 16-bit PCM, so for un-augmented clips
the round trip is the quantisation the data already had; after waveform-level augmentation (gain, waveform mix-up) it adds
2^-16 of full scale - opt-in for that reason.  The conversion to int16 belongs in the DataLoader WORKERS (`Int16Waveform`
wraps the dataset); the prefetcher converts a float batch itself only as a fallback (one host pass over the batch).

There is no CPU mode: without a GPU it raises, like the rest of the package.
"""
import numpy as np
import torch

from ._lib import EatHipError

I16_SCALE = 32767.0


def to_int16(wave):
    """float waveform in [-1, 1] (numpy or torch) -> int16 PCM, round to nearest, clipped."""
    if torch.is_tensor(wave):
        return (wave.clamp(-1.0, 1.0) * I16_SCALE).round_().to(torch.int16)
    return np.rint(np.clip(wave, -1.0, 1.0) * I16_SCALE).astype(np.int16)


class Int16Waveform(torch.utils.data.Dataset):
    """Dataset wrapper: item[0] (the waveform of the reference's `(wave, name, target[, index])` tuples) as int16 PCM,
    converted in the DataLoader worker that produced the item."""

    def __init__(self, ds):
        self.ds = ds

    def __len__(self):
        return len(self.ds)

    def __getitem__(self, index):
        item = self.ds[index]
        return (to_int16(item[0]),) + tuple(item[1:])


class _Slot:
    def __init__(self):
        self.pinned = {}          # position in the batch tuple -> pinned staging tensor
        self.device = {}          # position -> device tensor
        self.copied = torch.cuda.Event()
        self.released = None      # recorded on the consumer's stream when the batch after this one is requested

    def stage(self, batch, device, stream, transport="fp32"):
        out = list(batch)
        if transport == "int16" and torch.is_tensor(batch[0]) and batch[0].is_floating_point():
            batch = (to_int16(batch[0]),) + tuple(batch[1:])      # fallback: the workers did not convert
        if self.released is not None:
            stream.wait_event(self.released)                      # device buffers of this slot are free again
        for i, item in enumerate(batch):
            if not torch.is_tensor(item):
                continue
            item = item.contiguous()
            if item.is_pinned():
                # already page-locked (DataLoader(pin_memory=True), a pinned ring of the caller): no staging memcpy - the
                # slot keeps the tensor alive until its H2D has completed (`copied` is waited on before the slot is reused)
                dv = self.device.get(i)
                if dv is None or dv.shape != item.shape or dv.dtype != item.dtype:
                    with torch.cuda.stream(stream):
                        self.device[i] = torch.empty(item.shape, dtype=item.dtype, device=device)
                self.copied.synchronize()
                self.pinned[i] = item
                with torch.cuda.stream(stream):
                    self.device[i].copy_(item, non_blocking=True)
                out[i] = self.device[i]
                continue
            pin = self.pinned.get(i)
            if pin is None or pin.shape != item.shape or pin.dtype != item.dtype:
                pin = torch.empty(item.shape, dtype=item.dtype, pin_memory=True)
                self.pinned[i] = pin
                # allocated ON the copy stream: the caching allocator orders a recycled block after the work queued on the
                # stream it is handed out on, so the first copy into it cannot race kernels that still read the block's
                # previous contents (e.g. an unsynchronised training epoch that just ended on the compute stream is
                # ordered by the wait_stream in DevicePrefetcher.__iter__)
                with torch.cuda.stream(stream):
                    self.device[i] = torch.empty(item.shape, dtype=item.dtype, device=device)
            self.copied.synchronize()                             # the previous H2D out of this pinned buffer is done
            pin.copy_(item)
            with torch.cuda.stream(stream):
                self.device[i].copy_(pin, non_blocking=True)
            out[i] = self.device[i]
        self.copied.record(stream)
        return out


class DevicePrefetcher:
    """for wave, names, y, idx in DevicePrefetcher(loader, device): ...   (tensors arrive resident in HBM)."""

    def __init__(self, loader, device="cuda", depth=2, transport="fp32"):
        if not torch.cuda.is_available():
            raise EatHipError("DevicePrefetcher needs a GPU: efficientat_amd has no CPU path")
        if depth < 1:
            raise ValueError("depth must be >= 1")
        if transport not in ("fp32", "int16"):
            raise ValueError("transport must be 'fp32' or 'int16'")
        self.loader, self.device, self.depth, self.transport = loader, torch.device(device), depth, transport
        self.stream = torch.cuda.Stream(device=self.device)
        # depth batches in flight + the one the consumer holds
        self.slots = [_Slot() for _ in range(depth + 1)]

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        it = iter(self.loader)
        queue = []                                                # (slot index, staged batch)
        free = list(range(len(self.slots)))
        held = None                                               # slot whose tensors the consumer is using
        # a new pass may start while kernels of the previous one (or of whatever ran before on the compute stream) still
        # read the slots' device buffers: order every copy of this pass after the work queued so far
        self.stream.wait_stream(torch.cuda.current_stream(self.device))

        def fill():
            while free and len(queue) < self.depth:
                try:
                    batch = next(it)
                except StopIteration:
                    return
                s = free.pop(0)
                queue.append((s, self.slots[s].stage(batch, self.device, self.stream, self.transport)))

        fill()
        while queue:
            if held is not None:                                  # the consumer is done with the previous batch
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                self.slots[held].released = ev
                free.append(held)
            s, staged = queue.pop(0)
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(self.slots[s].copied)
            for t in staged:                                      # allocated on the copy stream, consumed on this one
                if torch.is_tensor(t):
                    t.record_stream(cur)
            held = s
            fill()                                                # next copies overlap with the consumer's kernels
            yield tuple(staged)
        if held is not None:                                      # the last batch of the pass: its slot gets a fresh event too
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.slots[held].released = ev
