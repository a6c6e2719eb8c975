"""Input feed for the MIM / V-JEPA steps: raw volumes travel host -> device in their storage dtype (int16 HU,
4x fewer PCIe bytes than the 335.5 MB fp32 tensor the reference's DataLoader hands to ``.to(device)``), through pinned double-buffered
staging on a copy stream, and the tail of the reference's MONAI pipeline (``src/dataloader/mim.py:154-169, 88-93``: ScaleIntensityRanged
-> SpatialPadd -> CenterSpatialCropd -> PermuteImage) runs as one HIP kernel (``smbv_prepare_volume``) on the device.

    feeder = VolumeFeeder(source, device, image_size=512, depth=320)     # source yields (raw ndarray/tensor [H0,W0,D0] or a
                                                                           # volcache.VolumeRecord, extras)
    for pixel_values, extras in feeder:                                    # pixel_values: fp32 [1, D, 1, H, W] on the device
        out = model(pixel_values=pixel_values, bool_masked_pos=extras["mask"][None].to(device))

While the model works on volume i the host copies volume i+1 into pinned memory, the copy stream uploads it and converts it; the compute
stream only waits on the "volume i+1 is ready" event.  ``depth_slots`` device buffers are cycled, so a buffer is rewritten only after
the consumer's stream has passed an event recorded when the following volume was handed out.  No CPU fallback: a non-HIP device raises.
"""
from __future__ import annotations

import threading
import queue
from typing import Iterable, Iterator, Optional, Tuple

import numpy as np
import torch

from . import _lib

_DTYPES = {torch.int16: 0, torch.uint16: 1, torch.bfloat16: 2, torch.float32: 3}


def prepare_volume(raw: torch.Tensor, image_size: int, depth: int, a_min=-1000.0, a_max=1000.0, b_min=0.0, b_max=1.0, clip=True,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """raw: device tensor [B, H0, W0, D0] (or [H0, W0, D0]) int16 / uint16 / bf16 / fp32 -> fp32 [B, depth, 1, image_size, image_size]."""
    if not raw.is_cuda:
        raise _lib.SmbvError("smb_vision_amd.feed: the raw volume must live on a HIP device; there is no CPU fallback")
    if raw.dtype not in _DTYPES:
        raise ValueError(f"unsupported raw dtype {raw.dtype} (int16, uint16, bfloat16, float32)")
    if raw.dim() == 3:
        raw = raw[None]
    raw = raw.contiguous()
    B, H0, W0, D0 = raw.shape
    if out is None:
        out = torch.empty(B, depth, 1, image_size, image_size, dtype=torch.float32, device=raw.device)
    lib = _lib.load()
    with torch.cuda.device(raw.device):
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.smbv_prepare_volume(raw.data_ptr(), _DTYPES[raw.dtype], out.data_ptr(), B, H0, W0, D0, image_size, image_size, depth,
                                           float(a_min), float(a_max), float(b_min), float(b_max), int(bool(clip)), stream), "smbv_prepare_volume")
    return out


class VolumeFeeder:
    """Iterates ``source`` (an iterable of ``(raw, extras)``; raw: numpy array or CPU tensor [H0, W0, D0]) and yields
    ``(pixel_values fp32 [1, depth, 1, image_size, image_size] on device, extras)`` with upload + conversion overlapped with the consumer.

    Three stages, each one volume ahead of the next: a host thread copies raw volume i+2 into a ring of pinned buffers; the copy stream
    uploads volume i+1 (in its storage dtype) and converts it; the consumer's stream works on volume i."""

    def __init__(self, source: Iterable, device, image_size: int, depth: int, a_min=-1000.0, a_max=1000.0, b_min=0.0, b_max=1.0, clip=True,
                 slots: int = 2, prefetch: int = 2):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.SmbvError("smb_vision_amd.feed: VolumeFeeder needs a HIP device; there is no CPU fallback")
        _lib.load()
        self.source, self.image_size, self.depth = source, image_size, depth
        self.scale = (a_min, a_max, b_min, b_max, clip)
        self.slots = max(2, slots)            # device staging + output buffers
        self.prefetch = max(1, prefetch)      # raw volumes the host thread may run ahead
        self.nhost = self.prefetch + 2        # pinned buffers (see _producer for why +2)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._pinned = [None] * self.nhost
        self._borrowed = [False] * self.nhost
        self._h2d_done = [None] * self.nhost  # event after the upload that last read pinned buffer k
        self._staging, self._out, self._free_ev = [None] * self.slots, [None] * self.slots, [None] * self.slots

    def _producer(self, q: "queue.Queue", stop: threading.Event):
        # Item i goes to pinned buffer i % nhost.  The queue holds at most `prefetch` items, so when item i is written, item
        # i - nhost = i - prefetch - 2 has been taken off the queue and its upload enqueued (self._h2d_done[k] set); wait for that upload.
        try:
            for i, (raw, extras) in enumerate(self.source):
                if stop.is_set():
                    return
                k = i % self.nhost
                if hasattr(raw, "read_into"):  # a record that fills a host buffer itself (volcache.VolumeRecord): no intermediate copy
                    if raw.dtype not in _DTYPES:
                        raise ValueError(f"unsupported raw dtype {raw.dtype} (int16, uint16, bfloat16, float32)")
                    if self._pinned[k] is None or tuple(self._pinned[k].shape) != tuple(raw.shape) or self._pinned[k].dtype != raw.dtype or self._borrowed[k]:
                        self._pinned[k] = torch.empty(tuple(raw.shape), dtype=raw.dtype).pin_memory()
                    ev = self._h2d_done[k]
                    if ev is not None:
                        ev.synchronize()
                    raw.read_into(self._pinned[k])
                    self._borrowed[k] = False
                    q.put((k, extras))
                    continue
                t = torch.from_numpy(np.ascontiguousarray(raw)) if isinstance(raw, np.ndarray) else raw.contiguous()
                if t.dtype not in _DTYPES:
                    raise ValueError(f"unsupported raw dtype {t.dtype} (int16, uint16, bfloat16, float32)")
                if t.is_pinned():  # a loader that already reads into page-locked memory: uploaded from where it is (the loader owns its reuse)
                    self._pinned[k] = t
                else:
                    if self._pinned[k] is None or self._pinned[k].shape != t.shape or self._pinned[k].dtype != t.dtype or self._borrowed[k]:
                        self._pinned[k] = torch.empty(t.shape, dtype=t.dtype).pin_memory()
                    ev = self._h2d_done[k]
                    if ev is not None:
                        ev.synchronize()
                    self._pinned[k].copy_(t)
                self._borrowed[k] = t.is_pinned()
                q.put((k, extras))
        except BaseException as e:  # surfaced in the consumer
            q.put(e)
        finally:
            q.put(None)

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, object]]:
        q: "queue.Queue" = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()
        th = threading.Thread(target=self._producer, args=(q, stop), daemon=True)
        th.start()
        try:
            pending = self._launch(q.get(), 0)
            i = 0
            while pending is not None:
                out, extras, ready, slot = pending
                torch.cuda.current_stream(self.device).wait_event(ready)  # the consumer's stream, not the host, waits for upload + conversion
                nxt_item = q.get()
                err = nxt_item if isinstance(nxt_item, BaseException) else None  # a failing source surfaces at ITS item, not one early
                nxt = self._launch(nxt_item, i + 1) if (nxt_item is not None and err is None) else None
                yield out, extras
                # everything the consumer enqueued for this volume precedes this event: the slot's buffers may be rewritten after it
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                self._free_ev[slot] = ev
                if err is not None:
                    raise err
                pending = nxt
                i += 1
        finally:
            stop.set()
            while th.is_alive():
                try:
                    q.get_nowait()
                except queue.Empty:
                    th.join(timeout=0.05)

    def _launch(self, item, i):
        if item is None:
            return None
        if isinstance(item, BaseException):
            raise item
        k, extras = item
        host = self._pinned[k]
        slot = i % self.slots
        if self._staging[slot] is None or self._staging[slot].shape != host.shape or self._staging[slot].dtype != host.dtype:
            self._staging[slot] = torch.empty(host.shape, dtype=host.dtype, device=self.device)
            self._out[slot] = torch.empty(1, self.depth, 1, self.image_size, self.image_size, dtype=torch.float32, device=self.device)
        with torch.cuda.stream(self.copy_stream):
            if self._free_ev[slot] is not None:
                self.copy_stream.wait_event(self._free_ev[slot])  # the consumer of the volume that used this slot is done with it
            self._staging[slot].copy_(host, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.copy_stream)
            self._h2d_done[k] = done
            a_min, a_max, b_min, b_max, clip = self.scale
            prepare_volume(self._staging[slot], self.image_size, self.depth, a_min, a_max, b_min, b_max, clip, out=self._out[slot])
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        return self._out[slot], extras, ready, slot
