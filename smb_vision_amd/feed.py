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
    """Iterates ``source`` (an iterable of ``(raw, extras)``; raw: numpy array, CPU tensor [H0, W0, D0] or ``volcache.VolumeRecord`` - or a LIST of B of
    them, one batch) and yields ``(pixel_values fp32 [B, depth, 1, image_size, image_size] on device, extras)`` with upload + conversion overlapped
    with the consumer.

    Three stages, each one volume ahead of the next: a host thread copies raw volume i+2 into a ring of pinned buffers; the copy stream
    uploads volume i+1 (in its storage dtype) and converts it; the consumer's stream works on volume i."""

    def __init__(self, source: Iterable, device, image_size: int, depth: int, a_min=-1000.0, a_max=1000.0, b_min=0.0, b_max=1.0, clip=True,
                 slots: int = 2, prefetch: int = 2, record_window: bool = False, fresh_outputs: bool = False):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.SmbvError("smb_vision_amd.feed: VolumeFeeder needs a HIP device; there is no CPU fallback")
        _lib.load()
        self.source, self.image_size, self.depth = source, image_size, depth
        self.scale = (a_min, a_max, b_min, b_max, clip)
        self.record_window = record_window    # scale a volcache.VolumeRecord with the intensity window its header stores instead of the arguments above
        # fresh_outputs: every item gets a NEW output tensor from the caching allocator (handed over with record_stream) instead of a cycled slot - for
        # consumers that hold several items at once (HF Trainer fetches all micro-batches of an accumulation step before it computes the first)
        self.fresh_outputs = fresh_outputs
        self.slots = max(2, slots)            # device staging + output buffers
        self.prefetch = max(1, prefetch)      # raw volumes the host thread may run ahead
        self.nhost = self.prefetch + 2        # pinned buffers (see _producer for why +2)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._pinned = [[] for _ in range(self.nhost)]       # per ring entry: one page-locked tensor per volume of the batch
        self._borrowed = [[] for _ in range(self.nhost)]
        self._h2d_done = [None] * self.nhost  # event after the upload that last read pinned buffer k
        self._staging, self._out, self._free_ev = [None] * self.slots, [None] * self.slots, [None] * self.slots

    def _fill(self, k: int, b: int, raw):
        """Put one raw volume into pinned buffer (k, b); returns the page-locked tensor the upload will read."""
        cur = self._pinned[k][b] if b < len(self._pinned[k]) else None
        borrowed = self._borrowed[k][b] if b < len(self._borrowed[k]) else False
        if hasattr(raw, "read_into"):  # a record that fills a host buffer itself (volcache.VolumeRecord): no intermediate copy
            if raw.dtype not in _DTYPES:
                raise ValueError(f"unsupported raw dtype {raw.dtype} (int16, uint16, bfloat16, float32)")
            if cur is None or tuple(cur.shape) != tuple(raw.shape) or cur.dtype != raw.dtype or borrowed:
                cur = torch.empty(tuple(raw.shape), dtype=raw.dtype).pin_memory()
            raw.read_into(cur)
            return cur, False
        t = torch.from_numpy(np.ascontiguousarray(raw)) if isinstance(raw, np.ndarray) else raw.contiguous()
        if t.dtype not in _DTYPES:
            raise ValueError(f"unsupported raw dtype {t.dtype} (int16, uint16, bfloat16, float32)")
        if t.dim() != 3:
            raise ValueError(f"a raw volume is [H0, W0, D0]; got shape {tuple(t.shape)}")
        if t.is_pinned():  # a loader that already reads into page-locked memory: uploaded from where it is (the loader owns its reuse)
            return t, True
        if cur is None or cur.shape != t.shape or cur.dtype != t.dtype or borrowed:
            cur = torch.empty(t.shape, dtype=t.dtype).pin_memory()
        cur.copy_(t)
        return cur, False

    def _producer(self, q: "queue.Queue", stop: threading.Event):
        # Item i goes to pinned buffer set i % nhost.  The queue holds at most `prefetch` items, so when item i is written, item
        # i - nhost = i - prefetch - 2 has been taken off the queue and its upload enqueued (self._h2d_done[k] set); wait for that upload.
        # An item is ONE raw volume or a LIST of them (a batch: the volumes may differ in shape and dtype, the step receives [B, depth, 1, H, W]).
        try:
            # a new thread's current device is 0: under one process per GPU every rank > 0 would otherwise page-lock its buffers through, and create a
            # context on, GPU 0
            if self.device.type == "cuda":
                torch.cuda.set_device(self.device)
            for i, (raw, extras) in enumerate(self.source):
                if stop.is_set():
                    return
                k = i % self.nhost
                raws = list(raw) if isinstance(raw, (list, tuple)) else [raw]
                if not raws:
                    raise ValueError("VolumeFeeder: an empty batch of raw volumes")
                ev = self._h2d_done[k]
                if ev is not None:
                    ev.synchronize()
                bufs, borrowed, windows = [], [], []
                for b, r in enumerate(raws):
                    t, bo = self._fill(k, b, r)
                    bufs.append(t)
                    borrowed.append(bo)
                    windows.append(tuple(r.window) if (self.record_window and hasattr(r, "window")) else self.scale)
                self._pinned[k], self._borrowed[k] = bufs, borrowed
                q.put((k, extras, windows))
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
                if self.fresh_outputs:
                    out.record_stream(torch.cuda.current_stream(self.device))
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
        k, extras, windows = item
        hosts = self._pinned[k]
        slot = i % self.slots
        B = len(hosts)
        st = self._staging[slot]
        if st is None or len(st) != B or any(a.shape != h.shape or a.dtype != h.dtype for a, h in zip(st, hosts)):
            self._staging[slot] = [torch.empty(h.shape, dtype=h.dtype, device=self.device) for h in hosts]
        if not self.fresh_outputs and (self._out[slot] is None or self._out[slot].shape[0] != B):
            self._out[slot] = torch.empty(B, self.depth, 1, self.image_size, self.image_size, dtype=torch.float32, device=self.device)
        with torch.cuda.stream(self.copy_stream):
            if self.fresh_outputs:
                self._out[slot] = torch.empty(B, self.depth, 1, self.image_size, self.image_size, dtype=torch.float32, device=self.device)
            if self._free_ev[slot] is not None:
                self.copy_stream.wait_event(self._free_ev[slot])  # the consumer of the volume that used this slot is done with it
            for st, host in zip(self._staging[slot], hosts):
                st.copy_(host, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.copy_stream)
            self._h2d_done[k] = done
            for b, (st, (a_min, a_max, b_min, b_max, clip)) in enumerate(zip(self._staging[slot], windows)):
                prepare_volume(st, self.image_size, self.depth, a_min, a_max, b_min, b_max, clip, out=self._out[slot][b:b + 1])
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        out = self._out[slot]
        if self.fresh_outputs:
            self._out[slot] = None
        return out, extras, ready, slot


RAW_KEY = "smbv_raw_volumes"   # batch key under which a collate function ships RAW volumes (tensors / arrays / volcache.VolumeRecord) instead of pixel values


def raw_batch_to_device(raws, device, image_size: int, depth: int, window=(-1000.0, 1000.0, 0.0, 1.0, True), record_window: bool = True) -> torch.Tensor:
    """The feeder's work for ONE batch, inline on the current stream (no ring, no overlap): what evaluation and the first partial epoch of a resumed run
    use, where HF hands the step a plain DataLoader's batch.  Same device kernel, same bits as ``VolumeFeeder``."""
    device = torch.device(device)
    out = torch.empty(len(raws), depth, 1, image_size, image_size, dtype=torch.float32, device=device)
    for b, r in enumerate(raws):
        w = tuple(r.window) if (record_window and hasattr(r, "window")) else window
        if hasattr(r, "read_into"):
            t = r.read_into(torch.empty(tuple(r.shape), dtype=r.dtype))
        else:
            t = torch.from_numpy(np.ascontiguousarray(r)) if isinstance(r, np.ndarray) else r
        prepare_volume(t.to(device), image_size, depth, *w, out=out[b:b + 1])
    return out
