"""Cached-volume file format: the on-disk format on the input side of the path.

The reference caches pre-processed samples as MONAI ``PersistentDataset`` ``.pt`` pickles (``src/dataloader/load.py:95-138``,
``src/dataloader/mim.py:226-235``): a pickled dict holding the 335.5 MB fp32 ``(D, 1, H, W)`` tensor - unpickled, copied and re-pinned
for every sample.  This format is flat and needs no deserialiser:

    offset 0     : 4096-byte header = magic b"SMBVOL01" + little-endian fields (struct ``HEADER``): dtype code (0 int16, 1 uint16,
                   2 bf16, 3 fp32 - the codes of ``smbv_prepare_volume``), shape (H0, W0, D0), voxel spacing, the intensity window
                   (a_min, a_max, b_min, b_max, clip) the volume is meant to be scaled with, number of mask tokens, byte offsets
    offset 4096  : the raw volume, C order (H0, W0, D0) - what EnsureChannelFirstd / Orientationd / Spacingd leave, BEFORE intensity
                   scaling, padding, cropping and permuting: those run on the device (``smbv_prepare_volume``), so the file holds
                   2 bytes per voxel and any (image_size, depth) can be cut from it later
    after it     : optional MIM mask, one bit per token (page aligned)

``VolumeRecord.read_into(pinned)`` fills a page-locked buffer straight from the file (one ``readinto``; no intermediate tensor), which is
what ``VolumeFeeder`` calls on its own pinned ring; ``VolumeRecord.array()`` is a zero-copy ``numpy.memmap`` view.

Masks.  In the reference ``GenerateMask`` is a deterministic MONAI ``Transform`` (``mim.py:72``), so ``PersistentDataset`` caches its output
with the image: every sample keeps ONE fixed mask for all epochs (a side effect of caching, not a modelling choice).  This format makes
the choice explicit: ``write_volume(..., mask=None)`` (default) stores no mask and the loader draws a fresh one per visit, as MIM
pre-training intends; ``mask=bool[N]`` stores it and ``CachedVolumeSource`` returns it, reproducing the reference's behaviour.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass
from typing import Iterable, Iterator, Optional, Sequence, Tuple

import numpy as np
import torch

MAGIC = b"SMBVOL01"
HEADER_BYTES = 4096
HEADER = struct.Struct("<8sIIQQQfffffffIIQQ")  # magic, version, dtype, H0, W0, D0, spacing[3], a_min, a_max, b_min, b_max, clip, n_mask, data_off, mask_off
_NP = {0: np.dtype("<i2"), 1: np.dtype("<u2"), 2: np.dtype("<u2"), 3: np.dtype("<f4")}
_TORCH = {0: torch.int16, 1: torch.uint16, 2: torch.bfloat16, 3: torch.float32}
_CODE = {np.dtype("int16"): 0, np.dtype("uint16"): 1, np.dtype("float32"): 3}


def _page_up(n: int) -> int:
    return (n + 4095) & ~4095


def write_volume(path: str, volume, spacing=(1.5, 1.5, 3.0), a_min=-1000.0, a_max=1000.0, b_min=0.0, b_max=1.0, clip=True,
                 mask: Optional[np.ndarray] = None) -> None:
    """volume: numpy array or CPU tensor [H0, W0, D0] of int16 / uint16 / float32 / (torch) bfloat16."""
    if isinstance(volume, torch.Tensor):
        if volume.dtype == torch.bfloat16:
            code, arr = 2, volume.contiguous().view(torch.int16).numpy().view(np.uint16)
        else:
            arr = volume.contiguous().numpy()
            code = _CODE.get(arr.dtype)
    else:
        arr = np.ascontiguousarray(volume)
        code = _CODE.get(arr.dtype)
    if code is None or arr.ndim != 3:
        raise ValueError("write_volume: need a 3-D int16 / uint16 / float32 / bfloat16 volume")
    data_off = HEADER_BYTES
    mask_off, n_mask = 0, 0
    if mask is not None:
        mask = np.asarray(mask).astype(bool).ravel()
        n_mask = mask.size
        mask_off = _page_up(data_off + arr.nbytes)
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        hdr = HEADER.pack(MAGIC, 1, code, arr.shape[0], arr.shape[1], arr.shape[2], *[float(x) for x in spacing], float(a_min), float(a_max),
                          float(b_min), float(b_max), int(bool(clip)), n_mask, data_off, mask_off)
        f.write(hdr + b"\0" * (HEADER_BYTES - len(hdr)))
        f.write(arr.astype(_NP[code], copy=False).tobytes() if arr.dtype != _NP[code] else memoryview(arr).cast("B"))
        if mask is not None:
            f.write(b"\0" * (mask_off - data_off - arr.nbytes))
            f.write(np.packbits(mask, bitorder="little").tobytes())
    os.replace(tmp, path)  # readers never see a half-written file


@dataclass
class VolumeRecord:
    path: str
    code: int
    shape: Tuple[int, int, int]
    spacing: Tuple[float, float, float]
    window: Tuple[float, float, float, float, bool]
    n_mask: int
    data_off: int
    mask_off: int

    @property
    def dtype(self) -> torch.dtype:
        return _TORCH[self.code]

    @property
    def nbytes(self) -> int:
        return int(np.prod(self.shape)) * _NP[self.code].itemsize

    def array(self) -> np.ndarray:
        """Zero-copy view of the raw volume (bf16 volumes come back as their uint16 bit patterns)."""
        return np.memmap(self.path, dtype=_NP[self.code], mode="r", offset=self.data_off, shape=self.shape)

    def read_into(self, dst: torch.Tensor) -> torch.Tensor:
        """Fill ``dst`` (CPU tensor of this record's shape and dtype, ideally page-locked) straight from the file."""
        if tuple(dst.shape) != tuple(self.shape) or dst.dtype != self.dtype or not dst.is_contiguous() or dst.is_cuda:
            raise ValueError(f"read_into: need a contiguous CPU {self.dtype} tensor of shape {self.shape}")
        buf = memoryview(dst.view(torch.int16 if self.code in (1, 2) else dst.dtype).numpy()).cast("B")
        with open(self.path, "rb", buffering=0) as f:
            f.seek(self.data_off)
            got = 0
            while got < self.nbytes:
                n = f.readinto(buf[got:])
                if not n:
                    raise IOError(f"{self.path}: truncated volume ({got} of {self.nbytes} bytes)")
                got += n
        return dst

    def mask(self) -> Optional[torch.Tensor]:
        if not self.n_mask:
            return None
        with open(self.path, "rb") as f:
            f.seek(self.mask_off)
            bits = np.frombuffer(f.read((self.n_mask + 7) // 8), dtype=np.uint8)
        return torch.from_numpy(np.unpackbits(bits, bitorder="little")[: self.n_mask].astype(bool))


def open_volume(path: str) -> VolumeRecord:
    with open(path, "rb") as f:
        raw = f.read(HEADER.size)
    if len(raw) < HEADER.size:
        raise ValueError(f"{path}: not a cached volume (file too short)")
    magic, version, code, h, w, d, s0, s1, s2, a_min, a_max, b_min, b_max, clip, n_mask, data_off, mask_off = HEADER.unpack(raw)
    if magic != MAGIC:
        raise ValueError(f"{path}: not a cached volume (bad magic)")
    if version != 1 or code not in _NP:
        raise ValueError(f"{path}: unsupported version {version} / dtype code {code}")
    rec = VolumeRecord(path, code, (int(h), int(w), int(d)), (s0, s1, s2), (a_min, a_max, b_min, b_max, bool(clip)), int(n_mask), int(data_off), int(mask_off))
    if os.path.getsize(path) < rec.data_off + rec.nbytes:
        raise ValueError(f"{path}: truncated volume")
    return rec


class CachedVolumeSource:
    """Iterable over cached volumes for ``VolumeFeeder``: yields ``(VolumeRecord, {"path", "mask"})``; the feeder reads each record
    straight into its own pinned ring (``read_into``), so the volume is touched once on the host."""

    def __init__(self, paths: Sequence[str], epochs: Optional[int] = 1, mask_generator=None):
        self.paths, self.epochs, self.mask_generator = list(paths), epochs, mask_generator

    def __iter__(self) -> Iterator:
        e = 0
        while self.epochs is None or e < self.epochs:
            for p in self.paths:
                rec = open_volume(p)
                m = rec.mask()
                if m is None and self.mask_generator is not None:
                    m = self.mask_generator()
                yield rec, {"path": p, "mask": m}
            e += 1


class CachedVolumeDataset(torch.utils.data.Dataset):
    """Map-style dataset over a directory (or list) of ``.smbvol`` files: the role of ``MIMDataset(json_path, cache_dir).setup("train")``
    (``src/run_mim.py:279-290``, ``src/dataloader/load.py:70-142``) for the harnesses' ``--volume_cache_dir``.  An item is
    ``{"volume": VolumeRecord, "mask": bool[N]}`` - the RECORD, not the voxels: the trainer's loader (``trainer.VolumeFeedLoader``) reads the file straight
    into the feeder's page-locked ring and scales / pads / crops / permutes on the device, so a DataLoader worker never touches (or pickles) 168-335 MB.

    Masks: the one the file stores if it stores one (the reference's cached-mask behaviour); else, with ``mask_generator``, a draw from the legacy numpy
    stream seeded with ``mask_seed + index`` (a fixed mask per sample as well - and what lets a resumed run see the uninterrupted run's batches), or a
    fresh draw from the GLOBAL stream per visit when ``mask_seed`` is None.  ``item_transform`` (e.g. ``VJEPAMaskGenerator``) is applied to the item last."""

    def __init__(self, source, mask_generator=None, mask_seed: Optional[int] = 0, item_transform=None):
        if isinstance(source, (str, os.PathLike)):
            d = os.fspath(source)
            if not os.path.isdir(d):
                raise FileNotFoundError(f"volume cache directory {d!r} does not exist")
            self.paths = sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".smbvol"))
            if not self.paths:
                raise FileNotFoundError(f"no .smbvol files in {d!r}")
        else:
            self.paths = [os.fspath(p) for p in source]
        self.mask_generator, self.mask_seed, self.item_transform = mask_generator, mask_seed, item_transform

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, i):
        rec = open_volume(self.paths[i])
        item = {"volume": rec}
        m = rec.mask()
        if m is None and self.mask_generator is not None:
            if self.mask_seed is None:
                m = self.mask_generator()
            else:
                state = np.random.get_state()
                np.random.seed((self.mask_seed + i) % (2 ** 32))
                try:
                    m = self.mask_generator()
                finally:
                    np.random.set_state(state)
        if m is not None:
            item["mask"] = m
        return self.item_transform(item) if self.item_transform is not None else item
