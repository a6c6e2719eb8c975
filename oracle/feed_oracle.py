"""CPU restatement of the per-volume tail of the reference's MIM data pipeline - TEST INFRASTRUCTURE ONLY (imported by tests/, never by
smb_vision_amd/).

Follows ``src/dataloader/mim.py:154-169`` (ScaleIntensityRanged(a_min=-1000, a_max=1000, b_min=0, b_max=1, clip=True) -> SpatialPadd
(img_size, img_size, depth) -> CenterSpatialCropd(same)) and ``:88-93`` (PermuteImage: (C, H, W, D) -> (D, C, H, W)).

PARITY UNPINNED for the MONAI-specific conventions: ``monai`` is a third-party dependency (reference pin ``monai>=1.4.0``,
pyproject.toml) that is absent from the build image and cannot be installed, and the reference's own tests hold no fixture for these
transforms.  The three conventions below restate MONAI's published behaviour:
  * ScaleIntensityRange: ``img = (img - a_min) / (a_max - a_min)``; ``img = img * (b_max - b_min) + b_min``; ``clip(img, b_min, b_max)``
    when clip; float32 throughout.
  * SpatialPad(method="symmetric", mode="constant"): per axis ``pad = max(target - size, 0)``, ``pad // 2`` voxels of 0 in front, the
    rest behind (the pad runs AFTER the intensity scaling, so padded voxels are 0.0, not b_min-mapped air).
  * CenterSpatialCrop(roi): centre ``size // 2``, start ``size // 2 - roi // 2``, length ``roi`` (axes shorter than roi were padded first).
"""
from __future__ import annotations

import numpy as np


def scale_intensity_range(x: np.ndarray, a_min=-1000.0, a_max=1000.0, b_min=0.0, b_max=1.0, clip=True) -> np.ndarray:
    v = x.astype(np.float32)
    v = (v - np.float32(a_min)) / np.float32(np.float32(a_max) - np.float32(a_min))
    v = v * np.float32(np.float32(b_max) - np.float32(b_min)) + np.float32(b_min)
    if clip:
        v = np.clip(v, np.float32(b_min), np.float32(b_max))
    return v.astype(np.float32)


def spatial_pad_symmetric(x: np.ndarray, target) -> np.ndarray:
    """x: (C, H, W, D); target: (H, W, D)."""
    widths = [(0, 0)]
    for size, tgt in zip(x.shape[1:], target):
        pad = max(tgt - size, 0)
        widths.append((pad // 2, pad - pad // 2))
    return np.pad(x, widths, mode="constant", constant_values=0)


def center_spatial_crop(x: np.ndarray, roi) -> np.ndarray:
    sl = [slice(None)]
    for size, r in zip(x.shape[1:], roi):
        start = max(size // 2 - r // 2, 0)
        sl.append(slice(start, start + r))
    return x[tuple(sl)]


def permute_image(x: np.ndarray) -> np.ndarray:
    """mim.py:88-93: (C, H, W, D) -> (D, C, H, W)."""
    return np.ascontiguousarray(np.transpose(x, (3, 0, 1, 2)))


def prepare_volume(raw: np.ndarray, img_size: int, depth: int, a_min=-1000.0, a_max=1000.0, b_min=0.0, b_max=1.0, clip=True) -> np.ndarray:
    """raw: (H0, W0, D0) of any numeric dtype (bf16 passed as float32 of the rounded values) -> float32 (depth, 1, img_size, img_size)."""
    x = scale_intensity_range(raw[None], a_min, a_max, b_min, b_max, clip)
    x = spatial_pad_symmetric(x, (img_size, img_size, depth))
    x = center_spatial_crop(x, (img_size, img_size, depth))
    return permute_image(x)
