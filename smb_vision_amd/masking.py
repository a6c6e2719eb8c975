"""Host-side block-mask generator for MIM pre-training (product code; stays on the host with numpy so that the random
stream is bit-identical to the reference).

Mirrors ``MaskGenerator`` of the reference (``src/dataloader/mim.py:25-69``, duplicate ``src/dataloader/transforms.py:20-63``):
same constructor arguments, same errors, same draws from the GLOBAL legacy numpy RNG, same output (``torch.bool [N]`` in
(D',H',W') row-major token order).  ``GenerateMask`` / ``PermuteImage`` mirror ``mim.py:72-93`` without the MONAI base class.
"""
from __future__ import annotations

import numpy as np
import torch


class MaskGenerator:
    def __init__(self, input_size=224, depth=96, mask_patch_size=32, model_patch_size=16, mask_ratio=0.6):
        self.input_size = input_size
        self.depth = depth
        self.mask_patch_size = mask_patch_size
        self.model_patch_size = model_patch_size
        self.mask_ratio = mask_ratio
        if self.input_size % self.mask_patch_size != 0:
            raise ValueError("Input size must be divisible by mask patch size")
        if self.depth % self.mask_patch_size != 0:
            raise ValueError("Depth must be divisible by mask patch size")
        if self.mask_patch_size % self.model_patch_size != 0:
            raise ValueError("Mask patch size must be divisible by model patch size")
        self.rand_size = self.input_size // self.mask_patch_size
        self.rand_depth = self.depth // self.mask_patch_size
        self.scale = self.mask_patch_size // self.model_patch_size
        self.token_count = self.rand_size ** 2 * self.rand_depth
        self.mask_count = int(np.ceil(self.token_count * self.mask_ratio))

    @property
    def num_tokens(self) -> int:
        return self.token_count * self.scale ** 3

    @property
    def num_masked(self) -> int:
        return self.mask_count * self.scale ** 3

    def __call__(self) -> torch.Tensor:
        coarse = np.zeros(self.token_count, dtype=bool)
        coarse[np.random.permutation(self.token_count)[: self.mask_count]] = True
        coarse = coarse.reshape(self.rand_depth, self.rand_size, self.rand_size)
        s = self.scale
        fine = np.broadcast_to(coarse[:, None, :, None, :, None], (self.rand_depth, s, self.rand_size, s, self.rand_size, s))
        return torch.from_numpy(np.array(fine).reshape(-1))


class GenerateMask:
    """``inputs["mask"] = MaskGenerator(...)()`` (reference ``mim.py:72-85``)."""

    def __init__(self, input_size=224, depth=96, mask_patch_size=32, model_patch_size=16, mask_ratio=0.75):
        self.mask_generator = MaskGenerator(input_size, depth, mask_patch_size, model_patch_size, mask_ratio)

    def __call__(self, inputs):
        inputs["mask"] = self.mask_generator()
        return inputs


class PermuteImage:
    """``(C,H,W,D) -> (D,C,H,W)`` (reference ``mim.py:88-93``)."""

    def __call__(self, data):
        data["image"] = data["image"].permute(3, 0, 1, 2)
        return data


def split_mask_indices(bool_masked_pos: torch.Tensor, num_masked: int):
    """bool [B,N] -> (vis_idx int32 [B,N-num_masked], mask_idx int32 [B,num_masked]), each ascending — the order
    ``embeddings[~bool_masked_pos]`` / ``[bool_masked_pos]`` produce (reference modeling_videomae.py:134-137, :811-812).
    Stable sort of the 0/1 key: no host synchronisation, works on any device."""
    order = torch.argsort(bool_masked_pos.to(torch.uint8), dim=1, stable=True)
    n_vis = bool_masked_pos.shape[1] - num_masked
    return order[:, :n_vis].to(torch.int32).contiguous(), order[:, n_vis:].to(torch.int32).contiguous()
