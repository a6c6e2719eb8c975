"""Host-side mask generators (MIM block masks, V-JEPA context/target index masks) (product code; stays on the host with numpy so that the random
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


def split_mask_indices_host(bool_masked_pos) -> "tuple[torch.Tensor, torch.Tensor]":
    """The same two ascending index lists, built on the HOST from the mask the data pipeline just generated (a CPU bool tensor / numpy array
    [B,N]): two ``flatnonzero`` calls per sample instead of a device sort per step.  Returns CPU int32 tensors ``(vis_idx [B,N-M], mask_idx [B,M])``;
    raises if the samples mask different numbers of tokens (reference: ``embeddings[~mask].reshape(B, -1, C)`` would fail the same way)."""
    m = bool_masked_pos.numpy() if torch.is_tensor(bool_masked_pos) else np.asarray(bool_masked_pos)
    m = m.astype(bool, copy=False)
    if m.ndim == 1:
        m = m[None]
    counts = m.sum(axis=1)
    if (counts != counts[0]).any():
        raise ValueError("bool_masked_pos: all samples in a batch must mask the same number of tokens")
    vis = np.stack([np.flatnonzero(~r) for r in m]).astype(np.int32)
    msk = np.stack([np.flatnonzero(r) for r in m]).astype(np.int32)
    return torch.from_numpy(vis), torch.from_numpy(msk)


class VJEPAMaskGenerator:
    """Context / target index masks for V-JEPA (reference ``src/dataloader/transforms.py:96-217``, default flags): same constructor
    arguments, same draws (the per-call seed and the block origins from the GLOBAL torch RNG, scale and aspect ratio from a local
    ``torch.Generator``), same outputs ``data["context_mask"]`` (tokens outside every block) / ``data["target_mask"]`` (inside), both
    ascending int64 over the (input[0]/p[0], input[1]/p[1], input[2]/p[2]) grid flattened row-major."""

    def __init__(self, input_size=(224, 224, 160), patch_size=(16, 16, 16), pred_mask_scale=(0.2, 0.8), aspect_ratio=(0.3, 3.0), num_blocks=1,
                 max_keep=None, inv_block=False, full_complement=False, pred_full_complement=False):
        if not isinstance(input_size, tuple):
            input_size = (input_size,) * 3
        if not isinstance(patch_size, tuple):
            patch_size = (patch_size,) * 3
        if full_complement or pred_full_complement:
            raise NotImplementedError("full_complement / pred_full_complement are not used by the shipped pipelines")
        self.input_size, self.patch_size = input_size, patch_size
        self.depth, self.height, self.width = (input_size[i] // patch_size[i] for i in range(3))
        self.pred_mask_scale, self.aspect_ratio, self.num_blocks = pred_mask_scale, aspect_ratio, num_blocks
        self.max_keep, self.inv_block = max_keep, inv_block

    def _sample_block_size(self, generator):
        import math
        lo, hi = self.pred_mask_scale
        mask_scale = lo + torch.rand(1, generator=generator).item() * (hi - lo)
        num_keep = int(self.depth * self.height * self.width * mask_scale)
        lo, hi = self.aspect_ratio
        ar = lo + torch.rand(1, generator=generator).item() * (hi - lo)
        ar1, ar2 = ar, 1.0 / ar
        d = int(round(math.pow(num_keep * ar1 * ar2, 1 / 3)))
        h, w = int(round(d * ar1)), int(round(d * ar2))
        return min(d, self.depth), min(h, self.height), min(w, self.width)

    def __call__(self, data):
        seed = torch.randint(0, 2 ** 32, (1,)).item()
        generator = torch.Generator()
        generator.manual_seed(seed)
        d, h, w = self._sample_block_size(generator)
        keep = torch.ones((self.depth, self.height, self.width), dtype=torch.int32)
        for _ in range(self.num_blocks):
            sd = torch.randint(0, self.depth - d + 1, (1,))
            sh = torch.randint(0, self.height - h + 1, (1,))
            sw = torch.randint(0, self.width - w + 1, (1,))
            keep[sd:sd + d, sh:sh + h, sw:sw + w] = 0
        flat = keep.flatten()
        mask_p = torch.argwhere(flat == 0).squeeze()
        mask_e = torch.nonzero(flat).squeeze()
        if self.max_keep is not None:
            mask_e, mask_p = mask_e[: self.max_keep], mask_p[: self.max_keep]
        data["context_mask"], data["target_mask"] = (mask_p, mask_e) if self.inv_block else (mask_e, mask_p)
        return data
