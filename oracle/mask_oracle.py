"""CPU oracle for the MIM block-mask generator.  TEST INFRASTRUCTURE ONLY (see oracle/mim_oracle.py header).

Follows ``/root/reference/src/dataloader/mim.py:33-69`` (duplicate at ``src/dataloader/transforms.py:28-63``):
draw ``np.random.permutation(token_count)[:mask_count]`` from the GLOBAL legacy numpy RNG, scatter ones on the
coarse (depth/mps, size/mps, size/mps) grid, repeat ``scale`` times along each axis, flatten in (D',H',W') order.
Written with explicit loops on purpose (independent of the product's vectorised version).

Pinned by ``tests/golden/masks.npz`` (outputs of the reference class executed in the build container).
"""
import math

import numpy as np


def mask_counts(input_size, depth, mask_patch_size, model_patch_size, mask_ratio):
    if input_size % mask_patch_size != 0:
        raise ValueError("Input size must be divisible by mask patch size")
    if depth % mask_patch_size != 0:
        raise ValueError("Depth must be divisible by mask patch size")
    if mask_patch_size % model_patch_size != 0:
        raise ValueError("Mask patch size must be divisible by model patch size")
    rs, rd = input_size // mask_patch_size, depth // mask_patch_size
    tc = rs * rs * rd
    return rs, rd, mask_patch_size // model_patch_size, tc, int(math.ceil(tc * mask_ratio))


def generate_mask(input_size=224, depth=96, mask_patch_size=32, model_patch_size=16, mask_ratio=0.6):
    rs, rd, scale, tc, mc = mask_counts(input_size, depth, mask_patch_size, model_patch_size, mask_ratio)
    perm = np.random.permutation(tc)
    coarse = np.zeros(tc, dtype=np.int64)
    for i in range(mc):
        coarse[perm[i]] = 1
    fine = np.zeros((rd * scale, rs * scale, rs * scale), dtype=bool)
    for d in range(rd * scale):
        for h in range(rs * scale):
            for w in range(rs * scale):
                fine[d, h, w] = coarse[((d // scale) * rs + (h // scale)) * rs + (w // scale)] == 1
    return fine.reshape(-1)
