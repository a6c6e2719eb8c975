"""Cached-volume format: round trips, zero-copy view, direct read into a host buffer, stored masks, and the
rejection of foreign / damaged files.  Pure host code."""
import os

import numpy as np
import pytest
import torch

from smb_vision_amd import volcache as vc
from smb_vision_amd.masking import MaskGenerator


@pytest.mark.parametrize("dtype", ["int16", "uint16", "float32", "bfloat16"])
def test_round_trip_and_views(tmp_path, dtype):
    rng = np.random.default_rng(0)
    shape = (9, 17, 5)
    if dtype == "bfloat16":
        vol = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).bfloat16()
        ref_bits = vol.view(torch.int16).numpy().view(np.uint16)
    else:
        vol = (rng.standard_normal(shape) * 500).astype(dtype)
        ref_bits = vol
    p = str(tmp_path / "v.smbvol")
    vc.write_volume(p, vol, spacing=(1.0, 2.0, 3.0), a_min=-900, a_max=1100, clip=False)
    rec = vc.open_volume(p)
    assert rec.shape == shape and rec.spacing == (1.0, 2.0, 3.0) and rec.window == (-900.0, 1100.0, 0.0, 1.0, False)
    assert rec.data_off % 4096 == 0 and rec.n_mask == 0 and rec.mask() is None
    assert np.array_equal(np.asarray(rec.array()), ref_bits)
    dst = torch.empty(shape, dtype=rec.dtype)
    rec.read_into(dst)
    if dtype == "bfloat16":
        assert torch.equal(dst, vol)
    else:
        assert np.array_equal(dst.view(torch.int16).numpy().view(np.uint16) if dtype == "uint16" else dst.numpy(), vol)
    with pytest.raises(ValueError):
        rec.read_into(torch.empty((9, 17, 4), dtype=rec.dtype))


def test_stored_mask_reproduces_fixed_mask_per_sample(tmp_path):
    gen = MaskGenerator(input_size=64, depth=32, mask_patch_size=16, model_patch_size=16, mask_ratio=0.6)
    np.random.seed(0)
    m = gen()
    p = str(tmp_path / "m.smbvol")
    vc.write_volume(p, np.zeros((8, 8, 4), dtype=np.int16), mask=m.numpy())
    rec = vc.open_volume(p)
    assert rec.mask_off % 4096 == 0 and rec.n_mask == m.numel() and torch.equal(rec.mask(), m)
    items = list(vc.CachedVolumeSource([p, p], epochs=2))
    assert len(items) == 4 and all(torch.equal(e["mask"], m) for _, e in items)          # the reference's cached-mask behaviour
    q = str(tmp_path / "n.smbvol")
    vc.write_volume(q, np.zeros((8, 8, 4), dtype=np.int16))
    np.random.seed(1)
    fresh = [e["mask"] for _, e in vc.CachedVolumeSource([q], epochs=3, mask_generator=gen)]
    assert not torch.equal(fresh[0], fresh[1]) and fresh[0].sum() == fresh[1].sum() == m.sum()   # a new draw per visit otherwise


def test_rejects_foreign_and_truncated_files(tmp_path):
    p = str(tmp_path / "x.bin")
    open(p, "wb").write(b"not a volume" * 400)
    with pytest.raises(ValueError, match="bad magic"):
        vc.open_volume(p)
    open(p, "wb").write(b"short")
    with pytest.raises(ValueError, match="too short"):
        vc.open_volume(p)
    good = str(tmp_path / "g.smbvol")
    vc.write_volume(good, np.ones((16, 16, 16), dtype=np.int16))
    data = open(good, "rb").read()
    open(good, "wb").write(data[:-100])
    with pytest.raises(ValueError, match="truncated"):
        vc.open_volume(good)
    with pytest.raises(ValueError):
        vc.write_volume(str(tmp_path / "bad"), np.zeros((4, 4), dtype=np.int16))
    assert not os.path.exists(str(tmp_path / "bad"))
