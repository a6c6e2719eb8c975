"""Input-feed oracle (CPU): the restated ScaleIntensityRange / SpatialPad / CenterSpatialCrop / PermuteImage conventions on hand-checked
cases (monai is absent here: PARITY UNPINNED for its conventions, see oracle/feed_oracle.py), and the host-side error behaviour."""
import numpy as np
import pytest
import torch

from oracle import feed_oracle as fo


def test_scale_intensity_range_values():
    x = np.array([-2000, -1000, -500, 0, 500, 1000, 3000], dtype=np.int16)
    v = fo.scale_intensity_range(x)
    assert v.dtype == np.float32
    assert np.array_equal(v, np.array([0.0, 0.0, 0.25, 0.5, 0.75, 1.0, 1.0], dtype=np.float32))
    v = fo.scale_intensity_range(x, clip=False)
    assert np.allclose(v, [-0.5, 0.0, 0.25, 0.5, 0.75, 1.0, 2.0])
    v = fo.scale_intensity_range(np.array([0, 50, 100], dtype=np.float32), a_min=0, a_max=100, b_min=-1, b_max=1)
    assert np.array_equal(v, np.array([-1.0, 0.0, 1.0], dtype=np.float32))


def test_symmetric_pad_and_center_crop_indices():
    x = np.arange(1 * 3 * 5 * 4, dtype=np.float32).reshape(1, 3, 5, 4) + 1  # all > 0 so that pad zeros are recognisable
    p = fo.spatial_pad_symmetric(x, (6, 4, 7))     # H: 3 -> 6 (1 front, 2 back); W: 5 stays; D: 4 -> 7 (1 front, 2 back)
    assert p.shape == (1, 6, 5, 7)
    assert np.all(p[:, 0] == 0) and np.all(p[:, 4:] == 0) and np.all(p[:, :, :, 0] == 0) and np.all(p[:, :, :, 5:] == 0)
    assert np.array_equal(p[:, 1:4, :, 1:5], x)
    c = fo.center_spatial_crop(p, (6, 4, 7))       # W: 5 -> 4, start 5//2 - 4//2 = 0
    assert c.shape == (1, 6, 4, 7) and np.array_equal(c, p[:, :, 0:4])
    c = fo.center_spatial_crop(np.arange(10, dtype=np.float32).reshape(1, 10, 1, 1), (5, 1, 1))
    assert np.array_equal(c.ravel(), np.arange(3, 8))  # start 10//2 - 5//2 = 3
    c = fo.center_spatial_crop(np.arange(9, dtype=np.float32).reshape(1, 9, 1, 1), (4, 1, 1))
    assert np.array_equal(c.ravel(), np.arange(2, 6))  # start 9//2 - 4//2 = 2


def test_prepare_volume_layout_matches_permute_image():
    rng = np.random.default_rng(0)
    raw = rng.integers(-1500, 1500, size=(6, 7, 5), dtype=np.int16)
    out = fo.prepare_volume(raw, img_size=8, depth=4)
    assert out.shape == (4, 1, 8, 8) and out.dtype == np.float32
    # centre voxel bookkeeping: H 6 -> 8: 1 in front; W 7 -> 8: 0 in front; D 5 -> 4: crop start 5//2 - 4//2 = 0
    h, w, d = 2, 3, 1
    expect = np.float32(min(max((np.float32(raw[h, w, d]) + np.float32(1000)) / np.float32(2000), 0), 1))
    assert out[d, 0, h + 1, w] == expect
    assert np.all(out[:, 0, 0, :] == 0) and np.all(out[:, 0, 7, :] == 0) and np.all(out[:, 0, :, 7] == 0)


def test_feed_has_no_cpu_fallback():
    from smb_vision_amd import _lib, feed
    with pytest.raises(_lib.SmbvError, match="no CPU fallback"):
        feed.prepare_volume(torch.zeros(4, 4, 4, dtype=torch.int16), 4, 4)
    with pytest.raises(_lib.SmbvError, match="no CPU fallback"):
        feed.VolumeFeeder([], "cpu", 4, 4)
