"""Input feed on the HIP path: smbv_prepare_volume (through the C ABI) bit-exact against the CPU restatement of the reference's
ScaleIntensityRanged -> SpatialPadd -> CenterSpatialCropd -> PermuteImage tail (oracle/feed_oracle.py; MONAI conventions: parity
unpinned, see there), and the double-buffered VolumeFeeder delivering volumes in order while the consumer is busy."""
import numpy as np
import pytest
import torch

from oracle import feed_oracle as fo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def feed():
    from smb_vision_amd import feed as f, _lib
    _lib.load()
    return f


CASES = [  # (H0, W0, D0, image_size, depth): pad everywhere, crop everywhere, mixed, ragged tiles, exact fit
    (20, 24, 10, 32, 16), (70, 90, 80, 64, 32), (50, 80, 20, 64, 32), (33, 129, 67, 65, 66), (64, 64, 32, 64, 32), (5, 3, 2, 4, 4),
]


@pytest.mark.parametrize("shape", CASES)
@pytest.mark.parametrize("dtype", ["int16", "uint16", "float32", "bfloat16"])
def test_prepare_volume_bit_exact(feed, shape, dtype):
    H0, W0, D0, S, D = shape
    rng = np.random.default_rng(H0 * 1000 + W0)
    if dtype == "int16":
        raw = rng.integers(-2000, 2500, size=(H0, W0, D0), dtype=np.int16)
        dev, ref_in, kw = torch.from_numpy(raw).cuda(), raw, {}
    elif dtype == "uint16":
        raw = rng.integers(0, 4000, size=(H0, W0, D0), dtype=np.uint16)
        dev, ref_in, kw = torch.from_numpy(raw.view(np.int16)).cuda().view(torch.uint16), raw, dict(a_min=0.0, a_max=3000.0)
    elif dtype == "float32":
        raw = (rng.standard_normal((H0, W0, D0)) * 900).astype(np.float32)
        dev, ref_in, kw = torch.from_numpy(raw).cuda(), raw, dict(b_min=-1.0, b_max=3.0)
    else:
        t = (torch.from_numpy(rng.standard_normal((H0, W0, D0)).astype(np.float32)) * 900).bfloat16()
        dev, ref_in, kw = t.cuda(), t.float().numpy(), dict(clip=False)
    got = feed.prepare_volume(dev, S, D, **kw)
    ref = fo.prepare_volume(ref_in, S, D, **kw)
    assert got.shape == (1, D, 1, S, S)
    assert np.array_equal(got[0].cpu().numpy(), ref)


@pytest.mark.parametrize("seed", list(range(32)))
def test_prepare_volume_random_geometry_bit_exact(feed, seed):
    """Random source boxes (1..150 per axis), target sizes (odd and even: both centring conventions of pad and crop), intensity windows, clip on / off,
    batch 1-2: bit-exact against the CPU restatement."""
    rng = np.random.default_rng(9000 + seed)
    H0, W0, D0 = (int(rng.integers(1, 151)) for _ in range(3))
    S, D = int(rng.integers(2, 140)), int(rng.integers(2, 140))
    B = int(rng.integers(1, 3))
    a_min = float(rng.choice([-1000.0, -200.0, 0.0]))
    a_max = a_min + float(rng.choice([300.0, 1200.0, 2000.0]))
    kw = dict(a_min=a_min, a_max=a_max, b_min=float(rng.choice([0.0, -1.0])), b_max=float(rng.choice([1.0, 2.5])), clip=bool(rng.integers(0, 2)))
    if seed % 3 == 0:
        raw = rng.integers(-2000, 2500, size=(B, H0, W0, D0), dtype=np.int16)
        dev = torch.from_numpy(raw).cuda()
    elif seed % 3 == 1:
        raw = (rng.standard_normal((B, H0, W0, D0)) * 900).astype(np.float32)
        dev = torch.from_numpy(raw).cuda()
    else:
        raw = rng.integers(0, 4000, size=(B, H0, W0, D0), dtype=np.uint16)
        dev = torch.from_numpy(raw.view(np.int16)).cuda().view(torch.uint16)
    got = feed.prepare_volume(dev, S, D, **kw)
    assert got.shape == (B, D, 1, S, S)
    for b in range(B):
        assert np.array_equal(got[b].cpu().numpy(), fo.prepare_volume(raw[b], S, D, **kw)), (seed, H0, W0, D0, S, D, kw)


def test_prepare_volume_batch_and_full_size_property(feed):
    # batch of 2, and the size-independent property at the real geometry: a constant HU volume maps to one value inside the source box
    raw = torch.full((2, 40, 40, 24), 500, dtype=torch.int16).cuda()
    got = feed.prepare_volume(raw, 64, 32)
    assert got.shape == (2, 32, 1, 64, 64)
    inside = got[:, 4:28, 0, 12:52, 12:52]
    assert float(inside.min()) == 0.75 and float(inside.max()) == 0.75
    assert float(got.sum()) == pytest.approx(0.75 * 2 * 40 * 40 * 24, rel=1e-6)
    big = torch.zeros(1, 300, 512, 256, dtype=torch.int16).cuda()   # H padded 300 -> 512, D padded 256 -> 320, W exact
    out = feed.prepare_volume(big, 512, 320)
    assert out.shape == (1, 320, 1, 512, 512)
    assert float(out[0, 32:288, 0, 106:406].min()) == 0.5 and float(out.sum()) == pytest.approx(0.5 * 300 * 512 * 256, rel=1e-6)


def test_volume_feeder_order_and_overlap(feed):
    S, D, n = 64, 32, 7
    rng = np.random.default_rng(3)
    raws = [rng.integers(-1200, 1200, size=(60 + i, 64, 30 + i), dtype=np.int16) for i in range(n)]   # shapes change: buffers re-made
    def source():
        for i, r in enumerate(raws):
            yield r, {"i": i}
    seen = []
    sums = []
    work = torch.randn(2048, 2048, device="cuda")
    for pix, extras in feed.VolumeFeeder(source(), "cuda:0", S, D):
        seen.append(extras["i"])
        sums.append(pix.double().sum())      # consumer work enqueued on the current stream
        for _ in range(3):
            work = work @ work * 1e-3       # keep the consumer's stream busy while the next volume is uploaded
    torch.cuda.synchronize()
    assert seen == list(range(n))
    for i, r in enumerate(raws):
        assert float(sums[i]) == pytest.approx(float(fo.prepare_volume(r, S, D).astype(np.float64).sum()), rel=1e-9)


def test_volume_feeder_propagates_source_errors(feed):
    def bad():
        yield np.zeros((4, 4, 4), dtype=np.int16), None
        raise RuntimeError("source failed")
    it = iter(feed.VolumeFeeder(bad(), "cuda:0", 4, 4))
    next(it)
    with pytest.raises(RuntimeError, match="source failed"):
        next(it)


def test_cached_volumes_through_the_feeder(feed, tmp_path):
    from smb_vision_amd import volcache as vc
    rng = np.random.default_rng(5)
    raws, paths = [], []
    for i in range(4):
        r = rng.integers(-1500, 1500, size=(40 + 3 * i, 48, 20 + i), dtype=np.int16)
        p = str(tmp_path / f"v{i}.smbvol")
        vc.write_volume(p, r)
        raws.append(r)
        paths.append(p)
    got = []
    for pix, extras in feed.VolumeFeeder(vc.CachedVolumeSource(paths, epochs=2), "cuda:0", 64, 32):
        got.append((extras["path"], pix[0].cpu().numpy().copy()))
    assert [p for p, _ in got] == paths + paths
    for (p, a) in got:
        assert np.array_equal(a, fo.prepare_volume(raws[paths.index(p)], 64, 32))
