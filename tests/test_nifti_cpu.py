"""NIfTI-1 -> .smbvol converter: header parsing (both byte orders, gzip, value scaling, sform / qform / pixdim affines), reorientation
to RAS, resampling to the reference's spacing, and the hand-over to the cached-volume dataset.  Pure host code; the NIfTI files are built here, byte by byte,
from the published header layout (nibabel / MONAI are absent: the two MONAI conventions - output shape and sampling positions of ``Spacing`` - are parity
unpinned and stated in smb_vision_amd/nifti.py)."""
import gzip
import json
import struct

import numpy as np
import pytest

from smb_vision_amd import nifti, volcache as vc


def write_nifti(path, vol, affine=None, bo="<", slope=0.0, inter=0.0, qform=None, pixdim=(1.0, 1.0, 1.0), gz=False):
    code = {np.dtype("uint8"): 2, np.dtype("int16"): 4, np.dtype("int32"): 8, np.dtype("float32"): 16, np.dtype("float64"): 64, np.dtype("uint16"): 512}[vol.dtype]
    h = bytearray(352)
    struct.pack_into(bo + "i", h, 0, 348)
    struct.pack_into(bo + "8h", h, 40, 3, vol.shape[0], vol.shape[1], vol.shape[2], 1, 1, 1, 1)
    struct.pack_into(bo + "hh", h, 70, code, vol.dtype.itemsize * 8)
    qfac = 1.0
    if qform is not None:
        qfac = qform[0]
    struct.pack_into(bo + "8f", h, 76, qfac, pixdim[0], pixdim[1], pixdim[2], 0, 0, 0, 0)
    struct.pack_into(bo + "f", h, 108, 352.0)
    struct.pack_into(bo + "2f", h, 112, slope, inter)
    if affine is not None:
        struct.pack_into(bo + "2h", h, 252, 0, 1)
        for r, off in enumerate((280, 296, 312)):
            struct.pack_into(bo + "4f", h, off, *[float(x) for x in affine[r]])
    elif qform is not None:
        struct.pack_into(bo + "2h", h, 252, 1, 0)
        struct.pack_into(bo + "3f", h, 256, *qform[1:4])
        struct.pack_into(bo + "3f", h, 268, *qform[4:7])
    h[344:348] = b"n+1\0"
    data = vol.astype(vol.dtype.newbyteorder(bo)).tobytes(order="F")
    blob = bytes(h) + data
    if gz:
        with gzip.open(path, "wb") as f:
            f.write(blob)
    else:
        open(path, "wb").write(blob)


def ramp(shape):
    x, y, z = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), np.arange(shape[2]), indexing="ij")
    return x, y, z


@pytest.mark.parametrize("bo", ["<", ">"])
@pytest.mark.parametrize("gz", [False, True])
def test_header_byte_orders_gzip_and_value_scaling(tmp_path, bo, gz):
    rng = np.random.default_rng(0)
    vol = rng.integers(-1000, 2000, size=(7, 5, 3), dtype=np.int16)
    p = str(tmp_path / ("a.nii.gz" if gz else "a.nii"))
    aff = np.diag([2.0, 3.0, 4.0, 1.0]); aff[:3, 3] = [10, 20, 30]
    write_nifti(p, vol, affine=aff, bo=bo, gz=gz)
    got, a = nifti.read_nifti(p)
    assert got.dtype == np.int16 and np.array_equal(got, vol) and np.allclose(a, aff)
    write_nifti(p, vol, affine=aff, bo=bo, slope=0.5, inter=-1024.0, gz=gz)
    got, _ = nifti.read_nifti(p)
    assert got.dtype == np.float32 and np.allclose(got, vol.astype(np.float32) * 0.5 - 1024.0)
    with pytest.raises(ValueError):
        open(p, "wb").write(b"not nifti" * 100)
        nifti.read_nifti(p)


def test_reorientation_to_ras_from_sform_qform_and_pixdim(tmp_path):
    """Orientation("RAS"): an LPS-stored volume (x and y flipped), a volume stored with permuted axes, a slightly oblique one, the qform path, the bare-pixdim path."""
    shape = (6, 5, 4)
    x, y, z = ramp(shape)
    world = (100 * x + 10 * y + z).astype(np.int16)          # a value that names its own RAS voxel
    p = str(tmp_path / "v.nii")
    # (a) RAS already
    write_nifti(p, world, affine=np.diag([1.5, 1.5, 3.0, 1.0]))
    v, sp = nifti.to_ras(*nifti.read_nifti(p))
    assert np.array_equal(v, world) and np.allclose(sp, [1.5, 1.5, 3.0])
    # (b) LPS storage: array axes 0 and 1 run the other way
    write_nifti(p, world[::-1, ::-1, :].copy(), affine=np.diag([-1.5, -1.5, 3.0, 1.0]))
    v, sp = nifti.to_ras(*nifti.read_nifti(p))
    assert np.array_equal(v, world) and np.allclose(sp, [1.5, 1.5, 3.0])
    # (c) array axes (S, R, A) with a flip on A, anisotropic, 3 degrees oblique
    stored = np.transpose(world, (2, 0, 1))[:, :, ::-1].copy()   # stored[k, i, j'] = world[i, 4 - j', k]
    aff = np.zeros((4, 4)); aff[3, 3] = 1
    aff[2, 0], aff[0, 1], aff[1, 2] = 3.0, 0.8, -0.9           # array axis 0 -> S, 1 -> R, 2 -> -A
    c, s_ = np.cos(np.deg2rad(3.0)), np.sin(np.deg2rad(3.0))
    rot = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1.0]])
    aff[:3, :3] = rot @ aff[:3, :3]
    write_nifti(p, stored, affine=aff)
    v, sp = nifti.to_ras(*nifti.read_nifti(p))
    assert np.array_equal(v, world) and np.allclose(sp, [0.8, 0.9, 3.0], atol=1e-5)
    # (d) qform: a half turn about z (quaternion b = c = 0, d = 1): x and y flipped, as LPS
    write_nifti(p, world[::-1, ::-1, :].copy(), qform=(1.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0), pixdim=(1.5, 1.5, 3.0))
    v, sp = nifti.to_ras(*nifti.read_nifti(p))
    assert np.array_equal(v, world) and np.allclose(sp, [1.5, 1.5, 3.0], atol=1e-5)
    # (d') the same half turn stored a little long (|(b, c, d)| = 1.0002: single-precision storage of a 180-degree rotation), and a half turn about the
    # diagonal of x / y (b = c = 0.70712 > 1 / sqrt 2): the standard renormalises (b, c, d) and takes a = 0 instead of giving up the rotation's scale
    write_nifti(p, world[::-1, ::-1, :].copy(), qform=(1.0, 0.0, 0.0, 1.0002, 0.0, 0.0, 0.0), pixdim=(1.5, 1.5, 3.0))
    v, sp = nifti.to_ras(*nifti.read_nifti(p))
    assert np.array_equal(v, world) and np.allclose(sp, [1.5, 1.5, 3.0], atol=1e-5)
    write_nifti(p, np.transpose(world, (1, 0, 2))[:, :, ::-1].copy(), qform=(1.0, 0.70712, 0.70712, 0.0, 0.0, 0.0, 0.0), pixdim=(1.5, 1.5, 3.0))
    _, aff = nifti.read_nifti(p)
    assert np.allclose(aff[:3, :3], np.array([[0, 1.5, 0], [1.5, 0, 0], [0, 0, -3.0]]), atol=1e-4)   # x <-> y swapped, z reversed: an exact rotation again
    v, sp = nifti.to_ras(nifti.read_nifti(p)[0], aff)
    assert np.array_equal(v, world) and np.allclose(sp, [1.5, 1.5, 3.0], atol=1e-4)
    # (e) neither form: pixdim alone
    write_nifti(p, world, pixdim=(0.7, 0.7, 5.0))
    v, sp = nifti.to_ras(*nifti.read_nifti(p))
    assert np.array_equal(v, world) and np.allclose(sp, [0.7, 0.7, 5.0], atol=1e-6)


def test_spacing_resample_is_trilinear_with_coinciding_first_voxels():
    """Spacing(pixdim=(1.5, 1.5, 3.0), "bilinear"): a linear field is reproduced exactly at the new voxel centres (trilinear interpolation is exact on it), the shape is
    round((n - 1) * in / out + 1) - numpy's round-half-to-even applied AFTER the + 1, as MONAI's compute_shape_offset sizes it -, the last samples are clamped to the border,
    and a volume already at the target spacing comes back untouched in its own dtype."""
    shape, sp = (21, 17, 9), (0.75, 1.0, 4.0)
    x, y, z = ramp(shape)
    f = (2.0 * x * sp[0] - 3.0 * y * sp[1] + 0.5 * z * sp[2]).astype(np.float32)   # linear in millimetres
    out = nifti.resample_spacing(f, sp, (1.5, 1.5, 3.0))
    exp_shape = tuple(int(np.round((n - 1) * (s / p) + 1.0)) for n, s, p in zip(shape, sp, (1.5, 1.5, 3.0)))
    assert out.shape == exp_shape == (11, 12, 12) and out.dtype == np.float32
    xo, yo, zo = ramp(exp_shape)
    mm = [np.minimum(o * p, (n - 1) * s) for o, p, n, s in zip((xo, yo, zo), (1.5, 1.5, 3.0), shape, sp)]   # positions past the last voxel clamp to it
    assert np.allclose(out, 2.0 * mm[0] - 3.0 * mm[1] + 0.5 * mm[2], atol=1e-4)
    same = np.arange(24, dtype=np.int16).reshape(2, 3, 4)
    assert nifti.resample_spacing(same, (1.5, 1.5, 3.0)) is same
    # extents that end in .5 (every even voxel count at 0.75 mm onto 1.5 mm): np.round(extent + 1.0) rounds the half to even AFTER the + 1 - 512 voxels become 256, not 257
    for n, want in ((20, 10), (22, 12), (512, 256), (511, 256), (513, 257)):
        assert nifti.resample_spacing(np.zeros((n, 2, 2), np.float32), (0.75, 1.5, 3.0)).shape == (want, 2, 2), n


def test_convert_feeds_the_cached_volume_dataset(tmp_path):
    """The CLI over the reference's dataset json (src/dataloader/mim.py:130-135): one .smbvol per study under <out>/<split>/, read back by CachedVolumeDataset with the
    header's spacing and intensity window; int16 studies at the target spacing stay int16, resampled ones become float32 (or whole-HU int16 on request)."""
    rng = np.random.default_rng(3)
    a = rng.integers(-1000, 1500, size=(12, 10, 6), dtype=np.int16)
    b = rng.integers(-1000, 1500, size=(20, 20, 8), dtype=np.int16)
    pa, pb = str(tmp_path / "study_a.nii.gz"), str(tmp_path / "study_b.nii")
    write_nifti(pa, a[::-1].copy(), affine=np.diag([-1.5, 1.5, 3.0, 1.0]), gz=True)     # LAS storage at the target spacing
    write_nifti(pb, b, affine=np.diag([0.75, 0.75, 3.0, 1.0]))                           # finer in-plane
    js = str(tmp_path / "data.json")
    json.dump({"train": [{"image": pa}, {"image": pb}], "validation": [{"image": pa}]}, open(js, "w"))
    assert nifti.main(["--json_path", js, "--out_dir", str(tmp_path / "cache")]) == 0
    ds = vc.CachedVolumeDataset(str(tmp_path / "cache" / "train"))
    assert len(ds) == 2 and len(vc.CachedVolumeDataset(str(tmp_path / "cache" / "validation"))) == 1
    ra, rb = ds[0]["volume"], ds[1]["volume"]
    assert ra.shape == (12, 10, 6) and ra.code == 0 and np.array_equal(np.asarray(ra.array()), a)                  # flipped back to RAS, still int16
    assert ra.spacing == (1.5, 1.5, 3.0) and ra.window == (-1000.0, 1000.0, 0.0, 1.0, True)
    assert rb.shape == (10, 10, 8) and rb.code == 3                                                               # round(19 * 0.75 / 1.5 + 1) = round(10.5) = 10 (half to even, after the + 1), float32
    ref = nifti.resample_spacing(b, (0.75, 0.75, 3.0))
    assert np.allclose(np.asarray(rb.array()), ref)
    assert np.allclose(np.asarray(rb.array())[:, :, 0][::1, ::1][0, 0], b[0, 0, 0])
    nifti.convert(pb, str(tmp_path / "b16.smbvol"), dtype="int16")
    r16 = vc.open_volume(str(tmp_path / "b16.smbvol"))
    assert r16.code == 0 and np.abs(np.asarray(r16.array()).astype(np.float32) - ref).max() <= 0.5 + 1e-3
