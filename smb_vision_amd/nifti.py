"""NIfTI-1 -> ``.smbvol`` converter ("+ a NIfTI -> cache converter; nibabel absent here").

The head of the reference's per-sample pipeline (``src/dataloader/mim.py:142-153``: ``LoadImaged`` -> ``EnsureChannelFirstd`` ->
``Orientationd(axcodes="RAS")`` -> ``Spacingd(pixdim=(1.5, 1.5, 3.0), mode="bilinear")``) run ONCE per study on the host, its result written as a
flat cached volume (``volcache.write_volume``); everything after it - intensity window, pad, crop, permute - runs on the device every step
(``smbv_prepare_volume``).  Pure numpy + the standard library: nibabel and MONAI are absent from the image and cannot be installed.

What is restated, from the published NIfTI-1 layout and MONAI's documented behaviour (**parity unpinned** for the two MONAI conventions: the library is
absent and the reference holds no fixture; ``tests/test_nifti_cpu.py`` pins this file against hand-built volumes instead):

* the 348-byte NIfTI-1 header (either byte order, ``.nii`` and ``.nii.gz``, single-file magic ``n+1``), data in Fortran order (x fastest), the
  ``scl_slope`` / ``scl_inter`` value scaling, the voxel-to-world affine from the sform rows, else from the qform quaternion, else from ``pixdim``;
* ``Orientation("RAS")``: the closest axis permutation / flips of the affine's rotation part (polar factor by SVD, one output axis per input axis by largest
  absolute component), applied as flips and a transpose of the array;
* ``Spacing(pixdim, mode="bilinear")``: output shape ``round((n - 1) * in_spacing / out_spacing + 1)`` per axis (numpy's round-half-to-even applied AFTER the + 1, as MONAI's ``compute_shape_offset`` does; first voxel centres coincide), trilinear
  interpolation at ``i * out_spacing / in_spacing`` clamped to the volume ("border" padding), evaluated axis by axis in float32.

    python -m smb_vision_amd.nifti --json_path dataset.json --out_dir /data/cache      # {"train": [{"image": path}, ...], "validation": [...]}
    python -m smb_vision_amd.run_mim --volume_cache_dir /data/cache/train --eval_volume_cache_dir /data/cache/validation ...
"""
from __future__ import annotations

import argparse
import gzip
import json
import os
import struct
import sys
from typing import Optional, Sequence, Tuple

import numpy as np

_DTYPES = {2: "u1", 4: "i2", 8: "i4", 16: "f4", 64: "f8", 256: "i1", 512: "u2", 768: "u4"}


def _read_all(path: str) -> bytes:
    with open(path, "rb") as f:
        head = f.read(2)
        f.seek(0)
        if head == b"\x1f\x8b":
            with gzip.GzipFile(fileobj=f) as g:
                return g.read()
        return f.read()


def read_nifti(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """-> (volume [X, Y, Z] in its stored dtype, or float32 when the header scales it; affine 4x4 float64 voxel -> world mm)."""
    raw = _read_all(path)
    if len(raw) < 352:
        raise ValueError(f"{path}: too short for a NIfTI-1 header")
    for bo in ("<", ">"):
        if struct.unpack_from(bo + "i", raw, 0)[0] == 348:
            break
    else:
        raise ValueError(f"{path}: not a NIfTI-1 file (sizeof_hdr != 348 in either byte order)")
    magic = raw[344:348]
    if magic[:3] != b"n+1":
        raise ValueError(f"{path}: only single-file NIfTI-1 (magic 'n+1') is read, found {magic!r}")
    dim = struct.unpack_from(bo + "8h", raw, 40)
    nd = dim[0]
    if nd < 3 or nd > 7 or any(d != 1 for d in dim[4:1 + nd]):
        raise ValueError(f"{path}: expected one 3-D volume, dim = {dim}")
    shape = tuple(int(d) for d in dim[1:4])
    datatype = struct.unpack_from(bo + "h", raw, 70)[0]
    if datatype not in _DTYPES:
        raise ValueError(f"{path}: unsupported NIfTI datatype code {datatype}")
    pixdim = struct.unpack_from(bo + "8f", raw, 76)
    vox_offset = int(struct.unpack_from(bo + "f", raw, 108)[0])
    slope, inter = struct.unpack_from(bo + "2f", raw, 112)
    qform_code, sform_code = struct.unpack_from(bo + "2h", raw, 252)
    dt = np.dtype(bo + _DTYPES[datatype])
    n = shape[0] * shape[1] * shape[2]
    if vox_offset < 352 or len(raw) < vox_offset + n * dt.itemsize:
        raise ValueError(f"{path}: truncated data ({len(raw)} bytes, need {vox_offset + n * dt.itemsize})")
    vol = np.frombuffer(raw, dtype=dt, count=n, offset=vox_offset).reshape(shape, order="F")
    vol = vol.astype(dt.newbyteorder("="), copy=False)
    if np.isfinite(slope) and slope != 0.0 and not (slope == 1.0 and inter == 0.0):
        vol = vol.astype(np.float32) * np.float32(slope) + np.float32(inter)

    aff = np.eye(4)
    if sform_code > 0:
        aff[0, :] = struct.unpack_from(bo + "4f", raw, 280)
        aff[1, :] = struct.unpack_from(bo + "4f", raw, 296)
        aff[2, :] = struct.unpack_from(bo + "4f", raw, 312)
    elif qform_code > 0:
        b, c, d = struct.unpack_from(bo + "3f", raw, 256)
        off = struct.unpack_from(bo + "3f", raw, 268)
        # the NIfTI-1 standard's quatern -> matrix rule (nifti1.h "QUATERNION REPRESENTATION OF ROTATION MATRIX"): a = sqrt(1 - b^2 - c^2 - d^2); when that
        # is not positive (a 180-degree rotation stored in single precision lands slightly above 1) (b, c, d) is renormalised and a = 0
        a2 = 1.0 - (b * b + c * c + d * d)
        if a2 < 1e-7:
            n = 1.0 / np.sqrt(b * b + c * c + d * d)
            b, c, d, a = b * n, c * n, d * n, 0.0
        else:
            a = np.sqrt(a2)
        R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                      [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                      [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
        qfac = -1.0 if pixdim[0] < 0 else 1.0
        aff[:3, :3] = R * np.array([pixdim[1], pixdim[2], pixdim[3] * qfac])
        aff[:3, 3] = off
    else:
        aff[:3, :3] = np.diag([abs(p) if p else 1.0 for p in pixdim[1:4]])
    return vol, aff


def _axis_map(aff: np.ndarray):
    """For each array axis: (world axis it runs along, +1 / -1) - the closest permutation-with-flips of the affine's rotation part."""
    rzs = aff[:3, :3].astype(np.float64)
    zooms = np.sqrt((rzs * rzs).sum(axis=0))
    zooms[zooms == 0] = 1.0
    u, _, vt = np.linalg.svd(rzs / zooms)
    rot = u @ vt                                     # polar factor: the orthogonal matrix closest to the (possibly sheared) direction cosines
    out = [None] * 3
    rot = rot.copy()
    for in_ax in range(3):   # nibabel's io_orientation walks the array axes in order and removes a world axis once it is taken (its row is zeroed): mirrored for oblique volumes
        col = rot[:, in_ax]
        w = int(np.argmax(np.abs(col)))
        out[in_ax] = (w, 1.0 if col[w] >= 0 else -1.0)
        rot[w, :] = 0.0
    return out, zooms


def to_ras(vol: np.ndarray, aff: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """``Orientation(axcodes="RAS")``: -> (volume with axis 0 running Left->Right, 1 Posterior->Anterior, 2 Inferior->Superior; voxel spacing (3,) of those axes)."""
    amap, zooms = _axis_map(aff)
    for ax, (_, sign) in enumerate(amap):
        if sign < 0:
            vol = np.flip(vol, axis=ax)
    order = [None] * 3
    for ax, (w, _) in enumerate(amap):
        order[w] = ax
    vol = np.transpose(vol, order)
    return vol, zooms[order]


def _lerp_axis(x: np.ndarray, axis: int, n_out: int, ratio: float) -> np.ndarray:
    n_in = x.shape[axis]
    pos = np.clip(np.arange(n_out, dtype=np.float64) * ratio, 0.0, n_in - 1.0)
    i0 = np.minimum(np.floor(pos).astype(np.int64), max(n_in - 2, 0))
    w = (pos - i0).astype(np.float32)
    i1 = np.minimum(i0 + 1, n_in - 1)
    shp = [1, 1, 1]
    shp[axis] = n_out
    w = w.reshape(shp)
    return np.take(x, i0, axis=axis) * (np.float32(1.0) - w) + np.take(x, i1, axis=axis) * w


def resample_spacing(vol: np.ndarray, spacing: Sequence[float], pixdim: Sequence[float] = (1.5, 1.5, 3.0), tol: float = 1e-3) -> np.ndarray:
    """``Spacing(pixdim, mode="bilinear")``.  A volume already at ``pixdim`` (within ``tol`` mm) is returned as it is, in its stored dtype."""
    spacing = [float(s) for s in spacing]
    if all(abs(s - p) <= tol for s, p in zip(spacing, pixdim)):
        return vol
    out = vol.astype(np.float32)
    for ax in range(3):
        n_in = out.shape[ax]
        # MONAI sizes the output as np.round(extent + 1.0) (monai/data/utils.py compute_shape_offset, scale_extent=False), the extent being the distance between the
        # first and last voxel centres in output voxels - NOT round(extent) + 1: numpy rounds halves to even, so the two differ whenever the extent ends in .5
        # (512 voxels at 0.75 mm onto 1.5 mm: extent 255.5 -> 256 voxels, not 257)
        n_out = int(np.round((n_in - 1) * (spacing[ax] / pixdim[ax]) + 1.0))
        out = _lerp_axis(out, ax, max(n_out, 1), pixdim[ax] / spacing[ax])
    return np.ascontiguousarray(out)


def convert(src: str, dst: str, pixdim: Sequence[float] = (1.5, 1.5, 3.0), dtype: str = "auto", window=(-1000.0, 1000.0, 0.0, 1.0, True),
            mask: Optional[np.ndarray] = None) -> Tuple[int, int, int]:
    """One study: NIfTI -> RAS -> ``pixdim`` -> ``dst`` (.smbvol).  dtype "auto": the stored integer type when nothing was interpolated or scaled, float32
    otherwise (exactly what the reference's float32 pipeline holds at this point); "int16": rounded to whole HU (a quarter of the bytes per upload, at most
    0.5 HU = 2.5e-4 of the [0, 1] intensity range away); "float32"."""
    from .volcache import write_volume
    vol, aff = read_nifti(src)
    vol, zooms = to_ras(vol, aff)
    vol = resample_spacing(vol, zooms, pixdim)
    if dtype not in ("auto", "int16", "float32"):
        raise ValueError("dtype must be 'auto', 'int16' or 'float32'")
    if dtype == "float32" or (dtype == "auto" and vol.dtype.kind == "f"):
        vol = vol.astype(np.float32)
    elif vol.dtype in (np.dtype(np.int16), np.dtype(np.uint16)):
        pass                                                    # stored as it is (volcache codes 0 / 1)
    elif vol.dtype.kind == "f":                                 # "int16" asked for an interpolated / scaled volume
        vol = np.clip(np.rint(vol), -32768, 32767).astype(np.int16)
    elif dtype == "auto" and (int(vol.min()) < -32768 or int(vol.max()) > 32767):
        vol = vol.astype(np.float32)                            # a wider integer type whose values do not fit 16 bits
    else:
        vol = np.clip(vol, -32768, 32767).astype(np.int16)
    a_min, a_max, b_min, b_max, clip = window
    write_volume(dst, np.ascontiguousarray(vol), spacing=tuple(float(p) for p in pixdim), a_min=a_min, a_max=a_max, b_min=b_min, b_max=b_max, clip=clip, mask=mask)
    return tuple(int(s) for s in vol.shape)


def main(argv=None):
    ap = argparse.ArgumentParser(description="NIfTI-1 studies -> .smbvol cached volumes (RAS, resampled to --pixdim)")
    ap.add_argument("--json_path", help='the reference\'s dataset file (src/dataloader/mim.py:130-135): {"train": [{"image": path}, ...], "validation": [...]} or a plain list')
    ap.add_argument("--images", nargs="*", default=[], help="NIfTI files, instead of / besides --json_path (split 'train')")
    ap.add_argument("--out_dir", required=True)
    ap.add_argument("--pixdim", nargs=3, type=float, default=[1.5, 1.5, 3.0])
    ap.add_argument("--dtype", default="auto", choices=("auto", "int16", "float32"))
    args = ap.parse_args(argv)
    splits = {}
    if args.json_path:
        data = json.load(open(args.json_path))
        splits = data if isinstance(data, dict) else {"train": data}
    if args.images:
        splits.setdefault("train", [])
        splits["train"] = list(splits["train"]) + [{"image": p} for p in args.images]
    n = 0
    for split, items in splits.items():
        d = os.path.join(args.out_dir, split)
        os.makedirs(d, exist_ok=True)
        for i, it in enumerate(items):
            src = it["image"] if isinstance(it, dict) else it
            base = os.path.basename(src)
            for ext in (".nii.gz", ".nii"):
                if base.endswith(ext):
                    base = base[: -len(ext)]
            dst = os.path.join(d, f"{i:06d}_{base}.smbvol")
            shape = convert(src, dst, args.pixdim, args.dtype)
            print(f"{src} -> {dst} {shape}", file=sys.stderr)
            n += 1
    print(json.dumps({"converted": n, "out_dir": args.out_dir, "splits": {k: len(v) for k, v in splits.items()}}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
