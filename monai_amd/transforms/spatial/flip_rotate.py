"""``Flip`` / ``Rotate90`` (+ dictionary versions) -- monai/transforms/spatial/array.py:665-718, 1139-1200, functional.py:232-266, 397-447
-- on the strided-gather kernel of ``Orientation`` (``torch.flip`` / ``torch.rot90`` become one pass that writes a dense result).
The MetaTensor affine is updated with the exact integer voxel map ``new index -> old index`` (the reference builds the same matrix from
rotation matrices, i.e. to 1e-16)."""

from __future__ import annotations

from collections.abc import Sequence

import numpy as np
import torch

from ... import ops
from ...data.meta_tensor import affine_np, is_meta
from ..lazy import LazyCapable, LazyCapableDict, materialize, peek_affine, peek_shape, push_pending
from ...utils.misc import as_gather_f32, ensure_tuple

__all__ = ["Flip", "Flipd", "FlipD", "FlipDict", "Rotate90", "Rotate90d", "Rotate90D", "Rotate90Dict"]

_INTS = (torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64, torch.bool)


def _voxel_map(r: int, sr: int, perm, flips, size) -> np.ndarray:
    """(r+1) x (r+1) matrix `new index -> old index` of the signed axis permutation"""
    xform = np.zeros((r + 1, r + 1))
    xform[-1, -1] = 1.0
    for k in range(r):
        a = perm[k] if k < sr else k                 # axes of the affine beyond the image's rank are left alone
        f = flips[a] if a < sr else False
        xform[a, k] = -1.0 if f else 1.0
        if f:
            xform[a, -1] = size[a] - 1
    return xform


def _flip_permute(img, perm, flips, record, transform=None, lazy: bool = False):
    """channel-first image with 1-3 spatial axes: output spatial axis k shows input axis perm[k], input axis a reversed when flips[a]"""
    if lazy:               # functional.py:232-266 / 397-447 with lazy=True: only the record
        size = peek_shape(img)
        r = max(len(peek_affine(img)) - 1, len(size)) if is_meta(img) else len(size)
        return push_pending(img, transform, _voxel_map(min(r, 3), len(size), perm, flips, size), [size[p] for p in perm], record.get("extra_info"),
                            orig_size=size)
    img = materialize(img)
    data = img.as_tensor() if is_meta(img) else torch.as_tensor(img)
    sr = data.dim() - 1
    if sr < 1 or sr > 3:
        raise NotImplementedError(f"monai_amd flip / rotate90: 1-3 spatial axes are on the HIP path, got shape {tuple(data.shape)}")
    if data.dtype != torch.float32 and data.dtype not in _INTS:
        raise NotImplementedError(f"monai_amd flip / rotate90: {data.dtype} images are not on the HIP path (float32 and integer images are)")
    size = tuple(int(v) for v in data.shape[1:])
    pad = 3 - sr
    x32, _, restore = as_gather_f32(data)             # int32 as bit patterns: exact for every value
    x4 = x32.reshape((data.shape[0],) + (1,) * pad + size).contiguous()
    out = ops.flip_permute(x4, list(range(pad)) + [p + pad for p in perm], [False] * pad + list(flips))
    out = restore(out.reshape((data.shape[0],) + tuple(size[p] for p in perm)))
    if not is_meta(img):
        return out
    res = type(img)(out, meta=dict(img.meta), applied_operations=list(getattr(img, "applied_operations", [])))
    aff = affine_np(img)
    r = aff.shape[0] - 1
    xform = _voxel_map(r, sr, perm, flips, size)
    res.meta["affine"] = torch.as_tensor(aff @ xform, dtype=torch.float64)
    res.applied_operations.append(dict(record, orig_size=size))
    return res


class Flip(LazyCapable):
    """Reverse the order of elements along the given spatial axes (``None``: all of them; negative axes count from the end)."""

    def __init__(self, spatial_axis: Sequence[int] | int | None = None, lazy: bool = False) -> None:
        self.spatial_axis = spatial_axis
        self.lazy = lazy

    def __call__(self, img, lazy: bool | None = None):
        lazy_ = self.lazy if lazy is None else lazy
        sr = (img.as_tensor() if is_meta(img) else torch.as_tensor(img)).dim() - 1
        if self.spatial_axis is None:
            axes = list(range(sr))
        else:                                        # map_spatial_axes, monai/transforms/utils.py:1380-1412
            axes = []
            for a in ensure_tuple(self.spatial_axis):
                if not isinstance(a, int):          # the reference hands the axes to torch.flip, which raises TypeError (tests/transforms/test_flip.py:32)
                    raise TypeError("spatial_axis must be None, int or sequence of ints.")
                if a >= sr or a < -sr:
                    raise IndexError(f"spatial axis {a} is out of range for an image with {sr} spatial axes")
                axes.append(a if a >= 0 else sr + a)
        flips = [a in axes for a in range(sr)]
        return _flip_permute(img, list(range(sr)), flips, {"class": type(self).__name__, "extra_info": {"axes": self.spatial_axis}}, self, lazy_)

    def inverse(self, data):
        rec = data.applied_operations[-1]
        prev = type(data)(data.as_tensor(), meta=dict(data.meta), applied_operations=list(data.applied_operations[:-1]))
        out = Flip(spatial_axis=rec["extra_info"]["axes"])(prev)
        out.applied_operations = list(data.applied_operations[:-1])
        return out


class Rotate90(LazyCapable):
    """Rotate by ``k`` x 90 degrees in the plane of two spatial axes (``torch.rot90`` semantics)."""

    def __init__(self, k: int = 1, spatial_axes: tuple[int, int] = (0, 1), lazy: bool = False) -> None:
        self.k = (4 + (k % 4)) % 4
        axes = ensure_tuple(spatial_axes)
        if len(axes) != 2:
            raise ValueError(f"spatial_axes must be 2 numbers to define the plane to rotate, got {axes}.")
        self.spatial_axes = axes
        self.lazy = lazy

    def __call__(self, img, lazy: bool | None = None):
        lazy_ = self.lazy if lazy is None else lazy
        sr = (img.as_tensor() if is_meta(img) else torch.as_tensor(img)).dim() - 1
        a0, a1 = (a if a >= 0 else sr + a for a in self.spatial_axes)
        if not (0 <= a0 < sr and 0 <= a1 < sr) or a0 == a1:
            raise ValueError(f"spatial_axes {self.spatial_axes} do not name two different axes of an image with {sr} spatial axes")
        perm, flips = list(range(sr)), [False] * sr
        if self.k in (1, 3):                         # rot90 = flip one axis of the plane, then swap the two
            perm[a0], perm[a1] = a1, a0
            flips[a1 if self.k == 1 else a0] = True
        elif self.k == 2:
            flips[a0] = flips[a1] = True
        return _flip_permute(img, perm, flips, {"class": type(self).__name__, "extra_info": {"axes": [a0, a1], "k": self.k}}, self, lazy_)

    def inverse(self, data):
        rec = data.applied_operations[-1]
        prev = type(data)(data.as_tensor(), meta=dict(data.meta), applied_operations=list(data.applied_operations[:-1]))
        return self.inverse_transform(prev, rec)

    def inverse_transform(self, data, transform):
        """Undo the rotation described by the record `transform` on `data` (whose record has already been popped) -- the entry point the
        reference's RandRotate90.inverse uses (monai/transforms/spatial/array.py:1188-1196, 1254-1259); leaves no record of its own."""
        kept = list(getattr(data, "applied_operations", []) or [])
        out = Rotate90(k=4 - transform["extra_info"]["k"], spatial_axes=tuple(transform["extra_info"]["axes"]))(data)
        if hasattr(out, "applied_operations"):
            out.applied_operations = kept
        return out


class _Keyed(LazyCapableDict):
    _lazy_inner = ("transform",)

    def __init__(self, keys, transform, allow_missing_keys: bool = False) -> None:
        self.keys, self.allow_missing_keys, self.transform = ensure_tuple(keys), allow_missing_keys, transform
        self.lazy = bool(getattr(transform, "lazy", False))

    def _each(self, data, fn):
        d = dict(data)
        for key in self.keys:
            if key not in d:
                if self.allow_missing_keys:
                    continue
                raise KeyError(f"Key `{key}` of transform `{type(self).__name__}` was missing in the data and allow_missing_keys==False.")
            d[key] = fn(d[key])
        return d

    def __call__(self, data, lazy: bool | None = None):
        lazy_ = self.lazy if lazy is None else lazy
        return self._each(data, lambda v: self.transform(v, lazy=lazy_))

    def inverse(self, data):
        return self._each(data, self.transform.inverse)


class Flipd(_Keyed):
    """monai/transforms/spatial/dictionary.py:1471-1520"""

    def __init__(self, keys, spatial_axis=None, allow_missing_keys: bool = False, lazy: bool = False) -> None:
        super().__init__(keys, Flip(spatial_axis=spatial_axis, lazy=lazy), allow_missing_keys)
        self.flipper = self.transform


class Rotate90d(_Keyed):
    """monai/transforms/spatial/dictionary.py:613-665"""

    def __init__(self, keys, k: int = 1, spatial_axes: tuple[int, int] = (0, 1), allow_missing_keys: bool = False, lazy: bool = False) -> None:
        super().__init__(keys, Rotate90(k, spatial_axes, lazy=lazy), allow_missing_keys)
        self.rotator = self.transform


FlipD = FlipDict = Flipd
Rotate90D = Rotate90Dict = Rotate90d
