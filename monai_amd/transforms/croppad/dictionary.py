"""``CropForegroundd`` -- monai/transforms/croppad/dictionary.py:703-800: one bounding box from ``source_key``, every key
cropped (and padded) to it."""

from __future__ import annotations

from collections.abc import Callable, Sequence

from ...utils.misc import ensure_tuple, ensure_tuple_rep
from ..lazy import LazyCapableDict, materialize
from .array import BorderPad, CenterSpatialCrop, CropForeground, DivisiblePad, SpatialCrop, SpatialPad, is_positive

__all__ = ["CropForegroundd", "CropForegroundD", "CropForegroundDict", "SpatialPadd", "SpatialPadD", "SpatialPadDict", "BorderPadd", "BorderPadD",
           "BorderPadDict", "DivisiblePadd", "DivisiblePadD", "DivisiblePadDict", "SpatialCropd", "SpatialCropD", "SpatialCropDict",
           "CenterSpatialCropd", "CenterSpatialCropD", "CenterSpatialCropDict"]


class CropForegroundd(LazyCapableDict):
    _lazy_inner = ("cropper",)

    def __init__(self, keys, source_key: str, select_fn: Callable = is_positive, channel_indices=None, margin: Sequence[int] | int = 0,
                 allow_smaller: bool = False, k_divisible: Sequence[int] | int = 1, mode="constant",
                 start_coord_key: str | None = "foreground_start_coord", end_coord_key: str | None = "foreground_end_coord",
                 allow_missing_keys: bool = False, lazy: bool = False, **pad_kwargs) -> None:
        self.keys = ensure_tuple(keys)
        self.allow_missing_keys = allow_missing_keys
        self.source_key, self.start_coord_key, self.end_coord_key = source_key, start_coord_key, end_coord_key
        self.cropper = CropForeground(select_fn=select_fn, channel_indices=channel_indices, margin=margin, allow_smaller=allow_smaller,
                                      k_divisible=k_divisible, lazy=lazy, **pad_kwargs)
        self.mode = ensure_tuple_rep(mode, len(self.keys))
        self.lazy = lazy

    def __call__(self, data, lazy: bool | None = None):
        lazy_ = self.lazy if lazy is None else lazy
        d = dict(data)
        d[self.source_key] = materialize(d[self.source_key])     # requires_current_data: the box is computed on the current voxels
        box_start, box_end = self.cropper.compute_bounding_box(img=d[self.source_key])
        if self.start_coord_key is not None:
            d[self.start_coord_key] = box_start
        if self.end_coord_key is not None:
            d[self.end_coord_key] = box_end
        for key, m in zip(self.keys, self.mode):
            if key not in d:
                if self.allow_missing_keys:
                    continue
                raise KeyError(f"Key `{key}` of transform `{type(self).__name__}` was missing in the data and allow_missing_keys==False.")
            d[key] = self.cropper.crop_pad(img=materialize(d[key]), box_start=box_start, box_end=box_end, mode=m, lazy=bool(lazy_))
        return d

    def inverse(self, data):
        d = dict(data)
        for key in self.keys:
            if key in d:
                d[key] = self.cropper.inverse(d[key])
            elif not self.allow_missing_keys:
                raise KeyError(f"Key `{key}` of transform `{type(self).__name__}` was missing in the data and allow_missing_keys==False.")
        return d


CropForegroundD = CropForegroundDict = CropForegroundd


class _KeyedCropPad(LazyCapableDict):
    _lazy_inner = ("padder",)

    """``Padd`` / ``Cropd`` (monai/transforms/croppad/dictionary.py:113-186, 309-366): one array transform over the keys"""

    def __init__(self, keys, transform, allow_missing_keys: bool = False, mode=None) -> None:
        self.keys = ensure_tuple(keys)
        self.allow_missing_keys = allow_missing_keys
        self.padder = self.cropper = transform
        self.mode = None if mode is None else ensure_tuple_rep(mode, len(self.keys))
        self.lazy = bool(getattr(transform, "lazy", False))

    def _each(self, data, fn):
        d = dict(data)
        for i, key in enumerate(self.keys):
            if key not in d:
                if self.allow_missing_keys:
                    continue
                raise KeyError(f"Key `{key}` of transform `{type(self).__name__}` was missing in the data and allow_missing_keys==False.")
            d[key] = fn(d[key], i)
        return d

    def __call__(self, data, lazy: bool | None = None):
        lazy_ = self.lazy if lazy is None else lazy
        if self.mode is None:
            return self._each(data, lambda v, i: self.cropper(v, lazy=lazy_))
        return self._each(data, lambda v, i: self.padder(v, mode=self.mode[i], lazy=lazy_))

    def inverse(self, data):
        return self._each(data, lambda v, i: self.padder.inverse(v))


class SpatialPadd(_KeyedCropPad):
    def __init__(self, keys, spatial_size, method: str = "symmetric", mode="constant", allow_missing_keys: bool = False, lazy: bool = False, **kwargs) -> None:
        super().__init__(keys, SpatialPad(spatial_size, method, lazy=lazy, **kwargs), allow_missing_keys, mode)


class BorderPadd(_KeyedCropPad):
    def __init__(self, keys, spatial_border, mode="constant", allow_missing_keys: bool = False, lazy: bool = False, **kwargs) -> None:
        super().__init__(keys, BorderPad(spatial_border, lazy=lazy, **kwargs), allow_missing_keys, mode)


class DivisiblePadd(_KeyedCropPad):
    def __init__(self, keys, k, mode="constant", method: str = "symmetric", allow_missing_keys: bool = False, lazy: bool = False, **kwargs) -> None:
        super().__init__(keys, DivisiblePad(k, method=method, lazy=lazy, **kwargs), allow_missing_keys, mode)


class SpatialCropd(_KeyedCropPad):
    def __init__(self, keys, roi_center=None, roi_size=None, roi_start=None, roi_end=None, roi_slices=None, allow_missing_keys: bool = False,
                 lazy: bool = False) -> None:
        super().__init__(keys, SpatialCrop(roi_center, roi_size, roi_start, roi_end, roi_slices, lazy=lazy), allow_missing_keys)


class CenterSpatialCropd(_KeyedCropPad):
    def __init__(self, keys, roi_size, allow_missing_keys: bool = False, lazy: bool = False) -> None:
        super().__init__(keys, CenterSpatialCrop(roi_size, lazy=lazy), allow_missing_keys)


SpatialPadD = SpatialPadDict = SpatialPadd
BorderPadD = BorderPadDict = BorderPadd
DivisiblePadD = DivisiblePadDict = DivisiblePadd
SpatialCropD = SpatialCropDict = SpatialCropd
CenterSpatialCropD = CenterSpatialCropDict = CenterSpatialCropd
