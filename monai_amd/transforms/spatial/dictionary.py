"""``Spacingd`` -- dictionary wrapper with the reference's signature (monai/transforms/spatial/dictionary.py:365-531)."""

from __future__ import annotations

import warnings
from collections.abc import Hashable, Mapping, Sequence

import numpy as np

from ...utils.misc import ensure_tuple, ensure_tuple_rep
from ..lazy import LazyCapableDict, peek_shape
from .array import Spacing

__all__ = ["Spacingd", "SpacingD", "SpacingDict"]


class Spacingd(LazyCapableDict):
    _lazy_inner = ("spacing_transform",)

    def __init__(self, keys, pixdim, diagonal: bool = False, mode="bilinear", padding_mode="border", align_corners=False, dtype=np.float64,
                 scale_extent: bool = False, recompute_affine: bool = False, min_pixdim=None, max_pixdim=None, ensure_same_shape: bool = True,
                 allow_missing_keys: bool = False, lazy: bool = False) -> None:
        self.keys = ensure_tuple(keys)
        if not self.keys:
            raise ValueError("keys must be non empty.")
        self.allow_missing_keys = allow_missing_keys
        self.spacing_transform = Spacing(pixdim, diagonal=diagonal, recompute_affine=recompute_affine, min_pixdim=min_pixdim,
                                         max_pixdim=max_pixdim, lazy=lazy)
        n = len(self.keys)
        self.mode = ensure_tuple_rep(mode, n)
        self.padding_mode = ensure_tuple_rep(padding_mode, n)
        self.align_corners = ensure_tuple_rep(align_corners, n)
        self.dtype = ensure_tuple_rep(dtype, n)
        self.scale_extent = ensure_tuple_rep(scale_extent, n)
        self.ensure_same_shape = ensure_same_shape
        self.lazy = lazy

    def __call__(self, data: Mapping[Hashable, object], lazy=None) -> dict:
        lazy_ = self.lazy if lazy is None else lazy
        d = dict(data)
        _init_shape, _pixdim, should_match = None, None, False
        output_shape_k = None  # first key's output shape, reused so that image / label keep matching shapes (:503-512)
        for key, mode, padding_mode, align_corners, dtype, scale_extent in zip(self.keys, self.mode, self.padding_mode, self.align_corners,
                                                                            self.dtype, self.scale_extent):
            if key not in d:
                if self.allow_missing_keys:
                    continue
                raise KeyError(f"Key `{key}` of transform `{type(self).__name__}` was missing in the data and allow_missing_keys==False.")
            if self.ensure_same_shape and hasattr(d[key], "meta"):
                if _init_shape is None and _pixdim is None:
                    _init_shape, _pixdim = peek_shape(d[key]), _pixdim_of(d[key])
                else:
                    should_match = np.allclose(_init_shape, peek_shape(d[key])) and np.allclose(_pixdim, _pixdim_of(d[key]), atol=1e-3)
            d[key] = self.spacing_transform(d[key], mode=mode, padding_mode=padding_mode, align_corners=align_corners, dtype=dtype,
                                            scale_extent=scale_extent, output_spatial_shape=output_shape_k if should_match else None, lazy=lazy_)
            if output_shape_k is None:
                output_shape_k = peek_shape(d[key])
        return d

    def inverse(self, data: Mapping[Hashable, object]) -> dict:
        d = dict(data)
        for key in self.keys:
            if key in d:
                d[key] = self.spacing_transform.inverse(d[key])
        return d


def _pixdim_of(x):
    a = np.asarray(x.meta["affine"], dtype=np.float64)
    return np.sqrt(np.sum(a[:3, :3] * a[:3, :3], axis=0))


SpacingD = SpacingDict = Spacingd
