"""``Orientation`` / ``Orientationd`` -- monai/transforms/spatial/array.py:549-668, functional.py:187-229,
dictionary.py:534-606: re-orient a channel-first image to the requested axis codes, as one strided-gather HIP pass
(``torch.flip`` + ``permute`` fused) on the device tensor.

The reference delegates the orientation algebra to nibabel (``nibabel.orientations``: ``io_orientation``, ``axcodes2ornt``,
``ornt_transform``, ``inv_ornt_aff``, ``aff2axcodes``; requirements-dev.txt lists ``nibabel`` unpinned, and it is NOT
installed in the build container).  The functions below restate nibabel's published algorithms (nibabel/orientations.py,
5.x); parity is anchored on the reference's own known-answer tests, tests/transforms/test_orientation.py:29-187 and
test_orientationd.py, restated in tests/orientation_cases.py ("parity pinned to the reference's test vectors, not to a live
reference run").  Host math is numpy float64 on <= 4x4 matrices; no image data is touched on the host."""

from __future__ import annotations

import warnings
from collections.abc import Sequence

import numpy as np
import torch

from ... import ops
from ..lazy import LazyCapable, LazyCapableDict, materialize, peek_affine, peek_shape, push_pending
from ...data.meta_tensor import is_meta
from ...data.utils import to_affine_nd
from ...utils.misc import as_gather_f32, ensure_tuple

__all__ = ["Orientation", "Orientationd", "OrientationD", "OrientationDict", "io_orientation", "axcodes2ornt", "ornt_transform",
           "inv_ornt_aff", "aff2axcodes"]

_LABELS = (("L", "R"), ("P", "A"), ("I", "S"))


def io_orientation(affine, tol=None) -> np.ndarray:
    """nibabel.orientations.io_orientation: for each input axis the closest output axis and its direction (rows ``[axis, flip]``),
    from the polar-decomposition rotation of the zoom-normalised affine; each output axis is used once."""
    affine = np.asarray(affine, dtype=np.float64)
    q, p = affine.shape[0] - 1, affine.shape[1] - 1
    rzs = affine[:q, :p]
    zooms = np.sqrt(np.sum(rzs * rzs, axis=0))
    zooms[zooms == 0] = 1
    rs = rzs / zooms
    pm, s, qs = np.linalg.svd(rs, full_matrices=False)
    if tol is None:
        tol = s.max() * max(rs.shape) * np.finfo(s.dtype).eps
    keep = s > tol
    r = np.dot(pm[:, keep], qs[keep])
    ornt = np.ones((p, 2), dtype=np.int8) * np.nan
    for in_ax in range(p):
        col = r[:, in_ax]
        if not np.allclose(col, 0):
            out_ax = np.argmax(np.abs(col))
            ornt[in_ax, 0] = out_ax
            ornt[in_ax, 1] = -1 if col[out_ax] < 0 else 1
            r[out_ax, :] = 0          # this output axis is taken
    return ornt


def axcodes2ornt(axcodes, labels=None) -> np.ndarray:
    """nibabel.orientations.axcodes2ornt"""
    labels = list(zip("LPI", "RAS")) if labels is None else labels
    allowed = {c for pair in labels for c in pair} | {None}
    if len(allowed) != 2 * len(labels) + 1:
        raise ValueError(f"Duplicate labels in {labels}")
    if not set(axcodes).issubset(allowed):
        raise ValueError(f"Not all axis codes {list(axcodes)} in label set {allowed}")
    ornt = np.ones((len(axcodes), 2), dtype=np.int8) * np.nan
    for code_idx, code in enumerate(axcodes):
        for label_idx, codes in enumerate(labels):
            if code is None:
                continue
            if code in codes:
                ornt[code_idx, :] = [label_idx, -1 if code == codes[0] else 1]
                break
    return ornt


def ornt_transform(start_ornt, end_ornt) -> np.ndarray:
    """nibabel.orientations.ornt_transform: the orientation that takes an array in `start_ornt` to `end_ornt`"""
    start_ornt, end_ornt = np.asarray(start_ornt), np.asarray(end_ornt)
    if start_ornt.shape != end_ornt.shape:
        raise ValueError("The orientations must have the same shape")
    if start_ornt.shape[1] != 2:
        raise ValueError(f"Invalid shape for an orientation: {start_ornt.shape}")
    result = np.empty_like(start_ornt)
    for end_in_idx, (end_out_idx, end_flip) in enumerate(end_ornt):
        for start_in_idx, (start_out_idx, start_flip) in enumerate(start_ornt):
            if end_out_idx == start_out_idx:
                result[start_in_idx, :] = [end_in_idx, 1 if start_flip == end_flip else -1]
                break
        else:
            raise ValueError(f"Unable to find out axis {end_out_idx} in start_ornt")
    return result


def inv_ornt_aff(ornt, shape) -> np.ndarray:
    """nibabel.orientations.inv_ornt_aff: the affine from the re-oriented array's voxel space back to the original one"""
    ornt = np.asarray(ornt)
    if np.any(np.isnan(ornt)):
        raise ValueError("We cannot invert orientation transform")
    p = ornt.shape[0]
    shape = np.array(shape)[:p]
    axis_transpose = [int(v) for v in ornt[:, 0]]
    undo_reorder = np.eye(p + 1)[axis_transpose + [p], :]
    undo_flip = np.diag(list(ornt[:, 1]) + [1.0])
    center_trans = -(shape - 1) / 2.0
    undo_flip[:p, p] = (ornt[:, 1] * center_trans) - center_trans
    return np.dot(undo_flip, undo_reorder)


def aff2axcodes(aff, labels=None, tol=None):
    """nibabel.orientations.aff2axcodes"""
    labels = _LABELS if labels is None else labels
    codes = []
    for axno, direction in io_orientation(aff, tol):
        if np.isnan(axno):
            codes.append(None)
            continue
        codes.append(labels[int(np.round(axno))][1 if direction == 1 else 0])
    return tuple(codes)


class Orientation(LazyCapable):
    """Change the input image's orientation into the one given by ``axcodes`` (or the closest canonical one), updating the
    MetaTensor's affine.  Same constructor / call signature as the reference; ``lazy=True`` records the signed axis permutation as a
    pending operation (monai_amd/transforms/lazy.py)."""

    def __init__(self, axcodes: str | None = None, as_closest_canonical: bool = False, labels: Sequence[tuple[str, str]] | None = _LABELS,
                 lazy: bool = False) -> None:
        if axcodes is None and not as_closest_canonical:
            raise ValueError("Incompatible values: axcodes=None and as_closest_canonical=True.")
        if axcodes is not None and as_closest_canonical:
            warnings.warn("using as_closest_canonical=True, axcodes ignored.")
        self.axcodes, self.as_closest_canonical, self.labels = axcodes, as_closest_canonical, labels
        self.lazy = lazy

    def __call__(self, data_array, lazy: bool | None = None):
        lazy_ = self.lazy if lazy is None else lazy
        if not lazy_:
            data_array = materialize(data_array)
        data = data_array.as_tensor() if is_meta(data_array) else torch.as_tensor(data_array)
        spatial_shape = peek_shape(data_array) if is_meta(data_array) else tuple(int(v) for v in data.shape[1:])
        sr = len(spatial_shape)
        if sr <= 0:
            raise ValueError(f"data_array must have at least one spatial dimension, got {spatial_shape}.")
        if sr > 3:
            raise NotImplementedError(f"monai_amd.Orientation: {sr} spatial axes are not on the HIP path (1-3 are)")
        if is_meta(data_array):
            affine_np = peek_affine(data_array)
            affine_ = to_affine_nd(sr, affine_np)
        else:
            warnings.warn("`data_array` is not of type `MetaTensor, assuming affine to be identity.")
            affine_np = np.eye(sr + 1, dtype=np.float64)
            affine_ = np.eye(sr + 1, dtype=np.float64)
        src = io_orientation(affine_)
        if self.as_closest_canonical:
            spatial_ornt = src
        else:
            if self.axcodes is None:
                raise ValueError("Incompatible values: axcodes=None and as_closest_canonical=True.")
            if sr < len(self.axcodes):
                warnings.warn(f"axcodes ('{self.axcodes}') length is smaller than number of input spatial dimensions D={sr}.\n"
                              f"{self.__class__.__name__}: spatial shape = {spatial_shape}, channels = {data.shape[0]},"
                              "please make sure the input is in the channel-first format.")
            dst = axcodes2ornt(self.axcodes[:sr], labels=self.labels)
            if len(dst) < sr:
                raise ValueError(f"axcodes must match data_array spatially, got axcodes={len(self.axcodes)}D data_array={sr}D")
            spatial_ornt = ornt_transform(src, dst)
        xform = inv_ornt_aff(spatial_ornt, spatial_shape)                 # functional.py:189
        # output axis k shows input axis perm[k]; input axis a is reversed when its flip is -1 (functional.py:192-218)
        perm = [int(v) for v in np.argsort(spatial_ornt[:, 0])]
        flips = [bool(f == -1) for f in spatial_ornt[:, 1]]
        if lazy_:          # functional.py:219-228: only the record; apply_pending runs it as a flip / permute (or fused into a resampling)
            return push_pending(data_array, self, xform, [spatial_shape[p] for p in perm], {"original_affine": affine_np}, orig_size=spatial_shape)
        ints = (torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64, torch.bool)
        if data.dtype != torch.float32 and data.dtype not in ints:
            raise NotImplementedError(f"monai_amd.Orientation: {data.dtype} images are not on the HIP path (float32 and integer images are)")
        pad = 3 - sr
        x32, _, restore = as_gather_f32(data)         # label maps: int32 as bit patterns, exact for every value
        x4 = x32.reshape((data.shape[0],) + (1,) * pad + spatial_shape).contiguous()
        out = ops.flip_permute(x4, list(range(pad)) + [p + pad for p in perm], [False] * pad + flips)
        out = restore(out.reshape((data.shape[0],) + tuple(spatial_shape[p] for p in perm)))
        if not is_meta(data_array):
            return out
        res = type(data_array)(out, meta=dict(data_array.meta), applied_operations=list(getattr(data_array, "applied_operations", [])))
        full = np.asarray(affine_np, dtype=np.float64)
        res.meta["affine"] = torch.as_tensor(full @ to_affine_nd(full.shape[0] - 1, xform), dtype=torch.float64)
        res.applied_operations.append({"class": type(self).__name__, "orig_size": spatial_shape, "extra_info": {"original_affine": affine_np}})
        return res

    def inverse(self, data):
        """Back to the orientation recorded by the forward call (array.py:651-663)."""
        rec = data.applied_operations[-1]
        orig_axcodes = aff2axcodes(rec["extra_info"]["original_affine"])
        prev = type(data)(data.as_tensor(), meta=dict(data.meta), applied_operations=list(data.applied_operations[:-1]))
        out = Orientation(axcodes=orig_axcodes, as_closest_canonical=False, labels=self.labels)(prev)
        out.applied_operations = list(data.applied_operations[:-1])
        return out


class Orientationd(LazyCapableDict):
    """Dictionary version (monai/transforms/spatial/dictionary.py:534-606)."""

    _lazy_inner = ("ornt_transform",)

    def __init__(self, keys, axcodes: str | None = None, as_closest_canonical: bool = False, labels: Sequence[tuple[str, str]] | None = _LABELS,
                 allow_missing_keys: bool = False, lazy: bool = False) -> None:
        self.keys = ensure_tuple(keys)
        self.allow_missing_keys = allow_missing_keys
        self.ornt_transform = Orientation(axcodes=axcodes, as_closest_canonical=as_closest_canonical, labels=labels, lazy=lazy)
        self.lazy = lazy

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
        return self._each(data, lambda v: self.ornt_transform(v, lazy=lazy_))

    def inverse(self, data):
        return self._each(data, self.ornt_transform.inverse)


OrientationD = OrientationDict = Orientationd
