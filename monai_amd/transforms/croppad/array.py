"""``CropForeground`` -- monai/transforms/croppad/array.py:776-960 on two HIP kernels: the foreground bounding box
(``generate_spatial_bounding_box``, monai/transforms/utils.py:1069-1129) is a single read of the volume with per-workgroup
extrema, the crop and the constant pad of whatever part of the box sticks out of the image are one pass."""

from __future__ import annotations

from collections.abc import Callable, Sequence

import numpy as np
import torch

from ... import ops
from ...data.meta_tensor import affine_np, is_meta
from ..lazy import LazyCapable, materialize, peek_shape, push_pending
from ...utils.misc import as_gather_f32, ensure_tuple, ensure_tuple_rep, fall_back_tuple

__all__ = ["CropForeground", "Pad", "SpatialPad", "BorderPad", "DivisiblePad", "Crop", "SpatialCrop", "CenterSpatialCrop", "is_positive",
           "generate_spatial_bounding_box", "compute_divisible_spatial_size"]


def is_positive(img):
    """monai/transforms/utils.py:212-216"""
    return img > 0


def _as4(data: torch.Tensor) -> torch.Tensor:
    """channel-first image with 1-3 spatial axes -> contiguous fp32 [C, D, H, W] (leading spatial axes of extent 1)"""
    nsp = data.dim() - 1
    if nsp < 1 or nsp > 3:
        raise NotImplementedError(f"monai_amd.CropForeground: channel-first images with 1-3 spatial axes expected, got shape {tuple(data.shape)}")
    return data.reshape((data.shape[0],) + (1,) * (3 - nsp) + tuple(data.shape[1:])).contiguous()


_INT_DTYPES = (torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64, torch.bool)


def _run_crop_pad(data: torch.Tensor, start, size, value: float) -> torch.Tensor:
    """out[c, o] = data[c, o + start] inside `data`, `value` outside -- the crop + constant-pad kernel on a channel-first image with 1-3
    spatial axes.  The kernel copies 32-bit words: float32 directly, int32 as bit patterns, narrower integers / bool through float32
    (exact), int64 only while every value is below 2^24 (`as_gather_f32`)."""
    if data.dtype != torch.float32 and data.dtype not in _INT_DTYPES:
        raise NotImplementedError(f"monai_amd crop / pad: {data.dtype} images are not on the HIP path (float32 and integer images are)")
    nsp = data.dim() - 1
    size = [int(v) for v in size]
    if any(v < 1 for v in size) or not data.numel():
        return data.new_zeros((data.shape[0],) + tuple(max(v, 0) for v in size))
    x32, kval, restore = as_gather_f32(data, value)
    out = ops.crop_pad(_as4(x32), [0] * (3 - nsp) + [int(v) for v in start], [1] * (3 - nsp) + size, kval).reshape((data.shape[0],) + tuple(size))
    return restore(out)


def _wrap_crop_pad(img, out: torch.Tensor, start, cls_name: str, value: float):
    """MetaTensor bookkeeping of a crop / pad whose output voxel o shows source voxel o + start: affine @ translate(start), and the
    record `inverse` needs (what TraceableTransform.track_transform_meta keeps, monai/transforms/inverse.py:168-297)."""
    if not is_meta(img):
        return out
    nsp = out.dim() - 1
    res = type(img)(out, meta=dict(img.meta), applied_operations=list(getattr(img, "applied_operations", [])))
    aff = affine_np(img)
    shift = np.eye(aff.shape[0])
    r = min(nsp, aff.shape[0] - 1)
    shift[:r, -1] = [int(v) for v in start][:r]
    res.meta["affine"] = torch.as_tensor(aff @ shift, dtype=torch.float64)
    res.applied_operations.append({"class": cls_name, "orig_size": tuple(int(v) for v in img.shape[1:]),
                                   "extra_info": {"box_start": [int(v) for v in start], "pad_value": float(value)}})
    return res


def _crop_pad_op(img, transform, start, size, value: float, lazy: bool):
    """One crop / pad: output voxel o shows source voxel o + start (`value` outside the source).  Eager: the kernel pass + the
    MetaTensor bookkeeping; lazy (croppad/functional.py:151-248 with lazy=True): only the pending record translate(start) / size."""
    if lazy:
        r = len(size)
        shift = np.eye(r + 1)
        shift[:r, -1] = [int(v) for v in start]
        return push_pending(img, transform, shift, size, {"box_start": [int(v) for v in start], "pad_value": float(value)}, orig_size=peek_shape(img))
    data = img.as_tensor() if is_meta(img) else torch.as_tensor(img)
    return _wrap_crop_pad(img, _run_crop_pad(data, start, size, value), start, type(transform).__name__, value)


def _inverse_crop_pad(img, cls_name: str):
    """Undo the most recent crop / pad of `cls_name`: the same kernel with the negated start (cropped-away voxels come back as 0,
    padding is cut off) -- Pad.inverse / Crop.inverse, monai/transforms/croppad/array.py:190-204, 441-450."""
    if not is_meta(img) or not getattr(img, "applied_operations", None):
        raise RuntimeError(f"monai_amd.{cls_name}.inverse: a MetaTensor with the forward call's record is required")
    rec = img.applied_operations[-1]
    if rec.get("class") != cls_name:
        raise RuntimeError(f"monai_amd.{cls_name}.inverse: the most recent operation is {rec.get('class')!r}")
    info, orig = rec["extra_info"], tuple(int(v) for v in rec["orig_size"])
    if "box_start" in info:
        start = info["box_start"]
    elif "padded" in info:        # a record written by the reference's pad_func (croppad/functional.py:182): (before, after) per axis, channel first
        start = [-int(p[0]) for p in list(info["padded"])[1:]]
    elif "cropped" in info:       # ... by its crop_func (:233): before / after per spatial axis, flattened
        start = [int(v) for v in list(info["cropped"])[0::2]]
    else:
        raise NotImplementedError(f"monai_amd.{cls_name}.inverse: unknown record {sorted(info)}")
    start = (list(start) + [0] * len(orig))[:len(orig)]
    out = _run_crop_pad(img.as_tensor(), [-s for s in start], orig, 0.0)
    res = type(img)(out, meta=dict(img.meta), applied_operations=list(img.applied_operations[:-1]))
    aff = affine_np(img)
    shift = np.eye(aff.shape[0])
    r = min(len(orig), aff.shape[0] - 1)
    shift[:r, -1] = [-s for s in start][:r]
    res.meta["affine"] = torch.as_tensor(aff @ shift, dtype=torch.float64)
    return res


def _pad_value(mode, kwargs) -> float:
    if str(getattr(mode, "value", mode)).lower() != "constant":
        raise NotImplementedError(f"monai_amd crop / pad: padding mode {mode!r} is not on the HIP path (constant is)")
    kw = dict(kwargs)
    v = kw.pop("value", kw.pop("constant_values", 0.0))
    if isinstance(v, (tuple, list)) or getattr(v, "ndim", 0):
        raise NotImplementedError("monai_amd crop / pad: per-axis constant_values are not on the HIP path (one scalar value is)")
    return float(v)


class Pad(LazyCapable):
    """``monai.transforms.Pad`` (monai/transforms/croppad/array.py:84-204): pad by ``to_pad`` = ``[(before, after), ...]`` including the
    channel axis; constant mode on the crop + pad kernel."""

    def __init__(self, to_pad=None, mode: str = "constant", lazy: bool = False, **kwargs) -> None:
        self.to_pad, self.mode, self.kwargs = to_pad, mode, kwargs
        self.lazy = lazy

    def compute_pad_width(self, spatial_shape):
        raise NotImplementedError(f"subclass {self.__class__.__name__} must implement this method.")

    def __call__(self, img, to_pad=None, mode: str | None = None, lazy: bool | None = None, **kwargs):
        lazy_ = self.lazy if lazy is None else lazy
        if not lazy_:
            img = materialize(img)
        data = img.as_tensor() if is_meta(img) else torch.as_tensor(img)
        shape_ = peek_shape(img) if is_meta(img) else tuple(int(v) for v in data.shape[1:])
        to_pad_ = self.to_pad if to_pad is None else to_pad
        if to_pad_ is None:
            to_pad_ = self.compute_pad_width(shape_)
        kw = dict(self.kwargs)
        kw.update(kwargs)
        value = _pad_value(self.mode if mode is None else mode, kw)
        to_pad_ = [tuple(int(v) for v in p) for p in to_pad_]
        if len(to_pad_) != len(shape_) + 1 or to_pad_[0] != (0, 0):      # rank from the (pending) shape: lazy records may ride on an empty tensor
            raise NotImplementedError(f"monai_amd pad: to_pad must list every axis and leave the channel axis alone, got {to_pad_}")
        start = [-p[0] for p in to_pad_[1:]]
        size = [int(n) + p[0] + p[1] for n, p in zip(shape_, to_pad_[1:])]
        return _crop_pad_op(img, self, start, size, value, lazy_)

    def inverse(self, img):
        return _inverse_crop_pad(img, type(self).__name__)


class SpatialPad(Pad):
    """array.py:207-266: pad up to ``spatial_size`` (symmetric or at the end); axes already large enough are left alone."""

    def __init__(self, spatial_size, method: str = "symmetric", mode: str = "constant", lazy: bool = False, **kwargs) -> None:
        self.spatial_size = spatial_size
        self.method = str(getattr(method, "value", method)).lower()
        if self.method not in ("symmetric", "end"):
            raise ValueError(f"Unsupported method: {method}, available options are ['symmetric', 'end'].")
        super().__init__(mode=mode, lazy=lazy, **kwargs)

    def compute_pad_width(self, spatial_shape):
        spatial_size = fall_back_tuple(self.spatial_size, spatial_shape)
        if self.method == "symmetric":
            widths = [max(sp - spatial_shape[i], 0) for i, sp in enumerate(spatial_size)]
            return tuple([(0, 0)] + [(int(w // 2), int(w - w // 2)) for w in widths])
        return tuple([(0, 0)] + [(0, int(max(sp - spatial_shape[i], 0))) for i, sp in enumerate(spatial_size)])


class BorderPad(Pad):
    """array.py:269-327: the same border on every side, one per axis, or (before, after) per axis."""

    def __init__(self, spatial_border, mode: str = "constant", lazy: bool = False, **kwargs) -> None:
        self.spatial_border = spatial_border
        super().__init__(mode=mode, lazy=lazy, **kwargs)

    def compute_pad_width(self, spatial_shape):
        b = ensure_tuple(self.spatial_border)
        if not all(isinstance(v, int) for v in b):
            raise ValueError(f"self.spatial_border must contain only ints, got {b}.")
        b = tuple(max(0, v) for v in b)
        n = len(spatial_shape)
        if len(b) == 1:
            w = [(b[0], b[0])] * n
        elif len(b) == n:
            w = [(v, v) for v in b]
        elif len(b) == 2 * n:
            w = [(b[2 * i], b[2 * i + 1]) for i in range(n)]
        else:
            raise ValueError(f"Unsupported spatial_border length: {len(b)}, available options are [1, len(spatial_shape)={n}, 2*len(spatial_shape)={2 * n}].")
        return tuple([(0, 0)] + w)


class DivisiblePad(Pad):
    """array.py:330-376: pad every spatial extent up to the next multiple of ``k``."""

    def __init__(self, k, mode: str = "constant", method: str = "symmetric", lazy: bool = False, **kwargs) -> None:
        self.k, self.method = k, method
        super().__init__(mode=mode, lazy=lazy, **kwargs)

    def compute_pad_width(self, spatial_shape):
        return SpatialPad(compute_divisible_spatial_size(spatial_shape, self.k), method=self.method).compute_pad_width(spatial_shape)


class Crop(LazyCapable):
    """``monai.transforms.Crop`` (array.py:379-450): crop by per-axis slices (step 1); the result is a dense copy."""

    def __init__(self, lazy: bool = False):
        self.lazy = lazy

    @staticmethod
    def compute_slices(roi_center=None, roi_size=None, roi_start=None, roi_end=None, roi_slices=None):
        """array.py:388-424: slices from centre + size, start + end (clamped to >= 0, end >= start) or given slices"""
        if roi_slices:
            if not all(s.step is None or s.step == 1 for s in roi_slices):
                raise ValueError(f"only slice steps of 1/None are currently supported, got {roi_slices}.")
            return tuple(roi_slices)
        as_list = lambda v: [int(a) for a in np.atleast_1d(np.asarray(v.cpu() if isinstance(v, torch.Tensor) else v))]  # noqa: E731
        if roi_center is not None and roi_size is not None:
            c, sz = as_list(roi_center), as_list(roi_size)
            start = [max(a - b // 2, 0) for a, b in zip(c, sz)]
            end = [max(a + b, a) for a, b in zip(start, sz)]
        else:
            if roi_start is None or roi_end is None:
                raise ValueError("please specify either roi_center, roi_size or roi_start, roi_end.")
            start = [max(a, 0) for a in as_list(roi_start)]
            end = [max(a, b) for a, b in zip(as_list(roi_end), start)]
        return tuple(slice(a, b) for a, b in zip(start, end))

    def __call__(self, img, slices, lazy: bool | None = None):
        lazy_ = self.lazy if lazy is None else lazy
        if not lazy_:
            img = materialize(img)
        data = img.as_tensor() if is_meta(img) else torch.as_tensor(img)
        shape_ = peek_shape(img) if is_meta(img) else tuple(int(v) for v in data.shape[1:])
        sd = len(shape_)                                 # rank from the (pending) shape: the reference's Zoom sends a lazy record on an empty tensor
        slices_ = (list(slices) + [slice(None)] * sd)[:sd]
        rng = [s.indices(int(n)) for s, n in zip(slices_, shape_)]
        start = [r[0] for r in rng]
        size = [max(r[1] - r[0], 0) for r in rng]
        return _crop_pad_op(img, self, start, size, 0.0, lazy_)

    def inverse(self, img):
        return _inverse_crop_pad(img, type(self).__name__)


class SpatialCrop(Crop):
    """array.py:453-500"""

    def __init__(self, roi_center=None, roi_size=None, roi_start=None, roi_end=None, roi_slices=None, lazy: bool = False) -> None:
        super().__init__(lazy)
        self.slices = self.compute_slices(roi_center=roi_center, roi_size=roi_size, roi_start=roi_start, roi_end=roi_end, roi_slices=roi_slices)

    def __call__(self, img, lazy: bool | None = None):
        return super().__call__(img=img, slices=ensure_tuple(self.slices), lazy=lazy)


class CenterSpatialCrop(Crop):
    """array.py:503-539: a centred box of ``roi_size`` (non-positive components keep the axis whole)"""

    def __init__(self, roi_size, lazy: bool = False) -> None:
        super().__init__(lazy=lazy)
        self.roi_size = roi_size

    def compute_slices(self, spatial_size):      # the reference's override (array.py:506-509): slices of the centred box for this image size
        roi_size = fall_back_tuple(self.roi_size, spatial_size)
        return Crop.compute_slices(roi_center=[int(i) // 2 for i in spatial_size], roi_size=roi_size)

    def __call__(self, img, lazy: bool | None = None):
        lazy_ = self.lazy if lazy is None else lazy
        spatial = peek_shape(img) if (lazy_ and is_meta(img)) else tuple(int(v) for v in materialize(img).shape[1:])
        return super().__call__(img=img, slices=self.compute_slices(spatial), lazy=lazy)


def generate_spatial_bounding_box(img, select_fn: Callable = is_positive, channel_indices=None, margin: Sequence[int] | int = 0,
                                  allow_smaller: bool = False):
    """Start / end (exclusive) of the foreground box per spatial axis, with the reference's margin and clipping rules
    (monai/transforms/utils.py:1069-1129); ``[0] * ndim, [0] * ndim`` when nothing is selected."""
    data = img.as_tensor() if is_meta(img) else torch.as_tensor(img)
    spatial_size = tuple(int(v) for v in data.shape[1:])
    ndim = len(spatial_size)
    if channel_indices is not None:
        data = data[list(ensure_tuple(channel_indices))]
    if getattr(select_fn, "__name__", "") == "is_positive" and data.dtype == torch.float32:
        mask = data                                   # the kernel's own test is `> 0`
    else:
        mask = select_fn(data).to(torch.float32)      # an arbitrary selection runs on the device tensor, the kernel boxes its result
    margin = ensure_tuple_rep(margin, ndim)
    for m in margin:
        if m < 0:
            raise ValueError(f"margin value should not be negative number, got {margin}.")
    box = ops.foreground_bbox(_as4(mask))
    if box is None:
        return [0] * ndim, [0] * ndim
    lo, hi = box[:3][3 - ndim:], box[3:][3 - ndim:]
    box_start, box_end = [0] * ndim, [0] * ndim
    for di in range(ndim):
        min_d = lo[di] - margin[di]
        max_d = hi[di] + margin[di] + 1
        if allow_smaller:
            min_d = max(min_d, 0)
            max_d = min(max_d, spatial_size[di])
        box_start[di], box_end[di] = int(min_d), int(max_d)
    return box_start, box_end


def compute_divisible_spatial_size(spatial_shape: Sequence[int], k: Sequence[int] | int):
    """monai/transforms/utils.py:1803-1820"""
    k = fall_back_tuple(k, (1,) * len(spatial_shape))
    return tuple(int(np.ceil(dim / k_d) * k_d) if k_d > 0 else dim for k_d, dim in zip(k, spatial_shape))


class CropForeground(LazyCapable):
    """Crop an image to the bounding box of its foreground (``select_fn``, default ``> 0``), with ``margin``, ``k_divisible``
    and a constant pad where the box leaves the image.  Same constructor / call signature as the reference; ``mode`` other
    than ``"constant"`` is not on the HIP path.  ``lazy=True``: the box needs the CURRENT voxel values (``requires_current_data``, as in
    the reference): pending operations are executed first, then the crop + pad itself is recorded as one pending operation."""

    @property
    def requires_current_data(self):
        return True

    def __init__(self, select_fn: Callable = is_positive, channel_indices=None, margin: Sequence[int] | int = 0, allow_smaller: bool = False,
                 return_coords: bool = False, k_divisible: Sequence[int] | int = 1, mode: str = "constant", lazy: bool = False, **pad_kwargs) -> None:
        self.select_fn = select_fn
        self.channel_indices = ensure_tuple(channel_indices) if channel_indices is not None else None
        self.margin, self.allow_smaller, self.return_coords, self.k_divisible = margin, allow_smaller, return_coords, k_divisible
        self.mode, self.pad_kwargs = mode, pad_kwargs
        self.lazy = lazy

    def compute_bounding_box(self, img):
        """Box of the foreground, grown symmetrically to sizes divisible by ``k_divisible`` (array.py:847-867)."""
        box_start, box_end = generate_spatial_bounding_box(img, self.select_fn, self.channel_indices, self.margin, self.allow_smaller)
        box_start_ = np.asarray(box_start, dtype=np.int16)
        box_end_ = np.asarray(box_end, dtype=np.int16)
        orig_spatial_size = box_end_ - box_start_
        spatial_size = np.asarray(compute_divisible_spatial_size(orig_spatial_size.tolist(), k=self.k_divisible))
        box_start_ = box_start_ - np.floor_divide(np.asarray(spatial_size) - orig_spatial_size, 2)
        box_end_ = box_start_ + spatial_size
        return box_start_, box_end_

    def crop_pad(self, img, box_start: np.ndarray, box_end: np.ndarray, mode: str | None = None, lazy: bool = False, **pad_kwargs):
        """Crop to ``[max(start, 0), end)`` and pad what lies outside the image (array.py:884-927) -- one kernel pass."""
        kw = dict(self.pad_kwargs)
        kw.update(pad_kwargs)
        value = _pad_value(self.mode if mode is None else mode, kw)
        data = img.as_tensor() if is_meta(img) else torch.as_tensor(img)
        nsp = data.dim() - 1
        start = [int(v) for v in np.asarray(box_start).tolist()]
        end = [max(int(e), max(s, 0)) for s, e in zip(start, np.asarray(box_end).tolist())]       # Crop.compute_slices: end >= start >= 0
        if len(start) != nsp:
            raise ValueError(f"monai_amd.CropForeground: a {len(start)}-D box does not fit an image of shape {tuple(data.shape)}")
        size = [e - s for s, e in zip(start, end)]
        return _crop_pad_op(img, self, start, size, value, lazy)

    def inverse(self, img):
        """Undo the most recent crop + pad (array.py:950-960: crop the padding away, zero-pad back to the original size) -- the
        same kernel with the negated start: voxels outside the crop box come back as 0."""
        return _inverse_crop_pad(img, type(self).__name__)

    def __call__(self, img, mode: str | None = None, lazy: bool | None = None, **pad_kwargs):
        lazy_ = self.lazy if lazy is None else lazy
        img = materialize(img)                         # the box is computed on the current data (array.py:929-948)
        box_start, box_end = self.compute_bounding_box(img)
        cropped = self.crop_pad(img, box_start, box_end, mode, lazy=bool(lazy_), **pad_kwargs)
        if self.return_coords:
            return cropped, box_start, box_end
        return cropped
