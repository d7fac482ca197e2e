"""``CropForeground`` -- monai/transforms/croppad/array.py:776-960 on two HIP kernels: the foreground bounding box
(``generate_spatial_bounding_box``, monai/transforms/utils.py:1069-1129) is a single read of the volume with per-workgroup
extrema, the crop and the constant pad of whatever part of the box sticks out of the image are one pass."""

from __future__ import annotations

from collections.abc import Callable, Sequence

import numpy as np
import torch

from ... import ops
from ...data.meta_tensor import is_meta
from ...utils.misc import ensure_tuple, ensure_tuple_rep, fall_back_tuple

__all__ = ["CropForeground", "is_positive", "generate_spatial_bounding_box", "compute_divisible_spatial_size"]


def is_positive(img):
    """monai/transforms/utils.py:212-216"""
    return img > 0


def _as4(data: torch.Tensor) -> torch.Tensor:
    """channel-first image with 1-3 spatial axes -> contiguous fp32 [C, D, H, W] (leading spatial axes of extent 1)"""
    nsp = data.dim() - 1
    if nsp < 1 or nsp > 3:
        raise NotImplementedError(f"monai_amd.CropForeground: channel-first images with 1-3 spatial axes expected, got shape {tuple(data.shape)}")
    return data.reshape((data.shape[0],) + (1,) * (3 - nsp) + tuple(data.shape[1:])).contiguous()


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


class CropForeground:
    """Crop an image to the bounding box of its foreground (``select_fn``, default ``> 0``), with ``margin``, ``k_divisible``
    and a constant pad where the box leaves the image.  Same constructor / call signature as the reference; ``mode`` other
    than ``"constant"`` and lazy execution are not on the HIP path (they raise)."""

    def __init__(self, select_fn: Callable = is_positive, channel_indices=None, margin: Sequence[int] | int = 0, allow_smaller: bool = False,
                 return_coords: bool = False, k_divisible: Sequence[int] | int = 1, mode: str = "constant", lazy: bool = False, **pad_kwargs) -> None:
        if lazy:
            raise NotImplementedError("monai_amd.CropForeground: lazy execution is not implemented")
        self.select_fn = select_fn
        self.channel_indices = ensure_tuple(channel_indices) if channel_indices is not None else None
        self.margin, self.allow_smaller, self.return_coords, self.k_divisible = margin, allow_smaller, return_coords, k_divisible
        self.mode, self.pad_kwargs, self.lazy = mode, pad_kwargs, False

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
        if lazy:
            raise NotImplementedError("monai_amd.CropForeground: lazy execution is not implemented")
        mode = self.mode if mode is None else mode
        if str(getattr(mode, "value", mode)).lower() != "constant":
            raise NotImplementedError(f"monai_amd.CropForeground: padding mode {mode!r} is not on the HIP path (constant is)")
        kw = dict(self.pad_kwargs)
        kw.update(pad_kwargs)
        value = float(kw.pop("value", kw.pop("constant_values", 0.0)))
        data = img.as_tensor() if is_meta(img) else torch.as_tensor(img)
        if data.dtype != torch.float32:
            raise NotImplementedError(f"monai_amd.CropForeground: {data.dtype} images are not on the HIP path (float32 is)")
        nsp = data.dim() - 1
        start = [int(v) for v in np.asarray(box_start).tolist()]
        end = [max(int(e), max(s, 0)) for s, e in zip(start, np.asarray(box_end).tolist())]       # Crop.compute_slices: end >= start >= 0
        if len(start) != nsp:
            raise ValueError(f"monai_amd.CropForeground: a {len(start)}-D box does not fit an image of shape {tuple(data.shape)}")
        size = [e - s for s, e in zip(start, end)]
        if any(s < 1 for s in size):
            out = data.new_empty((data.shape[0],) + tuple(max(s, 0) for s in size))
        else:
            out = ops.crop_pad(_as4(data), [0] * (3 - nsp) + start, [1] * (3 - nsp) + size, value).reshape((data.shape[0],) + tuple(size))
        if not is_meta(img):
            return out
        res = type(img)(out, meta=dict(img.meta), applied_operations=list(getattr(img, "applied_operations", [])))
        aff = np.asarray(img.meta["affine"], dtype=np.float64) if "affine" in img.meta else np.eye(4)
        shift = np.eye(aff.shape[0])
        r = min(nsp, aff.shape[0] - 1)
        shift[:r, -1] = start[:r]                                # crop moves the origin by +start, the front pad by -pad: in total `start`
        res.meta["affine"] = torch.as_tensor(aff @ shift, dtype=torch.float64)
        res.applied_operations.append({"class": type(self).__name__, "orig_size": tuple(data.shape[1:]),
                                       "extra_info": {"box_start": start, "box_end": end, "pad_value": value}})
        return res

    def inverse(self, img):
        """Undo the most recent crop + pad (array.py:950-960: crop the padding away, zero-pad back to the original size) -- the
        same kernel with the negated start: voxels outside the crop box come back as 0."""
        if not is_meta(img) or not getattr(img, "applied_operations", None):
            raise RuntimeError("monai_amd.CropForeground.inverse: a MetaTensor with the forward call's record is required")
        rec = img.applied_operations[-1]
        if rec.get("class") != type(self).__name__:
            raise RuntimeError(f"monai_amd.CropForeground.inverse: the most recent operation is {rec.get('class')!r}")
        data = img.as_tensor()
        start, orig = rec["extra_info"]["box_start"], tuple(int(v) for v in rec["orig_size"])
        nsp = len(orig)
        out = ops.crop_pad(_as4(data), [0] * (3 - nsp) + [-s for s in start], [1] * (3 - nsp) + list(orig), 0.0).reshape((data.shape[0],) + orig)
        res = type(img)(out, meta=dict(img.meta), applied_operations=list(img.applied_operations[:-1]))
        aff = np.asarray(img.meta["affine"], dtype=np.float64)
        shift = np.eye(aff.shape[0])
        r = min(nsp, aff.shape[0] - 1)
        shift[:r, -1] = [-s for s in start][:r]
        res.meta["affine"] = torch.as_tensor(aff @ shift, dtype=torch.float64)
        return res

    def __call__(self, img, mode: str | None = None, lazy: bool | None = None, **pad_kwargs):
        if lazy:
            raise NotImplementedError("monai_amd.CropForeground: lazy execution is not implemented")
        box_start, box_end = self.compute_bounding_box(img)
        cropped = self.crop_pad(img, box_start, box_end, mode, **pad_kwargs)
        if self.return_coords:
            return cropped, box_start, box_end
        return cropped
