"""``SpatialResample`` and ``Spacing`` -- same call signatures and results as
monai/transforms/spatial/array.py:122-253 and :338-546 (eager mode), on the HIP resampling kernel."""

from __future__ import annotations

import warnings
from collections.abc import Sequence
from itertools import zip_longest

import numpy as np
import torch

from ...data.meta_tensor import MetaTensor, is_meta
from ...data.utils import AFFINE_TOL, affine_to_spacing, compute_shape_offset, to_affine_nd, zoom_affine
from ...utils.misc import ensure_tuple
from ... import ops
from ... import config
from ..lazy import LazyCapable, materialize, peek_affine, peek_shape, push_pending
from .functional import _mode_name, _pad_name, memo, resample_plan, spatial_resample

__all__ = ["SpatialResample", "Spacing", "Resample"]

_NP2T = {np.float64: torch.float64, np.float32: torch.float32, float: torch.float64, "float64": torch.float64, "float32": torch.float32}


def _torch_dtype(dtype, default):
    if dtype is None:
        return default
    if isinstance(dtype, torch.dtype):
        return dtype
    try:
        return _NP2T[dtype]
    except (KeyError, TypeError):
        return {np.dtype("float64"): torch.float64, np.dtype("float32"): torch.float32}.get(np.dtype(dtype), torch.float64)


def _wrap(out: torch.Tensor, src, new_affine, op_record):
    """Result as the input's MetaTensor type with the updated affine and the operation pushed on
    ``applied_operations`` (what ``TraceableTransform.track_transform_meta`` does, inverse.py:168-297)."""
    if not is_meta(src):
        return out
    res = type(src)(out, meta=dict(src.meta), applied_operations=list(getattr(src, "applied_operations", [])))
    if new_affine is not None:
        res.meta["affine"] = torch.as_tensor(new_affine, dtype=torch.float64)
    if op_record is not None:
        res.applied_operations.append(op_record)
    return res


class SpatialResample(LazyCapable):
    """Resample from the image's affine to ``dst_affine``: ``xform = solve(src_affine, dst_affine)``, then an affine
    pull with that matrix.  ``lazy=True`` records xform and the output size as a pending operation (monai_amd/transforms/lazy.py)."""

    def __init__(self, mode="bilinear", padding_mode="border", align_corners: bool = False, dtype=np.float64, lazy: bool = False):
        self.mode, self.padding_mode, self.align_corners, self.dtype = mode, padding_mode, align_corners, dtype
        self.lazy = lazy

    def __call__(self, img, dst_affine=None, spatial_size=None, mode=None, padding_mode=None, align_corners=None, dtype=None, lazy=None):
        lazy_ = self.lazy if lazy is None else lazy
        dtype_pt = _torch_dtype(dtype or self.dtype, img.dtype if img.dtype.is_floating_point else torch.float64)
        align_corners = self.align_corners if align_corners is None else align_corners
        mode = self.mode if mode is None else mode
        padding_mode = self.padding_mode if padding_mode is None else padding_mode
        if lazy_:          # functional.py:141-151: nothing is resampled, the composed affine is executed by apply_pending
            _mode_name(mode), _pad_name(padding_mode)         # the same argument errors as the eager call
            orig = peek_shape(img)
            out_size, xform, src_a, _ = resample_plan(orig, peek_affine(img), dst_affine, spatial_size)
            info = {"dtype": str(dtype_pt)[6:], "mode": getattr(mode, "value", mode), "padding_mode": getattr(padding_mode, "value", padding_mode),
                    "align_corners": align_corners, "src_affine": torch.as_tensor(src_a)}
            return push_pending(img, self, xform, [int(v) for v in out_size], info, orig_size=orig)
        img = materialize(img)
        orig_size = tuple(img.shape[1:])
        out, xform, out_size, src_a = spatial_resample(img, dst_affine, spatial_size, mode, padding_mode, align_corners, dtype_pt)
        record = None
        new_affine = None
        if is_meta(img):
            record = {
                "class": type(self).__name__, "orig_size": orig_size,
                "extra_info": {"dtype": str(dtype_pt)[6:], "mode": getattr(mode, "value", mode),
                               "padding_mode": getattr(padding_mode, "value", padding_mode), "align_corners": align_corners,
                               "src_affine": torch.as_tensor(src_a)},
            }
            if xform is not None:  # new affine = old affine @ xform, on the image's full (4x4) affine
                full = np.asarray(img.meta["affine"], dtype=np.float64)
                new_affine = full @ to_affine_nd(len(full) - 1, xform)
        return _wrap(out, img, new_affine, record)

    def inverse(self, data):
        rec = data.applied_operations[-1]
        info = rec["extra_info"]
        prev = type(data)(data.as_tensor(), meta=dict(data.meta), applied_operations=list(data.applied_operations[:-1]))
        out = SpatialResample.__call__(
            self, prev, dst_affine=info["src_affine"], spatial_size=rec["orig_size"], mode=info["mode"], padding_mode=info["padding_mode"],
            # MONAI's convert_applied_interp_mode (Invertd(nearest_interp=True)) rewrites the record with TraceKeys.NONE = "none"
            align_corners=bool(info["align_corners"]) if info["align_corners"] not in (None, "none") else False, dtype=getattr(torch, info["dtype"]),
        )
        if is_meta(out):
            out.applied_operations = list(data.applied_operations[:-1])
        return out


def _affine_bytes(a) -> bytes:
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return a.astype(np.float64, copy=False).tobytes() + bytes(str(a.shape), "ascii")


class Spacing(LazyCapable):
    """Resample the image to voxel size ``pixdim`` (array.py:338-546)."""

    def __init__(self, pixdim, diagonal: bool = False, mode="bilinear", padding_mode="border", align_corners: bool = False, dtype=np.float64,
                 scale_extent: bool = False, recompute_affine: bool = False, min_pixdim=None, max_pixdim=None, lazy: bool = False):
        self.pixdim = np.array(ensure_tuple(pixdim), dtype=np.float64)
        self.min_pixdim = np.array(ensure_tuple(min_pixdim), dtype=np.float64)
        self.max_pixdim = np.array(ensure_tuple(max_pixdim), dtype=np.float64)
        self.diagonal, self.scale_extent, self.recompute_affine = diagonal, scale_extent, recompute_affine
        for mn, mx in zip(self.min_pixdim, self.max_pixdim):
            if (not np.isnan(mn)) and (not np.isnan(mx)) and ((mx < mn) or (mn < 0)):
                raise ValueError(f"min_pixdim {self.min_pixdim} must be positive, smaller than max {self.max_pixdim}.")
        self.sp_resample = SpatialResample(mode=mode, padding_mode=padding_mode, align_corners=align_corners, dtype=dtype, lazy=lazy)
        self._lazy = lazy

    @property
    def lazy(self):
        return self._lazy

    @lazy.setter
    def lazy(self, val) -> None:
        object.__setattr__(self, "_lazy", val)
        self.sp_resample.lazy = val

    def __call__(self, data_array, mode=None, padding_mode=None, align_corners=None, dtype=None, scale_extent=None, output_spatial_shape=None,
                 lazy=None):
        lazy_ = self.lazy if lazy is None else lazy
        if not lazy_:
            data_array = materialize(data_array)
        original_shape = peek_shape(data_array)
        sr = len(original_shape)
        if sr <= 0:
            raise ValueError(f"data_array must have at least one spatial dimension, got {original_shape}.")
        if is_meta(data_array) and "affine" in data_array.meta:
            input_affine = peek_affine(data_array)
        else:
            warnings.warn("`data_array` is not of type MetaTensor, assuming affine to be identity.")
            input_affine = np.eye(sr + 1, dtype=np.float64)
        ac = self.sp_resample.align_corners if align_corners is None else align_corners
        scale_extent = self.scale_extent if scale_extent is None else scale_extent
        if not ac and scale_extent:
            warnings.warn("align_corners=False is not compatible with scale_extent=True.")
        # the target grid is a pure function of (shape, affine, this transform's parameters): computed once per combination (functional.memo)
        key = ("spacing", tuple(int(v) for v in original_shape), _affine_bytes(input_affine), self.pixdim.tobytes(), self.min_pixdim.tobytes(), self.max_pixdim.tobytes(),
               bool(self.diagonal), bool(scale_extent))
        new_affine, output_shape = memo(key, lambda: self._target_grid(original_shape, sr, input_affine, scale_extent))
        new_affine = new_affine.copy()
        actual_shape = list(output_shape) if output_spatial_shape is None else output_spatial_shape
        out = self.sp_resample(data_array, dst_affine=torch.as_tensor(new_affine), spatial_size=actual_shape, mode=mode, padding_mode=padding_mode,
                               align_corners=align_corners, dtype=dtype, lazy=lazy_)
        return self._recomputed_affine(out, original_shape, actual_shape, sr, lazy_)

    def _target_grid(self, original_shape, sr, input_affine, scale_extent):
        """(new affine, output shape) of array.py:497-527: pixdim clamped into [min_pixdim, max_pixdim] -> zoom_affine -> compute_shape_offset"""
        affine_ = to_affine_nd(sr, input_affine)
        out_d = self.pixdim[:sr].copy()
        if out_d.size < sr:
            out_d = np.append(out_d, [out_d[-1]] * (sr - out_d.size))
        orig_d = affine_to_spacing(affine_, sr)
        for idx, (_d, mn, mx) in enumerate(zip_longest(orig_d, self.min_pixdim[:sr], self.max_pixdim[:sr], fillvalue=np.nan)):
            target = out_d[idx]
            mn = target if np.isnan(mn) else min(mn, target)
            mx = target if np.isnan(mx) else max(mx, target)
            if mn > mx:
                raise ValueError(f"min_pixdim is larger than max_pixdim at dim {idx}: min {mn} max {mx} out {target}.")
            out_d[idx] = _d if (mn - AFFINE_TOL) <= _d <= (mx + AFFINE_TOL) else target
        new_affine = zoom_affine(affine_, out_d, diagonal=self.diagonal)
        output_shape, offset = compute_shape_offset(original_shape, affine_, new_affine, scale_extent)
        new_affine[:sr, -1] = offset[:sr]
        return new_affine, tuple(int(v) for v in output_shape)

    def _recomputed_affine(self, out, original_shape, actual_shape, sr, lazy_):
        if self.recompute_affine and is_meta(out):
            # array.py:538-542: the output affine becomes scale_affine(original shape, actual shape) -- the centred scaling between the two voxel
            # grids (transforms/utils.py:2093-2113), which reflects the quantisation of the output shape; host algebra on a 4 x 4 matrix
            if lazy_:
                raise NotImplementedError("recompute_affine is not supported with lazy evaluation.")
            ratio = [float(o) / float(max(int(n), 1)) for o, n in zip(original_shape, actual_shape)]
            scale = np.diag(ratio + [1.0])
            scale[:sr, -1] = (np.diag(scale)[:sr] - 1.0) / 2.0
            out.affine = torch.as_tensor(scale, dtype=torch.float64)
        return out

    def inverse(self, data):
        return self.sp_resample.inverse(data)


class Resample:
    """Sample ``img`` at an explicit coordinate ``grid`` -- the torch branch of monai/transforms/spatial/array.py:1962-2117.

    ``grid`` is ``(sr [+1], spatial...)`` in tensor-axis order; with ``norm_coords=True`` its values are voxel units centred
    on the image (``[-(size-1)/2, (size-1)/2]``), otherwise already normalised to ``[-1, 1]``.  The reference rescales the
    grid, reorders it to xyz and lets ``F.grid_sample`` unnormalise it; here both steps are one scale/offset per axis
    applied inside the sampling kernel.  Output is float32 (array.py:2117)."""

    def __init__(self, mode="bilinear", padding_mode="border", norm_coords: bool = True, device=None, align_corners: bool = False,
                 dtype=np.float64) -> None:
        self.mode, self.padding_mode, self.norm_coords = mode, padding_mode, norm_coords
        self.device, self.align_corners, self.dtype = device, align_corners, dtype

    def __call__(self, img, grid=None, mode=None, padding_mode=None, dtype=None, align_corners=None):
        if grid is None:
            return img
        data = img.as_tensor() if is_meta(img) else img
        dtype_pt = _torch_dtype(dtype or self.dtype, data.dtype if data.dtype.is_floating_point else torch.float64)
        ac = self.align_corners if align_corners is None else align_corners
        sr = min(data.dim() - 1, 3)
        if config.USE_COMPILED:
            return self._compiled(img, data, grid, sr, self.mode if mode is None else mode,
                                  self.padding_mode if padding_mode is None else padding_mode, dtype_pt, ac)
        if sr < 2:
            raise NotImplementedError("monai_amd.Resample: 1-D images are not on the HIP path")
        sizes = [int(v) for v in data.shape[1:1 + sr]]
        scale, offset = [], []
        for dim in sizes:
            s = 2.0 / max(2, dim) if self.norm_coords else 1.0          # array.py:2106-2108
            if ac:                                                       # grid_sample unnormalisation
                scale.append(s * (dim - 1) / 2.0); offset.append((dim - 1) / 2.0)
            else:
                scale.append(s * dim / 2.0); offset.append((dim - 1) / 2.0)
        pad = 3 - sr
        g = torch.as_tensor(grid)[:sr].to(device=data.device)
        if g.dtype not in (torch.float32, torch.float64):
            g = g.to(torch.float64)
        osp = tuple(int(v) for v in g.shape[1:])
        if pad:
            g = torch.cat([torch.zeros((pad,) + osp, dtype=g.dtype, device=g.device), g]).reshape((3,) + (1,) * pad + osp)
        x = data.to(torch.float32).contiguous().reshape((data.shape[0],) + (1,) * pad + tuple(sizes))
        out = ops.grid_resample(x, g.contiguous(), _mode_name(self.mode if mode is None else mode),
                                _pad_name(self.padding_mode if padding_mode is None else padding_mode), bool(ac), dtype_pt == torch.float64,
                                scale=[1.0] * pad + scale, offset=[0.0] * pad + offset)
        out = out.reshape((data.shape[0],) + osp)
        return _wrap(out, img, None, None)

    def _compiled(self, img, data, grid, sr, mode, padding_mode, dtype_pt, ac):
        """``USE_COMPILED`` branch (array.py:2076-2092): the grid is turned into voxel indices and sampled with the native
        ``grid_pull`` (extrapolate=True) in ``dtype``.  Mode mapping of ``resolves_modes(use_compiled=True)``
        (transforms/utils.py:2339-2347): *linear -> order 1, bicubic -> order 3 (a cubic B-spline, not ATen's bicubic),
        nearest -> 0; padding zeros -> zero, border -> replicate, reflection -> dct1."""
        from ...networks.layers import grid_pull

        m = str(getattr(mode, "value", mode)).lower()
        if isinstance(getattr(mode, "value", mode), (int, np.integer)) and not isinstance(mode, bool):
            raise NotImplementedError("monai_amd: spline-order (integer) interpolation modes use scipy in the reference and are not on the HIP path")
        if m.endswith("linear"):
            order = 1
        elif m == "bicubic":
            order = 3
        elif m in ("nearest", "nearest-exact"):
            order = "nearest"
        else:
            raise ValueError(f"Unsupported mode: {mode}, available options are ['bilinear', 'nearest', 'bicubic'].")
        p = _pad_name(padding_mode)
        bound = 1 if p == "reflection" else p
        g = torch.as_tensor(grid)[:sr].to(device=data.device)
        if not g.dtype.is_floating_point:
            g = g.to(torch.float64)
        g = g.clone(memory_format=torch.contiguous_format)
        for i, dim in enumerate(data.shape[1:1 + sr]):
            d_ = max(2, int(dim))
            t = (d_ - 1) / 2.0
            if self.norm_coords:
                g[i] = ((d_ - 1) / d_) * g[i] + t if ac else g[i] + t
            elif ac:
                g[i] = ((d_ - 1) / d_) * (g[i] + 0.5)
        x = data.to(dtype_pt)
        out = grid_pull(x.unsqueeze(0), torch.movedim(g, 0, -1).unsqueeze(0).to(x), bound=bound, extrapolate=True, interpolation=order)[0]
        return _wrap(out.to(torch.float32), img, None, None)
