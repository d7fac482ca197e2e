"""Lazy resampling: spatial transforms constructed / called with ``lazy=True`` only RECORD what they would do --
a voxel-space affine and an output shape, pushed on the MetaTensor's ``pending_operations`` -- and ``apply_pending`` composes the
recorded affines on the host and executes them as ONE resampling pass (``mh_affine_resample_f32``; pure flips / axis permutations /
integer shifts go through the exact gather kernels instead).  ``Orientationd -> Spacingd`` in front of a CT bundle's network moves
8 B per voxel once instead of once per transform, and interpolates once.

Reference: monai/transforms/lazy/functional.py:196-300 (``apply_pending``), monai/transforms/lazy/utils.py:148-229 (``resample``),
monai/transforms/inverse.py:168-297 (``track_transform_meta``: what a pending record holds), monai/data/meta_tensor.py:470-509
(``peek_pending_shape`` / ``peek_pending_affine``).  As in the reference a pending record carries no interpolation settings: the
fused resampling is bilinear / border in float64 coordinates unless ``overrides`` say otherwise, and records whose composed
matrix is a pure axis operation are executed without interpolation (zero padding).

With MONAI installed and ``monai_amd.patch.install()`` the reference's own ``Compose(..., lazy=True)`` drives this: the lazy-capable
classes become real ``LazyTrait`` subclasses and the reference's ``resample`` is rebound to the one below.
"""

from __future__ import annotations

import numpy as np
import torch

from ..data.meta_tensor import MetaTensor, is_meta
from ..data.utils import AFFINE_TOL, to_affine_nd

__all__ = ["LazyAttr", "LazyCapable", "LazyCapableDict", "apply_pending", "apply_pending_transforms", "resample", "ApplyPending", "ApplyPendingd", "push_pending",
           "materialize", "peek_shape", "peek_affine"]


class LazyAttr:
    """monai/utils/enums.py `LazyAttr` (plain strings: the records are shared with the reference's machinery)"""

    SHAPE = "lazy_shape"
    AFFINE = "lazy_affine"
    PADDING_MODE = "lazy_padding_mode"
    INTERP_MODE = "lazy_interpolation_mode"
    DTYPE = "lazy_dtype"
    ALIGN_CORNERS = "lazy_align_corners"
    RESAMPLE_MODE = "lazy_resample_mode"


class _Root:
    """so that `LazyCapable.__bases__` can be extended with the reference's `LazyTrait` when MONAI is installed (patch.py)"""


class LazyCapable(_Root):
    """Mixin of the transforms that can execute lazily: the ``lazy`` switch (monai/transforms/transform.py:300-325)."""

    _lazy = False

    @property
    def lazy(self):
        return self._lazy

    @lazy.setter
    def lazy(self, lazy):
        if lazy is not None and not isinstance(lazy, bool):
            raise TypeError(f"lazy must be a bool but is of type {type(lazy)}")
        object.__setattr__(self, "_lazy", lazy)

    @property
    def requires_current_data(self):
        return False


class LazyCapableDict(LazyCapable):
    """dictionary transforms: the switch is forwarded to the wrapped array transform(s) named in `_lazy_inner`"""

    _lazy_inner: tuple = ()

    @property
    def lazy(self):
        return self._lazy

    @lazy.setter
    def lazy(self, lazy):
        if lazy is not None and not isinstance(lazy, bool):
            raise TypeError(f"lazy must be a bool but is of type {type(lazy)}")
        object.__setattr__(self, "_lazy", lazy)
        for name in self._lazy_inner:
            inner = self.__dict__.get(name)
            if inner is not None:
                inner.lazy = lazy

    @property
    def requires_current_data(self):
        return any(getattr(self.__dict__.get(n), "requires_current_data", False) for n in self._lazy_inner)


def peek_shape(img):
    """spatial shape the next transform sees: after the pending operations"""
    if is_meta(img) and hasattr(img, "peek_pending_shape"):
        return tuple(int(v) for v in img.peek_pending_shape())
    return tuple(int(v) for v in img.shape[1:])


def peek_affine(img, rank: int | None = None):
    """affine (float64 numpy, on the host) the next transform sees: after the pending operations"""
    if is_meta(img):
        a = img.peek_pending_affine() if hasattr(img, "peek_pending_affine") else img.meta.get("affine")
        if a is not None:
            a = a.detach().cpu() if isinstance(a, torch.Tensor) else a
            return np.asarray(a, dtype=np.float64)
    r = (img.dim() - 1) if rank is None else rank
    return np.eye(min(r, 3) + 1)


def _pending(img) -> list:
    return list(getattr(img, "pending_operations", []) or []) if is_meta(img) else []


def push_pending(img, transform, xform, out_shape, extra_info: dict | None = None, orig_size=None):
    """`img` with one more pending operation: output voxel o shows input voxel ``xform @ (o, 1)``; the data is not touched."""
    if not is_meta(img):
        img = MetaTensor(torch.as_tensor(img))
    info = {
        "class": type(transform).__name__, "id": id(transform), "tracing": True, "do_transforms": True,
        "orig_size": tuple(int(v) for v in (peek_shape(img) if orig_size is None else orig_size)), "lazy": True,
        "extra_info": dict(extra_info or {}),
        LazyAttr.SHAPE: tuple(int(v) for v in out_shape),
        LazyAttr.AFFINE: torch.as_tensor(np.asarray(xform, dtype=np.float64)),
    }
    res = type(img)(img.as_tensor(), meta=dict(img.meta), applied_operations=list(getattr(img, "applied_operations", [])))
    for p in _pending(img):
        res.push_pending_operation(p)
    res.push_pending_operation(info)
    return res


def requires_interp(matrix, atol: float = AFFINE_TOL):
    """monai/transforms/lazy/utils.py:112-142: None when the matrix needs interpolation, else the source axis (1-based, channel
    first) of every output axis -- a signed permutation with an integer shift."""
    matrix = np.asarray(matrix, dtype=np.float64)
    s = matrix[:, -1]
    if not np.allclose(s, np.round(s), atol=atol):
        return None
    ndim = len(matrix) - 1
    ox, oy = [], [0]
    for x, r in enumerate(matrix[:ndim, :ndim]):
        for y, c in enumerate(r):
            if np.isclose(c, -1, atol=atol) or np.isclose(c, 1, atol=atol):
                if x in ox or y + 1 in oy:
                    return None
                ox.append(x)
                oy.append(y + 1)
            elif not np.isclose(c, 0.0, atol=atol):
                return None
    return oy


def _torch_pad_mode(padding_mode) -> str:
    """grid_sample / numpy padding names -> torch.nn.functional.pad modes (monai/transforms/croppad/functional.py:38-75,
    `_convert_pt_pad_mode`); None = constant zeros."""
    p = str(getattr(padding_mode, "value", padding_mode)).lower() if padding_mode is not None else "constant"
    table = {"zeros": "constant", "constant": "constant", "grid-constant": "constant", "border": "replicate", "replicate": "replicate", "edge": "replicate",
             "nearest": "replicate", "reflection": "reflect", "reflect": "reflect", "mirror": "reflect", "grid-mirror": "reflect",
             "wrap": "circular", "grid-wrap": "circular", "circular": "circular"}
    return table.get(p, "replicate")      # "nearest", "border", and others (the reference's last branch)


def resample(data, matrix, kwargs: dict | None = None):
    """Execute the composed voxel-space affine `matrix` on `data` (monai/transforms/lazy/utils.py:148-229).

    Pure axis operations (flip / permute / integer shift; `lazy_resample_mode` "auto", no align_corners): the exact gather kernels
    (`mh_flip_permute_f32`, `mh_crop_pad_f32`), zero padding.  Anything else: ONE `mh_affine_resample_f32` from the image's affine
    to ``affine @ matrix`` at `lazy_shape`, bilinear / border unless `lazy_interpolation_mode` / `lazy_padding_mode` are given,
    float64 coordinates unless `lazy_dtype` says float32.  The result is float32 with the new affine."""
    from .. import ops
    from .croppad.array import _run_crop_pad
    from .spatial.functional import spatial_resample

    kwargs = dict(kwargs or {})
    matrix = np.asarray(matrix.detach().cpu() if isinstance(matrix, torch.Tensor) else matrix, dtype=np.float64)
    if matrix.ndim != 2 or matrix.shape[0] != matrix.shape[1] or matrix.shape[0] not in (3, 4):
        raise NotImplementedError(f"Calling the dense grid resample API directly not implemented, {matrix.shape}.")
    atol = kwargs.get("atol", AFFINE_TOL)
    mode = kwargs.get(LazyAttr.RESAMPLE_MODE, "auto")
    align_corners = bool(kwargs.get(LazyAttr.ALIGN_CORNERS, False) or False)
    ndim = len(matrix) - 1
    img = data if is_meta(data) else MetaTensor(torch.as_tensor(data))
    x = img.as_tensor()
    r = x.dim() - 1
    if 1 <= r < ndim:
        # a 2-D image under a matrix `apply_pending` lifted to 3-D (lazy/functional.py:258-260): when the extra axes are untouched the
        # operation is the r-D one (the reference's axis-only branch slices its permutation / shapes to the image's rank, utils.py:199-203)
        rest = np.delete(np.delete(matrix, list(range(r)), axis=0), list(range(r)), axis=1)
        if np.allclose(rest, np.eye(len(rest)), atol=atol) and np.allclose(matrix[:r, r:ndim], 0.0, atol=atol) and np.allclose(matrix[r:ndim, :r], 0.0, atol=atol):
            sub = np.eye(r + 1)
            sub[:r, :r], sub[:r, -1] = matrix[:r, :r], matrix[:r, -1]
            matrix, ndim = sub, r
    full = np.asarray(img.meta["affine"].detach().cpu() if isinstance(img.meta.get("affine"), torch.Tensor) else img.meta.get("affine", np.eye(4)),
                      dtype=np.float64)
    init_affine = to_affine_nd(ndim, full)
    out_size = kwargs.get(LazyAttr.SHAPE)
    out_size = tuple(int(v) for v in (x.shape[1:ndim + 1] if out_size is None else out_size))
    dst_affine = init_affine @ matrix
    new_full = full @ to_affine_nd(len(full) - 1, matrix)

    def wrap(out):
        res = type(img)(out, meta=dict(img.meta), applied_operations=list(getattr(img, "applied_operations", [])))
        res.meta["affine"] = torch.as_tensor(new_full, dtype=torch.float64)
        return res

    if not all(o > 0 for o in out_size):
        raise ValueError(f"Resampling out_spatial_size should be positive, got {out_size}.")
    axes = requires_interp(matrix, atol=atol)
    if axes is not None and mode == "auto" and not align_corners and x.dim() - 1 == ndim:
        m = np.round(matrix)
        drive = [a - 1 for a in axes[1:]]                      # row x of the matrix = INPUT axis x, driven by output axis drive[x]
        perm = [drive.index(k) for k in range(ndim)]           # output axis k shows input axis perm[k]
        in_shape = [int(v) for v in x.shape[1:]]
        flips_out = [m[perm[k], k] == -1 for k in range(ndim)]
        # input index along axis perm[k]:  sign * o_k + shift ; a flip turns it into (n - 1 - i) = o_k + (n - 1 - shift)
        start = []
        for k in range(ndim):
            n, shift = in_shape[perm[k]], m[perm[k], -1]
            start.append(int(n - 1 - shift) if flips_out[k] else int(shift))
        y = x
        if perm != list(range(ndim)) or any(flips_out):
            pad = 3 - ndim
            flip_in = [False] * ndim
            for k in range(ndim):
                flip_in[perm[k]] = bool(flips_out[k])
            y4 = y.to(torch.float32).reshape((y.shape[0],) + (1,) * pad + tuple(in_shape)).contiguous()
            y = ops.flip_permute(y4, list(range(pad)) + [p + pad for p in perm], [False] * pad + flip_in)
            y = y.reshape((x.shape[0],) + tuple(in_shape[p] for p in perm))
        if any(s != 0 for s in start) or tuple(int(v) for v in y.shape[1:]) != out_size:
            cur = [int(v) for v in y.shape[1:]]
            lo = [max(-s, 0) for s in start]
            hi = [max(s + o - n, 0) for s, o, n in zip(start, out_size, cur)]
            pm = _torch_pad_mode(kwargs.get(LazyAttr.PADDING_MODE))
            if pm == "constant" or not (any(lo) or any(hi)):
                y = _run_crop_pad(y.to(torch.float32), start, out_size, 0.0)
            else:
                # `crop_or_pad_nd(..., mode=padding_mode)` of the reference (lazy/utils.py:222): the crop on the gather kernel, the
                # replicated / reflected / wrapped border by torch's pad on the device
                inner = [o - a - b for o, a, b in zip(out_size, lo, hi)]
                y = _run_crop_pad(y.to(torch.float32), [max(s, 0) for s in start], inner, 0.0)
                pads = [v for a, b in zip(reversed(lo), reversed(hi)) for v in (a, b)]
                y = torch.nn.functional.pad(y[None], pads, mode=pm)[0]
        return wrap(y.to(torch.float32))

    interp = kwargs.get(LazyAttr.INTERP_MODE) or "bilinear"
    padding = kwargs.get(LazyAttr.PADDING_MODE) or "border"
    dt = kwargs.get(LazyAttr.DTYPE, torch.float64)
    dt = torch.float32 if dt in (torch.float32, np.float32, "float32") else torch.float64
    out, _, _, _ = spatial_resample(img, dst_affine, out_size, interp, padding, align_corners, dt)
    return wrap(out)


def apply_pending(data, pending: list | None = None, overrides: dict | None = None):
    """Compose and execute the pending operations of `data` (monai/transforms/lazy/functional.py:196-300) -> (data, applied)."""
    overrides = dict(overrides or {})
    for k in overrides:
        if k not in ("mode", "padding_mode", "dtype", "align_corners", "resample_mode", "device"):
            raise ValueError(f"unsupported override {k!r}")
    if is_meta(data) and pending is None:
        pending = _pending(data)
        data = type(data)(data.as_tensor(), meta=dict(data.meta), applied_operations=list(getattr(data, "applied_operations", [])))
    pending = [] if pending is None else list(pending)
    if not pending:
        return data, []
    cumulative = np.eye(4)
    for p in pending:                                       # A(B(x)) -> (A B)(x): left to right
        nxt = p[LazyAttr.AFFINE] if isinstance(p, dict) else p
        nxt = np.asarray(nxt.detach().cpu() if isinstance(nxt, torch.Tensor) else nxt, dtype=np.float64)
        cumulative = cumulative @ to_affine_nd(3, nxt)
    rank = min((data.dim() - 1), 3)
    if rank < 3:
        cumulative = to_affine_nd(rank, cumulative)
    last = pending[-1] if isinstance(pending[-1], dict) else {}
    kwargs = {}
    if LazyAttr.SHAPE in last:
        kwargs[LazyAttr.SHAPE] = last[LazyAttr.SHAPE]
    for src, dst in (("mode", LazyAttr.INTERP_MODE), ("padding_mode", LazyAttr.PADDING_MODE), ("align_corners", LazyAttr.ALIGN_CORNERS),
                     ("resample_mode", LazyAttr.RESAMPLE_MODE)):
        if src in overrides:
            kwargs[dst] = overrides[src]
    od = overrides.get("dtype", torch.float64)
    kwargs[LazyAttr.DTYPE] = data.dtype if od is None else od
    if overrides.get("device") is not None:
        data = data.to(overrides["device"])
    out = resample(data, cumulative, kwargs)
    for p in pending:
        if isinstance(p, dict):
            out.applied_operations.append(p)
    return out, pending


def materialize(img, overrides: dict | None = None):
    """`img` with its pending operations executed (no-op without any): what an eager transform does first"""
    if _pending(img):
        return apply_pending(img, overrides=overrides)[0]
    return img


def apply_pending_transforms(data, keys=None, overrides: dict | None = None):
    """dictionary / tensor convenience of monai/transforms/lazy/functional.py:61-107"""
    if isinstance(data, dict):
        d = dict(data)
        for k in (keys if keys is not None else list(d.keys())):
            if k in d and _pending(d[k]):
                d[k] = apply_pending(d[k], overrides=(overrides or {}).get(k) if isinstance(overrides, dict) and k in (overrides or {}) else None)[0]
        return d
    return materialize(data, overrides)


class ApplyPending:
    """monai/transforms/lazy/array.py: marks the point where pending operations are executed"""

    def __call__(self, data, overrides: dict | None = None):
        return materialize(data, overrides)


class ApplyPendingd:
    """monai/transforms/lazy/dictionary.py"""

    def __init__(self, keys):
        from ..utils.misc import ensure_tuple

        self.keys = ensure_tuple(keys)

    def __call__(self, data, overrides: dict | None = None):
        return apply_pending_transforms(data, self.keys, overrides)
