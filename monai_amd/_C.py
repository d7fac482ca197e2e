"""Stand-in for the reference's native module ``monai._C`` (pybind11, monai/csrc/ext.cpp:20-80) on the MI355X kernels.

Exports the same names the reference's Python reaches for -- ``grid_pull`` and the ``BoundType`` /
``InterpolationType`` enums with ``__members__`` lookup including the aliases (used at
monai/networks/layers/spatial_transforms.py:123-127).  Entry points this build does not provide raise
``RuntimeError`` the way ``AT_ERROR("Not compiled with GPU support.")`` does (monai/csrc/resample/pushpull.h:95).
``monai_amd.patch.install()`` can register this module as ``monai._C`` when the real MONAI is installed without its
own compiled extension.
"""

from __future__ import annotations

import enum

import torch

from . import _lib

__all__ = ["BoundType", "InterpolationType", "grid_pull"]


class BoundType(enum.IntEnum):
    replicate = 0   # a a a | a b c d | d d d
    nearest = 0
    border = 0
    dct1 = 1        # d c b | a b c d | c b a
    mirror = 1
    dct2 = 2        # c b a | a b c d | d c b
    reflect = 2
    dst1 = 3        # -b -a 0 | a b c d | 0 -d -c
    antimirror = 3
    dst2 = 4        # -c -b -a | a b c d | -d -c -b
    antireflect = 4
    dft = 5         # b c d | a b c d | a b c
    wrap = 5
    zero = 7        # 0 0 0 | a b c d | 0 0 0
    zeros = 7


class InterpolationType(enum.IntEnum):
    nearest = 0
    linear = 1
    quadratic = 2
    cubic = 3
    fourth = 4
    fifth = 5
    sixth = 6
    seventh = 7


def _rep3(values, n):
    values = [int(v) for v in values]
    if not values:
        raise RuntimeError("bound/interpolation vector must not be empty")
    values = values + [values[-1]] * (n - len(values))
    return values[:n]


def grid_pull(input: torch.Tensor, grid: torch.Tensor, bound, interpolation, extrapolate: bool) -> torch.Tensor:
    """``monai._C.grid_pull``: input (B, C, X[, Y[, Z]]), grid (B, Xo[, Yo[, Zo]], D) voxel coordinates -> (B, C, Xo...)."""
    if input.dim() < 3 or input.dim() > 5:
        raise RuntimeError("grid_pull: input must be (B, C, spatial) with 1, 2 or 3 spatial dimensions")
    sd = input.dim() - 2
    if grid.dim() != sd + 2 or grid.shape[-1] != sd:
        raise RuntimeError(f"grid_pull: grid must be (B, spatial..., {sd}), got {tuple(grid.shape)}")
    if input.dtype != grid.dtype or input.device != grid.device:
        raise RuntimeError("grid_pull: input and grid must have the same dtype and device")
    if input.dtype not in (torch.float32, torch.float64):
        raise RuntimeError(f"grid_pull: dtype {input.dtype} is not built (float32 and float64 are)")
    _lib.require_device(input, grid, dtypes=(torch.float32, torch.float64))
    b, c = input.shape[:2]
    pad = 3 - sd
    src = input.contiguous().reshape((b, c) + tuple(input.shape[2:]) + (1,) * pad)
    g = grid.contiguous()
    if pad:
        z = torch.zeros(g.shape[:-1] + (pad,), dtype=g.dtype, device=g.device)
        g = torch.cat([g, z], dim=-1).reshape((b,) + tuple(grid.shape[1:-1]) + (1,) * pad + (3,)).contiguous()
    osp = tuple(g.shape[1:4])
    out = torch.empty((b, c) + osp, dtype=input.dtype, device=input.device)
    bd = _rep3(bound, sd) + [int(BoundType.replicate)] * pad
    it = _rep3(interpolation, sd)
    it = it + [it[0]] * pad  # padded size-1 axes sample at coordinate 0 exactly under either order
    _lib.lib().call(
        "mh_grid_pull", _lib.ptr(src), _lib.ptr(g), _lib.ptr(out), int(input.dtype == torch.float64), b, c, *[int(v) for v in src.shape[2:]],
        *[int(v) for v in osp], _lib.int_array(bd), _lib.int_array(it), int(bool(extrapolate)), _lib.stream_ptr(input),
    )
    return out.reshape((b, c) + tuple(grid.shape[1:-1]))


def _not_built(name):
    def f(*_a, **_k):
        raise RuntimeError(f"monai_amd._C.{name}: not built for the MI355X path (grid_pull orders 0/1 are)")

    f.__name__ = name
    return f


grid_pull_backward = _not_built("grid_pull_backward")
grid_push = _not_built("grid_push")
grid_push_backward = _not_built("grid_push_backward")
grid_count = _not_built("grid_count")
grid_count_backward = _not_built("grid_count_backward")
grid_grad = _not_built("grid_grad")
grid_grad_backward = _not_built("grid_grad_backward")
