"""Stand-in for the reference's native module ``monai._C`` (pybind11, monai/csrc/ext.cpp:20-80) on the MI355X kernels.

Exports the resampling surface of that module -- ``grid_pull`` / ``grid_push`` / ``grid_count`` / ``grid_grad`` and their
``*_backward`` functions (ext.cpp:67-74, interpolation orders 0-7, every index-remapping boundary condition) -- and
the ``BoundType`` / ``InterpolationType`` enums with ``__members__`` lookup including the aliases (used at
monai/networks/layers/spatial_transforms.py:123-127).  The "sliding" boundary condition (unfinished in the reference,
pushpull_cpu.cpp:29-34) raises ``RuntimeError``.
``monai_amd.patch.install()`` can register this module as ``monai._C`` when the real MONAI is installed without its
own compiled extension.
"""

from __future__ import annotations

import enum

import torch

from . import _lib

__all__ = [
    "BoundType", "InterpolationType", "grid_pull", "grid_pull_backward", "grid_push", "grid_push_backward", "grid_count",
    "grid_count_backward", "grid_grad", "grid_grad_backward",
]


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


def _ext3(values):
    """The reference extends bound / interpolation vectors to three entries by repeating the last one
    (PushPullAllocator constructor, monai/csrc/resample/pushpull_cpu.cpp:100-133); the nearest / linear fast paths are
    taken when all THREE orders are equal, whatever the dimensionality (:136)."""
    values = [int(v) for v in values]
    if not values:
        raise RuntimeError("bound/interpolation vector must not be empty")
    return (values + [values[-1]] * 3)[:3]


def _check(name, *tensors):
    first = tensors[0]
    for t in tensors:
        if t.dtype != first.dtype or t.device != first.device:
            raise RuntimeError(f"{name}: tensors must have the same dtype and device")
    if first.dtype not in (torch.float32, torch.float64):
        raise RuntimeError(f"{name}: dtype {first.dtype} is not built (float16, float32 and float64 are)")
    _lib.require_device(*tensors, dtypes=(torch.float32, torch.float64))


def _pad3(shape):
    shape = tuple(int(v) for v in shape)
    return shape + (1,) * (3 - len(shape))


def _pushpull(name, source, source_size, grid, target, bound, interpolation, extrapolate, do_pull=False, do_push=False,
              do_count=False, do_grad=False, do_sgrad=False):
    """The reference's `pushpull(source | source_size, grid[, target], ...)` (pushpull.h:24-50): returns the list of
    outputs in the order of PushPullAllocator::init_output (:390-480): [pull | sgrad | push | count][, grad]."""
    sd = grid.dim() - 2
    if sd < 1 or sd > 3 or grid.shape[-1] != sd:
        raise RuntimeError(f"{name}: grid must be (B, spatial..., D) with D = 1, 2 or 3 spatial dimensions, got {tuple(grid.shape)}")
    tensors = [t for t in (source, grid, target) if t is not None]
    if tensors and all(t.dtype == torch.float16 for t in tensors):
        # half precision (the reference's GPU build dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF, pushpull_cuda.cu:2195,2228, and does every
        # step -- coordinates, weights, sums -- in half): evaluated in fp32 here and rounded to half once at the end: same interface, never less accurate
        outs = _pushpull(name, None if source is None else source.float(), source_size, grid.float(), None if target is None else target.float(), bound,
                         interpolation, extrapolate, do_pull, do_push, do_count, do_grad, do_sgrad)
        return [o.half() for o in outs]
    _check(name, *tensors)
    b = int(grid.shape[0])
    osp = tuple(int(v) for v in grid.shape[1:-1])
    if source is not None:
        if source.dim() != sd + 2 or source.shape[0] != b:
            raise RuntimeError(f"{name}: source must be (B, C, {sd} spatial dims), got {tuple(source.shape)}")
        isp = tuple(int(v) for v in source.shape[2:])
        c = int(source.shape[1])
    else:
        isp = tuple(int(v) for v in source_size)
        if len(isp) != sd:
            raise RuntimeError(f"{name}: source size must have {sd} entries, got {isp}")
        c = int(target.shape[1]) if target is not None else 1
    if min(isp + osp) < 1:
        raise RuntimeError(f"{name}: empty spatial dimensions")
    tk = 0
    if target is not None:
        if target.dim() == sd + 3:
            tk = int(target.shape[-1])
            if tk != sd:
                raise RuntimeError(f"{name}: target gradient axis must have {sd} components")
        elif target.dim() != sd + 2:
            raise RuntimeError(f"{name}: target must be (B, C, spatial[, D])")
        if tuple(target.shape[2 : 2 + sd]) != osp or target.shape[0] != b:
            raise RuntimeError(f"{name}: target and grid shapes differ")
        if do_push or source is None:
            c = int(target.shape[1])
        elif int(target.shape[1]) != c:
            raise RuntimeError(f"{name}: source and target channel counts differ")
    dt, dev = grid.dtype, grid.device
    out = grad = None
    if do_pull:
        out = torch.empty((b, c) + osp, dtype=dt, device=dev)
    elif do_sgrad:
        out = torch.empty((b, c) + osp + (sd,), dtype=dt, device=dev)
    elif do_push:
        out = torch.empty((b, c) + isp, dtype=dt, device=dev)
    elif do_count:
        out = torch.empty((b, 1) + isp, dtype=dt, device=dev)
    if do_grad:
        grad = torch.empty((b,) + osp + (sd,), dtype=dt, device=dev)
    src = source.contiguous() if source is not None else None
    g = grid.contiguous()
    tg = target.contiguous() if target is not None else None
    x3, o3 = _pad3(isp), _pad3(osp)
    _lib.lib().call(
        "mh_pushpull", _lib.ptr(src), _lib.ptr(g), _lib.ptr(tg), _lib.ptr(out), _lib.ptr(grad), int(dt == torch.float64), sd, b, c,
        *x3, *o3, _lib.int_array(_ext3(bound)), _lib.int_array(_ext3(interpolation)), int(bool(extrapolate)), int(do_pull), int(do_push),
        int(do_count), int(do_grad), int(do_sgrad), tk, _lib.stream_ptr(grid),
    )
    return [t for t in (out, grad) if t is not None]


def grid_pull(input: torch.Tensor, grid: torch.Tensor, bound, interpolation, extrapolate: bool) -> torch.Tensor:
    """``monai._C.grid_pull`` (pushpull.h:58-110): input (B, C, X[, Y[, Z]]), grid (B, Xo[, Yo[, Zo]], D) voxel
    coordinates -> (B, C, Xo...)."""
    return _pushpull("grid_pull", input, None, grid, None, bound, interpolation, extrapolate, do_pull=True)[0]


def grid_pull_backward(grad, input, grid, bound, interpolation, extrapolate):
    """``monai._C.grid_pull_backward`` (pushpull.h:112-150): [d/d input if input.requires_grad][, d/d grid if grid.requires_grad]."""
    return _pushpull("grid_pull_backward", input, None, grid, grad, bound, interpolation, extrapolate,
                     do_push=input.requires_grad, do_grad=grid.requires_grad)


def grid_push(input, grid, source_size, bound, interpolation, extrapolate):
    """``monai._C.grid_push`` (pushpull.h:153-249): splat input (B, C, spatial of grid) into a (B, C, *source_size) volume."""
    size = tuple(source_size) if source_size is not None and len(source_size) else tuple(input.shape[2:])
    return _pushpull("grid_push", None, size, grid, input, bound, interpolation, extrapolate, do_push=True)[0]


def grid_push_backward(grad, input, grid, bound, interpolation, extrapolate):
    """``monai._C.grid_push_backward`` (pushpull.h:251-289)."""
    return _pushpull("grid_push_backward", grad, None, grid, input, bound, interpolation, extrapolate,
                     do_pull=input.requires_grad, do_grad=grid.requires_grad)


def grid_count(grid, source_size, bound, interpolation, extrapolate):
    """``monai._C.grid_count`` (pushpull.h:292-373): splat an image of ones -> (B, 1, *source_size)."""
    size = tuple(source_size) if source_size is not None and len(source_size) else tuple(grid.shape[1:-1])
    return _pushpull("grid_count", None, size, grid, None, bound, interpolation, extrapolate, do_count=True)[0]


def grid_count_backward(grad, grid, bound, interpolation, extrapolate):
    """``monai._C.grid_count_backward`` (pushpull.h:375-413): gradient with respect to the grid."""
    res = _pushpull("grid_count_backward", grad, None, grid, None, bound, interpolation, extrapolate, do_grad=grid.requires_grad)
    if not res:
        raise RuntimeError("grid_count_backward: grid does not require a gradient")
    return res[0]


def grid_grad(input, grid, bound, interpolation, extrapolate):
    """``monai._C.grid_grad`` (pushpull.h:415-467): spatial gradients of the sampled image -> (B, C, spatial of grid, D)."""
    return _pushpull("grid_grad", input, None, grid, None, bound, interpolation, extrapolate, do_sgrad=True)[0]


def grid_grad_backward(grad, input, grid, bound, interpolation, extrapolate):
    """``monai._C.grid_grad_backward`` (pushpull.h:469-507); grad is (B, C, spatial of grid, D)."""
    return _pushpull("grid_grad_backward", input, None, grid, grad, bound, interpolation, extrapolate,
                     do_push=input.requires_grad, do_grad=grid.requires_grad)
