"""``AffineTransform`` on the MI355X resampling kernel -- drop-in for
monai/networks/layers/spatial_transforms.py:439-592 (``F.affine_grid`` + ``F.grid_sample``) -- and the differentiable
``grid_pull`` / ``grid_push`` / ``grid_count`` / ``grid_grad`` functions of the same file (:35-436) on ``monai_amd._C``."""

from __future__ import annotations

from collections.abc import Sequence

import torch
import torch.nn as nn

from ... import _C, _lib, ops
from ...utils.misc import ensure_tuple, look_up_option
from ..utils import index_matrix

__all__ = ["AffineTransform", "grid_pull", "grid_push", "grid_count", "grid_grad"]


class AffineTransform(nn.Module):
    def __init__(
        self,
        spatial_size: Sequence[int] | int | None = None,
        normalized: bool = False,
        mode: str = "bilinear",
        padding_mode: str = "zeros",
        align_corners: bool = True,
        reverse_indexing: bool = True,
        zero_centered: bool | None = None,
    ) -> None:
        super().__init__()
        self.spatial_size = ensure_tuple(spatial_size) if spatial_size is not None else None
        self.normalized = normalized
        self.mode = look_up_option(mode, ("bilinear", "nearest"), "mode")
        self.padding_mode = look_up_option(padding_mode, ("zeros", "border", "reflection"), "padding_mode")
        self.align_corners = align_corners
        self.reverse_indexing = reverse_indexing
        if zero_centered is not None and self.normalized:
            raise ValueError("`normalized=True` is not compatible with the `zero_centered` option.")
        self.zero_centered = zero_centered if zero_centered is not None else False

    def forward(self, src: torch.Tensor, theta: torch.Tensor, spatial_size: Sequence[int] | int | None = None) -> torch.Tensor:
        """``src`` (N, C, spatial 2-D or 3-D); ``theta`` d x d, N x d x d, (d-1) x d or N x (d-1) x d.  The interpolation
        runs in fp64 when ``src`` is float64 and in fp32 otherwise; the result has ``src``'s dtype."""
        if not isinstance(theta, torch.Tensor):
            raise TypeError(f"theta must be torch.Tensor but is {type(theta).__name__}.")
        if theta.dim() not in (2, 3):
            raise ValueError(f"theta must be Nxdxd or dxd, got {theta.shape}.")
        if theta.dim() == 2:
            theta = theta[None]
        theta_shape = tuple(theta.shape[1:])
        if theta_shape not in ((2, 3), (3, 4), (3, 3), (4, 4)):
            raise ValueError(f"theta must be Nx3x3 or Nx4x4, got {theta.shape}.")
        if not torch.is_floating_point(theta):
            raise ValueError(f"theta must be floating point data, got {theta.dtype}")
        if not isinstance(src, torch.Tensor):
            raise TypeError(f"src must be torch.Tensor but is {type(src).__name__}.")
        sr = src.dim() - 2
        if sr not in (2, 3):
            raise ValueError(f"Unsupported src dimension: {sr}, available options are [2, 3].")
        src_size = tuple(src.shape)
        dst_sp = src_size[2:]
        if self.spatial_size is not None:
            dst_sp = tuple(self.spatial_size)
        if spatial_size is not None:
            dst_sp = tuple(ensure_tuple(spatial_size))
        n = src_size[0]
        if theta.shape[0] == 1 and n > 1:
            theta = theta.repeat(n, 1, 1)
        if theta.shape[0] != n:
            raise ValueError(f"affine and image batch dimension must match, got affine={theta.shape[0]} image={n}.")

        plain = src.as_tensor() if hasattr(src, "as_tensor") else src
        th = theta.detach().double().cpu().numpy()
        ms = [index_matrix(th[b], src_size[2:], dst_sp, self.normalized, self.align_corners, self.reverse_indexing, self.zero_centered) for b in range(n)]
        if plain.dtype != theta.dtype:
            # after the shape / matrix checks (ValueError), as in the reference: it builds the sampling grid in theta's dtype and
            # F.grid_sample refuses an input of another dtype (spatial_transforms.py:584-591; tests/networks/layers/test_affine_transform.py:313-333)
            raise RuntimeError(f"grid_sampler(): expected input and grid to have same dtype, but input has {plain.dtype} and grid has {theta.dtype}")
        f64 = plain.dtype == torch.float64
        x = plain.to(torch.float32).contiguous()
        _lib.require_device(x)
        pad = 3 - sr
        outs = []
        for b in range(n):
            m = ms[b]
            vol = x[b].reshape((x.shape[1],) + (1,) * pad + tuple(src_size[2:]))
            o = ops.affine_resample(vol, m.reshape(-1), (1,) * pad + tuple(int(v) for v in dst_sp), self.mode, self.padding_mode,
                                    self.align_corners, f64)
            outs.append(o.reshape((x.shape[1],) + tuple(int(v) for v in dst_sp)))
        return torch.stack(outs).to(plain.dtype)


# ---------------------------------------------------------------------------------------------------------------------
# grid_pull / grid_push / grid_count / grid_grad (monai/networks/layers/spatial_transforms.py:35-436): spline sampling
# with respect to a deformation field in VOXEL coordinates, its adjoint (splatting), the splatted image of ones and the
# spatial gradients of the sampled image, each differentiable with respect to the image and the field.


def _modes(bound, interpolation):
    """Names / ints / enum members -> lists of enum members, the reference's conversion (:123-127)."""
    b = [_C.BoundType.__members__[v] if isinstance(v, str) else _C.BoundType(v) for v in ensure_tuple(bound)]
    i = [_C.InterpolationType.__members__[v] if isinstance(v, str) else _C.InterpolationType(v) for v in ensure_tuple(interpolation)]
    return b, i


def _like_input(out: torch.Tensor, input) -> torch.Tensor:
    from ...data.meta_tensor import MetaTensor

    if isinstance(input, MetaTensor):
        return MetaTensor(out).copy_meta_from(input)
    return out


def _two_grads(ctx, grads, n_extra):
    """Order of the reference's backward passes: [d input][, d grid], each present when the tensor required it."""
    extra = (None,) * n_extra
    if ctx.needs_input_grad[0]:
        return (grads[0], grads[1] if ctx.needs_input_grad[1] else None) + extra
    return (None, grads[0]) + extra


class _GridPull(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, grid, interpolation, bound, extrapolate):
        opt = (bound, interpolation, extrapolate)
        if input.requires_grad or grid.requires_grad:
            ctx.opt = opt
            ctx.save_for_backward(input, grid)
        return _C.grid_pull(input, grid, *opt)

    @staticmethod
    def backward(ctx, grad):
        if not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            return None, None, None, None, None
        return _two_grads(ctx, _C.grid_pull_backward(grad, *ctx.saved_tensors, *ctx.opt), 3)


class _GridPush(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, grid, shape, interpolation, bound, extrapolate):
        opt = (bound, interpolation, extrapolate)
        if input.requires_grad or grid.requires_grad:
            ctx.opt = opt
            ctx.save_for_backward(input, grid)
        return _C.grid_push(input, grid, shape, *opt)

    @staticmethod
    def backward(ctx, grad):
        if not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            return None, None, None, None, None, None
        return _two_grads(ctx, _C.grid_push_backward(grad, *ctx.saved_tensors, *ctx.opt), 4)


class _GridCount(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grid, shape, interpolation, bound, extrapolate):
        opt = (bound, interpolation, extrapolate)
        if grid.requires_grad:
            ctx.opt = opt
            ctx.save_for_backward(grid)
        return _C.grid_count(grid, shape, *opt)

    @staticmethod
    def backward(ctx, grad):
        if ctx.needs_input_grad[0]:
            return _C.grid_count_backward(grad, *ctx.saved_tensors, *ctx.opt), None, None, None, None
        return None, None, None, None, None


class _GridGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, grid, interpolation, bound, extrapolate):
        opt = (bound, interpolation, extrapolate)
        if input.requires_grad or grid.requires_grad:
            ctx.opt = opt
            ctx.save_for_backward(input, grid)
        return _C.grid_grad(input, grid, *opt)

    @staticmethod
    def backward(ctx, grad):
        if not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            return None, None, None, None, None
        return _two_grads(ctx, _C.grid_grad_backward(grad, *ctx.saved_tensors, *ctx.opt), 3)


def grid_pull(input: torch.Tensor, grid: torch.Tensor, interpolation="linear", bound="zero", extrapolate: bool = True) -> torch.Tensor:
    """Sample ``input`` (B, C, Wi, Hi, Di) at the voxel coordinates ``grid`` (B, Wo, Ho, Do, 1|2|3) -> (B, C, Wo, Ho, Do).

    ``interpolation``: 0-7 or "nearest" / "linear" / "quadratic" / "cubic" / "fourth" ... "seventh" (B-spline order),
    ``bound``: "replicate"|"nearest"|"border" (0), "dct1"|"mirror" (1), "dct2"|"reflect" (2), "dst1"|"antimirror" (3),
    "dst2"|"antireflect" (4), "dft"|"wrap" (5), "zero"|"zeros" (7); either may be a list in the order [W, H, D].
    ``extrapolate=False`` zeroes samples whose coordinate lies outside the field of view.  Reference:
    monai/networks/layers/spatial_transforms.py:60-132."""
    b, i = _modes(bound, interpolation)
    return _like_input(_GridPull.apply(input, grid, i, b, extrapolate), input)


def grid_push(input: torch.Tensor, grid: torch.Tensor, shape=None, interpolation="linear", bound="zero", extrapolate: bool = True):
    """Splat ``input`` (B, C, Wi, Hi, Di) along ``grid`` (B, Wi, Hi, Di, 1|2|3) into a volume of spatial ``shape``
    (default: the input's) -- the adjoint of ``grid_pull``.  Reference: spatial_transforms.py:160-237."""
    b, i = _modes(bound, interpolation)
    if shape is None:
        shape = tuple(input.shape[2:])
    return _like_input(_GridPush.apply(input, grid, shape, i, b, extrapolate), input)


def grid_count(grid: torch.Tensor, shape=None, interpolation="linear", bound="zero", extrapolate: bool = True):
    """Splat an image of ones along ``grid`` -> (B, 1, *shape).  Reference: spatial_transforms.py:261-337; like there the
    default ``shape`` is ``grid.shape[2:]`` (the trailing grid axes INCLUDING the coordinate axis), so pass ``shape``
    explicitly for anything but a quick look."""
    b, i = _modes(bound, interpolation)
    if shape is None:
        shape = tuple(grid.shape[2:])
    return _GridCount.apply(grid, shape, i, b, extrapolate)


def grid_grad(input: torch.Tensor, grid: torch.Tensor, interpolation="linear", bound="zero", extrapolate: bool = True):
    """Spatial gradients of ``input`` sampled at ``grid`` -> (B, C, Wo, Ho, Do, 1|2|3).  Reference:
    spatial_transforms.py:365-436."""
    b, i = _modes(bound, interpolation)
    return _like_input(_GridGrad.apply(input, grid, i, b, extrapolate), input)
