"""``AffineTransform`` on the MI355X resampling kernel -- drop-in for
monai/networks/layers/spatial_transforms.py:439-592 (``F.affine_grid`` + ``F.grid_sample``)."""

from __future__ import annotations

from collections.abc import Sequence

import torch
import torch.nn as nn

from ... import _lib, ops
from ...utils.misc import ensure_tuple, look_up_option
from ..utils import index_matrix

__all__ = ["AffineTransform"]


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
        f64 = plain.dtype == torch.float64
        x = plain.to(torch.float32).contiguous()
        _lib.require_device(x)
        th = theta.detach().double().cpu().numpy()
        pad = 3 - sr
        outs = []
        for b in range(n):
            m = index_matrix(th[b], src_size[2:], dst_sp, self.normalized, self.align_corners, self.reverse_indexing, self.zero_centered)
            vol = x[b].reshape((x.shape[1],) + (1,) * pad + tuple(src_size[2:]))
            o = ops.affine_resample(vol, m.reshape(-1), (1,) * pad + tuple(int(v) for v in dst_sp), self.mode, self.padding_mode,
                                    self.align_corners, f64)
            outs.append(o.reshape((x.shape[1],) + tuple(int(v) for v in dst_sp)))
        return torch.stack(outs).to(plain.dtype)
