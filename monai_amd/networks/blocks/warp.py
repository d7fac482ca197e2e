"""``Warp`` / ``DVF2DDF`` -- drop-ins for monai/networks/blocks/warp.py:31-184 on the MI355X resampling kernels.

With ``monai_amd.config.USE_COMPILED`` the image is sampled by the native ``grid_pull`` exactly as in the reference
(:157: voxel-coordinate grid, extrapolate=True) and is differentiable with respect to the image and the displacement
field.  Without it the reference normalises the grid and calls ``F.grid_sample(align_corners=True)`` (:147-155); here the
dense-grid kernel samples at the same voxel coordinates directly (no normalise / unnormalise round trip); that branch is
forward-only -- tensors that require a gradient need ``USE_COMPILED``.
"""

from __future__ import annotations

import warnings

import torch
import torch.nn as nn

from ... import config, ops
from ..layers import grid_pull

__all__ = ["Warp", "DVF2DDF"]

_MODES = ("bilinear", "nearest", "bicubic")
_PADS = ("zeros", "border", "reflection")


class Warp(nn.Module):
    def __init__(self, mode="bilinear", padding_mode="border", jitter: bool = False):
        super().__init__()
        mode = getattr(mode, "value", mode)
        padding_mode = getattr(padding_mode, "value", padding_mode)
        self.compiled = bool(config.USE_COMPILED)
        if self.compiled:     # names -> the native module's integers (:67-97); ints pass through
            if mode in _MODES:
                mode = {"bilinear": 1, "nearest": 0, "bicubic": 3}[mode]
            if padding_mode in _PADS:
                padding_mode = {"zeros": 7, "border": 0, "reflection": 1}[padding_mode]
        else:
            warnings.warn("monai_amd.networks.blocks.Warp: USE_COMPILED is off, using the dense-grid sampling kernel (forward only).")
            if mode not in _MODES:
                raise ValueError(f"'{mode}' is not a valid GridSampleMode")
            if padding_mode not in _PADS:
                raise ValueError(f"'{padding_mode}' is not a valid GridSamplePadMode")
        self._interp_mode, self._padding_mode = mode, padding_mode
        self.ref_grid = None
        self.jitter = jitter

    def get_reference_grid(self, ddf: torch.Tensor, jitter: bool = False, seed: int = 0) -> torch.Tensor:
        if self.ref_grid is not None and self.ref_grid.shape[0] == ddf.shape[0] and self.ref_grid.shape[1:] == ddf.shape[2:]:
            return self.ref_grid
        mesh = torch.meshgrid(*[torch.arange(0, dim) for dim in ddf.shape[2:]], indexing="ij")
        grid = torch.stack([torch.stack(mesh, dim=0)] * ddf.shape[0], dim=0)      # (batch, spatial_dims, ...)
        self.ref_grid = grid.to(ddf)
        if jitter:      # reference grid on non-integer positions (Likar & Pernus 2001)
            with torch.random.fork_rng(enabled=seed):
                torch.random.manual_seed(seed)
                grid += torch.rand_like(grid)
        self.ref_grid.requires_grad = False
        return self.ref_grid

    def forward(self, image: torch.Tensor, ddf: torch.Tensor):
        """image (batch, channels, H, W[, D]); ddf (batch, spatial_dims, H, W[, D]) -> the warped image, same shape."""
        sd = image.dim() - 2
        if sd not in (2, 3):
            raise NotImplementedError(f"got unsupported spatial_dims={sd}, currently support 2 or 3.")
        ddf_shape = (image.shape[0], sd) + tuple(image.shape[2:])
        if tuple(ddf.shape) != ddf_shape:
            raise ValueError(f"Given input {sd}-d image shape {image.shape}, the input DDF shape must be {ddf_shape}, Got {ddf.shape} instead.")
        grid = self.get_reference_grid(ddf, jitter=self.jitter) + ddf
        if self.compiled:
            grid = grid.permute([0] + list(range(2, 2 + sd)) + [1])               # (batch, ..., spatial_dims)
            return grid_pull(image, grid, bound=self._padding_mode, extrapolate=True, interpolation=self._interp_mode)
        if torch.is_grad_enabled() and (image.requires_grad or ddf.requires_grad):
            # falls through to the reference's F.grid_sample branch when MONAI is importable (boundary B3)
            raise NotImplementedError("monai_amd.networks.blocks.Warp: the non-compiled branch is forward-only; set monai_amd.config.USE_COMPILED "
                                      "(BUILD_MONAI=1) for a differentiable warp on the HIP kernels")
        if self._interp_mode == "bicubic":
            raise NotImplementedError("monai_amd.networks.blocks.Warp: bicubic needs USE_COMPILED (cubic B-spline grid_pull)")
        pad = 3 - sd
        out = torch.empty(image.shape, dtype=torch.float32, device=image.device)
        for b in range(image.shape[0]):
            coords = grid[b].to(torch.float64 if grid.dtype == torch.float64 else torch.float32)
            if pad:
                coords = torch.cat([torch.zeros((1,) + tuple(coords.shape[1:]), dtype=coords.dtype, device=coords.device), coords])
                coords = coords.reshape((3, 1) + tuple(image.shape[2:]))
            src = image[b].to(torch.float32).contiguous().reshape((image.shape[1],) + (1,) * pad + tuple(image.shape[2:]))
            res = ops.grid_resample(src, coords.contiguous(), self._interp_mode, self._padding_mode, True, grid.dtype == torch.float64)
            out[b] = res.reshape(image.shape[1:])
        return out.to(image.dtype)


class DVF2DDF(nn.Module):
    """Dense displacement field from a dense velocity field by scaling and squaring (warp.py:160-184)."""

    def __init__(self, num_steps: int = 7, mode="bilinear", padding_mode="zeros"):
        super().__init__()
        if num_steps <= 0:
            raise ValueError(f"expecting positive num_steps, got {num_steps}")
        self.num_steps = num_steps
        self.warp_layer = Warp(mode=mode, padding_mode=padding_mode)

    def forward(self, dvf: torch.Tensor) -> torch.Tensor:
        ddf: torch.Tensor = dvf / (2 ** self.num_steps)
        for _ in range(self.num_steps):
            ddf = ddf + self.warp_layer(image=ddf, ddf=ddf)
        return ddf
