"""``gaussian_1d`` / ``separable_filtering`` / ``GaussianFilter`` on the fused MI355X smoothing kernel -- drop-ins for
monai/networks/layers/convutils.py:78-131 and monai/networks/layers/simplelayers.py:207-249, :542-595."""

from __future__ import annotations

from collections.abc import Sequence

import numpy as np
import torch
import torch.nn as nn

from ... import ops

__all__ = ["gaussian_1d", "separable_filtering", "GaussianFilter"]


def gaussian_1d(sigma, truncated: float = 4.0, approx: str = "erf", normalize: bool = False) -> torch.Tensor:
    """1-D discrete Gaussian with ``tail = int(max(sigma * truncated, 0.5) + 0.5)`` taps each side, evaluated on the HOST
    in fp32 with the same torch operators as the reference ("erf": 0.5*(erf(t(x+.5)) - erf(t(x-.5))), t = 0.70710678/|sigma|,
    clamped at 0; "sampled": exp(-x^2 / 2 sigma^2) / (2.5066282 sigma)) so the taps are bit-identical to the reference's CPU
    kernel.  "scalespace" (discrete Gaussian, I_k(sigma^2) e^{-sigma^2}) uses scipy's exponentially scaled Bessel function in
    fp64 instead of the reference's fp32 polynomial recurrences: equal to ~1e-7."""
    sigma = torch.as_tensor(sigma, dtype=torch.float).detach().cpu()
    if truncated <= 0.0:
        raise ValueError(f"truncated must be positive, got {truncated}.")
    tail = int(max(float(sigma) * truncated, 0.5) + 0.5)
    kind = approx.lower()
    if kind == "erf":
        x = torch.arange(-tail, tail + 1, dtype=torch.float)
        t = 0.70710678 / torch.abs(sigma)
        out = 0.5 * ((t * (x + 0.5)).erf() - (t * (x - 0.5)).erf())
        out = out.clamp(min=0)
    elif kind == "sampled":
        x = torch.arange(-tail, tail + 1, dtype=torch.float)
        out = torch.exp(-0.5 / (sigma * sigma) * x**2)
        if not normalize:
            out = out / (2.5066282 * sigma)
    elif kind == "scalespace":
        from scipy.special import ive

        s2 = float(sigma) * float(sigma)
        pos = [float(ive(k, s2)) for k in range(tail + 1)]
        out = torch.tensor(pos[:0:-1] + pos, dtype=torch.float)
    else:
        raise NotImplementedError(f"Unsupported option: approx='{approx}'.")
    return out / out.sum() if normalize else out


def separable_filtering(x: torch.Tensor, kernels, mode: str = "zeros") -> torch.Tensor:
    """1-D convolution along every spatial axis of ``x`` (batch, channels, spatial...), zero padding, one fused pass."""
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"x must be a torch.Tensor but is {type(x).__name__}.")
    if mode != "zeros":
        raise NotImplementedError("monai_amd.separable_filtering: only mode='zeros' is on the HIP path")
    sd = x.dim() - 2
    if sd not in (2, 3):
        raise NotImplementedError("monai_amd.separable_filtering: 2-D and 3-D inputs only")
    if isinstance(kernels, torch.Tensor):
        kernels = [kernels] * sd
    ks = [np.asarray(torch.as_tensor(k).detach().cpu(), dtype=np.float32).reshape(-1) for k in kernels]
    if len(ks) != sd:
        raise ValueError(f"expected {sd} kernels, got {len(ks)}")
    for k in ks:
        if k.size % 2 == 0:
            raise NotImplementedError("monai_amd.separable_filtering: even-length kernels are not on the HIP path")
    ks = [np.ones(1, dtype=np.float32)] * (3 - sd) + ks
    data = x.as_tensor() if hasattr(x, "as_tensor") else x
    src = data.to(torch.float32).contiguous()
    b, c = src.shape[:2]
    sp = (1,) * (3 - sd) + tuple(src.shape[2:])
    out = ops.separable_filter3d(src.reshape((b * c,) + sp), ks)
    return out.reshape(src.shape)


class GaussianFilter(nn.Module):
    def __init__(self, spatial_dims: int, sigma, truncated: float = 4.0, approx: str = "erf", requires_grad: bool = False) -> None:
        if isinstance(sigma, (Sequence, np.ndarray)) and not isinstance(sigma, str):
            if len(sigma) != spatial_dims:
                raise ValueError
            sig = list(sigma)
        elif isinstance(sigma, torch.Tensor) and sigma.ndim > 0:
            if len(sigma) != spatial_dims:
                raise ValueError
            sig = list(sigma)
        else:
            sig = [sigma for _ in range(spatial_dims)]
        super().__init__()
        if requires_grad:
            raise NotImplementedError("monai_amd.GaussianFilter: trainable sigma is not on the (inference) HIP path")
        self.sigma = [torch.nn.Parameter(torch.as_tensor(s, dtype=torch.float).detach().cpu(), requires_grad=False) for s in sig]
        self.truncated, self.approx = truncated, approx
        for idx, param in enumerate(self.sigma):
            self.register_parameter(f"kernel_sigma_{idx}", param)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        kernels = [gaussian_1d(s, truncated=self.truncated, approx=self.approx) for s in self.sigma]
        return separable_filtering(x=x, kernels=kernels)
