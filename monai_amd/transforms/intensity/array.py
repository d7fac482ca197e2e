"""``GaussianSmooth`` -- monai/transforms/intensity/array.py:1590-1622 on the fused smoothing kernel."""

from __future__ import annotations

from collections.abc import Sequence

import torch

from ...data.meta_tensor import is_meta
from ...networks.layers.simplelayers import GaussianFilter

__all__ = ["GaussianSmooth"]


class GaussianSmooth:
    """Gaussian smoothing of a channel-first image; ``sigma`` is a scalar or one value per spatial axis."""

    def __init__(self, sigma: Sequence[float] | float = 1.0, approx: str = "erf") -> None:
        self.sigma, self.approx = sigma, approx

    def __call__(self, img):
        data = img.as_tensor() if is_meta(img) else torch.as_tensor(img)
        x = data.to(torch.float)
        sigma = list(self.sigma) if isinstance(self.sigma, Sequence) else self.sigma
        out = GaussianFilter(x.ndim - 1, sigma, approx=self.approx)(x.unsqueeze(0)).squeeze(0)
        if is_meta(img):
            return type(img)(out, meta=dict(img.meta), applied_operations=list(getattr(img, "applied_operations", [])))
        return out
