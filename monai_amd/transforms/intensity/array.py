"""``GaussianSmooth`` (monai/transforms/intensity/array.py:1590-1622, the fused smoothing kernel) and ``ScaleIntensityRange``
(array.py:958-1012, one element-wise pass)."""

from __future__ import annotations

from collections.abc import Sequence

import torch

from ...data.meta_tensor import is_meta
from ...networks.layers.simplelayers import GaussianFilter

__all__ = ["GaussianSmooth", "ScaleIntensityRange", "NormalizeIntensity", "ScaleIntensity"]


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


_NP2T = {"float32": torch.float32, "float64": torch.float64, "float16": torch.float16, "int16": torch.int16, "int32": torch.int32,
         "int64": torch.int64, "uint8": torch.uint8, "int8": torch.int8, "bool": torch.bool}


def _to_torch_dtype(dtype):
    if dtype is None or isinstance(dtype, torch.dtype):
        return dtype
    import numpy as np

    return _NP2T[np.dtype(dtype).name]


class ScaleIntensityRange:
    """``monai.transforms.ScaleIntensityRange`` (monai/transforms/intensity/array.py:958-1012) as one HIP pass over the device
    tensor: ``(img - a_min) / (a_max - a_min)``, optionally ``* (b_max - b_min) + b_min``, optionally clamped -- the reference's
    operator sequence with each of its fp32 roundings kept, so the results are bit-identical to the reference's.  Integer images
    are promoted to float32 first (what the reference's first subtraction does); float64 / half images are not on this path."""

    def __init__(self, a_min: float, a_max: float, b_min: float | None = None, b_max: float | None = None, clip: bool = False,
                 dtype=torch.float32) -> None:
        self.a_min, self.a_max, self.b_min, self.b_max, self.clip, self.dtype = a_min, a_max, b_min, b_max, clip, dtype

    def __call__(self, img):
        from ... import ops
        import warnings

        data = img.as_tensor() if is_meta(img) else torch.as_tensor(img)
        if data.dtype in (torch.float64, torch.float16, torch.bfloat16) or data.is_complex():
            raise NotImplementedError(f"monai_amd.ScaleIntensityRange: {data.dtype} images are not on the HIP path (float32 / integer images are)")
        x = data.to(torch.float32).contiguous()
        dtype = _to_torch_dtype(self.dtype) or x.dtype
        if self.a_max - self.a_min == 0.0:                       # array.py:999-1003: no division, no clip, no dtype conversion
            warnings.warn("Divide by zero (a_min == a_max)", Warning)
            out = ops.scale_intensity_range(x, self.a_min, 1.0, None if self.b_min is None else 1.0, self.b_min or 0.0, None, None)
        else:
            rescale = self.b_min is not None and self.b_max is not None
            if self.clip and self.b_min is None and self.b_max is None:       # array.py:1009 -> torch.clamp(img, None, None)
                raise RuntimeError("torch.clamp: At least one of 'min' or 'max' must not be None")
            out = ops.scale_intensity_range(
                x, self.a_min, self.a_max - self.a_min, (self.b_max - self.b_min) if rescale else None, self.b_min if rescale else 0.0,
                self.b_min if self.clip else None, self.b_max if self.clip else None)
            if out.dtype != dtype:
                out = out.to(dtype)
        if is_meta(img):
            return type(img)(out, meta=dict(img.meta), applied_operations=list(getattr(img, "applied_operations", [])))
        return out


class NormalizeIntensity:
    """``monai.transforms.NormalizeIntensity`` (monai/transforms/intensity/array.py:816-907): ``(img - mean) / std`` over the whole
    image or per channel, optionally over the non-zero voxels only (which are then the only ones rewritten), or with given
    ``subtrahend`` / ``divisor``.  Two HBM passes on the device tensor: fp64 {count, sum, sum of squares} per workgroup folded into
    the fp32 {mean, std} pair per channel ON THE DEVICE, then the apply pass reads that pair -- the reference's ``.item()`` round
    trips between the two do not exist here.  The reference's sums are fp32 tree sums, these are fp64: results agree to fp32
    rounding (<= 1e-6 relative, stated in the tests), not bit for bit.  Scalar / per-channel ``subtrahend`` / ``divisor`` only
    (voxel-wise arrays are not on the HIP path).  The input is never modified (the reference normalises float inputs in place)."""

    def __init__(self, subtrahend=None, divisor=None, nonzero: bool = False, channel_wise: bool = False, dtype=torch.float32) -> None:
        self.subtrahend, self.divisor, self.nonzero, self.channel_wise, self.dtype = subtrahend, divisor, nonzero, channel_wise, dtype

    @staticmethod
    def _per_channel(v, c: int, what: str):
        """None | scalar | sequence of c scalars -> list of c floats (or None)"""
        import numpy as np

        if v is None:
            return None
        a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
        if a.ndim == 0:
            return [float(a)] * c
        if a.ndim == 1 and a.shape[0] == c:
            return [float(x) for x in a]
        raise NotImplementedError(f"monai_amd.NormalizeIntensity: a {what} of shape {a.shape} is not on the HIP path (scalars / one value per channel are)")

    def __call__(self, img):
        from ... import ops

        data = img.as_tensor() if is_meta(img) else torch.as_tensor(img)
        if data.dtype in (torch.float64, torch.float16, torch.bfloat16) or data.is_complex():
            raise NotImplementedError(f"monai_amd.NormalizeIntensity: {data.dtype} images are not on the HIP path (float32 / integer images are)")
        dtype = _to_torch_dtype(self.dtype) or data.dtype
        if self.channel_wise:
            if self.subtrahend is not None and len(self.subtrahend) != len(data):
                raise ValueError(f"img has {len(data)} channels, but subtrahend has {len(self.subtrahend)} components.")
            if self.divisor is not None and len(self.divisor) != len(data):
                raise ValueError(f"img has {len(data)} channels, but divisor has {len(self.divisor)} components.")
        x = data.to(torch.float32).contiguous()
        c = int(x.shape[0]) if self.channel_wise and x.dim() > 0 else 1
        sub = self._per_channel(self.subtrahend, c, "subtrahend")
        div = self._per_channel(self.divisor, c, "divisor")
        if div is not None:
            div = [1.0 if d == 0.0 else d for d in div]                       # array.py:866-868
        if not x.numel():
            out = x
        else:
            n = x.numel() // c
            if sub is not None and div is not None:
                table = torch.tensor([[s, d] for s, d in zip(sub, div)], dtype=torch.float32).to(x.device)
            else:
                table = ops.normalize_stats(x, c, n, self.nonzero)
                if sub is not None:
                    table[:, 0] = torch.tensor(sub, dtype=torch.float32).to(x.device)
                if div is not None:
                    table[:, 1] = torch.tensor(div, dtype=torch.float32).to(x.device)
            out = ops.normalize_apply(x, c, n, self.nonzero, table)
        if out.dtype != dtype:
            out = out.to(dtype)
        if is_meta(img):
            return type(img)(out, meta=dict(img.meta), applied_operations=list(getattr(img, "applied_operations", [])))
        return out


class ScaleIntensity:
    """``monai.transforms.ScaleIntensity`` (monai/transforms/intensity/array.py:445-491): rescale to ``[minv, maxv]`` from the image's (or
    each channel's) own minimum / maximum, or multiply by ``1 + factor`` when both are None.  Min / max are exact reductions left in device
    memory, the apply pass keeps the reference's fp32 operator sequence: bit-identical results, no host round trip.  The reference
    converts the image to ``dtype`` BEFORE the arithmetic (rescale_array, monai/transforms/utils.py:229-251); float32 is on the HIP path."""

    def __init__(self, minv: float | None = 0.0, maxv: float | None = 1.0, factor: float | None = None, channel_wise: bool = False,
                 dtype=torch.float32) -> None:
        self.minv, self.maxv, self.factor, self.channel_wise, self.dtype = minv, maxv, factor, channel_wise, dtype

    def __call__(self, img):
        from ... import ops

        data = img.as_tensor() if is_meta(img) else torch.as_tensor(img)
        dtype = _to_torch_dtype(self.dtype)
        if data.dtype in (torch.float64, torch.float16, torch.bfloat16) or data.is_complex() or dtype not in (None, torch.float32):
            raise NotImplementedError("monai_amd.ScaleIntensity: float32 arithmetic only (float32 / integer images, dtype float32 or None)")
        if dtype is None and data.dtype != torch.float32:
            raise NotImplementedError("monai_amd.ScaleIntensity: dtype=None on an integer image is integer arithmetic in the reference; not on the HIP path")
        x = data.to(torch.float32).contiguous()
        if self.minv is not None or self.maxv is not None:
            c = int(x.shape[0]) if self.channel_wise and x.dim() > 0 else 1
            if not x.numel():
                out = x
            else:
                both = self.minv is not None and self.maxv is not None
                out = ops.minmax_scale(x, c, x.numel() // c, (self.maxv - self.minv) if both else None, self.minv if both else 0.0, self.minv)
        elif self.factor is not None:
            out = ops.scale_intensity_range(x, 0.0, 1.0, 1 + self.factor, 0.0, None, None)      # (x - 0) / 1 * (1 + factor) + 0: exact steps around one multiply
        else:
            out = x
        if is_meta(img):
            return type(img)(out, meta=dict(img.meta), applied_operations=list(getattr(img, "applied_operations", [])))
        return out
