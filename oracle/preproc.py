"""Oracle (test infrastructure): the pre-processing transforms in front of the path restated in plain torch-CPU / numpy.

Reference followed (paths relative to /root/reference):
  * ``ScaleIntensityRange.__call__``                 monai/transforms/intensity/array.py:993-1012
  * ``NormalizeIntensity._normalize`` / ``__call__``  monai/transforms/intensity/array.py:849-907
  * ``generate_spatial_bounding_box``                monai/transforms/utils.py:1069-1129
  * ``CropForeground.compute_bounding_box`` / ``crop_pad``   monai/transforms/croppad/array.py:847-927
  * ``orientation`` (flip + permute)                 monai/transforms/spatial/functional.py:187-229
"""

from __future__ import annotations

import numpy as np
import torch


def scale_intensity_range(img: torch.Tensor, a_min, a_max, b_min=None, b_max=None, clip=False) -> torch.Tensor:
    img = img.to(torch.float32) if not img.dtype.is_floating_point else img
    if a_max - a_min == 0.0:
        return img - a_min if b_min is None else img - a_min + b_min
    img = (img - a_min) / (a_max - a_min)
    if b_min is not None and b_max is not None:
        img = img * (b_max - b_min) + b_min
    if clip:
        img = torch.clamp(img, b_min, b_max)
    return img


def normalize_intensity(img: torch.Tensor, nonzero=False, channel_wise=False) -> torch.Tensor:
    img = img.to(torch.float32).clone()

    def one(t):
        mask = t != 0 if nonzero else torch.ones_like(t, dtype=torch.bool)
        if not mask.any():
            return t
        v = t[mask]
        sub, div = torch.mean(v).item(), torch.std(v, unbiased=False).item()
        div = 1.0 if div == 0.0 else div
        t[mask] = (v - sub) / div
        return t

    if channel_wise:
        for i in range(len(img)):
            img[i] = one(img[i])
        return img
    return one(img)


def foreground_box(img: torch.Tensor, margin=0, allow_smaller=False, k_divisible=1):
    """start / end of CropForeground's box for the default ``select_fn`` (values > 0 in any channel)"""
    data = (img > 0).any(0).numpy()
    nd = data.ndim
    margin = [margin] * nd if np.isscalar(margin) else list(margin)
    if not data.any():
        start, end = [0] * nd, [0] * nd
    else:
        start, end = [], []
        for d in range(nd):
            idx = np.nonzero(data.any(axis=tuple(a for a in range(nd) if a != d)))[0]
            lo, hi = int(idx[0]) - margin[d], int(idx[-1]) + margin[d] + 1
            if allow_smaller:
                lo, hi = max(lo, 0), min(hi, data.shape[d])
            start.append(lo)
            end.append(hi)
    start, end = np.asarray(start), np.asarray(end)
    k = np.asarray([k_divisible] * nd if np.isscalar(k_divisible) else k_divisible)
    size = end - start
    new = np.where(k > 0, np.ceil(size / np.maximum(k, 1)) * np.maximum(k, 1), size).astype(int)
    start = start - (new - size) // 2
    return start, start + new


def crop_pad(img: torch.Tensor, start, end, value=0.0) -> torch.Tensor:
    nd = img.dim() - 1
    start = [int(s) for s in start]
    end = [max(int(e), max(s, 0)) for s, e in zip(start, end)]
    out = torch.full((img.shape[0],) + tuple(e - s for s, e in zip(start, end)), value, dtype=img.dtype)
    src = tuple(slice(max(s, 0), min(e, img.shape[d + 1])) for d, (s, e) in enumerate(zip(start, end)))
    dst = tuple(slice(sl.start - s, sl.stop - s) for sl, s in zip(src, start))
    if all(sl.stop > sl.start for sl in src):
        out[(slice(None),) + dst] = img[(slice(None),) + src]
    return out


def flip_permute(img: torch.Tensor, perm, flips) -> torch.Tensor:
    axes = [a + 1 for a, f in enumerate(flips) if f]
    out = torch.flip(img, axes) if axes else img
    return out.permute([0] + [int(p) + 1 for p in perm]).contiguous()
