"""Host-side index math of the sliding-window path (tiny; runs once per call on the CPU).

Same results as the reference helpers in monai/data/utils.py -- ``dense_patch_slices`` :166-206,
``get_valid_patch_size`` :343-354, ``compute_importance_map`` :1084-1134 -- but organised around the
per-axis start lists that the blend / gather kernels take (the window grid is their cartesian product).
"""

from __future__ import annotations

import itertools
import math
from collections.abc import Sequence

import torch

from ..utils.misc import ensure_tuple_rep, look_up_option

__all__ = ["window_starts", "dense_patch_slices", "get_valid_patch_size", "compute_importance_map"]


def get_valid_patch_size(image_size: Sequence[int], patch_size) -> tuple:
    """A patch dimension that is 0/None, missing, or larger than the image becomes the image dimension."""
    ps = tuple(patch_size) if isinstance(patch_size, (Sequence, torch.Size)) else (patch_size,)
    ps = (ps + (0,) * len(image_size))[: len(image_size)]  # ensure_tuple_size semantics: missing axes -> whole axis
    return tuple(min(int(m), int(p) if p else int(m)) for m, p in zip(image_size, ps))


def window_starts(image_size: Sequence[int], patch_size: Sequence[int], scan_interval: Sequence[int]) -> list:
    """Per-axis ascending window start indices.

    Along one axis, window k sits at ``k * interval`` and is pulled back so that it ends inside the image; windows
    are generated up to and including the first one that reaches the image end."""
    patch = get_valid_patch_size(image_size, patch_size)
    steps = tuple(scan_interval) + (0,) * (len(image_size) - len(tuple(scan_interval)))
    axes = []
    for size, p, step in zip(image_size, patch, steps):
        if step == 0:
            count = 1
        else:
            count = 1
            for k in range(int(math.ceil(float(size) / step))):
                if k * step + p >= size:
                    count = k + 1
                    break
        axes.append([min(k * step, size - p) for k in range(count)])
    return axes


def dense_patch_slices(image_size, patch_size, scan_interval, return_slice: bool = True) -> list:
    """All window slices, row-major over the per-axis starts (last axis fastest)."""
    patch = get_valid_patch_size(image_size, patch_size)
    axes = window_starts(image_size, patch, scan_interval)
    if return_slice:
        return [tuple(slice(s, s + patch[d]) for d, s in enumerate(w)) for w in itertools.product(*axes)]
    return [tuple((s, s + patch[d]) for d, s in enumerate(w)) for w in itertools.product(*axes)]


def compute_importance_map(patch_size, mode="constant", sigma_scale=0.125, device="cpu", dtype=torch.float32) -> torch.Tensor:
    """Window weight map.  "constant": ones.  "gaussian": outer product over the axes of
    ``exp(x^2 / (-2 sigma^2))`` with ``x`` centred integer offsets and ``sigma = sigma_scale * size``; clamped from
    below by ``max(min(map), 1e-3)``.

    The 1-D factors are always evaluated with torch on the HOST in fp32 and the outer product / clamp in fp32 as
    well, so the map is bit-identical to the reference's CPU map whatever `device` is (a device-side ``expf`` may
    differ by an ulp; SURVEY.md section 7 "reproducibility quirks")."""
    mode = look_up_option(mode, ("constant", "gaussian"), "mode")
    patch_size = tuple(int(p) for p in patch_size)
    if mode == "constant":
        weights = torch.ones(patch_size, dtype=torch.float)
    else:
        scales = ensure_tuple_rep(sigma_scale, len(patch_size))
        weights = None
        for axis, (n, s) in enumerate(zip(patch_size, scales)):
            sigma = n * s
            offs = torch.arange(start=-(n - 1) / 2.0, end=(n - 1) / 2.0 + 1, dtype=torch.float)
            g = torch.exp(offs**2 / (-2 * sigma**2))
            weights = g if axis == 0 else weights.unsqueeze(-1) * g[(None,) * axis]
    floor = max(torch.min(weights).item(), 1e-3)
    weights = torch.clamp_(weights.to(torch.float), min=floor).to(dtype)
    return weights.to(device)
