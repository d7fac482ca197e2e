"""Host-side helpers with the reference's semantics (monai/utils/misc.py:162-299, monai/utils/module.py:61)."""

from __future__ import annotations

from collections.abc import Iterable, Sequence
from typing import Any

import numpy as np
import torch


def _issequence(x: Any) -> bool:
    if isinstance(x, torch.Tensor):
        return x.ndim > 0
    return isinstance(x, Iterable) and not isinstance(x, (str, bytes))


def ensure_tuple(vals: Any) -> tuple:
    """monai/utils/misc.py:162-167"""
    if isinstance(vals, (torch.Tensor, np.ndarray)) and vals.ndim == 0:
        return (vals.item(),)
    return tuple(vals) if _issequence(vals) else (vals,)


def ensure_tuple_rep(tup: Any, dim: int) -> tuple:
    """monai/utils/misc.py:190-228: a scalar is repeated `dim` times, a sequence must have length `dim`."""
    if isinstance(tup, torch.Tensor):
        tup = tup.detach().cpu().numpy()
    if isinstance(tup, np.ndarray):
        tup = tup.tolist()
    if not _issequence(tup):
        return (tup,) * dim
    if len(tup) == dim:
        return tuple(tup)
    raise ValueError(f"Sequence must have length {dim}, got {len(tup)}.")


def fall_back_tuple(user_provided: Any, default: Sequence, func=lambda x: x and x > 0) -> tuple:
    """monai/utils/misc.py:256-299: invalid (None / non-positive) components fall back to `default`."""
    ndim = len(default)
    user = ensure_tuple_rep(user_provided, ndim)
    return tuple(u if func(u) else d for d, u in zip(default, user))


def look_up_option(opt: Any, supported: Sequence[str], name: str = "option") -> str:
    """String-valued restatement of monai/utils/module.py:61-130 (accepts enum members through `.value`)."""
    val = getattr(opt, "value", opt)
    if isinstance(val, str):
        val = val.strip().lower()
    for s in supported:
        if val == s:
            return s
    raise ValueError(f"Unsupported {name}: {opt}, available options are {list(supported)}.")


def as_gather_f32(data, pad_value: float = 0.0):
    """View / convert a tensor for the pure-gather kernels (crop + pad, flip + permute), which copy 32-bit words without arithmetic:
    float32 as is; int32 as its BIT PATTERN (exact for every value; pad value 0); narrower integers
    / bool through float32 (exact: every value < 2^24); int64 through float32 only when every value is below 2^24 in magnitude --
    otherwise ``NotImplementedError`` (with MONAI installed the call falls through to the reference, whose slicing is exact for every
    dtype).  Returns (float32 tensor, pad value to hand to the kernel, restore function)."""
    import torch

    dt = data.dtype
    if dt == torch.float32:
        return data, float(pad_value), (lambda out: out)
    if dt == torch.int32 and pad_value == 0:
        return data.contiguous().view(torch.float32), 0.0, (lambda out: out.view(torch.int32))
    if dt in (torch.uint8, torch.int8, torch.int16, torch.bool):
        return data.to(torch.float32), float(pad_value), (lambda out: out.to(dt))
    if dt in (torch.int64, torch.int32):      # int32 with a non-zero pad value: small ints are denormal / NaN bit patterns, no float argument carries them
        if data.numel() and int(data.abs().max()) >= (1 << 24):
            raise NotImplementedError(f"monai_amd: {dt} images with values >= 2^24 are not on the HIP gather path here (float32 would lose bits)")
        return data.to(torch.float32), float(pad_value), (lambda out: out.to(dt))
    raise NotImplementedError(f"monai_amd: {dt} images are not on the HIP path (float32 and integer images are)")
