"""``Activations`` / ``AsDiscrete`` -- the step after the sliding-window inferer in every segmentation bundle -- on HIP
kernels (csrc/kernels/post.h).  Drop-ins for monai/transforms/post/array.py:61-237: same arguments, defaults and errors.
Channel-first tensors (no batch axis), reductions over ``dim=0`` (the reference's default; other axes are not on the HIP
path).  Outputs are float32 like the reference's (``convert_to_dst_type(..., dtype=torch.float)``)."""

from __future__ import annotations

from collections.abc import Callable

import torch

from ... import ops
from ...utils.misc import look_up_option

__all__ = ["Activations", "AsDiscrete"]


def _is_meta(x) -> bool:
    return type(x) is not torch.Tensor and hasattr(x, "as_tensor")


def _like(out: torch.Tensor, img):
    if _is_meta(img):
        return type(img)(out).copy_meta_from(img)
    return out


def _plain(img) -> torch.Tensor:
    if not isinstance(img, torch.Tensor):
        raise TypeError(f"monai_amd: a device tensor is required, got {type(img).__name__} (no CPU / numpy path in the product)")
    return img.as_tensor() if _is_meta(img) else img


class Activations:
    def __init__(self, sigmoid: bool = False, softmax: bool = False, other: Callable | None = None, **kwargs) -> None:
        self.sigmoid = sigmoid
        self.softmax = softmax
        self.kwargs = kwargs
        if other is not None and not callable(other):
            raise TypeError(f"other must be None or callable but is {type(other).__name__}.")
        self.other = other

    def __call__(self, img, sigmoid: bool | None = None, softmax: bool | None = None, other: Callable | None = None):
        if sigmoid and softmax:
            raise ValueError("Incompatible values: sigmoid=True and softmax=True.")
        if other is not None and not callable(other):
            raise TypeError(f"other must be None or callable but is {type(other).__name__}.")
        t = _plain(img).to(torch.float32).contiguous()
        if sigmoid or self.sigmoid:
            t = ops.pointwise("sigmoid", t)
        if softmax or self.softmax:
            if self.kwargs.get("dim", 0) != 0:
                raise NotImplementedError("monai_amd.Activations: softmax over dim=0 (the channel axis) is what the HIP path implements")
            t = ops.channel_reduce("softmax", t)
        act_func = self.other if other is None else other
        if act_func is not None:
            t = act_func(t)
        return _like(t, img)


class AsDiscrete:
    def __init__(self, argmax: bool = False, to_onehot: int | None = None, threshold: float | None = None, rounding: str | None = None, **kwargs) -> None:
        self.argmax = argmax
        if isinstance(to_onehot, bool):
            raise ValueError("`to_onehot=True/False` is deprecated, please use `to_onehot=num_classes` instead.")
        self.to_onehot = to_onehot
        self.threshold = threshold
        self.rounding = rounding
        self.kwargs = kwargs

    def __call__(self, img, argmax: bool | None = None, to_onehot: int | None = None, threshold: float | None = None, rounding: str | None = None):
        if isinstance(to_onehot, bool):
            raise ValueError("`to_onehot=True/False` is deprecated, please use `to_onehot=num_classes` instead.")
        if self.kwargs.get("dim", 0) != 0 or not self.kwargs.get("keepdim", True) or self.kwargs.get("dtype", torch.float) not in (torch.float, torch.float32):
            raise NotImplementedError("monai_amd.AsDiscrete: dim=0, keepdim=True, dtype=float32 (the reference's defaults) are what the HIP path implements")
        t = _plain(img).to(torch.float32).contiguous()
        argmax = self.argmax if argmax is None else argmax
        if argmax:
            t = ops.channel_reduce("argmax", t)
        to_onehot = self.to_onehot if to_onehot is None else to_onehot
        if to_onehot is not None:
            if not isinstance(to_onehot, int):
                raise ValueError(f"the number of classes for One-Hot must be an integer, got {type(to_onehot)}.")
            if t.dim() < 2:      # a scalar / 1-D label: the reference's one_hot reshapes it (networks/utils.py:170-220); not a volume for the kernel
                raise NotImplementedError("monai_amd.AsDiscrete: one-hot of a label without spatial axes is not on the HIP path")
            if t.shape[0] != 1:
                raise AssertionError("labels should have a channel with length equal to one.")
            t = ops.onehot(t, to_onehot)
        threshold = self.threshold if threshold is None else threshold
        if threshold is not None:
            t = ops.pointwise("threshold", t, float(threshold))
        rounding = self.rounding if rounding is None else rounding
        if rounding is not None:
            look_up_option(rounding, ["torchrounding"])
            t = ops.pointwise("round", t)
        return _like(t, img)
