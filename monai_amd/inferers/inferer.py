"""Inferer classes with the reference's public surface (monai/inferers/inferer.py:62-97, :399-552)."""

from __future__ import annotations

import warnings
from abc import ABC, abstractmethod
from collections.abc import Callable, Sequence
from typing import Any

import torch

from ..data.utils import compute_importance_map
from ..utils.misc import ensure_tuple, look_up_option
from .utils import sliding_window_inference

__all__ = ["Inferer", "SlidingWindowInferer"]


class Inferer(ABC):
    """A callable ``inferer(inputs, network, *args, **kwargs)`` (reference: inferer.py:62-97)."""

    @abstractmethod
    def __call__(self, inputs: torch.Tensor, network: Callable, *args: Any, **kwargs: Any) -> Any:
        raise NotImplementedError(f"Subclass {self.__class__.__name__} must implement this method.")


class SlidingWindowInferer(Inferer):
    """Sliding-window inference with ``sw_batch_size`` windows per ``network`` call.

    Constructor and call signatures are those of the reference class (inferer.py:455-552), so a bundle's
    ``{"_target_": "SlidingWindowInferer", "roi_size": [96, 96, 96], "sw_batch_size": 4, "overlap": 0.5, ...}``
    instantiates this class unchanged once ``monai_amd.patch`` has installed it.  The arithmetic runs in
    ``monai_amd.inferers.utils.sliding_window_inference``.
    """

    def __init__(
        self,
        roi_size: Sequence[int] | int,
        sw_batch_size: int = 1,
        overlap: Sequence[float] | float = 0.25,
        mode: str = "constant",
        sigma_scale: Sequence[float] | float = 0.125,
        padding_mode: str = "constant",
        cval: float = 0.0,
        sw_device=None,
        device=None,
        progress: bool = False,
        cache_roi_weight_map: bool = False,
        cpu_thresh: int | None = None,
        buffer_steps: int | None = None,
        buffer_dim: int = -1,
        with_coord: bool = False,
    ) -> None:
        super().__init__()
        self.roi_size = roi_size
        self.sw_batch_size = sw_batch_size
        self.overlap = overlap
        self.mode = look_up_option(mode, ("constant", "gaussian"), "mode")  # ValueError on anything else, like BlendMode(mode)
        self.sigma_scale = sigma_scale
        self.padding_mode = padding_mode
        self.cval = cval
        self.sw_device = sw_device
        self.device = device
        self.progress = progress
        self.cpu_thresh = cpu_thresh
        self.buffer_steps = buffer_steps
        self.buffer_dim = buffer_dim
        self.with_coord = with_coord

        # the weight map of a static roi can be computed once (inferer.py:488-505)
        self.roi_weight_map = None
        try:
            if cache_roi_weight_map and isinstance(roi_size, Sequence) and min(roi_size) > 0:
                self.roi_weight_map = compute_importance_map(ensure_tuple(self.roi_size), mode=mode, sigma_scale=sigma_scale, device="cpu")
            if cache_roi_weight_map and self.roi_weight_map is None:
                warnings.warn("cache_roi_weight_map=True, but cache is not created. (dynamic roi_size?)")
        except BaseException as e:
            raise RuntimeError(
                f"roi size {self.roi_size}, mode={mode}, sigma_scale={sigma_scale}, device={device}\n"
                "Seems to be OOM. Please try smaller patch size or mode='constant' instead of mode='gaussian'."
            ) from e

    def __call__(self, inputs: torch.Tensor, network: Callable, *args: Any, **kwargs: Any):
        """``device`` / ``buffer_steps`` / ``buffer_dim`` may be overridden per call (inferer.py:525-527)."""
        device = kwargs.pop("device", self.device)
        buffer_steps = kwargs.pop("buffer_steps", self.buffer_steps)
        buffer_dim = kwargs.pop("buffer_dim", self.buffer_dim)
        if device is None and self.cpu_thresh is not None and inputs.shape[2:].numel() > self.cpu_thresh:
            device = "cpu"  # hand the stitched volume back in host memory for very large images
        return sliding_window_inference(
            inputs,
            self.roi_size,
            self.sw_batch_size,
            network,
            self.overlap,
            self.mode,
            self.sigma_scale,
            self.padding_mode,
            self.cval,
            self.sw_device,
            device,
            self.progress,
            self.roi_weight_map,
            None,
            buffer_steps,
            buffer_dim,
            self.with_coord,
            *args,
            **kwargs,
        )
