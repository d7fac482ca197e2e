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

__all__ = ["Inferer", "SlidingWindowInferer", "SlidingWindowInfererAdapt", "SliceInferer"]


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


class SlidingWindowInfererAdapt(SlidingWindowInferer):
    """``SlidingWindowInferer`` that survives an HBM out-of-memory error (reference: inferer.py:555-641).

    The reference degrades GPU stitching -> buffered stitching -> CPU stitching.  Here stitching never holds a
    partial volume, so the ladder has two rungs: (1) everything in HBM; (2) on ``OutOfMemoryError`` retry with the
    stitched volume handed back on the CPU (``device="cpu"``), remembering the image size in ``cpu_thresh`` so the next
    image of that size goes there directly -- the same externally visible behaviour (the output may land on the CPU)."""

    def __call__(self, inputs: torch.Tensor, network: Callable, *args: Any, **kwargs: Any):
        if self.device is not None:
            return super().__call__(inputs, network, *args, **kwargs)
        cpu_cond = self.cpu_thresh is not None and inputs.shape[2:].numel() > self.cpu_thresh
        gpu_stitching = inputs.is_cuda and not cpu_cond
        for _ in range(2):
            try:
                return super().__call__(inputs, network, *args, device=inputs.device if gpu_stitching else torch.device("cpu"), **kwargs)
            except RuntimeError as e:
                if not gpu_stitching or "OutOfMemoryError" not in type(e).__name__:
                    raise
                warnings.warn(f"GPU stitching failed, attempting on CPU, image dim {tuple(inputs.shape)}.")
                gpu_stitching = False
                self.cpu_thresh = inputs.shape[2:].numel() - 1
                torch.cuda.empty_cache()
        raise RuntimeError(f"SlidingWindowInfererAdapt could not finish: cpu_cond={cpu_cond} gpu_stitching={gpu_stitching}")


class SliceInferer(SlidingWindowInferer):
    """Slice-by-slice (2-D network) inference over a 3-D volume (reference: inferer.py:691-771): a 2-D ``roi_size`` gets a
    singleton inserted at ``spatial_dim`` and the network sees the windows with that axis squeezed."""

    def __init__(self, spatial_dim: int = 0, *args: Any, **kwargs: Any) -> None:
        self.spatial_dim = spatial_dim
        super().__init__(*args, **kwargs)
        self.orig_roi_size = ensure_tuple(self.roi_size)

    def __call__(self, inputs: torch.Tensor, network: Callable, *args: Any, **kwargs: Any):
        if self.spatial_dim > 2:
            raise ValueError("`spatial_dim` can only be `0, 1, 2` with `[H, W, D]` respectively.")
        self.roi_size = ensure_tuple(self.roi_size)
        if len(self.orig_roi_size) == 2 and len(inputs.shape[2:]) == 3:
            self.roi_size = list(self.orig_roi_size)
            self.roi_size.insert(self.spatial_dim, 1)
        else:
            raise RuntimeError(
                f"Currently, only 2D `roi_size` ({self.orig_roi_size}) with 3D `inputs` tensor (shape={inputs.shape}) is supported."
            )
        return super().__call__(inputs=inputs, network=lambda x: self.network_wrapper(network, x, *args, **kwargs))

    def network_wrapper(self, network: Callable, x: torch.Tensor, *args: Any, **kwargs: Any):
        from collections.abc import Mapping

        x = x.squeeze(dim=self.spatial_dim + 2)
        out = network(x, *args, **kwargs)
        if isinstance(out, torch.Tensor):
            return out.unsqueeze(dim=self.spatial_dim + 2)
        if isinstance(out, Mapping):
            return {k: v.unsqueeze(dim=self.spatial_dim + 2) for k, v in out.items()}
        return tuple(o.unsqueeze(dim=self.spatial_dim + 2) for o in out)
