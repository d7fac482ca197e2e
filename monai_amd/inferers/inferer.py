"""Inferer classes with the reference's public surface (monai/inferers/inferer.py:62-97, :399-552)."""

from __future__ import annotations

import warnings
from abc import ABC, abstractmethod
from collections.abc import Callable, Sequence
from typing import Any

import torch

from ..data.utils import compute_importance_map
from ..utils.misc import ensure_tuple, look_up_option
from .merger import AvgMerger, Merger
from .splitter import Splitter
from .utils import sliding_window_inference

__all__ = ["Inferer", "SlidingWindowInferer", "SlidingWindowInfererAdapt", "SliceInferer"]


class Inferer(ABC):
    """A callable ``inferer(inputs, network, *args, **kwargs)`` (reference: inferer.py:62-97)."""

    @abstractmethod
    def __call__(self, inputs: torch.Tensor, network: Callable, *args: Any, **kwargs: Any) -> Any:
        raise NotImplementedError(f"Subclass {self.__class__.__name__} must implement this method.")


class PatchInferer(Inferer):
    """Inference on patches: ``splitter`` -> batches of patches -> ``network`` -> one ``Merger`` per output.  Drop-in for
    monai/inferers/inferer.py:100-370 with the same arguments; ``buffer_size`` (a prefetch thread in the reference) is
    accepted and has no effect here -- the patches are views of a tensor that is already in HBM."""

    def __init__(
        self,
        splitter: Splitter | None = None,
        merger_cls: type[Merger] | str = AvgMerger,
        batch_size: int = 1,
        preprocessing: Callable | None = None,
        postprocessing: Callable | None = None,
        output_keys: Sequence | None = None,
        match_spatial_shape: bool = True,
        buffer_size: int = 0,
        **merger_kwargs: Any,
    ) -> None:
        Inferer.__init__(self)
        if not isinstance(splitter, (Splitter, type(None))):
            raise TypeError(
                "'splitter' should be a `Splitter` object that returns: "
                "an iterable of pairs of (patch, location) or a MetaTensor that has `PatchKeys.LOCATION` metadata)."
                f"{type(splitter)} is given."
            )
        self.splitter = splitter
        if isinstance(merger_cls, str):
            from pydoc import locate

            from . import merger as _merger_mod

            found = getattr(_merger_mod, merger_cls, None) or locate(merger_cls)
            if found is None:
                raise ValueError(f"The requested `merger_cls` ['{merger_cls}'] does not exist.")
            merger_cls = found
        if not (isinstance(merger_cls, type) and issubclass(merger_cls, Merger)):
            raise TypeError(f"'merger' should be a subclass of `Merger`, {merger_cls} is given.")
        self.merger_cls = merger_cls
        self.merger_kwargs = merger_kwargs
        if preprocessing is not None and not callable(preprocessing):
            raise TypeError(f"'preprocessing' should be a callable object, {type(preprocessing)} is given.")
        self.preprocessing = preprocessing
        if postprocessing is not None and not callable(postprocessing):
            raise TypeError(f"'postprocessing' should be a callable object, {type(postprocessing)} is given.")
        self.postprocessing = postprocessing
        if batch_size < 1:
            raise ValueError(f"`batch_size` must be a positive number, {batch_size} is given.")
        self.batch_size = batch_size
        self.output_keys = output_keys
        self.match_spatial_shape = match_spatial_shape
        self.buffer_size = buffer_size

    def _batch_sampler(self, patches):
        """(batch of patches, their locations, number of patches in the batch)."""
        if hasattr(patches, "meta") and isinstance(patches, torch.Tensor):     # MetaTensor of already split patches
            total = len(patches)
            for i in range(0, total, self.batch_size):
                n = min(self.batch_size, total - i)
                yield patches[i : i + n], patches[i : i + n].meta["location"], n
            return
        batch, locs = [], []
        for patch, loc in patches:
            batch.append(patch)
            locs.append(loc)
            if len(batch) == self.batch_size:
                yield torch.cat(batch), locs, len(batch)
                batch, locs = [], []
        if batch:
            yield torch.cat(batch), locs, len(batch)

    def _ensure_tuple_outputs(self, outputs: Any) -> tuple:
        if isinstance(outputs, dict):
            if self.output_keys is None:
                self.output_keys = list(outputs.keys())
            return tuple(outputs[k] for k in self.output_keys)
        return tuple(outputs) if isinstance(outputs, (list, tuple)) else (outputs,)

    def _run_inference(self, network: Callable, patch: torch.Tensor, *args: Any, **kwargs: Any) -> tuple:
        if self.preprocessing:
            patch = self.preprocessing(patch)
        outputs = network(patch, *args, **kwargs)
        if self.postprocessing:
            outputs = self.postprocessing(outputs)
        return self._ensure_tuple_outputs(outputs)

    def _get_merged_shapes(self, inputs, out_patch, ratio):
        if self.splitter is None:
            return None, None
        original = self.splitter.get_input_shape(inputs)
        padded = self.splitter.get_padded_shape(inputs)
        cropped_shape = tuple(out_patch.shape[:2]) + tuple(round(s * r) for s, r in zip(original, ratio))
        merged_shape = tuple(out_patch.shape[:2]) + tuple(round(s * r) for s, r in zip(padded, ratio))
        if not self.match_spatial_shape:
            cropped_shape = merged_shape
        return cropped_shape, merged_shape

    def _initialize_mergers(self, inputs, outputs, patches, batch_size):
        in_patch = torch.chunk(patches, batch_size)[0]
        mergers, ratios = [], []
        for out_patch_batch in outputs:
            out_patch = torch.chunk(out_patch_batch, batch_size)[0]
            ratio = tuple(op / ip for ip, op in zip(in_patch.shape[2:], out_patch.shape[2:]))
            merger_kwargs = self.merger_kwargs.copy()
            cropped_shape, merged_shape = self._get_merged_shapes(inputs, out_patch, ratio)
            if "merged_shape" not in merger_kwargs:
                merger_kwargs["merged_shape"] = merged_shape
                if merger_kwargs["merged_shape"] is None:
                    raise ValueError("`merged_shape` cannot be `None`.")
            if "cropped_shape" not in merger_kwargs:
                merger_kwargs["cropped_shape"] = cropped_shape
            mergers.append(self.merger_cls(**merger_kwargs))
            ratios.append(ratio)
        return mergers, ratios

    def _aggregate(self, outputs, locations, batch_size, mergers, ratios):
        for output_patches, merger, ratio in zip(outputs, mergers, ratios):
            for in_loc, out_patch in zip(locations, torch.chunk(output_patches, batch_size)):
                merger.aggregate(out_patch, [round(l * r) for l, r in zip(in_loc, ratio)])

    def __call__(self, inputs, network: Callable, *args: Any, **kwargs: Any) -> Any:
        if self.splitter is None:
            if isinstance(inputs, torch.Tensor):
                if hasattr(inputs, "meta"):
                    if "location" not in inputs.meta:
                        raise ValueError(
                            "`PatchKey.LOCATION` does not exists in `inputs.meta`. "
                            "If the inputs are already split into patches, the location of patches needs to be "
                            "provided as `PatchKey.LOCATION` metadata in a MetaTensor. "
                            "If the input is not already split, please provide `splitter`."
                        )
                else:
                    raise ValueError(
                        "`splitter` should be set if the input is not already split into patches. "
                        "For inputs that are split, the location of patches needs to be provided as "
                        "(image, location) pairs, or as `PatchKey.LOCATION` metadata in a MetaTensor. "
                        f"The provided inputs type is {type(inputs)}."
                    )
            patches_locations = inputs
        else:
            patches_locations = self.splitter(inputs)
        ratios: list = []
        mergers: list = []
        for patches, locations, batch_size in self._batch_sampler(patches_locations):
            outputs = self._run_inference(network, patches, *args, **kwargs)
            if not mergers:
                mergers, ratios = self._initialize_mergers(inputs, outputs, patches, batch_size)
            self._aggregate(outputs, locations, batch_size, mergers, ratios)
        merged_outputs = [merger.finalize() for merger in mergers]
        if self.output_keys:
            return dict(zip(self.output_keys, merged_outputs))
        if len(merged_outputs) == 1:
            return merged_outputs[0]
        return merged_outputs


class SlidingWindowInferer(Inferer):
    """Sliding-window inference with ``sw_batch_size`` windows per ``network`` call.

    Constructor and call signatures are those of the reference class (inferer.py:455-552), so a bundle's
    ``{"_target_": "SlidingWindowInferer", "roi_size": [96, 96, 96], "sw_batch_size": 4, "overlap": 0.5, ...}``
    instantiates this class unchanged once ``monai_amd.patch`` has installed it.  The arithmetic runs in
    ``monai_amd.inferers.utils.sliding_window_inference``.

    ``sw_batch_size``: a network with ``forward_into`` (the HIP conv engines) is handed up to 64 windows per launch instead
    -- the result does not depend on it, peak memory does.  ``MONAI_AMD_STRICT_SW_BATCH=1`` keeps this argument as given,
    ``MONAI_AMD_SW_BATCH=n`` sets the launch size.  half / bfloat16 volumes are widened to fp32 for the kernels and the result
    returned in the caller's dtype; a CPU volume with a ROCm ``sw_device`` is moved to HBM once.
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


    def argmax(self, inputs: torch.Tensor, network: Callable, *args: Any, labels_dtype=torch.float32, **kwargs: Any):
        """``AsDiscrete(argmax=True)(self(inputs, network))`` per batch element with the argmax fused into the blend epilogue
        (``monai_amd.inferers.utils.sliding_window_argmax``): returns ``[B, 1, *spatial]`` labels, never writes the logits volume."""
        return self.__call__(inputs, network, *args, _monai_amd_argmax=labels_dtype, **kwargs)


class SlidingWindowArgmaxInferer(SlidingWindowInferer):
    """A ``SlidingWindowInferer`` whose result is the label map: ``"_target_": "monai_amd.inferers.SlidingWindowArgmaxInferer"`` in a
    bundle replaces the pair (SlidingWindowInferer, AsDiscreted(argmax=True)) -- same constructor arguments plus ``labels_dtype``."""

    def __init__(self, *args: Any, labels_dtype=torch.float32, **kwargs: Any) -> None:
        super().__init__(*args, **kwargs)
        self.labels_dtype = labels_dtype

    def __call__(self, inputs: torch.Tensor, network: Callable, *args: Any, **kwargs: Any):
        kwargs.setdefault("_monai_amd_argmax", self.labels_dtype)
        return super().__call__(inputs, network, *args, **kwargs)


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
