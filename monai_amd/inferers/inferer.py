"""Inferer classes with the reference's public surface (monai/inferers/inferer.py:62-97, :399-552)."""

from __future__ import annotations

import itertools
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


class _Sink:
    """One network output on its way into a merged volume: the Merger, the output / input resolution ratio, and how a batch of
    output patches reaches the merger -- in one kernel launch when the merger can take batches (AvgMerger on the device), patch by
    patch through the public ``Merger.aggregate`` otherwise."""

    def __init__(self, merger: Merger, ratio: tuple):
        self.merger, self.ratio = merger, ratio

    def place(self, location) -> list:
        return [round(v * r) for v, r in zip(location, self.ratio)]

    def add(self, out_batch: torch.Tensor, locations: Sequence) -> None:
        where = [self.place(loc) for loc in locations]
        if hasattr(self.merger, "aggregate_batch"):
            self.merger.aggregate_batch(out_batch, where)
            return
        for out_patch, loc in zip(torch.chunk(out_batch, len(where)), where):
            self.merger.aggregate(out_patch, loc)


def _resolve_merger_class(merger_cls):
    """a Merger subclass, or its name: in this package's merger module first, then as a dotted path (inferer.py:153-163)"""
    if isinstance(merger_cls, str):
        from pydoc import locate

        from . import merger as here

        found = getattr(here, merger_cls, None) or locate(merger_cls)
        if found is None:
            raise ValueError(f"The requested `merger_cls` ['{merger_cls}'] does not exist.")
        merger_cls = found
    if not (isinstance(merger_cls, type) and issubclass(merger_cls, Merger)):
        raise TypeError(f"'merger' should be a subclass of `Merger`, {merger_cls} is given.")
    return merger_cls


def _require_callable_or_none(fn, what: str) -> None:
    if fn is not None and not callable(fn):
        raise TypeError(f"'{what}' should be a callable object, {type(fn)} is given.")


class PatchInferer(Inferer):
    """Inference on patches: ``splitter`` -> batches of patches -> ``network`` -> one ``Merger`` per output.  Drop-in for
    monai/inferers/inferer.py:100-370 (same arguments, outputs pinned bit for bit by tests/golden/patch_inferer.npz and by the
    reference's own tests/inferers/test_patch_inferer.py).  Own design: the patch stream is a generator of (batch, locations) --
    from ``SlidingWindowSplitter.split_batches`` (one gather launch for all patches, contiguous batch slices), from any other
    ``Splitter`` by grouping its pairs, or from already split inputs -- and each network output flows into a ``_Sink`` that hands a
    whole batch of output patches to the merger in ONE launch (``AvgMerger.aggregate_batch`` -> ``mh_patch_accumulate_batch_f32``,
    in patch order: the reference's bits).  ``buffer_size`` (a prefetch thread in the reference) is accepted and has no effect --
    the patches are already in HBM."""

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
        if splitter is not None and not isinstance(splitter, Splitter):
            raise TypeError("'splitter' should be a `Splitter` object that returns: "
                            "an iterable of pairs of (patch, location) or a MetaTensor that has `PatchKeys.LOCATION` metadata)."
                            f"{type(splitter)} is given.")
        self.merger_cls = _resolve_merger_class(merger_cls)
        _require_callable_or_none(preprocessing, "preprocessing")
        _require_callable_or_none(postprocessing, "postprocessing")
        if batch_size < 1:
            raise ValueError(f"`batch_size` must be a positive number, {batch_size} is given.")
        self.splitter, self.merger_kwargs = splitter, merger_kwargs
        self.preprocessing, self.postprocessing = preprocessing, postprocessing
        self.batch_size, self.output_keys = batch_size, output_keys
        self.match_spatial_shape, self.buffer_size = match_spatial_shape, buffer_size

    # ---- the patch stream ----------------------------------------------------------------------------
    def _stream(self, inputs):
        """generator of (patches [n * B, C, ...], n locations)"""
        if self.splitter is not None:
            if hasattr(self.splitter, "split_batches"):
                yield from self.splitter.split_batches(inputs, self.batch_size)
                return
            pairs = self.splitter(inputs)
        elif isinstance(inputs, torch.Tensor):
            # already split: a MetaTensor whose metadata carries the locations (PatchKeys.LOCATION)
            if not hasattr(inputs, "meta"):
                raise ValueError("`splitter` should be set if the input is not already split into patches. "
                                 "For inputs that are split, the location of patches needs to be provided as "
                                 "(image, location) pairs, or as `PatchKey.LOCATION` metadata in a MetaTensor. "
                                 f"The provided inputs type is {type(inputs)}.")
            if "location" not in inputs.meta:
                raise ValueError("`PatchKey.LOCATION` does not exists in `inputs.meta`. "
                                 "If the inputs are already split into patches, the location of patches needs to be "
                                 "provided as `PatchKey.LOCATION` metadata in a MetaTensor. "
                                 "If the input is not already split, please provide `splitter`.")
            for i in range(0, len(inputs), self.batch_size):
                part = inputs[i : i + self.batch_size]
                yield part, part.meta["location"]
            return
        else:
            pairs = inputs                                              # an iterable of (patch, location)
        it = iter(pairs)
        while True:
            group = list(itertools.islice(it, self.batch_size))
            if not group:
                return
            yield (torch.cat([g[0] for g in group]) if len(group) > 1 else group[0][0]), [g[1] for g in group]

    def _forward(self, network: Callable, patches: torch.Tensor, args, kwargs) -> tuple:
        if self.preprocessing:
            patches = self.preprocessing(patches)
        result = network(patches, *args, **kwargs)
        if self.postprocessing:
            result = self.postprocessing(result)
        if isinstance(result, dict):
            if self.output_keys is None:
                self.output_keys = list(result.keys())
            return tuple(result[k] for k in self.output_keys)
        return tuple(result) if isinstance(result, (list, tuple)) else (result,)

    def _open_sinks(self, inputs, patches: torch.Tensor, outputs: tuple, n: int) -> list:
        """one Merger per output, sized from the first batch: the output / input patch extents give the resolution ratio, the splitter
        the original and the padded extent of the whole input"""
        sinks = []
        in_extent = patches.shape[2:]
        for out in outputs:
            lead = (out.shape[0] // n, out.shape[1])
            ratio = tuple(o / i for i, o in zip(in_extent, out.shape[2:]))
            kw = dict(self.merger_kwargs)
            if self.splitter is not None:
                full = lead + tuple(round(v * r) for v, r in zip(self.splitter.get_padded_shape(inputs), ratio))
                crop = lead + tuple(round(v * r) for v, r in zip(self.splitter.get_input_shape(inputs), ratio))
                kw.setdefault("merged_shape", full)
                kw.setdefault("cropped_shape", crop if self.match_spatial_shape else full)
            elif "merged_shape" not in kw:
                raise ValueError("`merged_shape` cannot be `None`.")
            sinks.append(_Sink(self.merger_cls(**kw), ratio))
        return sinks

    def __call__(self, inputs, network: Callable, *args: Any, **kwargs: Any) -> Any:
        sinks: list = []
        for patches, locations in self._stream(inputs):
            outputs = self._forward(network, patches, args, kwargs)
            if not sinks:
                sinks = self._open_sinks(inputs, patches, outputs, len(locations))
            for sink, out in zip(sinks, outputs):
                sink.add(out, locations)
        merged = [sink.merger.finalize() for sink in sinks]
        if self.output_keys:
            return dict(zip(self.output_keys, merged))
        return merged[0] if len(merged) == 1 else merged


# constructor arguments of SlidingWindowInferer that are also arguments of sliding_window_inference (+ cpu_thresh), in positional order
_SW_ARGS = ("roi_size", "sw_batch_size", "overlap", "mode", "sigma_scale", "padding_mode", "cval", "sw_device", "device", "progress", "cpu_thresh",
            "buffer_steps", "buffer_dim", "with_coord")


def _positional_call(inputs, network, inf, per_call, args, kwargs):
    """Everything positional: extra positional arguments for the network travel behind sliding_window_inference's 17 own positionals
    (utils.py:42-62), and a network keyword that happens to be called like one of them (``mode=...``) stays the network's."""
    return sliding_window_inference(inputs, inf.roi_size, inf.sw_batch_size, network, inf.overlap, inf.mode, inf.sigma_scale, inf.padding_mode, inf.cval,
                                    inf.sw_device, per_call["device"], inf.progress, inf.roi_weight_map, None, per_call["buffer_steps"],
                                    per_call["buffer_dim"], inf.with_coord, *args, **kwargs)


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
        Inferer.__init__(self)
        mode = look_up_option(mode, ("constant", "gaussian"), "mode")      # ValueError on anything else, like the reference's BlendMode(mode)
        # the public attributes of the reference object (bundles and SliceInferer read and rewrite them), in its positional order
        for name, value in zip(_SW_ARGS, (roi_size, sw_batch_size, overlap, mode, sigma_scale, padding_mode, cval, sw_device, device, progress,
                                          cpu_thresh, buffer_steps, buffer_dim, with_coord)):
            setattr(self, name, value)
        self.roi_weight_map = self._cached_weight_map() if cache_roi_weight_map else None

    def _cached_weight_map(self):
        """``cache_roi_weight_map=True``: the importance map of a static roi, evaluated once on the host (inferer.py:488-505); a roi with a
        non-positive entry is resolved per image, so nothing can be cached for it -- the reference's warning"""
        roi = self.roi_size
        if not (isinstance(roi, Sequence) and min(roi) > 0):
            warnings.warn("cache_roi_weight_map=True, but cache is not created. (dynamic roi_size?)")
            return None
        try:
            return compute_importance_map(ensure_tuple(roi), mode=self.mode, sigma_scale=self.sigma_scale, device="cpu")
        except BaseException as e:
            raise RuntimeError(f"roi size {roi}, mode={self.mode}, sigma_scale={self.sigma_scale}, device={self.device}\n"
                               "Seems to be OOM. Please try smaller patch size or mode='constant' instead of mode='gaussian'.") from e

    def _too_large_for_device_output(self, inputs: torch.Tensor) -> bool:
        return self.cpu_thresh is not None and inputs.shape[2:].numel() > self.cpu_thresh

    def __call__(self, inputs: torch.Tensor, network: Callable, *args: Any, **kwargs: Any):
        """``device`` / ``buffer_steps`` / ``buffer_dim`` may be overridden per call (inferer.py:525-527)."""
        per_call = {k: kwargs.pop(k, getattr(self, k)) for k in ("device", "buffer_steps", "buffer_dim")}
        if per_call["device"] is None and self._too_large_for_device_output(inputs):
            per_call["device"] = "cpu"             # hand the stitched volume back in host memory for very large images
        return _positional_call(inputs, network, self, per_call, args, kwargs)

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
        if self.device is not None:                # the caller fixed the output device: nothing to adapt
            return SlidingWindowInferer.__call__(self, inputs, network, *args, **kwargs)
        rungs = [inputs.device] if (inputs.is_cuda and not self._too_large_for_device_output(inputs)) else []
        rungs.append(torch.device("cpu"))
        for i, where in enumerate(rungs):
            try:
                return SlidingWindowInferer.__call__(self, inputs, network, *args, device=where, **kwargs)
            except RuntimeError as err:
                if i + 1 == len(rungs) or "OutOfMemoryError" not in type(err).__name__:
                    raise
                warnings.warn(f"GPU stitching failed, attempting on CPU, image dim {tuple(inputs.shape)}.")
                self.cpu_thresh = inputs.shape[2:].numel() - 1      # the next image of this size goes to the host rung directly
                torch.cuda.empty_cache()


class SliceInferer(SlidingWindowInferer):
    """Slice-by-slice (2-D network) inference over a 3-D volume (reference: inferer.py:691-771): a 2-D ``roi_size`` gets a
    singleton inserted at ``spatial_dim`` and the network sees the windows with that axis squeezed."""

    def __init__(self, spatial_dim: int = 0, *args: Any, **kwargs: Any) -> None:
        SlidingWindowInferer.__init__(self, *args, **kwargs)
        self.spatial_dim, self.orig_roi_size = spatial_dim, ensure_tuple(self.roi_size)

    def __call__(self, inputs: torch.Tensor, network: Callable, *args: Any, **kwargs: Any):
        axis = self.spatial_dim
        if axis > 2:
            raise ValueError("`spatial_dim` can only be `0, 1, 2` with `[H, W, D]` respectively.")
        if len(self.orig_roi_size) != 2 or inputs.dim() != 5:
            raise RuntimeError(f"Currently, only 2D `roi_size` ({self.orig_roi_size}) with 3D `inputs` tensor (shape={inputs.shape}) is supported.")
        plane = self.orig_roi_size
        self.roi_size = list(plane[:axis]) + [1] + list(plane[axis:])          # one slice thick along `spatial_dim`

        def on_slices(windows):
            return self.network_wrapper(network, windows, *args, **kwargs)

        return SlidingWindowInferer.__call__(self, inputs=inputs, network=on_slices)

    def network_wrapper(self, network: Callable, x: torch.Tensor, *args: Any, **kwargs: Any):
        """the 2-D network sees the windows without their singleton axis; whatever it returns (tensor, mapping, sequence) gets the axis back"""
        from collections.abc import Mapping

        at = self.spatial_dim + 2
        result = network(x.squeeze(dim=at), *args, **kwargs)
        if isinstance(result, Mapping):
            return {key: value.unsqueeze(dim=at) for key, value in result.items()}
        if isinstance(result, torch.Tensor):
            return result.unsqueeze(dim=at)
        return tuple(value.unsqueeze(dim=at) for value in result)
