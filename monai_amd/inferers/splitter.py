"""Patch splitters of the PatchInferer family -- drop-ins for monai/inferers/splitter.py:36-293 (``Splitter``,
``SlidingWindowSplitter``): same constructor arguments, same (patch, location) pairs in the same (row-major) order, same
error types (pinned by tests/golden/patch_inferer.npz from the real reference and by the reference's own
tests/inferers/test_sliding_window_splitter.py, which runs over this class in tests/test_reference_suites_emu.py).

Own design.  A split is described once, per axis, by a ``_GridAxis`` (patch extent, stride, padding on both sides, the
patch starts) -- ``get_padded_shape``, the padding itself and the locations all read that one plan, and the whole set of
locations is available as a tensor (``locations``).  A 3-D single-image fp32 volume in HBM is cut by ONE launch of the
window-gather kernel (``mh_window_extract_f32``: the plan's per-axis start lists are exactly its window grid) into a
dense ``[n_patches, C, ...]`` buffer, so ``split_batches`` -- what ``PatchInferer`` consumes -- hands the network
contiguous slices of that buffer instead of ``torch.cat``-ing ``batch_size`` views per call."""

from __future__ import annotations

import inspect
import itertools
from abc import ABC, abstractmethod
from collections.abc import Callable, Iterable, Iterator, Sequence
from typing import Any, NamedTuple

import torch

from ..utils.misc import ensure_tuple, ensure_tuple_rep

__all__ = ["Splitter", "SlidingWindowSplitter"]


class Splitter(ABC):
    """A callable that yields (patch, location) pairs (monai/inferers/splitter.py:36-91)."""

    def __init__(self, patch_size: Sequence[int] | int, device: torch.device | str | None = None) -> None:
        self.patch_size = patch_size
        self.device = device

    @abstractmethod
    def get_input_shape(self, inputs: Any) -> tuple:
        raise NotImplementedError(f"Subclass {self.__class__.__name__} must implement this method.")

    @abstractmethod
    def get_padded_shape(self, inputs: Any) -> tuple:
        raise NotImplementedError(f"Subclass {self.__class__.__name__} must implement this method.")

    @abstractmethod
    def __call__(self, inputs: Any) -> Iterable[tuple[torch.Tensor, Sequence[int]]]:
        raise NotImplementedError(f"Subclass {self.__class__.__name__} must implement this method.")


class _GridAxis(NamedTuple):
    """One axis of a sliding-window split.  Coordinates are those of the UNPADDED input: ``starts`` may begin below 0
    (negative offset) and patches may end beyond ``size``; ``lead`` / ``tail`` are the paddings that make them fit."""

    size: int
    patch: int
    stride: int
    lead: int
    tail: int
    starts: tuple

    @property
    def padded(self) -> int:
        return self.lead + self.size + self.tail


def _stride(patch: int, overlap, relative: bool) -> int:
    """Distance of neighbouring patch starts: a relative overlap is a fraction of the patch, an absolute one a number of elements
    (iter_patch_position, monai/data/utils.py:241-245 -- ``round(p * (1.0 - o))``, NOT the ``round(p - p * o)`` of the padding rule:
    the two land on opposite sides of a .5 tie for some (patch, overlap) pairs).  `relative` is ONE decision for all axes -- the type of the
    FIRST overlap entry, as the reference takes it (``isinstance(overlap[0], float)``), not a per-axis one"""
    return round(patch * (1.0 - overlap)) if relative else patch - overlap


def _pad_modulus(patch: int, overlap) -> int:
    """The modulus of the tail padding (splitter.py:184-188): ``round(ps - ps * ov)`` / ``round(ps - ov)``"""
    return round(patch - patch * overlap) if isinstance(overlap, float) else round(patch - overlap)


def _require_two_argument_callable(fn) -> None:
    """filter_fn(patch, location) -> bool: at least two parameters, at most two without a default (splitter.py:151-171)"""
    if fn is None:
        return
    if not callable(fn):
        raise ValueError(f"`filter_fn` should be a callable with two input parameters (patch, location). {type(fn)} is given.")
    params = list(inspect.signature(fn).parameters.values())
    required = sum(1 for q in params if q.default is inspect.Parameter.empty)
    if len(params) < 2:
        raise ValueError(f"`filter_fn` requires to accept at least two parameters (patch, location).The provided callable ({fn}) has {len(params)} parameters.")
    if required > 2:
        raise ValueError(f"`filter_fn` can have at most two positional parameters (patch, location).The provided callable ({fn}) has {required} positional parameters.")


class SlidingWindowSplitter(Splitter):
    """Regular grid of patches: ``overlap`` as a fraction in [0, 1) or a number of elements, a start ``offset`` (negative
    offsets and ragged ends are padded with ``pad_mode`` / ``pad_value``; ``pad_mode=None`` keeps only the patches that lie
    inside the input) and an optional ``filter_fn(patch, location) -> bool``.  Reference: splitter.py:94-293."""

    def __init__(
        self,
        patch_size: Sequence[int] | int,
        overlap: Sequence[float] | float | Sequence[int] | int = 0.0,
        offset: Sequence[int] | int = 0,
        filter_fn: Callable | None = None,
        pad_mode: str | None = "constant",
        pad_value: float | int = 0,
        device: torch.device | str | None = None,
    ) -> None:
        super().__init__(patch_size=patch_size, device=device)
        ov = ensure_tuple(overlap)
        if isinstance(ov[0], float):
            if not all(0.0 <= v < 1.0 for v in ov):
                raise ValueError(f"Relative overlap must be between 0.0 and 1.0 but {overlap} is given. "
                                 "If you wish to use number of pixels as overlap, please provide integer numbers.")
        elif min(ov) < 0:
            raise ValueError(f"Number of pixels for overlap cannot be negative. {overlap} is given. ")
        if not pad_mode and min(ensure_tuple(offset)) < 0:
            raise ValueError(f"Negative `offset`requires a valid padding mode but `mode` is set to {pad_mode}.")
        _require_two_argument_callable(filter_fn)
        self.overlap, self.offset, self.filter_fn, self.pad_mode, self.pad_value = overlap, offset, filter_fn, pad_mode, pad_value

    # ---- the plan ------------------------------------------------------------------------------------
    def plan(self, spatial_shape: Sequence[int]) -> tuple:
        """One ``_GridAxis`` per spatial axis of an input of this shape (validates overlap / offset against it)."""
        nd = len(spatial_shape)
        patches = ensure_tuple_rep(self.patch_size, nd)
        overlaps = ensure_tuple_rep(self.overlap, nd)
        zero = type(overlaps[0])(0)
        relative = isinstance(overlaps[0], float)
        overlaps = tuple(o if p else zero for o, p in zip(overlaps, patches))          # a 0 patch extent = the whole axis, no overlap
        if any(o > p for o, p in zip(overlaps, patches)):
            raise ValueError(f"`overlap` ({overlaps}) cannot be larger than patch size ({patches}).")
        axes = []
        for size, patch, ov, off in zip(spatial_shape, patches, overlaps, ensure_tuple_rep(self.offset, nd)):
            size, off = int(size), int(off)
            if off < -patch:
                raise ValueError(f"Negative `offset` ({off}) cannot be larger than `patch_size` ({patch}) in magnitude.")
            if off >= size:
                raise ValueError(f"`offset` ({off}) cannot be larger than inputs size ({size}).")
            if not patch:                               # whole axis: one patch at the offset (the reference's get_valid_patch_size rule)
                axes.append(_GridAxis(size, size, max(size, 1), 0, 0, tuple(range(off, 1, max(size, 1)))))
                continue
            stride = _stride(patch, ov, relative)
            lead = tail = 0
            if self.pad_mode:
                lead = max(-off, 0)
                tail = (off - size + patch) % _pad_modulus(patch, ov)    # smallest extension after which the last patch ends on the (padded) border
            starts = tuple(range(off, size + tail - patch + 1, stride))
            axes.append(_GridAxis(size, patch, stride, lead, tail, starts))
        return tuple(axes)

    def locations(self, spatial_shape: Sequence[int]) -> torch.Tensor:
        """All patch locations of an input of this spatial shape as an int64 tensor [n_patches, ndim], row-major (last axis fastest)
        -- in the unpadded input's coordinates, i.e. what ``__call__`` yields next to each patch."""
        axes = self.plan(spatial_shape)
        grids = torch.meshgrid(*[torch.tensor(a.starts, dtype=torch.int64) for a in axes], indexing="ij")
        return torch.stack([g.reshape(-1) for g in grids], dim=1)

    def get_input_shape(self, inputs: Any) -> tuple:
        return tuple(inputs.shape[2:])

    def get_padded_shape(self, inputs: Any) -> tuple:
        shape = self.get_input_shape(inputs)
        return tuple(a.padded for a in self.plan(shape)) if self.pad_mode else shape

    # ---- splitting -----------------------------------------------------------------------------------
    def _padded(self, inputs: torch.Tensor, axes) -> torch.Tensor:
        pads = [p for a in reversed(axes) for p in (a.lead, a.tail)]                  # F.pad order: last axis first, (front, back)
        return torch.nn.functional.pad(inputs, pads, mode=self.pad_mode, value=self.pad_value) if any(pads) else inputs

    def _gathered(self, src: torch.Tensor, axes):
        """All patches of a 3-D single-image fp32 device volume by ONE launch of the window-gather kernel -> [n_patches, C, pd, ph, pw]
        (None when the input is not of that kind: the caller slices views instead)."""
        if src.dim() != 5 or src.shape[0] != 1:
            return None
        from .. import _lib, ops

        try:
            _lib.require_device(src)               # fp32 in HBM; anything else is sliced as views
        except _lib.UnsupportedOnDevice:
            return None

        grid = [[s + a.lead for s in a.starts] for a in axes]
        n = len(grid[0]) * len(grid[1]) * len(grid[2])
        if n == 0:
            return None
        roi = tuple(a.patch for a in axes)
        # the dense buffer is the volume times the overlap factor (8x at overlap 0.5): only while it is a small part of the free HBM -- beyond that the
        # patches are sliced lazily as views, as the reference does (it keeps `batch_size` patches alive, nothing more)
        need = 4 * n * src.shape[1] * roi[0] * roi[1] * roi[2]
        if src.is_cuda and need > torch.cuda.mem_get_info(src.device)[0] // 4:
            return None
        try:
            out = torch.empty((n, src.shape[1]) + roi, dtype=torch.float32, device=src.device)
            return ops.window_extract(src[0].contiguous(), grid, 0, n, roi, out)
        except (_lib.KernelRejected, torch.cuda.OutOfMemoryError):
            # a gather too large for one launch (MH_ERR_UNSUPPORTED), or no room for the dense buffer: views.
            # Launch failures / HIP errors are NOT caught: a broken gather must fail, not degrade into a slow path
            return None

    def _pairs(self, inputs: torch.Tensor) -> Iterator[tuple[torch.Tensor, tuple]]:
        if not isinstance(inputs, torch.Tensor):
            raise ValueError(f"The input should be a tensor. {type(inputs)} is given.")
        axes = self.plan(inputs.shape[2:])
        src = self._padded(inputs, axes) if self.pad_mode else inputs
        dense = self._gathered(src, axes)
        for i, loc in enumerate(itertools.product(*[a.starts for a in axes])):
            if dense is not None:
                patch = dense[i : i + 1]
            else:
                patch = src[(slice(None), slice(None)) + tuple(slice(s + a.lead, s + a.lead + a.patch) for s, a in zip(loc, axes))]
            if self.device is not None:
                patch = patch.to(self.device)
            if self.filter_fn is None or self.filter_fn(patch, loc):
                yield patch, loc

    def __call__(self, inputs: Any) -> Iterable[tuple[torch.Tensor, Sequence[int]]]:
        return self._pairs(inputs)

    def split_batches(self, inputs: Any, batch_size: int) -> Iterator[tuple[torch.Tensor, list]]:
        """(``[<= batch_size * B, C, ...]`` patches, their locations) per network call.  Consecutive patches of the gathered buffer are
        handed over as ONE contiguous slice (no ``torch.cat``); views and filtered sequences are concatenated like the reference does."""
        held, locs = [], []
        for patch, loc in self._pairs(inputs):
            held.append(patch)
            locs.append(loc)
            if len(held) == batch_size:
                yield _join(held), locs
                held, locs = [], []
        if held:
            yield _join(held), locs


def _join(patches: list) -> torch.Tensor:
    """torch.cat of the patches -- or, when they are consecutive rows of one dense buffer, the slice that already holds them"""
    first = patches[0]
    base = first._base
    # one image per patch only: the batched view's dim-0 stride is the distance of consecutive patches, which a multi-image patch (dim-0 stride = the
    # source's batch stride) cannot express -- and `is_contiguous()` ignores the stride of a size-1 dim, so it is written out, never taken from `first`
    if base is not None and len(patches) > 1 and first.shape[0] == 1 and first.is_contiguous() and all(p._base is base for p in patches):
        step = first.numel()
        o0 = first.storage_offset()
        if all(p.storage_offset() == o0 + i * step and p.shape == first.shape and p.stride()[1:] == first.stride()[1:] for i, p in enumerate(patches)):
            return base.as_strided((len(patches),) + tuple(first.shape[1:]), (step,) + tuple(first.stride()[1:]), o0)
    return torch.cat(patches) if len(patches) > 1 else first
