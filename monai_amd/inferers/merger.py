"""Patch mergers of the PatchInferer family -- drop-ins for monai/inferers/merger.py:38-205 (``Merger``, ``AvgMerger``).
``AvgMerger`` keeps its aggregation tensors in HBM and runs ``values[slice] += patch; counts[slice] += 1`` and the final
``values /= counts`` as HIP kernels (csrc/kernels/sliding.h) -- the same accumulation order as the reference (patch
order), so results are bit-identical to it."""

from __future__ import annotations

from abc import ABC, abstractmethod
from collections.abc import Sequence
from typing import Any

import torch

from .. import ops

__all__ = ["Merger", "AvgMerger"]


class Merger(ABC):
    """Base class (merger.py:38-100): ``aggregate(values, location)`` per patch batch element, ``finalize()`` once."""

    def __init__(self, merged_shape: Sequence[int], cropped_shape: Sequence[int] | None = None, device: torch.device | str | None = None) -> None:
        if merged_shape is None:
            raise ValueError("Argument `merged_shape` must be provided")
        self.merged_shape: tuple[int, ...] = tuple(merged_shape)
        self.cropped_shape: tuple[int, ...] = tuple(self.merged_shape if cropped_shape is None else cropped_shape)
        self.device = device
        self.is_finalized = False

    @abstractmethod
    def aggregate(self, values: torch.Tensor, location: Sequence[int]) -> None:
        raise NotImplementedError(f"Subclass {self.__class__.__name__} must implement this method.")

    @abstractmethod
    def finalize(self) -> Any:
        raise NotImplementedError(f"Subclass {self.__class__.__name__} must implement this method.")


class AvgMerger(Merger):
    """Average of the overlapping patches (merger.py:103-205).  ``device`` defaults to the device of the first aggregated
    patch (the reference's default "cpu" has no counterpart on this path); ``value_dtype`` float32 and ``count_dtype``
    uint8 (the reference's defaults) are the combination the kernels implement."""

    def __init__(
        self,
        merged_shape: Sequence[int],
        cropped_shape: Sequence[int] | None = None,
        value_dtype: torch.dtype = torch.float32,
        count_dtype: torch.dtype = torch.uint8,
        device: torch.device | str | None = None,
    ) -> None:
        super().__init__(merged_shape=merged_shape, cropped_shape=cropped_shape, device=device)
        if not self.merged_shape:
            raise ValueError(f"`merged_shape` must be provided for `AvgMerger`. {self.merged_shape} is give.")
        if value_dtype != torch.float32 or count_dtype != torch.uint8:
            raise NotImplementedError("monai_amd.AvgMerger: value_dtype=float32 with count_dtype=uint8 is what the HIP path implements")
        self.value_dtype = value_dtype
        self.count_dtype = count_dtype
        self.values: torch.Tensor | None = None
        self.counts: torch.Tensor | None = None
        if device is not None and torch.device(device).type != "cpu":
            self._allocate(torch.device(device))

    def _allocate(self, device) -> None:
        self.values = torch.zeros(self.merged_shape, dtype=self.value_dtype, device=device)
        self.counts = torch.zeros(self.merged_shape, dtype=self.count_dtype, device=device)

    def aggregate(self, values: torch.Tensor, location: Sequence[int]) -> None:
        if self.is_finalized:
            raise ValueError("`AvgMerger` is already finalized. Please instantiate a new object to aggregate.")
        if self.values is None:
            self._allocate(values.device)
        ops.patch_accumulate(self.values, self.counts, values.to(self.value_dtype).contiguous(), location)

    def aggregate_batch(self, values: torch.Tensor, locations: Sequence[Sequence[int]]) -> None:
        """Extension over the reference API: ``len(locations)`` patches at once (``values`` = their torch.cat), accumulated by ONE kernel launch in
        patch order -- bit-identical to ``aggregate`` called patch by patch, also where patches of the batch overlap each other."""
        if self.is_finalized:
            raise ValueError("`AvgMerger` is already finalized. Please instantiate a new object to aggregate.")
        if self.values is None:
            self._allocate(values.device)
        ops.patch_accumulate_batch(self.values, self.counts, values.to(self.value_dtype).contiguous(), locations)

    def finalize(self) -> torch.Tensor:
        """values /= counts, cropped to ``cropped_shape``; idempotent like the reference's."""
        if not self.is_finalized:
            if self.values is None:
                raise ValueError("`AvgMerger.finalize` called before any patch was aggregated.")
            ops.avg_finalize(self.values, self.counts)
            self.values = self.values[tuple(slice(0, end) for end in self.cropped_shape)]
            self.is_finalized = True
        return self.values

    def get_output(self) -> torch.Tensor:
        return self.finalize()

    def get_values(self) -> torch.Tensor:
        return self.values

    def get_counts(self) -> torch.Tensor:
        return self.counts
