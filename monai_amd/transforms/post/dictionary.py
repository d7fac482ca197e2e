"""Dictionary wrappers ``Activationsd`` / ``AsDiscreted`` (monai/transforms/post/dictionary.py:102-212)."""

from __future__ import annotations

from collections.abc import Callable, Hashable, Mapping, Sequence

from ...utils.misc import ensure_tuple, ensure_tuple_rep
from .array import Activations, AsDiscrete

__all__ = ["Activationsd", "ActivationsD", "ActivationsDict", "AsDiscreted", "AsDiscreteD", "AsDiscreteDict"]


class _MapTransform:
    def __init__(self, keys, allow_missing_keys: bool = False) -> None:
        self.keys = ensure_tuple(keys)
        self.allow_missing_keys = allow_missing_keys
        if not self.keys:
            raise ValueError("keys must be non empty.")

    def key_iterator(self, data: Mapping[Hashable, object], *extra):
        """(key, *per-key options) for the keys present in `data`; a missing key raises unless allow_missing_keys."""
        for key, *vals in zip(self.keys, *extra):
            if key in data:
                yield (key, *vals)
            elif not self.allow_missing_keys:
                raise KeyError(f"Key `{key}` of transform `{self.__class__.__name__}` was missing in the data and allow_missing_keys==False.")


class Activationsd(_MapTransform):
    def __init__(self, keys, sigmoid: Sequence[bool] | bool = False, softmax: Sequence[bool] | bool = False,
                 other: Sequence[Callable] | Callable | None = None, allow_missing_keys: bool = False, **kwargs) -> None:
        super().__init__(keys, allow_missing_keys)
        self.sigmoid = ensure_tuple_rep(sigmoid, len(self.keys))
        self.softmax = ensure_tuple_rep(softmax, len(self.keys))
        self.other = ensure_tuple_rep(other, len(self.keys))
        self.converter = Activations()
        self.converter.kwargs = kwargs

    def __call__(self, data):
        d = dict(data)
        for key, sigmoid, softmax, other in self.key_iterator(d, self.sigmoid, self.softmax, self.other):
            d[key] = self.converter(d[key], sigmoid, softmax, other)
        return d


class AsDiscreted(_MapTransform):
    def __init__(self, keys, argmax: Sequence[bool] | bool = False, to_onehot: Sequence[int | None] | int | None = None,
                 threshold: Sequence[float | None] | float | None = None, rounding: Sequence[str | None] | str | None = None,
                 allow_missing_keys: bool = False, **kwargs) -> None:
        super().__init__(keys, allow_missing_keys)
        self.argmax = ensure_tuple_rep(argmax, len(self.keys))
        self.to_onehot = []
        for flag in ensure_tuple_rep(to_onehot, len(self.keys)):
            if isinstance(flag, bool):
                raise ValueError("`to_onehot=True/False` is deprecated, please use `to_onehot=num_classes` instead.")
            self.to_onehot.append(flag)
        self.threshold = []
        for flag in ensure_tuple_rep(threshold, len(self.keys)):
            if isinstance(flag, bool):
                raise ValueError("`threshold_values=True/False` is deprecated, please use `threshold=value` instead.")
            self.threshold.append(flag)
        self.rounding = ensure_tuple_rep(rounding, len(self.keys))
        self.converter = AsDiscrete()
        self.converter.kwargs = kwargs

    def __call__(self, data):
        d = dict(data)
        for key, argmax, to_onehot, threshold, rounding in self.key_iterator(d, self.argmax, self.to_onehot, self.threshold, self.rounding):
            d[key] = self.converter(d[key], argmax, to_onehot, threshold, rounding)
        return d


ActivationsD = ActivationsDict = Activationsd
AsDiscreteD = AsDiscreteDict = AsDiscreted
