"""Dictionary forms ``Activationsd`` / ``AsDiscreted`` of the post-processing transforms (contract: monai/transforms/post/dictionary.py:102-212 --
constructor signatures, per-key broadcasting of scalar options, the ``converter`` attribute with its ``kwargs``, missing-key behaviour).

Both are the same thing: ONE array transform (array.py, HIP kernels) applied to every key with that key's call options.  `_PerKey` holds the
option table -- one row per key -- and the loop; a dictionary transform only names its array transform and its options."""

from __future__ import annotations

from collections.abc import Callable, Hashable, Mapping, Sequence

from ...utils.misc import ensure_tuple, ensure_tuple_rep
from .array import Activations, AsDiscrete

__all__ = ["Activationsd", "ActivationsD", "ActivationsDict", "AsDiscreted", "AsDiscreteD", "AsDiscreteDict"]


class _PerKey:
    """keys + a table of per-key call options for `array_transform`; ``self.<option>`` is the column of that option (a tuple, one entry per key)"""

    array_transform: type = object
    # options that used to be boolean switches in old MONAI releases and now take a value: option name -> (the switch's old name, what to pass instead)
    valued_options: Mapping[str, tuple] = {}

    def _bind(self, keys, allow_missing_keys: bool, converter_kwargs: dict, **options) -> None:
        self.keys = ensure_tuple(keys)
        self.allow_missing_keys = allow_missing_keys
        if not self.keys:
            raise ValueError("keys must be non empty.")
        self._columns = tuple(options)
        for name, value in options.items():
            column = ensure_tuple_rep(value, len(self.keys))
            if name in self.valued_options and any(isinstance(v, bool) for v in column):
                old, new = self.valued_options[name]
                raise ValueError(f"`{old}=True/False` is deprecated, please use `{name}={new}` instead.")
            setattr(self, name, column)
        self.converter = self.array_transform()
        self.converter.kwargs = converter_kwargs

    def key_iterator(self, data: Mapping[Hashable, object], *extra):
        """(key, *per-key values) for the keys present in `data`; a missing key raises unless allow_missing_keys"""
        for row in zip(self.keys, *extra):
            if row[0] in data:
                yield row
            elif not self.allow_missing_keys:
                raise KeyError(f"Key `{row[0]}` of transform `{type(self).__name__}` was missing in the data and allow_missing_keys==False.")

    def __call__(self, data):
        out = dict(data)
        for key, *opts in self.key_iterator(out, *(getattr(self, c) for c in self._columns)):
            out[key] = self.converter(out[key], *opts)
        return out


class Activationsd(_PerKey):
    array_transform = Activations

    def __init__(self, keys, sigmoid: Sequence[bool] | bool = False, softmax: Sequence[bool] | bool = False,
                 other: Sequence[Callable] | Callable | None = None, allow_missing_keys: bool = False, **kwargs) -> None:
        self._bind(keys, allow_missing_keys, kwargs, sigmoid=sigmoid, softmax=softmax, other=other)


class AsDiscreted(_PerKey):
    array_transform = AsDiscrete
    valued_options = {"to_onehot": ("to_onehot", "num_classes"), "threshold": ("threshold_values", "value")}      # the reference's messages, post/dictionary.py:190-197

    def __init__(self, keys, argmax: Sequence[bool] | bool = False, to_onehot: Sequence[int | None] | int | None = None,
                 threshold: Sequence[float | None] | float | None = None, rounding: Sequence[str | None] | str | None = None,
                 allow_missing_keys: bool = False, **kwargs) -> None:
        self._bind(keys, allow_missing_keys, kwargs, argmax=argmax, to_onehot=to_onehot, threshold=threshold, rounding=rounding)


ActivationsD = ActivationsDict = Activationsd
AsDiscreteD = AsDiscreteDict = AsDiscreted
