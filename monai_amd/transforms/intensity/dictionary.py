"""``GaussianSmoothd`` (monai/transforms/intensity/dictionary.py:1184-1216) and ``ScaleIntensityRanged`` (dictionary.py:855-892)."""

from __future__ import annotations

from ...utils.misc import ensure_tuple
from .array import GaussianSmooth, NormalizeIntensity, ScaleIntensity, ScaleIntensityRange

__all__ = ["GaussianSmoothd", "GaussianSmoothD", "GaussianSmoothDict", "ScaleIntensityRanged", "ScaleIntensityRangeD", "ScaleIntensityRangeDict",
           "NormalizeIntensityd", "NormalizeIntensityD", "NormalizeIntensityDict", "ScaleIntensityd", "ScaleIntensityD", "ScaleIntensityDict"]


class GaussianSmoothd:
    def __init__(self, keys, sigma=1.0, approx: str = "erf", allow_missing_keys: bool = False) -> None:
        self.keys = ensure_tuple(keys)
        self.allow_missing_keys = allow_missing_keys
        self.converter = GaussianSmooth(sigma, approx=approx)

    def __call__(self, data):
        d = dict(data)
        for key in self.keys:
            if key not in d:
                if self.allow_missing_keys:
                    continue
                raise KeyError(f"Key `{key}` of transform `{type(self).__name__}` was missing in the data and allow_missing_keys==False.")
            d[key] = self.converter(d[key])
        return d


GaussianSmoothD = GaussianSmoothDict = GaussianSmoothd


class ScaleIntensityRanged(GaussianSmoothd):
    """Dictionary version of :class:`ScaleIntensityRange` (same key handling as the class above)."""

    def __init__(self, keys, a_min: float, a_max: float, b_min=None, b_max=None, clip: bool = False, dtype="float32", allow_missing_keys: bool = False) -> None:

        self.keys = ensure_tuple(keys)
        self.allow_missing_keys = allow_missing_keys
        self.scaler = self.converter = ScaleIntensityRange(a_min, a_max, b_min, b_max, clip, dtype)


ScaleIntensityRangeD = ScaleIntensityRangeDict = ScaleIntensityRanged


class NormalizeIntensityd(GaussianSmoothd):
    """Dictionary version of :class:`NormalizeIntensity` (monai/transforms/intensity/dictionary.py:782-820)."""

    def __init__(self, keys, subtrahend=None, divisor=None, nonzero: bool = False, channel_wise: bool = False, dtype="float32",
                 allow_missing_keys: bool = False) -> None:
        self.keys = ensure_tuple(keys)
        self.allow_missing_keys = allow_missing_keys
        self.normalizer = self.converter = NormalizeIntensity(subtrahend, divisor, nonzero, channel_wise, dtype)


NormalizeIntensityD = NormalizeIntensityDict = NormalizeIntensityd


class ScaleIntensityd(GaussianSmoothd):
    """Dictionary version of :class:`ScaleIntensity` (monai/transforms/intensity/dictionary.py:547-590)."""

    def __init__(self, keys, minv=0.0, maxv=1.0, factor=None, channel_wise: bool = False, dtype="float32", allow_missing_keys: bool = False) -> None:
        self.keys = ensure_tuple(keys)
        self.allow_missing_keys = allow_missing_keys
        self.scaler = self.converter = ScaleIntensity(minv, maxv, factor, channel_wise, dtype)


ScaleIntensityD = ScaleIntensityDict = ScaleIntensityd
