from .array import GaussianSmooth, NormalizeIntensity, ScaleIntensity, ScaleIntensityRange  # noqa: F401
from .dictionary import (  # noqa: F401
    GaussianSmoothD, GaussianSmoothDict, GaussianSmoothd, NormalizeIntensityD, NormalizeIntensityDict, NormalizeIntensityd, ScaleIntensityD,
    ScaleIntensityDict, ScaleIntensityRangeD, ScaleIntensityRangeDict, ScaleIntensityRanged, ScaleIntensityd,
)
