from .array import GaussianSmooth, NormalizeIntensity, ScaleIntensityRange  # noqa: F401
from .dictionary import (  # noqa: F401
    GaussianSmoothD, GaussianSmoothDict, GaussianSmoothd, NormalizeIntensityD, NormalizeIntensityDict, NormalizeIntensityd, ScaleIntensityRangeD,
    ScaleIntensityRangeDict, ScaleIntensityRanged,
)
