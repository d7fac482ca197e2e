from .array import GaussianSmooth, ScaleIntensityRange  # noqa: F401
from .dictionary import (  # noqa: F401
    GaussianSmoothD, GaussianSmoothDict, GaussianSmoothd, ScaleIntensityRangeD, ScaleIntensityRangeDict, ScaleIntensityRanged,
)
