from .array import GaussianSmooth  # noqa: F401
from .dictionary import GaussianSmoothD, GaussianSmoothDict, GaussianSmoothd  # noqa: F401
