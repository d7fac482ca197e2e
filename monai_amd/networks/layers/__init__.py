from .simplelayers import GaussianFilter, gaussian_1d, separable_filtering  # noqa: F401
from .spatial_transforms import AffineTransform  # noqa: F401
