from .simplelayers import GaussianFilter, gaussian_1d, separable_filtering  # noqa: F401
from .spatial_transforms import AffineTransform, grid_count, grid_grad, grid_pull, grid_push  # noqa: F401
