from .spatial_transforms import AffineTransform  # noqa: F401
