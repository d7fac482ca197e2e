from .spatial import SpatialResample, Spacing, SpacingD, SpacingDict, Spacingd, spatial_resample  # noqa: F401
