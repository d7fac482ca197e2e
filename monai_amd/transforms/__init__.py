from .spatial import Resample, SpatialResample, Spacing, SpacingD, SpacingDict, Spacingd, spatial_resample  # noqa: F401
