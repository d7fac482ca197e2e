from .array import Resample, SpatialResample, Spacing  # noqa: F401
from .dictionary import SpacingD, SpacingDict, Spacingd  # noqa: F401
from .functional import spatial_resample  # noqa: F401
from .orientation import Orientation, OrientationD, OrientationDict, Orientationd  # noqa: F401
from .flip_rotate import Flip, FlipD, FlipDict, Flipd, Rotate90, Rotate90D, Rotate90Dict, Rotate90d  # noqa: F401
