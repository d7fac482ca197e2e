from .croppad import (  # noqa: F401
    BorderPad, BorderPadD, BorderPadDict, BorderPadd, CenterSpatialCrop, CenterSpatialCropD, CenterSpatialCropDict, CenterSpatialCropd, Crop,
    CropForeground, CropForegroundD, CropForegroundDict, CropForegroundd, DivisiblePad, DivisiblePadD, DivisiblePadDict, DivisiblePadd, Pad,
    SpatialCrop, SpatialCropD, SpatialCropDict, SpatialCropd, SpatialPad, SpatialPadD, SpatialPadDict, SpatialPadd,
)
from .intensity import (  # noqa: F401
    GaussianSmooth, GaussianSmoothD, GaussianSmoothDict, GaussianSmoothd, NormalizeIntensity, NormalizeIntensityD, NormalizeIntensityDict,
    NormalizeIntensityd, ScaleIntensity, ScaleIntensityD, ScaleIntensityDict, ScaleIntensityd, ScaleIntensityRange, ScaleIntensityRangeD, ScaleIntensityRangeDict,
    ScaleIntensityRanged,
)
from .spatial import Flip, FlipD, FlipDict, Flipd, Rotate90, Rotate90D, Rotate90Dict, Rotate90d  # noqa: F401
from .spatial import Orientation, OrientationD, OrientationDict, Orientationd, Resample, SpatialResample, Spacing, SpacingD, SpacingDict, Spacingd, spatial_resample  # noqa: F401
from .post import Activations, ActivationsD, ActivationsDict, Activationsd, AsDiscrete, AsDiscreteD, AsDiscreteDict, AsDiscreted  # noqa: F401
from .lazy import ApplyPending, ApplyPendingd, apply_pending, apply_pending_transforms  # noqa: F401
