from .array import BorderPad, CenterSpatialCrop, Crop, CropForeground, DivisiblePad, Pad, SpatialCrop, SpatialPad  # noqa: F401
from .dictionary import (  # noqa: F401
    BorderPadD, BorderPadDict, BorderPadd, CenterSpatialCropD, CenterSpatialCropDict, CenterSpatialCropd, CropForegroundD, CropForegroundDict,
    CropForegroundd, DivisiblePadD, DivisiblePadDict, DivisiblePadd, SpatialCropD, SpatialCropDict, SpatialCropd, SpatialPadD, SpatialPadDict,
    SpatialPadd,
)
