from .array import CropForeground  # noqa: F401
from .dictionary import CropForegroundD, CropForegroundDict, CropForegroundd  # noqa: F401
