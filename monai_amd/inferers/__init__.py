from .inferer import Inferer, SliceInferer, SlidingWindowInferer, SlidingWindowInfererAdapt  # noqa: F401
from .utils import sliding_window_inference  # noqa: F401
