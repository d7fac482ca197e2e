from .inferer import Inferer, SlidingWindowInferer  # noqa: F401
from .utils import sliding_window_inference  # noqa: F401
