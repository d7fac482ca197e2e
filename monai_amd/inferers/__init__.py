from .inferer import Inferer, PatchInferer, SliceInferer, SlidingWindowArgmaxInferer, SlidingWindowInferer, SlidingWindowInfererAdapt  # noqa: F401
from .merger import AvgMerger, Merger  # noqa: F401
from .splitter import SlidingWindowSplitter, Splitter  # noqa: F401
from .utils import sliding_window_argmax, sliding_window_inference  # noqa: F401
