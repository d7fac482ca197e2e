"""monai_amd -- MI355X-native (gfx950) implementation of MONAI's 3-D sliding-window segmentation hot path.

Public names mirror the reference (`monai.inferers.SlidingWindowInferer`, `monai.networks.nets.BasicUNet`,
...); the arithmetic runs in hand-written HIP kernels behind the C ABI of ``include/monai_amd.h``.
"""

__version__ = "0.1.0"

from . import _fallback as _fallback  # noqa: E402

_fallback.apply_all()      # every class falls through to the reference object of the same name for calls outside the HIP path (B3)
