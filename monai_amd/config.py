"""Build flags mirroring monai/config/deviceconfig.py:29-34.

``HAS_EXT``: the native module is available (here: the gfx950 library behind ``monai_amd._C``).  ``USE_COMPILED``: route
resampling transforms through the native ``grid_pull`` instead of ``F.grid_sample`` -- in the reference
``HAS_EXT and os.getenv("BUILD_MONAI", "0") == "1"``; the same rule here.  Transforms read the attribute at call time, so
it can be switched at run time (``monai_amd.config.USE_COMPILED = True``)."""
import os

HAS_EXT = True
USE_COMPILED = HAS_EXT and os.getenv("BUILD_MONAI", "0") == "1"
