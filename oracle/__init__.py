"""CPU oracle for the MONAI sliding-window segmentation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``monai_amd/`` may import this package: it is the
checker for the HIP path, never the thing measured or shipped.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it.

What it is: a restatement, in plain torch-CPU / numpy, of the reference algorithm for every row of
SURVEY.md section 8(a).  All floating-point arithmetic on this path in the reference is PyTorch
ATen, so the restatement calls the *same* ATen CPU operators in the *same* order
(``F.conv3d``, ``F.instance_norm``, ``F.leaky_relu``, ``F.max_pool3d``, ``F.conv_transpose3d``,
in-place ``*=`` / ``+=`` / ``/=`` for the blend) and re-derives the host-side index math
(window starts, importance map, scan interval) from the reference source, cited per function.

Parity pinning: ``tests/golden/make_golden.py`` imports the real reference from
``/root/reference`` (only in the build container), runs it on seeded inputs and stores its outputs
under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this oracle against those vectors
(bit-exact for the blend and the network, since both sides run the same ATen kernels) and against
the known-answer tables of the reference's own unit tests (restated with file:line citations).
"""

from .sliding_window import (  # noqa: F401
    compute_importance_map,
    dense_patch_starts,
    get_scan_interval,
    sliding_window_inference,
)
from .basic_unet import basic_unet_forward, make_basic_unet_state  # noqa: F401
from .parity import assert_label_parity, label_parity  # noqa: F401
