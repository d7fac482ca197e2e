"""Golden outputs of the reference's SpatialPad / BorderPad / DivisiblePad / SpatialCrop / CenterSpatialCrop (+ dictionary versions,
monai/transforms/croppad/{array,dictionary}.py) on the cases of tests/croppad_cases.py, CPU.
Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_croppad.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")
import monai.transforms as ref  # noqa: E402
from monai.data import MetaTensor  # noqa: E402
from croppad_cases import run_all  # noqa: E402

out = {k: np.asarray(v) for k, v in run_all(ref, "cpu", lambda t, a: MetaTensor(t, affine=a)).items()}
np.savez_compressed(os.path.join(HERE, "croppad.npz"), **out)
print("croppad golden:", len(out), "arrays", {k: v.shape for k, v in out.items() if "__" not in k})
