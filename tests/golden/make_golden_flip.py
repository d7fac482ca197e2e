"""Golden outputs of the reference's Flip / Rotate90 (+ dictionary versions, monai/transforms/spatial/array.py:665-718, 1139-1200) on the
cases of tests/flip_cases.py, CPU.  Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_flip.py"""
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
from flip_cases import run_all  # noqa: E402

out = {k: np.asarray(v) for k, v in run_all(ref, "cpu", lambda t, a: MetaTensor(t, affine=a)).items()}
np.savez_compressed(os.path.join(HERE, "flip_rotate.npz"), **out)
print("flip / rotate90 golden:", len(out), "arrays")
