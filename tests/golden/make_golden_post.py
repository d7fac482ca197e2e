"""Golden outputs of the reference's Activations / AsDiscrete (+ dictionary versions), monai/transforms/post/*.py, on the
cases of tests/post_cases.py, CPU.  Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_post.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")
import monai.transforms as ref  # noqa: E402
from post_cases import run_all  # noqa: E402

out = {k: np.asarray(v) for k, v in run_all(ref, "cpu").items()}
np.savez_compressed(os.path.join(HERE, "post_transforms.npz"), **out)
print("post_transforms golden:", len(out), "arrays")
