"""Golden outputs of the reference's ScaleIntensityRange(d) / CropForeground(d) (monai/transforms/intensity/array.py:958-1012,
monai/transforms/croppad/array.py:776-960) on the cases of tests/preproc_cases.py, CPU.
Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_preproc.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")
import monai.transforms as ref  # noqa: E402
from monai.data import MetaTensor  # noqa: E402
from preproc_cases import run_all  # noqa: E402

out = {k: np.asarray(v) for k, v in run_all(ref, "cpu", lambda t, a: MetaTensor(t, affine=a)).items()}
np.savez_compressed(os.path.join(HERE, "preproc.npz"), **out)
for k, v in out.items():
    if k.endswith(("start", "end")):
        print(k, v.dtype, v.tolist())
print("preproc golden:", len(out), "arrays")
