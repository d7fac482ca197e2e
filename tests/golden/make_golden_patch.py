"""Golden outputs of the reference's PatchInferer + SlidingWindowSplitter + AvgMerger (monai/inferers/{inferer,splitter,merger}.py)
on the cases of tests/patch_cases.py, CPU.  Build container only:
PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_patch.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")
import monai.inferers as ref  # noqa: E402
from patch_cases import PATCH_CASES, run_case  # noqa: E402


def main():
    out = {}
    for k, case in enumerate(PATCH_CASES):
        for key, v in run_case(ref, k, case, "cpu").items():
            out[f"pi_{k}_{key}"] = v
    out["n"] = np.asarray(len(PATCH_CASES))
    np.savez_compressed(os.path.join(HERE, "patch_inferer.npz"), **out)
    print("patch_inferer golden:", len(PATCH_CASES), "cases")


if __name__ == "__main__":
    main()
