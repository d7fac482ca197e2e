"""Golden vectors for the USE_COMPILED branch of ``Resample`` (monai/transforms/spatial/array.py:2076-2092): the real
reference transform, imported from /root/reference with ``BUILD_MONAI=1`` and its native module ``monai._C`` provided
by the reference's own C++ sources compiled for the CPU (oracle/_ref, oracle/build_ref.py).
Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_resample_compiled.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["BUILD_MONAI"] = "1"
from oracle import build_ref  # noqa: E402

build_ref.build()
sys.modules["monai._C"] = build_ref.load()          # before `import monai`: optional_import("monai._C") then finds it
sys.path.insert(0, "/root/reference")
import monai  # noqa: E402
from monai.transforms import Resample  # noqa: E402
from transform_cases import RC_CASES, WARP_CASES, rc_inputs, warp_inputs  # noqa: E402

assert monai.config.USE_COMPILED and monai.transforms.spatial.array.USE_COMPILED


def main():
    out = {}
    for k, case in enumerate(RC_CASES):
        img, grid = rc_inputs(case)
        tr = Resample(mode=case["mode"], padding_mode=case["padding_mode"], norm_coords=case["norm_coords"],
                      align_corners=case["align_corners"], dtype=case["dtype"])
        y = tr(img, grid)
        out[f"rc_{k}"] = np.asarray(y)
    out["rc_n"] = np.asarray(len(RC_CASES))
    # Warp / DVF2DDF (monai/networks/blocks/warp.py) in both build modes; the module reads USE_COMPILED at call time
    import warnings

    import monai.networks.blocks.warp as W

    for k, case in enumerate(WARP_CASES):
        image, ddf = warp_inputs(case)
        W.USE_COMPILED = bool(case["compiled"])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            layer = W.DVF2DDF(num_steps=3, mode=case["mode"], padding_mode=case["padding_mode"]) if case["dvf"] else W.Warp(mode=case["mode"], padding_mode=case["padding_mode"])
        y = layer(ddf) if case["dvf"] else layer(image, ddf)
        out[f"warp_{k}"] = y.detach().numpy()
    W.USE_COMPILED = True
    out["warp_n"] = np.asarray(len(WARP_CASES))
    np.savez_compressed(os.path.join(HERE, "resample_compiled.npz"), **out)
    print("resample_compiled golden:", len(RC_CASES), "Resample cases,", len(WARP_CASES), "Warp cases")


if __name__ == "__main__":
    main()
