"""Golden outputs of the CT-bundle-shaped pipeline of tests/pipeline_ct_case.py, produced by the REAL reference classes on the CPU.
Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_pipeline_ct.py"""
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")
from monai.data import MetaTensor  # noqa: E402
from monai.inferers import SlidingWindowInferer  # noqa: E402
from monai.networks.nets import DynUNet  # noqa: E402
from monai.transforms import AsDiscreted, CropForegroundd, ScaleIntensityRanged, Spacingd  # noqa: E402
from dynunet_cases import build  # noqa: E402
from pipeline_ct_case import run_pipeline  # noqa: E402

net, _ = build(DynUNet, "basic")
ns = SimpleNamespace(MetaTensor=MetaTensor, ScaleIntensityRanged=ScaleIntensityRanged, CropForegroundd=CropForegroundd, Spacingd=Spacingd,
                     SlidingWindowInferer=SlidingWindowInferer, AsDiscreted=AsDiscreted)
out = run_pipeline(ns, net, "cpu")
np.savez_compressed(os.path.join(HERE, "pipeline_ct.npz"), **out)
print("ct pipeline golden:", {k: v.shape for k, v in out.items()}, out["crop_start"], out["crop_end"])
