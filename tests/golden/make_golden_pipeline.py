"""Golden outputs of the bundle-shaped pipeline of tests/pipeline_case.py through the REAL reference classes (CPU).
Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_pipeline.py"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")
from monai.data import MetaTensor  # noqa: E402
from monai.inferers import SlidingWindowInferer  # noqa: E402
from monai.networks.nets import BasicUNet  # noqa: E402
from monai.transforms import Activationsd, AsDiscreted, GaussianSmoothd, Spacingd  # noqa: E402
from pipeline_case import run_pipeline  # noqa: E402

torch.manual_seed(11)           # same seed and construction order as tests/e2e_cases.py:make_net
net = BasicUNet(3, 1, 3, features=(16, 16, 32, 32, 64, 16)).eval()
ns = SimpleNamespace(MetaTensor=MetaTensor, Spacingd=Spacingd, GaussianSmoothd=GaussianSmoothd, SlidingWindowInferer=SlidingWindowInferer,
                     Activationsd=Activationsd, AsDiscreted=AsDiscreted)
out = run_pipeline(ns, net, "cpu")
np.savez_compressed(os.path.join(HERE, "pipeline.npz"), **out)
print("pipeline golden:", {k: v.shape for k, v in out.items()})
