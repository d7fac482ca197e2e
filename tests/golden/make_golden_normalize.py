"""Golden outputs of the reference's NormalizeIntensity(d) (monai/transforms/intensity/array.py:816-907) on the cases of
tests/normalize_cases.py, and of the MRI-bundle-shaped pipeline of the same file through the REAL reference classes, CPU.
Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_normalize.py"""
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")
import monai.transforms as ref  # noqa: E402
from monai.inferers import SlidingWindowInferer  # noqa: E402
from monai.networks.nets import SegResNet  # noqa: E402
from normalize_cases import make_net, run_all, run_pipeline, run_scale_intensity  # noqa: E402

out = {k: np.asarray(v) for k, v in run_all(ref, "cpu").items()}
np.savez_compressed(os.path.join(HERE, "normalize.npz"), **out)
print("normalize golden:", len(out), "arrays")
si = {k: np.asarray(v) for k, v in run_scale_intensity(ref, "cpu").items()}
np.savez_compressed(os.path.join(HERE, "scale_intensity.npz"), **si)
print("scale_intensity golden:", len(si), "arrays")
ns = SimpleNamespace(NormalizeIntensityd=ref.NormalizeIntensityd, SlidingWindowInferer=SlidingWindowInferer, Activationsd=ref.Activationsd,
                     AsDiscreted=ref.AsDiscreted)
pipe = run_pipeline(ns, make_net(SegResNet), "cpu")
np.savez_compressed(os.path.join(HERE, "pipeline_mri.npz"), **pipe)
print("mri pipeline golden:", {k: v.shape for k, v in pipe.items()})
