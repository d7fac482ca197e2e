"""Golden vectors for the 2-D ``BasicUNet`` and ``SliceInferer`` over it from the REAL reference, CPU.
Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_basic_unet2d.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
from monai.inferers import SliceInferer  # noqa: E402
from monai.networks.nets import BasicUNet  # noqa: E402
from e2e_cases import BASIC2D, BASIC2D_SLICE, basic2d_build, basic2d_input, basic2d_volume  # noqa: E402

out = {}
for name in BASIC2D:
    net, init = basic2d_build(BasicUNet, name)
    out[f"{name}_keys"] = np.asarray(list(net.state_dict().keys()))
    out[f"{name}_init_sha256"] = np.asarray(init)
    with torch.no_grad():
        out[f"{name}_out"] = net(basic2d_input(name)).numpy()
        if name == "plain":
            out[f"{name}_slice_out"] = SliceInferer(**BASIC2D_SLICE)(basic2d_volume(), net).numpy()
    print(name, out[f"{name}_out"].shape)
np.savez_compressed(os.path.join(HERE, "basic_unet2d.npz"), **out)
print("basic_unet2d golden written")
