"""Golden vectors for the 2-D ``DynUNet`` and ``SliceInferer`` over it from the REAL reference, CPU.
Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_dynunet2d.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")
from monai.inferers import SliceInferer  # noqa: E402
from monai.networks.nets import DynUNet  # noqa: E402
from dynunet_cases import CFGS_2D, SLICE, build, inputs, slice_volume  # noqa: E402

out = {}
for name in CFGS_2D:
    net, init = build(DynUNet, name)
    out[f"{name}_keys"] = np.asarray(list(net.state_dict().keys()))
    out[f"{name}_init_sha256"] = np.asarray(init)
    with torch.no_grad():
        out[f"{name}_out"] = net(inputs(name)).numpy()
        if name == "2d_basic":
            out[f"{name}_slice_out"] = SliceInferer(**SLICE)(slice_volume(), net).numpy()
    print(name, len(out[f"{name}_keys"]), "keys", out[f"{name}_out"].shape)
np.savez_compressed(os.path.join(HERE, "dynunet2d.npz"), **out)
print("dynunet2d golden written")
