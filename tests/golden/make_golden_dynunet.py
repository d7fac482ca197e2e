"""Golden vectors for ``DynUNet`` from the REAL reference (monai/networks/nets/dynunet.py), CPU.
Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_dynunet.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")
from monai.networks.nets import DynUNet  # noqa: E402
from monai.inferers import SlidingWindowInferer  # noqa: E402
from dynunet_cases import CFGS, CFGS_2D, SW, build, inputs, sw_volume  # noqa: E402

out = {}
for name in (n for n in CFGS if n not in CFGS_2D):       # the 2-D configurations: make_golden_dynunet2d.py
    net, init = build(DynUNet, name)
    out[f"{name}_keys"] = np.asarray(list(net.state_dict().keys()))
    out[f"{name}_init_sha256"] = np.asarray(init)
    with torch.no_grad():
        out[f"{name}_out"] = net(inputs(name)).numpy()
        if name in ("basic", "stride0"):
            out[f"{name}_sw_out"] = SlidingWindowInferer(**SW)(sw_volume(), net).numpy()
    print(name, len(out[f"{name}_keys"]), "keys", out[f"{name}_out"].shape)
np.savez_compressed(os.path.join(HERE, "dynunet.npz"), **out)
print("dynunet golden written")
