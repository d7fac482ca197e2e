"""Golden vectors for ``SegResNet`` from the REAL reference (monai/networks/nets/segresnet.py), CPU.
Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_segresnet.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")
from monai.inferers import SlidingWindowInferer  # noqa: E402
from monai.networks.nets import SegResNet  # noqa: E402
from dynunet_cases import SW, sw_volume  # noqa: E402
from segresnet_cases import CFGS, build, inputs  # noqa: E402

out = {}
for name in CFGS:
    net, init = build(SegResNet, name)
    out[f"{name}_keys"] = np.asarray(list(net.state_dict().keys()))
    out[f"{name}_init_sha256"] = np.asarray(init)
    with torch.no_grad():
        out[f"{name}_out"] = net(inputs(name)).numpy()
        if name == "default":
            out["default_sw_out"] = SlidingWindowInferer(**SW)(sw_volume(), net).numpy()
    print(name, len(out[f"{name}_keys"]), "keys", out[f"{name}_out"].shape)
np.savez_compressed(os.path.join(HERE, "segresnet.npz"), **out)
print("segresnet golden written")
