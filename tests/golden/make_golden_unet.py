"""Golden vectors for MONAI ``UNet`` from the REAL reference.  Build container only."""
import hashlib
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
from monai.networks.nets import UNet  # noqa: E402


def digest(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def main():
    out = {}
    cfgs = {
        "res2": dict(channels=(16, 32, 64, 128, 256), strides=(2, 2, 2, 2), num_res_units=2, shape=(2, 1, 32, 32, 32), seed=4),
        "plain": dict(channels=(8, 16, 32), strides=(2, 2), num_res_units=0, shape=(1, 1, 24, 16, 16), seed=5),
        "mixed": dict(channels=(8, 16, 32), strides=(2, 1), num_res_units=1, shape=(1, 1, 16, 12, 20), seed=6),
    }
    for name, c in cfgs.items():
        torch.manual_seed(c["seed"])
        net = UNet(spatial_dims=3, in_channels=1, out_channels=3, channels=c["channels"], strides=c["strides"], num_res_units=c["num_res_units"]).eval()
        with torch.no_grad():
            for k, v in net.state_dict().items():   # make the PReLU slopes distinguishable from the default
                if k.endswith("adn.A.weight"):
                    v.fill_(0.1 + 0.01 * (len(k) % 7))
        out[f"{name}_keys"] = np.asarray(list(net.state_dict().keys()))
        torch.manual_seed(c["seed"])
        fresh = UNet(spatial_dims=3, in_channels=1, out_channels=3, channels=c["channels"], strides=c["strides"], num_res_units=c["num_res_units"])
        out[f"{name}_init_sha256"] = np.asarray(digest(fresh.state_dict()))
        torch.manual_seed(100 + c["seed"])
        x = torch.rand(c["shape"])
        with torch.no_grad():
            out[f"{name}_out"] = net(x).numpy()
    np.savez_compressed(os.path.join(HERE, "unet.npz"), **out)
    print("unet golden written")


if __name__ == "__main__":
    main()
