"""Golden vectors for MONAI ``UNet`` with other activations and `adn_ordering` values (tests/e2e_cases.py: UNET_VARIANTS) from the REAL reference.  Build container only."""
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
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))      # repo root: tests/e2e_cases.py imports the oracle package
    sys.path.insert(0, os.path.dirname(HERE))
    from e2e_cases import UNET_VARIANTS as UNET_CFGS, perturb_unet, unet_kwargs

    out = {}
    for name, c in UNET_CFGS.items():
        torch.manual_seed(c["seed"])
        net = perturb_unet(UNet(**unet_kwargs(name)), name).eval()
        out[f"{name}_keys"] = np.asarray(list(net.state_dict().keys()))
        torch.manual_seed(c["seed"])
        out[f"{name}_init_sha256"] = np.asarray(digest(UNet(**unet_kwargs(name)).state_dict()))
        torch.manual_seed(100 + c["seed"])
        x = torch.rand(c["shape"])
        with torch.no_grad():
            out[f"{name}_out"] = net(x).numpy()
    np.savez_compressed(os.path.join(HERE, "unet_variants.npz"), **out)
    print("unet variants golden written")


if __name__ == "__main__":
    main()
