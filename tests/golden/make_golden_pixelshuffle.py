"""Golden vectors for BasicUNet(upsample="pixelshuffle") -- UpSample -> SubpixelUpsample (monai/networks/blocks/upsample.py:186-288: k3 convolution to 8 x the channels,
pixelshuffle of monai/networks/utils.py:370-412, zero pad in front + average pooling) -- from the REAL reference: its parameters (ICNR-initialised sub-pixel convolutions,
trained-looking norm affines), two inputs (even extents; odd extents at a lower level: UpCat's replicate padding behind the shuffle) and its logits.  Build container only."""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
from monai.networks.nets import BasicUNet  # noqa: E402

FEATURES = (16, 16, 16, 16, 32, 16)


def main():
    torch.manual_seed(31)
    net = BasicUNet(spatial_dims=3, in_channels=1, out_channels=3, features=FEATURES, upsample="pixelshuffle").eval()
    gen = torch.Generator().manual_seed(32)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.endswith("adn.N.weight"):
                p.copy_(torch.rand(p.shape, generator=gen) + 0.5)
            elif name.endswith("adn.N.bias"):
                p.copy_(torch.randn(p.shape, generator=gen) * 0.2)
    sd = net.state_dict()
    out = {"keys": np.array(list(sd.keys())), "features": np.array(FEATURES)}
    for k, v in sd.items():
        out["p:" + k] = v.numpy()
    xs = {"even": torch.rand((1, 1, 32, 32, 48), generator=gen), "odd": torch.rand((2, 1, 16, 24, 40), generator=gen)}      # 24: 3 at level 3; 40: 5 at level 3
    with torch.no_grad():
        for name, x in xs.items():
            out["x_" + name] = x.numpy()
            out["y_" + name] = net(x).numpy()
    # two spatial dimensions (the engine's one-plane form): 4 sub-pixels per channel; 40: 5 at level 3 (odd)
    torch.manual_seed(33)
    net2 = BasicUNet(spatial_dims=2, in_channels=2, out_channels=3, features=FEATURES, upsample="pixelshuffle").eval()
    sd2 = net2.state_dict()
    out["keys2"] = np.array(list(sd2.keys()))
    for k, v in sd2.items():
        out["p2:" + k] = v.numpy()
    x2 = torch.rand((2, 2, 48, 40), generator=gen)
    with torch.no_grad():
        out["x_2d"] = x2.numpy()
        out["y_2d"] = net2(x2).numpy()
    np.savez_compressed(os.path.join(HERE, "basic_unet_pixelshuffle.npz"), **out)
    print("pixelshuffle golden written", {k: tuple(v.shape) for k, v in out.items() if k.startswith("y_")})


if __name__ == "__main__":
    main()
