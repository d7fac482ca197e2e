"""Golden vector for BasicUNet on a window with ODD extents at three levels (UpCat's replicate padding,
monai/networks/nets/basic_unet.py:163-170) from the REAL reference.  Build container only."""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
from monai.networks.nets import BasicUNet  # noqa: E402


def main():
    torch.manual_seed(1)
    net = BasicUNet(spatial_dims=3, in_channels=1, out_channels=5).eval()
    torch.manual_seed(23)
    x = torch.rand(1, 1, 40, 36, 34)      # 40: 5 at level 3; 36: 9 at level 2; 34: 17 at level 1
    with torch.no_grad():
        y = net(x)
    np.savez_compressed(os.path.join(HERE, "net5_odd.npz"), out=y.numpy())
    print("net5_odd golden written", tuple(y.shape))


if __name__ == "__main__":
    main()
