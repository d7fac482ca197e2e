"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Run only in the build container (the reference lives at /root/reference, which does not exist on
the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports ``monai`` from /root/reference (read-only; nothing is copied from it), runs the reference
classes on seeded inputs and stores inputs-by-seed + outputs as small ``.npz`` fixtures.  The oracle
(``oracle/``) and the HIP path are both tested against these files.
"""

from __future__ import annotations

import hashlib
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

import monai  # noqa: E402
from monai.data.utils import compute_importance_map, dense_patch_slices  # noqa: E402
from monai.inferers import SlidingWindowInferer, sliding_window_inference  # noqa: E402
from monai.inferers.utils import _get_scan_interval  # noqa: E402
from monai.networks.nets import BasicUNet  # noqa: E402


def sd_digest(sd) -> str:
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def toy_predictor(k_out):
    """Cheap deterministic stand-in network: K channels, each a different pointwise function."""

    def f(x):
        chans = [torch.sin(x[:, :1] * (1.0 + 0.37 * k)) + 0.05 * k * x[:, :1] for k in range(k_out)]
        return torch.cat(chans, dim=1)

    return f


def main():
    torch.set_num_threads(os.cpu_count())
    out = {}

    # ---- 1. host index math -------------------------------------------------------------
    cases = [
        ((512, 512, 512), (96, 96, 96), 0.5),
        ((64, 64, 64), (32, 32, 32), 0.5),
        ((40, 37, 29), (16, 16, 16), 0.25),
        ((125, 512, 200), (96, 97, 98), 0.25),
        ((7, 7), (3, 3), 0.5),
        ((33, 20, 21), (16, 20, 8), 0.6),
    ]
    for i, (img, roi, ov) in enumerate(cases):
        itv = _get_scan_interval(img, roi, len(img), (ov,) * len(img))
        sl = dense_patch_slices(img, roi, itv)
        out[f"slices_{i}_img"] = np.asarray(img)
        out[f"slices_{i}_roi"] = np.asarray(roi)
        out[f"slices_{i}_ov"] = np.asarray(ov)
        out[f"slices_{i}_interval"] = np.asarray(itv)
        out[f"slices_{i}_starts"] = np.asarray([[s.start for s in w] for w in sl], dtype=np.int32)
    for i, (ps, mode, sig) in enumerate(
        [((96, 96, 96), "gaussian", 0.125), ((32, 32, 32), "gaussian", 0.125), ((3, 3), "gaussian", 1.0),
         ((16, 20, 8), "gaussian", (0.125, 0.25, 0.5)), ((5, 4, 3), "constant", 0.125)]
    ):
        m = compute_importance_map(ps, mode=mode, sigma_scale=sig, device="cpu", dtype=torch.float32)
        out[f"imp_{i}_ps"] = np.asarray(ps)
        out[f"imp_{i}_mode"] = np.asarray(mode)
        out[f"imp_{i}_sigma"] = np.asarray(sig, dtype=np.float64)
        if np.prod(ps) <= 40000:
            out[f"imp_{i}_map"] = m.numpy()
        else:  # big maps: store a strided sample + an fp64 checksum
            out[f"imp_{i}_sample"] = m.numpy()[::7, ::5, ::3].copy()
            out[f"imp_{i}_sum"] = np.asarray(m.double().sum().item())
            out[f"imp_{i}_minmax"] = np.asarray([m.min().item(), m.max().item()])
    np.savez_compressed(os.path.join(HERE, "host_math.npz"), **out)

    # ---- 2. blend only (toy predictor), ragged 3-D sizes, both modes ---------------------
    out = {}
    blend_cases = [
        dict(shape=(1, 1, 40, 37, 29), roi=(16, 16, 16), ov=0.25, mode="gaussian", sw=4, k=3, seed=10),
        dict(shape=(2, 1, 33, 20, 21), roi=(16, 20, 8), ov=0.6, mode="constant", sw=3, k=2, seed=11),
        dict(shape=(1, 1, 48, 48, 48), roi=(32, 32, 32), ov=0.5, mode="gaussian", sw=4, k=5, seed=12),
        dict(shape=(1, 1, 20, 9, 12), roi=(24, 16, 16), ov=0.5, mode="gaussian", sw=2, k=2, seed=13),  # padded
        dict(shape=(1, 1, 31, 45), roi=(8, 16), ov=0.5, mode="gaussian", sw=5, k=4, seed=14),  # 2-D
    ]
    for i, c in enumerate(blend_cases):
        torch.manual_seed(c["seed"])
        x = torch.rand(c["shape"])
        y = sliding_window_inference(x, c["roi"], c["sw"], toy_predictor(c["k"]), overlap=c["ov"], mode=c["mode"],
                                     padding_mode="constant", cval=-0.5)
        for k, v in c.items():
            out[f"blend_{i}_{k}"] = np.asarray(v)
        out[f"blend_{i}_out"] = y.numpy()
    np.savez_compressed(os.path.join(HERE, "blend.npz"), **out)

    # ---- 3. config 0: BasicUNet(1->2) on rand 64^3, roi 32^3, sw 4, ov .5, gaussian ------
    out = {}
    torch.manual_seed(0)
    net = BasicUNet(spatial_dims=3, in_channels=1, out_channels=2).eval()
    x = torch.rand(1, 1, 64, 64, 64)
    with torch.no_grad():
        y = SlidingWindowInferer(roi_size=(32, 32, 32), sw_batch_size=4, overlap=0.5, mode="gaussian")(x, net)
    out["cfg0_state_sha256"] = np.asarray(sd_digest(net.state_dict()))
    out["cfg0_x_sum"] = np.asarray(x.double().sum().item())
    out["cfg0_out"] = y.numpy()
    np.savez_compressed(os.path.join(HERE, "config0.npz"), **out)

    # ---- 4. BasicUNet 5-class (the bench weights, seed 1): one window forward ------------
    out = {}
    torch.manual_seed(1)
    net = BasicUNet(spatial_dims=3, in_channels=1, out_channels=5).eval()
    out["net5_state_sha256"] = np.asarray(sd_digest(net.state_dict()))
    out["net5_keys"] = np.asarray(list(net.state_dict().keys()))
    out["net5_shapes"] = np.asarray([",".join(map(str, v.shape)) for v in net.state_dict().values()])
    torch.manual_seed(21)
    x = torch.rand(2, 1, 32, 32, 32)
    with torch.no_grad():
        out["net5_win32_out"] = net(x).numpy()
    torch.manual_seed(22)
    x = torch.rand(1, 1, 48, 32, 16)  # anisotropic window: distinguishes the three axes
    with torch.no_grad():
        out["net5_win48x32x16_out"] = net(x).numpy()
    # sliding window over a small volume with 5-class net: roi 32, ov 0.5 on 48x40x32
    torch.manual_seed(23)
    x = torch.rand(1, 1, 48, 40, 32)
    with torch.no_grad():
        y = SlidingWindowInferer(roi_size=(32, 32, 32), sw_batch_size=4, overlap=0.5, mode="gaussian")(x, net)
    out["net5_sw_out"] = y.numpy()
    np.savez_compressed(os.path.join(HERE, "net5.npz"), **out)
    print("monai", monai.__version__, "torch", torch.__version__, "golden vectors written to", HERE)


if __name__ == "__main__":
    main()
