"""Does the x-extent of a workgroup's tile relative to the row length matter to the z-marching HBM kernels?  The fused 9-tap Gaussian (tile 16 rows x 256 columns)
on volumes of the SAME byte count whose rows are 256, 512, 1024 and 2048 floats long: with W = 256 a tile row is a whole image row (a plane's 24 halo rows are one
contiguous 24 KB piece), with W = 2048 it is one eighth of a row (1 KB pieces, 8 KB apart).  Likewise the separable resample (x scale 0.625, tile 128 output columns
= 83 source floats).  One JSON line."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monai_amd import ops  # noqa: E402
from monai_amd.networks.layers import gaussian_1d  # noqa: E402

dev = torch.device("cuda")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters)
    return best


res = {"gaussian_9_taps": [], "separable_resample_fp32": []}
k = gaussian_1d(1.0).numpy()
m = np.array([[1.25, 0, 0, 0], [0, 1.25, 0, 0], [0, 0, 0.625, 0]], dtype=np.float64)
for shape in ((1024, 512, 256), (512, 512, 512), (512, 256, 1024), (256, 256, 2048), (2048, 256, 256)):
    raw = torch.rand(1, *shape, device=dev)
    ms = timeit(lambda: ops.separable_filter3d(raw, [k, k, k]))
    res["gaussian_9_taps"].append({"shape": list(shape), "ms": round(ms, 4), "frac_of_8TBps": round(8.0 * raw.numel() / ms / 1e6 / 8000.0, 3)})
    osz = (int(shape[0] * 0.8 + 0.5), int(shape[1] * 0.8 + 0.5), int(shape[2] * 1.6 + 0.5))
    out = ops.affine_resample(raw, m.reshape(-1), osz, "bilinear", "border", False, False)
    ms = timeit(lambda: ops.affine_resample(raw, m.reshape(-1), osz, "bilinear", "border", False, False))
    res["separable_resample_fp32"].append({"shape": list(shape), "out": list(osz), "ms": round(ms, 4), "frac_of_8TBps": round(4.0 * (raw.numel() + out.numel()) / ms / 1e6 / 8000.0, 3)})
    del raw, out
print(json.dumps(res))
