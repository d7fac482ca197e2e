"""BASELINE.json configs[4]: Spacingd trilinear resample (affine diag(.8,.8,1.6) -> pixdim 1: 512^3 -> 410x410x819) and
GaussianSmooth(sigma=1) on a batch of 4 x 512^3 volumes resident in HBM -- the transform-only HBM-roofline run.
Prints one JSON document with per-kernel times and algorithmic GB/s (SURVEY.md 8d byte counts)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monai_amd.data import MetaTensor  # noqa: E402
from monai_amd.transforms import GaussianSmooth, Spacing  # noqa: E402

dev = torch.device("cuda")
N = int(os.environ.get("TB_BATCH", "4"))
E = int(os.environ.get("TB_EDGE", "512"))


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


aff = np.diag([0.8, 0.8, 1.6, 1.0])
vols = []
for s in range(N):
    torch.manual_seed(s)
    vols.append(MetaTensor(torch.rand(1, E, E, E).to(dev), affine=aff))
res = {"batch": N, "edge": E, "runs": []}
for dt, name in ((np.float64, "fp64 interpolation (reference default)"), (np.float32, "fp32 interpolation")):
    sp = Spacing(pixdim=(1.0, 1.0, 1.0), mode="bilinear", padding_mode="border", dtype=dt)
    out = sp(vols[0])
    ms = timeit(lambda: [sp(v) for v in vols])
    nbytes = 4.0 * N * (E ** 3 + out.numel())
    res["runs"].append({"op": f"Spacing bilinear/border {name}", "out_shape": list(out.shape), "ms": ms, "GBps": nbytes / ms / 1e6, "bytes": nbytes})
gs = GaussianSmooth(sigma=1.0)
plain = [v.as_tensor() for v in vols]
ms = timeit(lambda: [gs(v) for v in plain])
nbytes = 4.0 * N * 2 * E ** 3
res["runs"].append({"op": "GaussianSmooth sigma=1 (9 taps/axis, fused 3-axis pass)", "ms": ms, "GBps": nbytes / ms / 1e6, "bytes": nbytes})
print(json.dumps(res, indent=1))
