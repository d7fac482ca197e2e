"""Pre-processing kernels in front of the path (DESIGN_HISTORY.md 4.4) on one 512^3 CT-like volume resident in HBM: ScaleIntensityRange,
the CropForeground box + crop, Orientation (pure flips and a real axis permutation).  Prints one JSON document with per-op times and
algorithmic GB/s (the byte counts of DESIGN_HISTORY.md section 4's table) next to a device-to-device copy of the same volume."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monai_amd import ops  # noqa: E402
from monai_amd.data import MetaTensor  # noqa: E402
from monai_amd.transforms import CropForeground, NormalizeIntensity, Orientation, ScaleIntensity, ScaleIntensityRange  # noqa: E402

dev = torch.device("cuda")
E = int(os.environ.get("PB_EDGE", "512"))


def timeit(fn, iters=10, warm=3):
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


torch.manual_seed(0)
x = torch.full((1, E, E, E), -1000.0, device=dev)
lo, hi = E // 8, E - E // 8
x[:, lo:hi, lo:hi, lo:hi] = torch.rand((1, hi - lo, hi - lo, hi - lo), device=dev) * 500.0 - 200.0
vol = 4.0 * E ** 3
res = {"edge": E, "runs": []}


def add(op, ms, nbytes, **kw):
    res["runs"].append({"op": op, "ms": ms, "GBps": nbytes / ms / 1e6, "bytes": nbytes, **kw})


y = torch.empty_like(x)
add("device copy (read + write)", timeit(lambda: y.copy_(x)), 2 * vol)
sc = ScaleIntensityRange(-175.0, 250.0, 0.0, 1.0, clip=True)
add("ScaleIntensityRange (transform call)", timeit(lambda: sc(x)), 2 * vol)
add("scale_range_kernel", timeit(lambda: ops.scale_intensity_range(x, -175.0, 425.0, 1.0, 0.0, 0.0, 1.0)), 2 * vol)
si = ScaleIntensity(minv=0.0, maxv=1.0)
add("ScaleIntensity (min / max + fold + apply)", timeit(lambda: si(x)), 3 * vol)
nz = NormalizeIntensity(nonzero=True, channel_wise=True)
add("NormalizeIntensity nonzero channel_wise (stats + fold + apply)", timeit(lambda: nz(x)), 3 * vol)
add("masked_stats_kernel + fold", timeit(lambda: ops.normalize_stats(x, 1, x.numel(), True)), vol)
s = sc(x)
add("foreground box (two kernels + 24-byte read back)", timeit(lambda: ops.foreground_bbox(s)), vol, box=list(ops.foreground_bbox(s)))
cf = CropForeground(margin=4)
out = cf(s)
add("CropForeground (box + crop)", timeit(lambda: cf(s)), vol + 2 * 4.0 * out.numel(), out_shape=list(out.shape))
add("crop_pad_kernel", timeit(lambda: ops.crop_pad(s, [lo - 4] * 3, list(out.shape[1:]), 0.0)), 2 * 4.0 * out.numel())
m = MetaTensor(x, affine=np.diag([-0.8, -0.8, 1.6, 1.0]))                      # LPS -> RAS: two flips, no permutation
o1 = Orientation(axcodes="RAS")
add("Orientation LPS->RAS (flip z, y)", timeit(lambda: o1(m)), 2 * vol)
add("flip_permute_kernel flip x", timeit(lambda: ops.flip_permute(x, [0, 1, 2], [False, False, True])), 2 * vol)
add("flip_permute_kernel permute (x, y, z)", timeit(lambda: ops.flip_permute(x, [2, 1, 0], [False, False, False])), 2 * vol)
print(json.dumps(res))
