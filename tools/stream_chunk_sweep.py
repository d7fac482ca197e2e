"""Development aid: time the streaming resample / Gaussian kernels on one 512^3 volume (BASELINE configs[4] shapes) with
the launcher's own z-chunk rule or the count forced through MONAI_AMD_RS_CHUNKS / MONAI_AMD_GS_CHUNKS."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monai_amd import ops  # noqa: E402
from monai_amd.networks.layers import gaussian_1d  # noqa: E402


def timeit(fn, iters=20, warm=3):
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


dev = torch.device("cuda")
torch.manual_seed(0)
raw = torch.rand(1, 512, 512, 512, device=dev)
m = np.array([[1.25, 0, 0, 0], [0, 1.25, 0, 0], [0, 0, 0.625, 0]], dtype=np.float64).reshape(-1)
osz = (410, 410, 819)
k1, k2 = gaussian_1d(1.0).numpy(), gaussian_1d(2.0).numpy()
row = {"rs_chunks": os.environ.get("MONAI_AMD_RS_CHUNKS", "auto"), "gs_chunks": os.environ.get("MONAI_AMD_GS_CHUNKS", "auto")}
row["resample_f64_ms"] = round(timeit(lambda: ops.affine_resample(raw, m, osz, "bilinear", "border", False, True)), 4)
row["resample_f32_ms"] = round(timeit(lambda: ops.affine_resample(raw, m, osz, "bilinear", "border", False, False)), 4)
row["gauss9_ms"] = round(timeit(lambda: ops.separable_filter3d(raw, [k1, k1, k1])), 4)
row["gauss17_ms"] = round(timeit(lambda: ops.separable_filter3d(raw, [k2, k2, k2])), 4)
a = torch.empty(4, 512, 512, 512, device=dev)
b = torch.empty_like(a)
ms = timeit(lambda: b.copy_(a), iters=5)
row["device_copy_GBps"] = round(8.0 * a.numel() / ms / 1e6, 1)
print(json.dumps(row))
