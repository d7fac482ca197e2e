"""A/B of the streaming HBM kernels of BASELINE.json configs[4] between two builds of the library (MONAI_AMD_LIB names the one a process loads):
kernel-only times of the separable resample (512^3 -> 410 x 410 x 819, fp64 and fp32 interpolation) and the fused 9-tap Gaussian on one 512^3 volume,
with a wrapping-integer digest of every output so that two builds can be compared bit for bit.  One JSON line."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monai_amd import _lib, ops  # noqa: E402
from monai_amd.networks.layers import gaussian_1d  # noqa: E402

dev = torch.device("cuda")
E = int(os.environ.get("TB_EDGE", "512"))
ITERS = int(os.environ.get("TB_ITERS", "20"))


def timeit(fn, iters=ITERS, warm=3):
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


def digest(t):
    return int(t.contiguous().view(torch.int32).sum(dtype=torch.int32).item())


torch.manual_seed(0)
raw = torch.rand(1, E, E, E, device=dev)
m = np.array([[1.25, 0, 0, 0], [0, 1.25, 0, 0], [0, 0, 0.625, 0]], dtype=np.float64)
osz = (int(E * 0.8 + 0.5), int(E * 0.8 + 0.5), int(E * 1.6 + 0.5))
res = {"lib": os.path.basename(_lib.LIB_PATH), "edge": E, "runs": []}
for f64 in (True, False):
    out = ops.affine_resample(raw, m.reshape(-1), osz, "bilinear", "border", False, f64)
    ms = timeit(lambda: ops.affine_resample(raw, m.reshape(-1), osz, "bilinear", "border", False, f64))
    nb = 4.0 * (raw.numel() + out.numel())
    res["runs"].append({"op": f"separable resample {'fp64' if f64 else 'fp32'}", "ms": round(ms, 4), "frac_of_8TBps": round(nb / ms / 1e6 / 8000.0, 3), "digest": digest(out)})
for sigma, name in ((1.0, "9 taps"), (0.5, "5 taps")):
    k = gaussian_1d(sigma).numpy()
    out = ops.separable_filter3d(raw, [k, k, k])
    ms = timeit(lambda: ops.separable_filter3d(raw, [k, k, k]))
    res["runs"].append({"op": f"gaussian {name} ({len(k)})", "ms": round(ms, 4), "frac_of_8TBps": round(8.0 * raw.numel() / ms / 1e6 / 8000.0, 3), "digest": digest(out)})
a = torch.empty(1, E, E, E, device=dev)
b = torch.empty_like(a)
ms = timeit(lambda: b.copy_(a))
res["device_copy_frac_of_8TBps"] = round(8.0 * a.numel() / ms / 1e6 / 8000.0, 3)
print(json.dumps(res))
