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
# kernel-only timings (no host-side affine algebra / MetaTensor bookkeeping)
from monai_amd import ops  # noqa: E402
from monai_amd.networks.layers import gaussian_1d  # noqa: E402

raw = vols[0].as_tensor()
m = np.array([[1.25, 0, 0, 0], [0, 1.25, 0, 0], [0, 0, 0.625, 0]], dtype=np.float64)
osz = tuple(int(v) for v in out.shape[1:])
for f64 in (True, False):
    for tag, mm in (("separable", m), ("general", m + np.array([[0, 1e-9, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]]))):
        ms = timeit(lambda: ops.affine_resample(raw, mm.reshape(-1), osz, "bilinear", "border", False, f64))
        nb = 4.0 * (raw.numel() + osz[0] * osz[1] * osz[2])
        res["runs"].append({"op": f"kernel affine_resample {tag} {'fp64' if f64 else 'fp32'} (1 volume)", "ms": ms, "GBps": nb / ms / 1e6, "bytes": nb})
# the general kernel's two forms (row-mapped with compile-time mode / padding = default, linear-index = round 1) on the same matrix, and
# on a matrix that really rotates (10 degrees in the (y, x) plane about the volume centre: gathers leave the coalesced pattern)
th = np.deg2rad(10.0)
cy, cx = 0.5 * (raw.shape[2] - 1), 0.5 * (raw.shape[3] - 1)
rot = np.array([[1.25, 0, 0, 0], [0, 1.25 * np.cos(th), -0.625 * np.sin(th), 0], [0, 1.25 * np.sin(th), 0.625 * np.cos(th), 0]], dtype=np.float64)
rot[1, 3] = cy - (rot[1, 1] * 0.5 * (osz[1] - 1) + rot[1, 2] * 0.5 * (osz[2] - 1))
rot[2, 3] = cx - (rot[2, 1] * 0.5 * (osz[1] - 1) + rot[2, 2] * 0.5 * (osz[2] - 1))
for f64 in (True, False):
    for tag, mm in (("general", m + np.array([[0, 1e-9, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]])), ("rotated 10 deg", rot)):
        for impl in ("rows", "linear"):
            os.environ["MONAI_AMD_RS_GENERAL"] = impl
            ms = timeit(lambda: ops.affine_resample(raw, mm.reshape(-1), osz, "bilinear", "border", False, f64))
            nb = 4.0 * (raw.numel() + osz[0] * osz[1] * osz[2])
            res["runs"].append({"op": f"kernel affine_resample {tag} {'fp64' if f64 else 'fp32'}, {impl} kernel (1 volume)", "ms": ms, "GBps": nb / ms / 1e6, "bytes": nb})
os.environ.pop("MONAI_AMD_RS_GENERAL", None)
k = gaussian_1d(1.0).numpy()
ms = timeit(lambda: ops.separable_filter3d(raw, [k, k, k]))
res["runs"].append({"op": "kernel separable_filter3d 9 taps (1 volume)", "ms": ms, "GBps": 8.0 * raw.numel() / ms / 1e6, "bytes": 8.0 * raw.numel()})
k2 = gaussian_1d(2.0).numpy()
ms = timeit(lambda: ops.separable_filter3d(raw, [k2, k2, k2]))
res["runs"].append({"op": "kernel separable_filter3d 17 taps (1 volume)", "ms": ms, "GBps": 8.0 * raw.numel() / ms / 1e6, "bytes": 8.0 * raw.numel()})
# lazy resampling: Orientationd -> Spacingd eagerly (two passes over HBM, the first one a flip) vs lazily (one fused pass)
from monai_amd.transforms import Orientationd, Spacingd  # noqa: E402
from monai_amd.transforms import lazy as L  # noqa: E402

lps = np.diag([-0.8, -0.8, 1.6, 1.0])
v0 = MetaTensor(vols[0].as_tensor(), affine=lps)
chain = [Orientationd("image", axcodes="RAS"), Spacingd("image", pixdim=(1.0, 1.0, 1.0), mode="bilinear", padding_mode="border")]


def eager_chain():
    d = {"image": v0}
    for t in chain:
        d = t(d)
    return d


def lazy_chain():
    d = {"image": v0}
    for t in chain:
        d = t(d, lazy=True)
    return L.apply_pending_transforms(d, ("image",))


o_e, o_l = eager_chain()["image"], lazy_chain()["image"]
nb1 = 4.0 * (v0.numel() + o_l.numel())
res["lazy_fusion"] = {
    "chain": "Orientationd(RAS) -> Spacingd(1 mm) on one 512^3 LPS volume", "out_shape": list(o_l.shape),
    "eager_ms": timeit(eager_chain), "lazy_ms": timeit(lazy_chain), "max_abs_diff_eager_vs_lazy": float((o_e.as_tensor() - o_l.as_tensor()).abs().max()),
    "algorithmic_bytes_one_pass": nb1,
}
res["lazy_fusion"]["lazy_GBps"] = nb1 / res["lazy_fusion"]["lazy_ms"] / 1e6
# what a plain device-to-device copy of the same bytes reaches on this box (the practical HBM ceiling: read + write streams)
a = torch.empty(N, E, E, E, device=dev)
b = torch.empty_like(a)
ms = timeit(lambda: b.copy_(a))
res["device_copy"] = {"bytes": 8.0 * a.numel(), "ms": ms, "GBps": 8.0 * a.numel() / ms / 1e6}
for r in res["runs"]:
    r["frac_of_8TBps"] = r["GBps"] / 8000.0
    r["frac_of_device_copy"] = r["GBps"] / res["device_copy"]["GBps"]
print(json.dumps(res, indent=1))
