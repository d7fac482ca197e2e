"""A/B of the Gaussian smoothing kernels on the MI355X (development library: MONAI_AMD_LIB=.../libmonai_amd_dev.so, python -m monai_amd.build --dev).
Implementations by MONAI_AMD_GS_IMPL: t = round-1 tile kernel, v = round-2 row-vector kernel, default = forward accumulation + DPP x-pass
(profiles/r03_gauss_ab.json also holds two forms that were measured and removed: "l" = forward accumulation with the LDS x-pass, "8" = 8-row tiles).  Prints ms per 512^3 volume, the fraction of 8 TB/s (8 bytes per voxel), and whether the output equals the tile
kernel's bit for bit."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monai_amd import ops  # noqa: E402
from monai_amd.networks.layers import gaussian_1d  # noqa: E402


def run(impl, x, ks, reps=10):
    os.environ.pop("MONAI_AMD_GS_IMPL", None)
    if impl:
        os.environ["MONAI_AMD_GS_IMPL"] = impl
    y = ops.separable_filter3d(x, ks)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            y = ops.separable_filter3d(x, ks)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return y, best


def main():
    edge = int(os.environ.get("GA_EDGE", "512"))
    nvol = int(os.environ.get("GA_VOLS", "4"))
    out = {"volume": [nvol, edge, edge, edge], "runs": []}
    x = torch.rand((nvol, edge, edge, edge), device="cuda")
    for sigma in (1.0, 0.5):
        ks = [gaussian_1d(torch.tensor(sigma)).numpy()] * 3
        ref = None
        for impl in ("t", "v", ""):
            y, ms = run(impl, x, ks)
            if ref is None:
                ref = y
            per_vol = ms / nvol
            rec = {"sigma": sigma, "taps": len(ks[0]), "impl": impl or "default (dpp)", "ms_per_volume": round(per_vol, 4),
                   "frac_of_8TBps": round(edge ** 3 * 8 / (per_vol * 1e-3) / 8e12, 3), "bitwise_equal_to_tile": bool(torch.equal(y, ref))}
            out["runs"].append(rec)
            print(rec, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
