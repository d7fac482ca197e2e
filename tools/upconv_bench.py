"""UpCat's composite transposed convolution (csrc/kernels/upconv_h2.h) at the headline's top decoder level: 32 channels @ 48^3 -> 32 couts @ 96^3, 64 windows per launch.

    python tools/upconv_bench.py [--batch 64] [--reps 20]
    MONAI_AMD_LIB=/path/to/another/libmonai_amd.so python tools/upconv_bench.py      # the same measurement on another build of the C ABI (A/B on one box)

Prints one JSON line: median / min / max ms of the writing form, the accumulating form and the accumulating form with statistics, and a checksum of each result
(bit-identical builds give identical checksums)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monai_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--edge", type=int, default=48)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(7)
B, E = args.batch, args.edge
low = torch.randn(B, 32, E, E, E, device=dev)
ln = torch.zeros(B, 32, 4, device=dev)
ln[:, :, 0] = 1.1
ln[:, :, 1] = 0.1
ln[:, :, 2] = 0.1
ln[:, :, 3] = 8.0
w4, table = ops.upconv_k4s2_weights(torch.randn(32, 32, 2, 2, 2, device=dev) * 0.2, torch.randn(32, device=dev) * 0.1, torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05)
packed = ops.upconv_k4s2_pack(w4)
y = torch.empty(B, 32, 2 * E, 2 * E, 2 * E, device=dev)
tiles = ops.upconv_k4s2_stat_tiles(E, E, E)
stats = torch.empty(B * 32 * tiles * 3, device=dev)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return {"median_ms": round(ts[len(ts) // 2], 4), "min_ms": round(ts[0], 4), "max_ms": round(ts[-1], 4)}


def checksum(t):
    return float(t.double().sum().item()), float(t.double().abs().max().item())


out = {"batch": B, "edge": E, "lib": os.environ.get("MONAI_AMD_LIB", "product")}
out["write"] = timed(lambda: ops.upconv_k4s2(low, ln, packed, table, y, accumulate=False))
out["write_sum"] = checksum(y)
y.normal_()          # what the convolution's skip half left there
ops.upconv_k4s2(low, ln, packed, table, y, accumulate=True, stats=stats)
out["rmw_stats_sum"] = checksum(y)
out["stats_sum"] = float(stats.double().sum().item())
out["rmw"] = timed(lambda: ops.upconv_k4s2(low, ln, packed, table, y, accumulate=True))
out["rmw_stats"] = timed(lambda: ops.upconv_k4s2(low, ln, packed, table, y, accumulate=True, stats=stats))
print(json.dumps(out))
