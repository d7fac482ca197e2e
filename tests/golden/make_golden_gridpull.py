"""Golden vectors for ``grid_pull`` (monai._C) from (a) the reference's own compiled CPU resampler
(oracle/_ref, built by oracle/build_ref.py from the sources under /root/reference) on seeded inputs and (b) the
reference's golden file tests/testing_data/1D_BP_fwd.txt (rows for interpolation orders 0 and 1).
Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_gridpull.py"""
import os
import re
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402

BOUNDS = {"replicate": 0, "dct1": 1, "dct2": 2, "dst1": 3, "dst2": 4, "dft": 5, "zero": 7}


def main():
    build_ref.build()
    ref = build_ref.load()
    out = {}
    k = 0
    for seed, ishape, oshape in [(1, (2, 2, 7, 6, 5), (4, 5, 6)), (2, (1, 3, 9, 8), (7, 6)), (3, (1, 2, 11), (13,))]:
        torch.manual_seed(seed)
        sd = len(ishape) - 2
        for dtype in (torch.float32, torch.float64):
            inp = torch.randn(ishape, dtype=dtype)
            # coordinates well outside the field of view on both sides, to exercise every boundary rule
            grid = (torch.rand((ishape[0],) + oshape + (sd,), dtype=dtype) * 3.0 - 1.0) * torch.tensor(ishape[2:], dtype=dtype)
            for bname, b in BOUNDS.items():
                for interp in (0, 1):
                    for extrap in (True, False):
                        y = ref.grid_pull(inp, grid, [ref.BoundType(b)], [ref.InterpolationType(interp)], extrap)
                        out[f"gp_{k}_cfg"] = np.asarray([seed, int(dtype == torch.float64), b, interp, int(extrap)])
                        out[f"gp_{k}_out"] = y.numpy()
                        k += 1
            if sd == 3 and dtype == torch.float32:  # per-axis boundary conditions
                y = ref.grid_pull(inp, grid, [ref.BoundType(2), ref.BoundType(7), ref.BoundType(5)], [ref.InterpolationType(1)], True)
                out["gp_mixed_out"] = y.numpy()
    out["gp_n"] = np.asarray(k)
    # the reference's own golden rows (orders 0/1): input arange(10), grid arange(20)+0.5
    rows = {}
    with open("/root/reference/tests/testing_data/1D_BP_fwd.txt") as f:
        for line in f:
            m = re.search(r"#\s*InterpolationType\.(\w+)\s+BoundType\.(\w+)", line)
            if not m or m.group(1) not in ("nearest", "linear") or m.group(2) not in BOUNDS:
                continue
            vals = [float(v) for v in line.split("#")[0].split(",") if v.strip()]
            rows[(m.group(1), m.group(2))] = vals
    for (interp, bound), vals in rows.items():
        out[f"bp1d_{interp}_{bound}"] = np.asarray(vals, dtype=np.float64)
    out["bp1d_keys"] = np.asarray([f"{i}_{b}" for (i, b) in rows])
    np.savez_compressed(os.path.join(HERE, "grid_pull.npz"), **out)
    print("grid_pull golden:", k, "cases +", len(rows), "rows of 1D_BP_fwd.txt")


if __name__ == "__main__":
    main()
