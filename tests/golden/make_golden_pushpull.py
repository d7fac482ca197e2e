"""Golden vectors for the full ``monai._C`` resampling surface (pull / push / count / grad and their backward passes,
interpolation orders 0-7) from
 (a) the reference's own compiled CPU resampler (oracle/_ref, built by oracle/build_ref.py from the sources under
     /root/reference) on seeded inputs -- see tests/transform_cases.py::pp_inputs for the inputs, and
 (b) the reference's golden files tests/testing_data/1D_BP_fwd.txt / 1D_BP_bwd.txt (every row: 8 orders x 7 bounds,
     forward values and the four (input.requires_grad, grid.requires_grad) gradient rows each).
Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_pushpull.py"""
import os
import re
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import build_ref  # noqa: E402
from transform_cases import PP_CASES, pp_inputs, pp_run  # noqa: E402


def parse_rows(path):
    rows = []
    with open(path) as f:
        for line in f:
            m = re.search(r"#\s*InterpolationType\.(\w+)\s+BoundType\.(\w+)", line)
            if not m:
                continue
            vals = [float(v) for v in line.split("#")[0].split(",") if v.strip()]
            rows.append((m.group(1), m.group(2), vals))
    return rows


def main():
    build_ref.build()
    ref = build_ref.load()
    torch.set_num_threads(1)        # the reference's atomics-free accumulation order for push / count
    out = {}
    for k, case in enumerate(PP_CASES):
        res = pp_run(ref, case, pp_inputs(case), "cpu")
        for name, t in res.items():
            out[f"pp_{k}_{name}"] = t.numpy()
    out["pp_n"] = np.asarray(len(PP_CASES))
    fwd = parse_rows("/root/reference/tests/testing_data/1D_BP_fwd.txt")
    for interp, bound, vals in fwd:
        out[f"fwd_{interp}_{bound}"] = np.asarray(vals, dtype=np.float64)
    bwd = parse_rows("/root/reference/tests/testing_data/1D_BP_bwd.txt")
    seen = {}
    for interp, bound, vals in bwd:
        j = seen.get((interp, bound), 0)
        seen[(interp, bound)] = j + 1
        out[f"bwd_{interp}_{bound}_{j}"] = np.asarray(vals, dtype=np.float64)
    out["rows"] = np.asarray([f"{i}_{b}" for i, b, _ in fwd])
    assert all(v == 4 for v in seen.values()) and len(seen) == len(fwd), (len(seen), len(fwd))
    np.savez_compressed(os.path.join(HERE, "pushpull.npz"), **out)
    print("pushpull golden:", len(PP_CASES), "cases,", len(fwd), "forward rows,", len(bwd), "backward rows;",
          os.path.getsize(os.path.join(HERE, "pushpull.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
