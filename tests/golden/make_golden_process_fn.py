"""Golden vectors for `process_fn` (monai/inferers/utils.py:232-238, 270-275, 286) from the REAL reference: the callback
edits the window predictions and returns a per-batch weight map; the count map uses the first batch's.  Build container only."""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
from monai.inferers import sliding_window_inference  # noqa: E402


def toy(x):
    return torch.cat([torch.sin(x[:, :1] * (1.0 + 0.37 * k)) + 0.05 * k * x[:, :1] for k in range(3)], dim=1)


def make_process_fn():
    calls = {"n": 0}

    def process_fn(segs, win_data, imp):
        calls["n"] += 1
        segs = tuple(s * 0.5 + 0.25 * win_data.mean() for s in segs)           # edits the predictions
        w = imp * (1.0 + 0.125 * (calls["n"] % 3)) + 0.0625                    # a weight map that changes from batch to batch
        return segs, w

    return process_fn


def main():
    out = {}
    for i, (shape, roi, sw, ov, mode) in enumerate((((1, 1, 20, 24, 28), (8, 12, 16), 2, 0.5, "gaussian"), ((2, 1, 14, 18), (6, 8), 3, 0.25, "constant"))):
        torch.manual_seed(40 + i)
        x = torch.rand(shape)
        with torch.no_grad():
            y = sliding_window_inference(x, roi, sw, toy, overlap=ov, mode=mode, process_fn=make_process_fn())
        out[f"pf_{i}_shape"], out[f"pf_{i}_roi"], out[f"pf_{i}_sw"] = np.asarray(shape), np.asarray(roi), np.asarray(sw)
        out[f"pf_{i}_ov"], out[f"pf_{i}_mode"], out[f"pf_{i}_out"] = np.asarray(ov), np.asarray(mode), y.numpy()
    np.savez_compressed(os.path.join(HERE, "process_fn.npz"), **out)
    print("process_fn golden written")


if __name__ == "__main__":
    main()
