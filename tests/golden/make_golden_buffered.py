"""Golden outputs of the REAL reference's buffered sliding-window schedule (monai/inferers/utils.py:239-253, 276-284, 324-348) -- build container only
(`PYTHONPATH=/root/reference python tests/golden/make_golden_buffered.py`).  Toy predictor evaluated on the CPU (the same one as blend.npz), so the stored
outputs pin the BLEND arithmetic of the buffered path: window order sorted by the buffered axis, slab-wise partial sums, count map in the sorted order."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from monai.inferers import sliding_window_inference  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def toy_predictor(k_out):
    return lambda x: torch.cat([torch.sin(x[:, :1] * (1.0 + 0.37 * k)) + 0.05 * k * x[:, :1] for k in range(k_out)], dim=1)


CASES = [
    dict(shape=(1, 1, 40, 36, 32), roi=(16, 16, 16), ov=0.5, mode="gaussian", sw=4, k=3, seed=20, steps=1, dim=-1),
    dict(shape=(1, 1, 40, 36, 32), roi=(16, 16, 16), ov=0.5, mode="gaussian", sw=4, k=3, seed=20, steps=2, dim=0),
    dict(shape=(1, 1, 40, 36, 32), roi=(16, 16, 16), ov=0.5, mode="gaussian", sw=3, k=3, seed=20, steps=2, dim=1),
    dict(shape=(2, 1, 33, 20, 21), roi=(16, 20, 8), ov=0.6, mode="constant", sw=3, k=2, seed=21, steps=3, dim=2),
    dict(shape=(1, 1, 48, 48, 48), roi=(32, 32, 32), ov=0.25, mode="gaussian", sw=2, k=5, seed=22, steps=1, dim=-3),
    dict(shape=(1, 1, 20, 9, 12), roi=(24, 16, 16), ov=0.5, mode="gaussian", sw=2, k=2, seed=23, steps=2, dim=-1),   # padded
    dict(shape=(1, 1, 31, 45), roi=(8, 16), ov=0.5, mode="gaussian", sw=5, k=4, seed=24, steps=2, dim=-1),          # 2-D
    dict(shape=(1, 1, 64, 24, 24), roi=(16, 16, 16), ov=0.5, mode="gaussian", sw=4, k=2, seed=25, steps=100, dim=0),  # one group: every window in one slab
]

if __name__ == "__main__":
    out = {}
    for i, c in enumerate(CASES):
        torch.manual_seed(c["seed"])
        x = torch.rand(c["shape"])
        y = sliding_window_inference(x, c["roi"], c["sw"], toy_predictor(c["k"]), overlap=c["ov"], mode=c["mode"], padding_mode="constant", cval=-0.5,
                                     buffer_steps=c["steps"], buffer_dim=c["dim"])
        plain = sliding_window_inference(x, c["roi"], c["sw"], toy_predictor(c["k"]), overlap=c["ov"], mode=c["mode"], padding_mode="constant", cval=-0.5)
        for k, v in c.items():
            out[f"buf_{i}_{k}"] = np.asarray(v)
        out[f"buf_{i}_out"] = y.numpy()
        out[f"buf_{i}_differs_from_plain"] = np.asarray(int((y != plain).sum()))
        print(i, c, "voxels differing from the non-buffered run:", int((y != plain).sum()), "max", float((y - plain).abs().max()))
    np.savez_compressed(os.path.join(HERE, "buffered.npz"), **out)
