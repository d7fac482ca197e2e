"""Golden outputs of the REAL reference's buffered sliding-window schedule WITH callbacks in the loop (monai/inferers/utils.py:215-253, 324-348): `process_fn`,
`with_coord`, tuple / dict predictor outputs -- build container only (`PYTHONPATH=/root/reference python tests/golden/make_golden_buffered_calls.py`).
What the stored outputs pin beyond buffered.npz: the predictor is CALLED in the buffered order (sorted windows, batches that end at the slab boundaries, the sorted
slices as coordinates), only the first output is blended, the count map is the weight map of the batch that was current at the first flush.
The case functions are shared with tests/e2e_cases.py (same toy predictors / callbacks on both sides; every operation in them is a single IEEE-rounded op)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from buffered_call_cases import CASES, make_callbacks  # noqa: E402

if __name__ == "__main__":
    sys.path.insert(0, "/root/reference")
    from monai.inferers import sliding_window_inference

    out = {}
    for i, c in enumerate(CASES):
        torch.manual_seed(c["seed"])
        x = torch.rand(c["shape"])
        pred, process_fn = make_callbacks(c, cpu_math=False)
        y = sliding_window_inference(x, c["roi"], c["sw"], pred, overlap=c["ov"], mode=c["mode"], process_fn=process_fn, buffer_steps=c["steps"], buffer_dim=c["dim"],
                                     with_coord=c["coord"])
        if isinstance(y, dict):
            assert list(y) == ["a"], list(y)
            y = y["a"]
        assert isinstance(y, torch.Tensor), type(y)
        out[f"bc_{i}_out"] = y.numpy()
        print(i, c, tuple(y.shape), float(y.abs().max()))
    np.savez_compressed(os.path.join(HERE, "buffered_calls.npz"), **out)
