"""Shared between tests/golden/make_golden_buffered_calls.py (the REAL reference) and tests/e2e_cases.py (the product): the configurations and the toy predictor /
process_fn of the buffered-schedule-with-callbacks goldens.  Every operation is one IEEE-rounded elementwise op (or runs on the host on both sides: sin), so the two
sides compute the same bits when -- and only when -- the predictor is called with the same batches, coordinates and weight maps."""
import torch

CASES = [
    dict(shape=(1, 1, 40, 36, 32), roi=(16, 16, 16), ov=0.5, mode="gaussian", sw=4, k=3, seed=30, steps=2, dim=0, coord=False, proc="const", out="tensor"),
    dict(shape=(1, 1, 40, 36, 32), roi=(16, 16, 16), ov=0.5, mode="gaussian", sw=3, k=2, seed=31, steps=1, dim=-1, coord=True, proc=None, out="tensor"),
    dict(shape=(1, 1, 33, 20, 21), roi=(16, 20, 8), ov=0.6, mode="constant", sw=3, k=2, seed=32, steps=3, dim=2, coord=False, proc=None, out="tuple"),
    dict(shape=(1, 1, 33, 20, 21), roi=(16, 20, 8), ov=0.6, mode="gaussian", sw=4, k=2, seed=33, steps=2, dim=1, coord=False, proc=None, out="dict"),
    dict(shape=(2, 1, 40, 24, 24), roi=(16, 16, 16), ov=0.5, mode="gaussian", sw=5, k=2, seed=34, steps=2, dim=0, coord=True, proc="batch", out="tuple"),
    dict(shape=(1, 1, 31, 45), roi=(8, 16), ov=0.5, mode="gaussian", sw=5, k=3, seed=35, steps=2, dim=-1, coord=True, proc="const", out="tensor"),       # 2-D
]


def make_callbacks(c, cpu_math: bool):
    """-> (predictor, process_fn or None).  cpu_math: evaluate the predictor's sin on the host (the product's windows live on the device; the golden side is all-CPU)"""
    k_out = c["k"]

    def core(x):
        xc = x.cpu() if cpu_math else x
        y = torch.cat([torch.sin(xc[:, :1] * (1.0 + 0.37 * k)) + 0.05 * k * xc[:, :1] for k in range(k_out)], dim=1)
        return y.to(x.device)

    def shift(coords, like):      # one number per window from its coordinates: start of the first spatial slice * 2^-6 + image index * 0.5 (exact in fp32)
        v = torch.tensor([float(cs[2].start) * 0.015625 + float(cs[0].start) * 0.5 for cs in coords], dtype=like.dtype).to(like.device)
        return v.reshape((-1,) + (1,) * (like.dim() - 1))

    def pack(y):
        if c["out"] == "tuple":
            return y, y * 2.0
        if c["out"] == "dict":
            return {"b": y * 3.0, "a": y}
        return y

    if c["coord"]:
        predictor = lambda x, coords: pack(core(x) + shift(coords, x))          # noqa: E731
    else:
        predictor = lambda x: pack(core(x))                                      # noqa: E731
    process_fn = None
    if c["proc"] == "const":
        process_fn = lambda segs, win, imp: ([s * 0.5 + 0.125 for s in segs], imp * 0.75 + 0.015625)          # noqa: E731
    elif c["proc"] == "batch":      # the weight map depends on the batch (its size): the count map is the one current at the first flush
        process_fn = lambda segs, win, imp: ([s * 0.5 for s in segs], imp * (1.0 + 0.25 * float(win.shape[0])))     # noqa: E731
    return predictor, process_fn
