"""PatchInferer / SlidingWindowSplitter / AvgMerger cases shared by the golden generator (real reference, CPU) and the
emulator / MI355X tests."""
import os

import numpy as np
import torch
import torch.nn.functional as F

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _drop_row(patch, location):
    return location[0] != 8


NETS = {
    "affine": lambda x: x * 2.0 + 1.0,
    "two": lambda x: (x * 2.0, F.avg_pool2d(x, 2) if x.dim() == 4 else F.avg_pool3d(x, 2)),
    "dict": lambda x: {"a": x + 1.0, "b": x * x},
}

PATCH_CASES = [
    {"shape": (2, 3, 16, 20), "splitter": {"patch_size": (8, 8), "overlap": 0.5}, "inferer": {"batch_size": 3}, "net": "affine"},
    {"shape": (1, 2, 18, 21), "splitter": {"patch_size": (8, 8), "overlap": 0.25, "pad_mode": "constant", "pad_value": 0.5}, "inferer": {"batch_size": 2}, "net": "affine"},
    {"shape": (1, 2, 18, 21), "splitter": {"patch_size": (8, 8), "overlap": 2, "offset": (2, 1), "pad_mode": "replicate"}, "inferer": {"batch_size": 4}, "net": "affine"},
    {"shape": (1, 2, 18, 21), "splitter": {"patch_size": (8, 6), "overlap": 0.0, "pad_mode": None}, "inferer": {"batch_size": 1, "match_spatial_shape": True}, "net": "affine"},
    {"shape": (1, 1, 16, 16), "splitter": {"patch_size": (8, 8), "overlap": 0.0, "filter": "drop_row"}, "inferer": {"batch_size": 2}, "net": "affine"},
    {"shape": (1, 2, 12, 16, 10), "splitter": {"patch_size": (8, 8, 6), "overlap": (0.5, 0.25, 0.0)}, "inferer": {"batch_size": 2}, "net": "affine"},
    {"shape": (1, 2, 16, 16), "splitter": {"patch_size": (8, 8), "overlap": 0.5}, "inferer": {"batch_size": 2}, "net": "two"},
    {"shape": (1, 1, 8, 12, 8), "splitter": {"patch_size": (4, 4, 4), "overlap": 0.5}, "inferer": {"batch_size": 5}, "net": "two"},
    {"shape": (2, 2, 16, 12), "splitter": {"patch_size": (8, 8), "overlap": 0.5}, "inferer": {"batch_size": 2}, "net": "dict"},
    {"shape": (1, 2, 18, 21), "splitter": {"patch_size": (8, 8), "overlap": 0.25}, "inferer": {"batch_size": 2, "match_spatial_shape": False}, "net": "affine"},
]


def patch_input(k, case):
    gen = torch.Generator().manual_seed(900 + k)
    return torch.rand(*case["shape"], generator=gen)


def run_case(mod_inferers, k, case, device):
    """`mod_inferers` provides PatchInferer / SlidingWindowSplitter (monai.inferers or monai_amd.inferers)."""
    sk = dict(case["splitter"])
    filt = sk.pop("filter", None)
    if filt:
        sk["filter_fn"] = _drop_row
    splitter = mod_inferers.SlidingWindowSplitter(**sk)
    inferer = mod_inferers.PatchInferer(splitter=splitter, **case["inferer"])
    out = inferer(patch_input(k, case).to(device), NETS[case["net"]])
    if isinstance(out, dict):
        return {f"{key}": v.cpu().numpy() for key, v in out.items()}
    if isinstance(out, (list, tuple)):
        return {str(i): v.cpu().numpy() for i, v in enumerate(out)}
    return {"0": out.cpu().numpy()}


def case_patch_inferer_vs_reference(device):
    """PatchInferer + SlidingWindowSplitter + AvgMerger against the real reference classes (tests/golden/make_golden_patch.py):
    the accumulation order is the patch order in both, so the merged tensors are BIT-IDENTICAL (NaN where no patch landed)."""
    import monai_amd.inferers as ours

    g = np.load(os.path.join(GOLDEN, "patch_inferer.npz"))
    assert int(g["n"]) == len(PATCH_CASES)
    for k, case in enumerate(PATCH_CASES):
        got = run_case(ours, k, case, device)
        for key, y in got.items():
            exp = g[f"pi_{k}_{key}"]
            assert y.shape == exp.shape, (k, key, y.shape, exp.shape)
            np.testing.assert_array_equal(y, exp, err_msg=f"case {k} output {key}")
    return len(PATCH_CASES)


def case_patch_inferer_api(device):
    """Argument validation and the splitter / merger objects on their own."""
    import pytest

    from monai_amd.inferers import AvgMerger, PatchInferer, SlidingWindowSplitter

    with pytest.raises(ValueError):
        SlidingWindowSplitter(patch_size=(8, 8), overlap=1.0)
    with pytest.raises(ValueError):
        SlidingWindowSplitter(patch_size=(8, 8), overlap=-1)
    with pytest.raises(ValueError):
        SlidingWindowSplitter(patch_size=(8, 8), offset=-1, pad_mode=None)
    with pytest.raises(ValueError):
        SlidingWindowSplitter(patch_size=(8, 8), filter_fn=lambda a: True)
    with pytest.raises(TypeError):
        PatchInferer(splitter="nope")
    with pytest.raises(ValueError):
        PatchInferer(splitter=None, batch_size=0)
    with pytest.raises(ValueError):
        PatchInferer(splitter=None, merger_cls="NoSuchMerger")
    with pytest.raises(ValueError):
        PatchInferer(splitter=None)(torch.zeros(1, 1, 4, 4), lambda x: x)      # not split, no splitter
    sp = SlidingWindowSplitter(patch_size=(4, 4), overlap=0.5, offset=(-2, 0))
    x = torch.arange(48, dtype=torch.float32).reshape(1, 1, 6, 8).to(device)
    locs = [loc for _, loc in sp(x)]
    assert locs[0] == (-2, 0) and sp.get_padded_shape(x) == (8, 8) and sp.get_input_shape(x) == (6, 8)
    # pre-split input: (patch, location) pairs and explicit merged shape
    m = AvgMerger(merged_shape=(1, 1, 6, 8))
    for patch, loc in SlidingWindowSplitter(patch_size=(2, 4), overlap=0)(x):
        m.aggregate(patch * 3.0, loc)
    assert torch.equal(m.finalize().cpu(), x.cpu() * 3.0) and torch.equal(m.get_counts().cpu(), torch.ones(1, 1, 6, 8, dtype=torch.uint8))
    with pytest.raises(ValueError):
        m.aggregate(x[..., :2, :4], (0, 0))
    pairs = list(SlidingWindowSplitter(patch_size=(2, 4), overlap=0)(x))
    out = PatchInferer(splitter=None, merged_shape=(1, 1, 6, 8))(pairs, lambda p: p + 1.0)
    assert torch.equal(out.cpu(), x.cpu() + 1.0)


def case_gathered_split_and_batched_merge(device):
    """The product's own pieces behind the reference API: a 3-D single-image volume is cut by ONE window-gather launch (patches = consecutive rows of
    one dense buffer, batches = contiguous slices of it), `locations` lists what `__call__` yields, and `AvgMerger.aggregate_batch` (one launch per
    batch, patches of a batch overlapping each other) leaves the bits of patch-by-patch `aggregate`."""
    from monai_amd.inferers import AvgMerger, SlidingWindowSplitter

    x = torch.rand(1, 2, 12, 16, 10, generator=torch.Generator().manual_seed(950)).to(device)
    sp = SlidingWindowSplitter((8, 8, 6), overlap=(0.5, 0.25, 0.0))
    pairs = list(sp(x))
    assert sp.get_padded_shape(x) == (12, 20, 12) and len(pairs) == 12
    assert [tuple(v) for v in sp.locations(x.shape[2:]).tolist()] == [tuple(loc) for _, loc in pairs]
    padded = F.pad(x, [0, 2, 0, 4, 0, 0])
    for p, loc in pairs:
        assert torch.equal(p, padded[:, :, loc[0]:loc[0] + 8, loc[1]:loc[1] + 8, loc[2]:loc[2] + 6]), loc
    batches = list(sp.split_batches(x, 5))
    assert [b.shape[0] for b, _ in batches] == [5, 5, 2] and all(b._base is not None for b, _ in batches), "batches must be slices of the gathered buffer"
    one, many = AvgMerger((1, 2, 12, 20, 12)), AvgMerger((1, 2, 12, 20, 12))
    for b, locs in batches:
        y = b * 1.5 + 0.25
        one.aggregate_batch(y, locs)
        for q, loc in zip(torch.chunk(y, len(locs)), locs):
            many.aggregate(q, loc)
    assert torch.equal(one.get_values(), many.get_values()) and torch.equal(one.get_counts(), many.get_counts())
    assert int(one.get_counts().max()) >= 4
    assert torch.equal(one.finalize(), many.finalize())
    # 2-D inputs, batches of images and filtered splits take the view path with the same pairs
    x2 = torch.rand(2, 3, 9, 11, generator=torch.Generator().manual_seed(951)).to(device)
    sp2 = SlidingWindowSplitter((4, 4), overlap=1, offset=(-1, 2), pad_mode="replicate")
    got = list(sp2(x2))
    assert [tuple(v) for v in sp2.locations(x2.shape[2:]).tolist()] == [tuple(loc) for _, loc in got] and got[0][1] == (-1, 2)
