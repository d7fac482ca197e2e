"""Flip / Rotate90 cases shared by the golden generator (real reference, CPU) and the emulator / MI355X tests."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
AFF = torch.tensor([[0.8, 0.1, 0.0, -12.0], [0.0, 0.9, 0.2, 7.0], [0.05, 0.0, 1.6, 30.0], [0.0, 0.0, 0.0, 1.0]], dtype=torch.float64)


def image(seed=0, shape=(2, 5, 7, 66)):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(3500 + seed))


CASES = [("flip_all", "Flip", {})] + [(f"flip_{a}", "Flip", {"spatial_axis": a}) for a in (0, 1, 2, -1, (0, 2), [1, -1])] + [
    (f"rot_k{k}_{a[0]}{a[1]}", "Rotate90", {"k": k, "spatial_axes": a}) for k in (0, 1, 2, 3, 5, -1) for a in ((0, 1), (1, 2), (0, 2), (2, 0), (-1, 0))]
CASES = [(n.replace("-", "m").replace("(", "").replace(")", "").replace(", ", "_").replace("[", "").replace("]", ""), c, kw) for n, c, kw in CASES]


def run_all(ns, device, make_meta):
    out = {}
    for name, cls, kw in CASES:
        tr = getattr(ns, cls)(**kw)
        m = tr(make_meta(image().to(device), AFF))
        out[name] = torch.as_tensor(m).cpu().numpy()
        out[name + "__affine"] = np.asarray(torch.as_tensor(m.affine).cpu(), dtype=np.float64)
        inv = tr.inverse(m)
        out[name + "__inverse_affine"] = np.asarray(torch.as_tensor(inv.affine).cpu(), dtype=np.float64)
        assert torch.equal(torch.as_tensor(inv).cpu(), image()), name            # both libraries: the inverse restores the data exactly
    out["flip_2d"] = torch.as_tensor(ns.Flip(1)(image(1, (3, 9, 14)).to(device))).cpu().numpy()
    out["rot_2d"] = torch.as_tensor(ns.Rotate90(1)(image(1, (3, 9, 14)).to(device))).cpu().numpy()
    out["rot_label"] = torch.as_tensor(ns.Rotate90(3, (1, 2))((image(2) * 3).to(torch.int16).to(device))).cpu().numpy()
    d = {"image": make_meta(image(3).to(device), AFF), "label": make_meta(image(4).to(device), AFF)}
    d = ns.Flipd(keys=["image", "label"], spatial_axis=0)(d)
    d = ns.Rotate90d(keys=["image", "label"], k=1, spatial_axes=(0, 2))(d)
    for k in ("image", "label"):
        out["dict_" + k] = torch.as_tensor(d[k]).cpu().numpy()
        out["dict_" + k + "__affine"] = np.asarray(torch.as_tensor(d[k].affine).cpu(), dtype=np.float64)
    return out


def case_flip_rotate_vs_reference(device):
    import monai_amd.transforms as ours
    from monai_amd.data.meta_tensor import MetaTensor

    g = np.load(os.path.join(GOLDEN, "flip_rotate.npz"))
    got = run_all(ours, device, lambda t, a: MetaTensor(t, affine=a))
    assert set(got) == set(g.files), set(got) ^ set(g.files)
    for name, y in got.items():
        exp = g[name]
        assert y.shape == exp.shape, (name, y.shape, exp.shape)
        if name.endswith("affine"):
            np.testing.assert_allclose(y, exp, rtol=0, atol=1e-9, err_msg=name)      # the reference composes sin / cos matrices (1e-16 off the integers)
        else:
            assert y.dtype == exp.dtype, (name, y.dtype, exp.dtype)
            np.testing.assert_array_equal(y, exp, err_msg=name)
    return len(got)


def case_flip_rotate_api(device):
    import pytest

    from monai_amd.transforms import Flip, Rotate90

    x = image().to(device)
    with pytest.raises(ValueError):
        Rotate90(1, (0, 1, 2))
    with pytest.raises(ValueError):
        Rotate90(1, (1, 1))(x)
    with pytest.raises(IndexError):
        Flip(3)(x)
    # lazy execution is supported (monai_amd/transforms/lazy.py): the switch is recorded, nothing raises
    assert Flip(0, lazy=True).lazy is True
    with pytest.raises(NotImplementedError):
        Flip(0)(x.double())
