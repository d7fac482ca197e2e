"""ScaleIntensityRange / CropForeground cases shared by the golden generator (real reference, CPU) and the emulator / MI355X tests."""
import os
import warnings

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def ct(seed=0, shape=(2, 9, 14, 19)):
    """CT-like intensities (HU) with specials: NaN, +-inf, exact range ends"""
    gen = torch.Generator().manual_seed(2300 + seed)
    x = torch.randn(*shape, generator=gen) * 400.0 - 100.0
    x.view(-1)[:6] = torch.tensor([float("nan"), float("inf"), float("-inf"), -175.0, 250.0, 0.0])
    return x


SCALE_CASES = [
    ("scale_ct", {"a_min": -175.0, "a_max": 250.0, "b_min": 0.0, "b_max": 1.0, "clip": True}),
    ("scale_noclip", {"a_min": -175.0, "a_max": 250.0, "b_min": 0.0, "b_max": 1.0}),
    ("scale_nob", {"a_min": -57.0, "a_max": 164.0}),
    ("scale_range", {"a_min": -1000.3, "a_max": 1000.7, "b_min": -1.1, "b_max": 3.3, "clip": True}),
    ("scale_half_b", {"a_min": 0.0, "a_max": 300.0, "b_min": 0.25, "clip": True}),
    ("scale_half_bmax", {"a_min": 0.0, "a_max": 300.0, "b_max": 0.75, "clip": True}),
    ("scale_degenerate", {"a_min": 7.0, "a_max": 7.0, "b_min": 0.5, "b_max": 1.0}),
    ("scale_degenerate_nob", {"a_min": 7.0, "a_max": 7.0}),
    ("scale_int_dtype", {"a_min": -175.0, "a_max": 250.0, "b_min": 0.0, "b_max": 255.0, "clip": True, "dtype": torch.uint8}),
]


def blob(seed=0, shape=(2, 12, 17, 23), box=((3, 9), (4, 13), (5, 20)), chans=(0, 1)):
    """zero background (with negative values and a NaN outside the box) and a positive blob inside `box`"""
    gen = torch.Generator().manual_seed(2400 + seed)
    x = -torch.rand(*shape, generator=gen)
    x[x > -0.3] = 0.0
    (z0, z1), (y0, y1), (x0, x1) = box
    for c in chans:
        x[c, z0, y0, x0] = 1.0
        x[c, z1 - 1, y1 - 1, x1 - 1] = 2.0
        x[c, (z0 + z1) // 2, y0:y1, (x0 + x1) // 2] = 0.5
    x[0, 0, 0, 0] = float("nan")
    return x


def _gt1(img):
    return img > 1.0


# name, image builder, CropForeground kwargs
CROP_CASES = [
    ("crop_plain", lambda: blob(0), {}),
    ("crop_margin", lambda: blob(1), {"margin": 2}),
    ("crop_margin_out", lambda: blob(2, box=((0, 5), (2, 17), (1, 22))), {"margin": [3, 2, 4]}),
    ("crop_margin_smaller", lambda: blob(2, box=((0, 5), (2, 17), (1, 22))), {"margin": [3, 2, 4], "allow_smaller": True}),
    ("crop_kdiv", lambda: blob(3), {"k_divisible": 4}),
    ("crop_kdiv_margin", lambda: blob(3), {"k_divisible": [8, 4, 16], "margin": 1}),
    ("crop_channel1", lambda: blob(4, chans=(1,)), {"channel_indices": 1}),
    ("crop_channel0_empty", lambda: blob(4, chans=(1,)), {"channel_indices": [0]}),
    ("crop_select", lambda: blob(5), {"select_fn": _gt1}),
    ("crop_padvalue", lambda: blob(6, box=((0, 4), (0, 6), (0, 7))), {"margin": 3, "value": -2.5}),
    ("crop_w4", lambda: blob(7, shape=(1, 6, 10, 32), box=((1, 5), (2, 9), (7, 29)), chans=(0,)), {}),
    ("crop_2d", lambda: blob(8)[:, 6].contiguous(), {"margin": 1}),
    ("crop_full", lambda: torch.ones(1, 4, 5, 6), {}),
]


def run_all(mod_transforms, device, make_meta):
    """-> {name: array}; `make_meta(tensor, affine)` builds the MetaTensor type of the library under test"""
    out = {}
    x = ct().to(device)
    for name, kw in SCALE_CASES:
        # float -> uint8 of NaN / inf is implementation-defined (CPU and GPU casts may differ): the integer-output case gets finite values
        xin = torch.nan_to_num(x, nan=0.0, posinf=1e4, neginf=-1e4) if "dtype" in kw else x
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            y = mod_transforms.ScaleIntensityRange(**kw)(xin)
        out[name] = torch.as_tensor(y).cpu().numpy()
    xi = (ct(1) * 0.5).to(torch.int16).to(device)
    out["scale_int16_input"] = torch.as_tensor(mod_transforms.ScaleIntensityRange(-175.0, 250.0, 0.0, 1.0, clip=True)(xi)).cpu().numpy()
    d = mod_transforms.ScaleIntensityRanged(keys=["image"], a_min=-175.0, a_max=250.0, b_min=0.0, b_max=1.0, clip=True)({"image": x, "label": x})
    out["scale_dict_image"] = torch.as_tensor(d["image"]).cpu().numpy()

    aff = torch.tensor([[0.8, 0.1, 0.0, -12.0], [0.0, 0.9, 0.2, 7.0], [0.05, 0.0, 1.6, 30.0], [0.0, 0.0, 0.0, 1.0]], dtype=torch.float64)
    for name, make, kw in CROP_CASES:
        img = make().to(device)
        y, s, e = mod_transforms.CropForeground(return_coords=True, **kw)(img)
        out[name] = torch.as_tensor(y).cpu().numpy()
        out[name + "__start"] = np.asarray(s)
        out[name + "__end"] = np.asarray(e)
        if img.dim() == 4:
            tr = mod_transforms.CropForeground(**kw)
            m = tr(make_meta(img, aff))
            out[name + "__affine"] = np.asarray(torch.as_tensor(m.affine).cpu(), dtype=np.float64)
            if m.numel():
                inv = tr.inverse(m)
                out[name + "__inverse"] = torch.as_tensor(inv).cpu().numpy()
                out[name + "__inverse_affine"] = np.asarray(torch.as_tensor(inv.affine).cpu(), dtype=np.float64)
    img, lab = blob(9).to(device), (blob(9) > 0).float().to(device)
    d = mod_transforms.CropForegroundd(keys=["image", "label"], source_key="label", margin=1, k_divisible=2)({"image": img, "label": lab})
    out["cropd_image"] = torch.as_tensor(d["image"]).cpu().numpy()
    out["cropd_label"] = torch.as_tensor(d["label"]).cpu().numpy()
    out["cropd_start"] = np.asarray(d["foreground_start_coord"])
    out["cropd_end"] = np.asarray(d["foreground_end_coord"])
    return out


def case_preproc_vs_reference(device):
    """ScaleIntensityRange(d) / CropForeground(d) against the real reference transforms (tests/golden/make_golden_preproc.py):
    everything bit-identical (the scale keeps the reference's roundings; crops are copies), box coordinates and the cropped
    MetaTensor's affine exact."""
    import monai_amd.transforms as ours
    from monai_amd.data.meta_tensor import MetaTensor

    g = np.load(os.path.join(GOLDEN, "preproc.npz"))
    got = run_all(ours, device, lambda t, a: MetaTensor(t, affine=a))
    assert set(got) == set(g.files), set(got) ^ set(g.files)
    for name, y in got.items():
        exp = g[name]
        assert y.shape == exp.shape, (name, y.shape, exp.shape)
        if name.endswith(("__start", "__end", "_start", "_end")):
            np.testing.assert_array_equal(y.astype(np.int64), exp.astype(np.int64), err_msg=name)
        elif name.endswith("affine"):
            np.testing.assert_allclose(y, exp, rtol=0, atol=1e-12, err_msg=name)
        else:
            assert y.dtype == exp.dtype, (name, y.dtype, exp.dtype)
            np.testing.assert_array_equal(y, exp, err_msg=name)
    return len(got)


def case_bbox_large(device):
    """The box kernel at a size with many workgroups and rows per wave, against numpy (vector and scalar row paths)."""
    from monai_amd import ops

    rng = np.random.RandomState(5)
    n = 0
    for shape in ((1, 40, 50, 64), (3, 33, 41, 67), (2, 70, 130, 36)):
        x = -rng.rand(*shape).astype(np.float32)
        assert ops.foreground_bbox(torch.from_numpy(x).to(device)) is None
        pts = [(rng.randint(shape[0]), rng.randint(5, shape[1] - 5), rng.randint(7, shape[2] - 3), rng.randint(2, shape[3] - 9)) for _ in range(5)]
        for p in pts:
            x[p] = 1e-30
        fg = (x > 0).any(0)
        zz, yy, xx = np.nonzero(fg)
        exp = (zz.min(), yy.min(), xx.min(), zz.max(), yy.max(), xx.max())
        assert ops.foreground_bbox(torch.from_numpy(x).to(device)) == tuple(int(v) for v in exp), shape
        n += 1
    return n


def case_preproc_api(device):
    import pytest

    from monai_amd.transforms import CropForeground, CropForegroundd, ScaleIntensityRange

    x = blob().to(device)
    with pytest.raises(ValueError):
        CropForeground(margin=-1)(x)
    with pytest.raises(NotImplementedError):
        CropForeground(mode="reflect")(x)
    # lazy execution is supported (monai_amd/transforms/lazy.py): the switch is recorded, nothing raises
    assert CropForeground(lazy=True).lazy is True
    with pytest.raises(NotImplementedError):
        ScaleIntensityRange(0.0, 1.0)(x.double())
    with pytest.raises(KeyError):
        CropForegroundd(keys=["missing"], source_key="image")({"image": x})
    d = CropForegroundd(keys=["missing"], source_key="image", allow_missing_keys=True, start_coord_key=None, end_coord_key=None)({"image": x})
    assert set(d) == {"image"}
    with pytest.warns(Warning):
        ScaleIntensityRange(1.0, 1.0)(x)
    with pytest.raises(RuntimeError):                 # the reference's torch.clamp(img, None, None) raises the same way
        ScaleIntensityRange(0.0, 1.0, clip=True)(x)


def case_preproc_full_size(device, edge=512):
    """Size-independent properties at the BASELINE volume size (edge^3, one channel): the scale against the same operator sequence in
    torch, NormalizeIntensity's output moments, the crop against a slice of the known box, Orientation against torch.flip and its
    inverse round trip."""
    from monai_amd.data.meta_tensor import MetaTensor
    from monai_amd.transforms import CropForeground, NormalizeIntensity, Orientation, ScaleIntensityRange

    gen = torch.Generator().manual_seed(77)
    lo, hi = edge // 8, edge - edge // 8 - 3
    x = torch.full((1, edge, edge, edge), -1000.0)
    x[:, lo:hi, lo + 1:hi, lo + 2:hi] = torch.rand((1, hi - lo, hi - lo - 1, hi - lo - 2), generator=gen) * 500.0 - 200.0
    x = x.to(device)
    y = ScaleIntensityRange(-175.0, 250.0, 0.0, 1.0, clip=True)(x)
    exp = torch.clamp(((x - (-175.0)) / 425.0) * 1.0 + 0.0, 0.0, 1.0)       # the device's own division may round differently: 1e-6
    assert float((y - exp).abs().max()) <= 1e-6 and float(y.min()) == 0.0 and float(y.max()) <= 1.0
    body = y[:, lo:hi, lo + 1:hi, lo + 2:hi]
    assert float(body.min()) >= 0.0 and float(y.sum()) == float(y.sum())    # no NaN
    # the body's corners are foreground with probability 1 - 1e-... only statistically; box the ">= 0" mask through a custom select_fn
    c, s, e = CropForeground(select_fn=lambda t: t > -500.0, return_coords=True, margin=0)(x)
    assert list(s) == [lo, lo + 1, lo + 2] and list(e) == [hi, hi, hi], (s, e)
    assert torch.equal(c, x[:, lo:hi, lo + 1:hi, lo + 2:hi])
    z = NormalizeIntensity(nonzero=True, channel_wise=True)(y)
    nz = z[y != 0]
    assert abs(float(nz.double().mean())) < 1e-4 and abs(float(nz.double().std(unbiased=False)) - 1.0) < 1e-4
    assert bool((z[y == 0] == 0).all())
    m = MetaTensor(x, affine=torch.as_tensor(np.diag([-0.8, -0.8, 1.6, 1.0])))
    tr = Orientation(axcodes="RAS")
    o = tr(m)
    assert torch.equal(o.as_tensor(), torch.flip(x, [1, 2]))
    assert np.allclose(np.asarray(o.affine), np.array([[0.8, 0, 0, -0.8 * (edge - 1)], [0, 0.8, 0, -0.8 * (edge - 1)], [0, 0, 1.6, 0], [0, 0, 0, 1.0]]))
    assert torch.equal(tr.inverse(o).as_tensor(), x)
    return edge
