"""NormalizeIntensity cases shared by the golden generator (real reference, CPU) and the emulator / MI355X tests, and an
MRI-bundle-shaped pipeline: NormalizeIntensityd(nonzero, channel_wise) -> SlidingWindowInferer(SegResNet, 4 modalities) ->
Activationsd(sigmoid) -> AsDiscreted(threshold 0.5)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 2e-6      # relative to values of order 1: the reference's mean / std are fp32 tree sums, the kernels' are fp64 sums


def mri(seed=0, shape=(4, 11, 18, 27)):
    """four "modalities" with different offsets / scales and a zero background outside the "brain" """
    gen = torch.Generator().manual_seed(3100 + seed)
    x = torch.randn(*shape, generator=gen)
    for c in range(shape[0]):
        x[c] = x[c] * (50.0 * (c + 1)) + 300.0 * (c + 1)
    x[:, :2] = 0.0
    x[:, :, :, -5:] = 0.0
    x[1, shape[1] // 2, 5, 5] = 0.0
    return x


NORM_CASES = [
    ("norm_plain", {}),
    ("norm_nonzero", {"nonzero": True}),
    ("norm_channel", {"channel_wise": True}),
    ("norm_nonzero_channel", {"nonzero": True, "channel_wise": True}),
    ("norm_given", {"subtrahend": 100.0, "divisor": 7.0}),
    ("norm_given_sub", {"subtrahend": 250.0, "nonzero": True}),
    ("norm_given_div0", {"subtrahend": 1.0, "divisor": 0.0}),
    ("norm_given_channel", {"subtrahend": [1.0, 2.0, 3.0, 4.0], "divisor": [2.0, 0.0, 4.0, 5.0], "channel_wise": True}),
    ("norm_given_div_channel", {"divisor": [2.0, 3.0, 4.0, 5.0], "channel_wise": True, "nonzero": True}),
]


def run_all(mod_transforms, device):
    out = {}
    for name, kw in NORM_CASES:
        x = mri().to(device)
        out[name] = torch.as_tensor(mod_transforms.NormalizeIntensity(**kw)(x)).cpu().numpy()
    out["norm_constant"] = torch.as_tensor(mod_transforms.NormalizeIntensity()(torch.full((1, 4, 5, 6), 3.5).to(device))).cpu().numpy()
    out["norm_all_zero_nonzero"] = torch.as_tensor(mod_transforms.NormalizeIntensity(nonzero=True)(torch.zeros(2, 3, 4, 5).to(device))).cpu().numpy()
    out["norm_int16"] = torch.as_tensor(mod_transforms.NormalizeIntensity(nonzero=True, channel_wise=True)(mri(1).to(torch.int16).to(device))).cpu().numpy()
    out["norm_odd"] = torch.as_tensor(mod_transforms.NormalizeIntensity(channel_wise=True)(mri(2, (3, 5, 7, 9)).to(device))).cpu().numpy()
    d = mod_transforms.NormalizeIntensityd(keys=["image"], nonzero=True, channel_wise=True)({"image": mri(3).to(device)})
    out["normd_image"] = torch.as_tensor(d["image"]).cpu().numpy()
    return out


def case_normalize_vs_reference(device):
    """NormalizeIntensity(d) against the real reference transform (tests/golden/make_golden_normalize.py), <= 2e-6 on values of
    order 1; zeros of the `nonzero` mode stay exactly zero."""
    import monai_amd.transforms as ours

    g = np.load(os.path.join(GOLDEN, "normalize.npz"))
    got = run_all(ours, device)
    assert set(got) == set(g.files), set(got) ^ set(g.files)
    worst = 0.0
    for name, y in got.items():
        exp = g[name]
        assert y.shape == exp.shape and y.dtype == exp.dtype, (name, y.shape, exp.shape, y.dtype, exp.dtype)
        err = float(np.abs(y.astype(np.float64) - exp).max() / max(1.0, float(np.abs(exp).max())))
        assert err < TOL, (name, err)
        np.testing.assert_array_equal(y == 0, exp == 0, err_msg=name)
        worst = max(worst, err)
    return worst


def case_normalize_api(device):
    import pytest

    from monai_amd.transforms import NormalizeIntensity

    x = mri().to(device)
    with pytest.raises(ValueError):
        NormalizeIntensity(subtrahend=[1.0, 2.0], channel_wise=True)(x)
    with pytest.raises(ValueError):
        NormalizeIntensity(divisor=[1.0, 2.0], channel_wise=True)(x)
    with pytest.raises(NotImplementedError):
        NormalizeIntensity(subtrahend=torch.zeros(4, 11, 18, 27))(x)
    with pytest.raises(NotImplementedError):
        NormalizeIntensity()(x.double())
    before = x.clone()
    NormalizeIntensity(channel_wise=True)(x)
    assert torch.equal(x, before)                       # the input is left alone


# ---------------------------------------------------------------------------------------------- MRI-bundle-shaped pipeline
ROI = (32, 32, 32)
SEG_KW = dict(init_filters=16, in_channels=4, out_channels=3, blocks_down=(1, 2, 2), blocks_up=(1, 1))


def mri_volume():
    return mri(7, (4, 40, 48, 36))


def make_net(cls):
    torch.manual_seed(77)
    return cls(spatial_dims=3, **SEG_KW).eval()


def run_pipeline(ns, net, device):
    d = ns.NormalizeIntensityd(keys=["image"], nonzero=True, channel_wise=True)({"image": mri_volume().to(device)})
    x = d["image"]
    with torch.no_grad():
        logits = ns.SlidingWindowInferer(roi_size=ROI, sw_batch_size=2, overlap=0.5, mode="gaussian")(x[None], net)
    d = ns.Activationsd(keys=["pred"], sigmoid=True)({"pred": logits[0]})
    prob = d["pred"]
    d = ns.AsDiscreted(keys=["pred"], threshold=0.5)(d)
    as_np = lambda t: (t.as_tensor() if hasattr(t, "as_tensor") else t).detach().cpu().numpy()  # noqa: E731
    return {"normalized": as_np(x), "logits": as_np(logits), "prob": as_np(prob), "mask": as_np(d["pred"])}


def case_mri_pipeline_vs_reference(device):
    from types import SimpleNamespace

    from monai_amd.inferers import SlidingWindowInferer
    from monai_amd.networks.nets import SegResNet
    from monai_amd.transforms import Activationsd, AsDiscreted, NormalizeIntensityd

    g = np.load(os.path.join(GOLDEN, "pipeline_mri.npz"))
    ns = SimpleNamespace(NormalizeIntensityd=NormalizeIntensityd, SlidingWindowInferer=SlidingWindowInferer, Activationsd=Activationsd, AsDiscreted=AsDiscreted)
    got = run_pipeline(ns, make_net(SegResNet).to(device), device)
    dn = float(np.abs(got["normalized"] - g["normalized"]).max())
    dl = float(np.abs(got["logits"] - g["logits"]).max())
    dp = float(np.abs(got["prob"] - g["prob"]).max())
    mism = got["mask"] != g["mask"]
    worst = float(np.abs(g["logits"][0][mism[...]]).max()) if mism.any() else 0.0          # masks may only differ where the reference's logit is ~0
    assert dn < 1e-5 and dl < 1e-4 and dp < 1e-4, (dn, dl, dp)
    assert worst < 2e-4, (int(mism.sum()), worst)
    return {"max_normalized_diff": dn, "max_logit_diff": dl, "max_prob_diff": dp, "mask_mismatches": int(mism.sum()), "voxels": int(mism.size)}


# ---------------------------------------------------------------------------------------------- ScaleIntensity
SCALE_INT_CASES = [
    ("si_default", {}),
    ("si_range", {"minv": -1.0, "maxv": 2.5}),
    ("si_channel", {"channel_wise": True}),
    ("si_channel_range", {"minv": 10.0, "maxv": 20.0, "channel_wise": True}),
    ("si_minv_only", {"minv": 3.0, "maxv": None}),
    ("si_maxv_only", {"minv": None, "maxv": 3.0}),
    ("si_factor", {"minv": None, "maxv": None, "factor": 0.37}),
    ("si_nothing", {"minv": None, "maxv": None}),
]


def run_scale_intensity(mod_transforms, device):
    out = {}
    for name, kw in SCALE_INT_CASES:
        out[name] = torch.as_tensor(mod_transforms.ScaleIntensity(**kw)(mri().to(device))).cpu().numpy()
    flat = mri()
    flat[2] = 4.25                                              # one constant channel: the reference's `min == max` branch
    out["si_flat_channel"] = torch.as_tensor(mod_transforms.ScaleIntensity(minv=2.0, maxv=3.0, channel_wise=True)(flat.to(device))).cpu().numpy()
    out["si_flat_noscale"] = torch.as_tensor(mod_transforms.ScaleIntensity(minv=None, maxv=1.0)(torch.full((1, 3, 4, 5), -2.0).to(device))).cpu().numpy()
    withnan = mri()
    withnan[1, 3, 3, 3] = float("nan")
    out["si_nan_channel"] = torch.as_tensor(mod_transforms.ScaleIntensity(channel_wise=True)(withnan.to(device))).cpu().numpy()
    out["si_int16"] = torch.as_tensor(mod_transforms.ScaleIntensity()(mri(1).to(torch.int16).to(device))).cpu().numpy()
    out["si_odd"] = torch.as_tensor(mod_transforms.ScaleIntensity(channel_wise=True)(mri(2, (3, 5, 7, 9)).to(device))).cpu().numpy()
    d = mod_transforms.ScaleIntensityd(keys=["image"], minv=0.0, maxv=255.0)({"image": mri(3).to(device)})
    out["sid_image"] = torch.as_tensor(d["image"]).cpu().numpy()
    return out


def case_scale_intensity_vs_reference(device):
    """ScaleIntensity(d) against the real reference transform: bit-identical (exact reductions, the reference's fp32 operator sequence)"""
    import monai_amd.transforms as ours

    g = np.load(os.path.join(GOLDEN, "scale_intensity.npz"))
    got = run_scale_intensity(ours, device)
    assert set(got) == set(g.files), set(got) ^ set(g.files)
    for name, y in got.items():
        assert y.shape == g[name].shape and y.dtype == g[name].dtype, name
        np.testing.assert_array_equal(y, g[name], err_msg=name)
    return len(got)
