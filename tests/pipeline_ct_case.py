"""A CT-bundle-shaped pipeline through the drop-in classes: ScaleIntensityRanged -> CropForegroundd -> Spacingd ->
SlidingWindowInferer(DynUNet) -> AsDiscreted(argmax), against the same chain of the REAL reference classes on the CPU
(tests/golden/make_golden_pipeline_ct.py).  The crop's affine update feeds Spacingd, so the chain also checks the metadata."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
AFFINE = np.array([[1.2, 0.0, 0.0, -30.0], [0.0, 0.9, 0.0, 12.0], [0.0, 0.0, 1.5, 4.0], [0.0, 0.0, 0.0, 1.0]])
ROI = (32, 32, 32)


def volume():
    gen = torch.Generator().manual_seed(5151)
    x = torch.full((1, 48, 56, 40), -1000.0)                      # air
    x[:, 6:41, 9:50, 5:33] = torch.rand((1, 35, 41, 28), generator=gen) * 500.0 - 200.0      # body: HU in [-200, 300)
    return x


def run_pipeline(ns, net, device):
    d = {"image": ns.MetaTensor(volume().to(device), affine=AFFINE)}
    d = ns.ScaleIntensityRanged(keys=["image"], a_min=-175.0, a_max=250.0, b_min=0.0, b_max=1.0, clip=True)(d)
    d = ns.CropForegroundd(keys=["image"], source_key="image", margin=2, k_divisible=1)(d)
    cropped = d["image"]
    d = ns.Spacingd(keys=["image"], pixdim=(1.0, 1.0, 1.0), mode="bilinear", padding_mode="border")(d)
    x = d["image"]
    with torch.no_grad():
        logits = ns.SlidingWindowInferer(roi_size=ROI, sw_batch_size=2, overlap=0.5, mode="gaussian")(x[None], net)
    lab = ns.AsDiscreted(keys=["pred"], argmax=True)({"pred": logits[0]})["pred"]
    as_np = lambda t: (t.as_tensor() if hasattr(t, "as_tensor") else t).detach().cpu().numpy()  # noqa: E731
    return {"crop_start": np.asarray(d["foreground_start_coord"]), "crop_end": np.asarray(d["foreground_end_coord"]),
            "cropped_affine": np.asarray(torch.as_tensor(cropped.affine).cpu(), dtype=np.float64),
            "resampled": as_np(x), "resampled_affine": np.asarray(torch.as_tensor(x.affine).cpu(), dtype=np.float64),
            "logits": as_np(logits), "label": as_np(lab)}


def case_ct_pipeline_vs_reference(device):
    from types import SimpleNamespace

    import dynunet_cases as dc
    from monai_amd.data import MetaTensor
    from monai_amd.inferers import SlidingWindowInferer
    from monai_amd.networks.nets import DynUNet
    from monai_amd.transforms import AsDiscreted, CropForegroundd, ScaleIntensityRanged, Spacingd

    g = np.load(os.path.join(GOLDEN, "pipeline_ct.npz"))
    net, _ = dc.build(DynUNet, "basic")
    ns = SimpleNamespace(MetaTensor=MetaTensor, ScaleIntensityRanged=ScaleIntensityRanged, CropForegroundd=CropForegroundd, Spacingd=Spacingd,
                         SlidingWindowInferer=SlidingWindowInferer, AsDiscreted=AsDiscreted)
    got = run_pipeline(ns, net.to(device), device)
    np.testing.assert_array_equal(got["crop_start"], g["crop_start"])
    np.testing.assert_array_equal(got["crop_end"], g["crop_end"])
    np.testing.assert_allclose(got["cropped_affine"], g["cropped_affine"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(got["resampled_affine"], g["resampled_affine"], rtol=0, atol=1e-9)
    assert got["resampled"].shape == g["resampled"].shape
    dr = float(np.abs(got["resampled"] - g["resampled"]).max())
    dl = float(np.abs(got["logits"] - g["logits"]).max())
    mism = got["label"] != g["label"]
    top2 = np.sort(g["logits"][0], axis=0)[-2:]
    worst_margin = float((top2[1] - top2[0])[mism[0]].max()) if mism.any() else 0.0
    assert dr < 2e-6 and dl < 1e-4, (dr, dl)
    assert worst_margin < 2e-4, (int(mism.sum()), worst_margin)
    return {"max_resampled_diff": dr, "max_logit_diff": dl, "label_mismatches": int(mism.sum()), "voxels": int(mism.size)}
