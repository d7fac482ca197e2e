"""One bundle-shaped pipeline through the drop-in classes: Spacingd -> GaussianSmoothd -> SlidingWindowInferer(BasicUNet)
-> Activationsd(softmax) -> AsDiscreted(argmax), against the same chain of the REAL reference classes on the CPU
(tests/golden/make_golden_pipeline.py).  Every stage is pinned on its own elsewhere; this checks that they compose."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
AFFINE = np.diag([1.25, 1.0, 0.8, 1.0])
ROI = (32, 32, 32)


def volume():
    gen = torch.Generator().manual_seed(4242)
    return torch.rand(1, 36, 44, 52, generator=gen)


def run_pipeline(ns, net, device):
    """`ns`: namespace with MetaTensor, Spacingd, GaussianSmoothd, SlidingWindowInferer, Activationsd, AsDiscreted."""
    img = ns.MetaTensor(volume().to(device), affine=AFFINE)
    d = {"image": img}
    d = ns.Spacingd(keys=["image"], pixdim=(1.0, 1.0, 1.0), mode="bilinear", padding_mode="border")(d)
    d = ns.GaussianSmoothd(keys=["image"], sigma=0.6)(d)
    x = d["image"]
    with torch.no_grad():
        logits = ns.SlidingWindowInferer(roi_size=ROI, sw_batch_size=2, overlap=0.5, mode="gaussian")(x[None], net)
    d = {"pred": logits[0]}
    d = ns.Activationsd(keys=["pred"], softmax=True)(d)
    prob = d["pred"]
    d = ns.AsDiscreted(keys=["pred"], argmax=True)(d)
    as_np = lambda t: (t.as_tensor() if hasattr(t, "as_tensor") else t).detach().cpu().numpy()  # noqa: E731
    return {"resampled_shape": np.asarray(x.shape), "logits": as_np(logits), "prob": as_np(prob), "label": as_np(d["pred"])}


def case_pipeline_vs_reference(device):
    from types import SimpleNamespace

    import e2e_cases as ec
    from monai_amd.data import MetaTensor
    from monai_amd.inferers import SlidingWindowInferer
    from monai_amd.transforms import Activationsd, AsDiscreted, GaussianSmoothd, Spacingd

    g = np.load(os.path.join(GOLDEN, "pipeline.npz"))
    net, _ = ec.make_net(11, 1, 3, device, features=(16, 16, 32, 32, 64, 16))
    ns = SimpleNamespace(MetaTensor=MetaTensor, Spacingd=Spacingd, GaussianSmoothd=GaussianSmoothd, SlidingWindowInferer=SlidingWindowInferer,
                         Activationsd=Activationsd, AsDiscreted=AsDiscreted)
    got = run_pipeline(ns, net, device)
    assert tuple(got["resampled_shape"]) == tuple(g["resampled_shape"])
    dl = float(np.abs(got["logits"] - g["logits"]).max())
    dp = float(np.abs(got["prob"] - g["prob"]).max())
    mism = got["label"] != g["label"]
    top2 = np.sort(g["logits"][0], axis=0)[-2:]
    margin = top2[1] - top2[0]
    worst_margin = float(margin[mism[0]].max()) if mism.any() else 0.0
    assert dl < 1e-4 and dp < 1e-4, (dl, dp)
    assert worst_margin < 2e-4, (int(mism.sum()), worst_margin)        # labels may only differ where the reference's own top-2 logits tie
    return {"max_logit_diff": dl, "max_prob_diff": dp, "label_mismatches": int(mism.sum()), "voxels": int(mism.size), "worst_margin": worst_margin}
