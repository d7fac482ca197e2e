"""Golden vectors for lazy resampling from the REAL reference (build container only):
    PYTHONPATH=/root/reference python tests/golden/make_golden_lazy.py
`Compose([...], lazy=True)` of monai/transforms/compose.py on CPU: the chains of SURVEY.md 8f-2 -- an orientation change (LPS <-> RAS is
a flip of two axes; the reference's own Orientationd needs nibabel, absent here, so the flip is spelled Flipd), Spacingd,
CropForegroundd, SpatialPadd -- executed lazily (Flip and Spacing fuse into ONE resampling; CropForeground reads the current data, so
its crop + pad and the following pad form a second, interpolation-free group).  Stores inputs, outputs, affines and the eager results."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from monai.data import MetaTensor  # noqa: E402
from monai.transforms import Compose, CropForegroundd, Flipd, Rotate90d, SpatialPadd, Spacingd  # noqa: E402

out = {}
torch.manual_seed(0)
vol = torch.zeros(1, 28, 30, 26)
vol[:, 5:22, 6:25, 4:20] = torch.rand(1, 17, 19, 16) + 0.1
lab = (vol > 0.6).float()
aff = torch.tensor([[-0.8, 0.0, 0.0, 10.0], [0.0, -0.9, 0.0, 12.0], [0.0, 0.0, 1.7, -5.0], [0.0, 0.0, 0.0, 1.0]], dtype=torch.float64)
out["image"], out["label"], out["affine"] = vol.numpy(), lab.numpy(), aff.numpy()


def chains():
    return {
        "a": [Flipd(("image", "label"), spatial_axis=[0, 1]), Spacingd(("image", "label"), pixdim=(1.0, 1.0, 1.0), mode=("bilinear", "nearest"))],
        "b": [Flipd(("image", "label"), spatial_axis=[0, 1]), Spacingd(("image", "label"), pixdim=(1.1, 0.7, 1.3), mode=("bilinear", "nearest")),
              CropForegroundd(("image", "label"), source_key="image", margin=2), SpatialPadd(("image", "label"), spatial_size=(48, 48, 48))],
        "c": [Rotate90d(("image", "label"), k=1, spatial_axes=(0, 2)), Flipd(("image", "label"), spatial_axis=1),
              SpatialPadd(("image", "label"), spatial_size=(32, 34, 36))],
    }


for lazy in (True, False):
    for name, ts in chains().items():
        d = {"image": MetaTensor(vol.clone(), affine=aff.clone()), "label": MetaTensor(lab.clone(), affine=aff.clone())}
        r = Compose(ts, lazy=lazy)(d)
        tag = f"{name}_{'lazy' if lazy else 'eager'}"
        for k in ("image", "label"):
            assert not r[k].pending_operations
            out[f"{tag}_{k}"] = r[k].as_tensor().numpy()
            out[f"{tag}_{k}_affine"] = r[k].affine.numpy()
        out[f"{tag}_ops"] = np.array([op["class"] for op in r["image"].applied_operations])
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lazy.npz"), **out)
for k, v in out.items():
    print(k, v.shape, v.dtype)
