"""SwinUNETR against the real reference (tests/golden/swin_unetr.npz, tests/golden/make_golden_swin.py): same
``state_dict`` keys, same seed -> same weights (digest), logits within the fp32 tolerance of the other networks."""
import hashlib
import os

import numpy as np
import torch

from monai_amd.networks.nets import SwinUNETR

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "swin_unetr.npz")
CASES = {
    "a": dict(kw=dict(in_channels=1, out_channels=3, feature_size=24), shape=(1, 1, 64, 64, 64), seed=5),
    "b": dict(kw=dict(in_channels=2, out_channels=4, feature_size=24, downsample="mergingv2", qkv_bias=False, normalize=False), shape=(2, 2, 64, 64, 96), seed=6),
    "c": dict(kw=dict(in_channels=1, out_channels=5, feature_size=48), shape=(1, 1, 96, 96, 96), seed=7),
}
TOL = 1e-4


def case_swin_unetr_vs_golden(device, names=("a", "b", "c")):
    g = np.load(GOLDEN)
    res = {}
    for name in names:
        c = CASES[name]
        torch.manual_seed(c["seed"])
        net = SwinUNETR(**c["kw"]).eval()
        sd = net.state_dict()
        assert sorted(sd) == list(g[f"{name}_keys"]), name
        h = hashlib.sha256()
        tabs = []
        for k in sorted(sd):
            if k.endswith("relative_position_bias_table"):      # truncated normal: erfinv_ differs in the last bit between CPU vector ISAs
                tabs.append(sd[k].detach().cpu().numpy().reshape(-1))
                continue
            h.update(k.encode())
            h.update(sd[k].detach().cpu().numpy().tobytes())
        assert h.digest() == bytes(g[f"{name}_digest"]), f"{name}: same seed must give the reference's initial weights"
        assert np.allclose(np.concatenate(tabs), g[f"{name}_tables"], rtol=0, atol=1e-6), name
        torch.manual_seed(100 + c["seed"])
        x = torch.rand(c["shape"])
        assert abs(float(x.double().sum()) - float(g[f"{name}_xsum"])) < 1e-6
        y = net.to(device)(x.to(device)).cpu()
        st = int(g[f"{name}_stride"])
        err = float((y[:, :, ::st, ::st, ::st] - torch.as_tensor(g[f"{name}_y"])).abs().max())
        assert err < TOL, (name, err)
        res[name] = err
    return res
