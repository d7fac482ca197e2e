"""SegResNet cases shared by the golden generator (real reference, CPU) and the emulator / MI355X tests."""
import os

import numpy as np
import torch

from dynunet_cases import LOGIT_TOL, digest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CFGS = {
    # the reference's defaults (8 filters: one channel per group at the top level, the direct kernel for the 8 -> 8 convolutions)
    "default": dict(kw=dict(in_channels=1, out_channels=2), shape=(1, 1, 16, 24, 32), seed=21),
    # 16 filters: groups of 2 .. 8 channels, matrix-core convolutions with fused statistics, batch of two
    "f16": dict(kw=dict(init_filters=16, in_channels=2, out_channels=3, blocks_down=(1, 2, 2), blocks_up=(1, 1), dropout_prob=0.2),
                shape=(2, 2, 32, 32, 32), seed=22),
    # instance norm (affine), leaky ReLU, transposed-conv upsampling
    "deconv_inst": dict(kw=dict(init_filters=8, in_channels=1, out_channels=4, act=("leakyrelu", {"negative_slope": 0.05}),
                                norm=("instance", {"affine": True}), blocks_down=(1, 1, 2), blocks_up=(2, 1), upsample_mode="deconv"),
                        shape=(1, 1, 16, 16, 24), seed=23),
    # no final convolution: the decoder features themselves
    "features": dict(kw=dict(init_filters=16, in_channels=1, out_channels=2, blocks_down=(1, 1), blocks_up=(1,), use_conv_final=False),
                     shape=(1, 1, 8, 16, 16), seed=24),
}


def build(cls, name):
    c = CFGS[name]
    torch.manual_seed(c["seed"])
    net = cls(spatial_dims=3, **c["kw"])
    init = digest(net.state_dict())
    gen = torch.Generator().manual_seed(900 + c["seed"])
    with torch.no_grad():
        for k, v in net.state_dict().items():       # non-default norm affine / bias values, so a dropped or swapped parameter shows
            if "norm" in k or k.startswith("conv_final.0"):
                v.copy_((1.0 if k.endswith("weight") else 0.0) + 0.2 * torch.randn(v.shape, generator=gen))
            elif k.endswith("bias"):
                v.copy_(0.1 * torch.randn(v.shape, generator=gen))
    return net.eval(), init


def inputs(name):
    return torch.rand(CFGS[name]["shape"], generator=torch.Generator().manual_seed(950 + CFGS[name]["seed"]))


def case_segresnet_vs_reference(device, names=tuple(CFGS)):
    """SegResNet against the real reference's output (tests/golden/make_golden_segresnet.py): state_dict keys, the same weights
    from the same seed, outputs within 1e-4."""
    from monai_amd.networks.nets import SegResNet

    g = np.load(os.path.join(GOLDEN, "segresnet.npz"))
    out = {}
    for name in names:
        net, init = build(SegResNet, name)
        assert list(net.state_dict().keys()) == list(g[f"{name}_keys"]), name
        assert init == str(g[f"{name}_init_sha256"]), f"{name}: same seed must give the reference's weights"
        y = net.to(device)(inputs(name).to(device)).cpu()
        exp = torch.from_numpy(g[f"{name}_out"])
        assert y.shape == exp.shape, (name, y.shape, exp.shape)
        out[name] = (y.double() - exp.double()).abs().max().item()
        assert out[name] < LOGIT_TOL, (name, out[name])
    return out


def case_segresnet_sliding_window(device):
    from dynunet_cases import SW, sw_volume
    from monai_amd.inferers import SlidingWindowInferer
    from monai_amd.networks.nets import SegResNet

    g = np.load(os.path.join(GOLDEN, "segresnet.npz"))
    net, _ = build(SegResNet, "default")
    y = SlidingWindowInferer(**SW)(sw_volume().to(device), net.to(device)).cpu()
    err = (y.double() - torch.from_numpy(g["default_sw_out"]).double()).abs().max().item()
    assert err < LOGIT_TOL, err
    return err


def case_segresnet_api(device):
    import pytest

    from monai_amd.networks.nets import SegResNet

    with pytest.raises(ValueError):
        SegResNet(spatial_dims=4)
    with pytest.raises(ValueError):
        SegResNet(norm_name="batch")
    with pytest.raises(NotImplementedError):
        SegResNet(spatial_dims=2)
    with pytest.raises(NotImplementedError):
        SegResNet(norm="batch")
    with pytest.raises(NotImplementedError):
        SegResNet(upsample_mode="pixelshuffle")
    net = SegResNet()
    with pytest.raises(NotImplementedError):
        net(torch.zeros(1, 1, 8, 8, 8, device=device))
    with pytest.raises(NotImplementedError):
        net.eval().to(device)(torch.zeros(1, 1, 12, 8, 8, device=device))
