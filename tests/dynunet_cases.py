"""DynUNet cases shared by the golden generator (real reference, CPU) and the emulator / MI355X tests."""
import hashlib
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LOGIT_TOL = 1e-4   # BASELINE.json north_star: fp32 logits within 1e-4 of the reference

K3 = [3, 3, 3, 3]
CFGS = {
    # the nnU-Net shape: basic blocks, affine instance norm, leaky ReLU 0.01, matrix-core convolutions at every level
    "basic": dict(kw=dict(kernel_size=K3, strides=[1, 2, 2, 2], upsample_kernel_size=[2, 2, 2], filters=[16, 32, 48, 64]),
                  shape=(2, 1, 32, 32, 32), seed=11),
    # residual blocks (strided 1x1 shortcut), deep-supervision heads present, few channels at the top (direct kernel)
    "res_ds": dict(kw=dict(kernel_size=K3, strides=[1, 2, 2, 2], upsample_kernel_size=[2, 2, 2], filters=[8, 16, 32, 64], res_block=True,
                           deep_supervision=True, deep_supr_num=2),
                   shape=(1, 2, 16, 24, 32), seed=12),
    # three levels, first block strided, plain instance norm, ReLU, biased transposed convs, per-block sequences for kernel / stride
    "stride0": dict(kw=dict(kernel_size=[[3, 3, 3]] * 3, strides=[2, [2, 2, 2], 2], upsample_kernel_size=[[2, 2, 2], 2], norm_name="instance",
                            act_name="relu", trans_bias=True),
                    shape=(1, 1, 16, 16, 24), seed=13),
    # an anisotropic nnU-Net plan (thick slices along the first axis): in-plane kernels / strides first, residual blocks with anisotropic
    # strided shortcuts, kernel == stride transposed convs of (1, 2, 2) and (2, 2, 1)
    "aniso": dict(kw=dict(kernel_size=[[1, 3, 3], [1, 3, 3], [3, 3, 3], [3, 3, 3]], strides=[[1, 1, 1], [1, 2, 2], [2, 2, 2], [2, 2, 1]],
                          upsample_kernel_size=[[1, 2, 2], [2, 2, 2], [2, 2, 1]], filters=[16, 32, 48, 64], res_block=True),
                  shape=(1, 1, 12, 32, 24), seed=14),
    "aniso_basic": dict(kw=dict(kernel_size=[[3, 3, 1], [3, 3, 3], [3, 3, 3]], strides=[1, [2, 2, 1], [2, 2, 2]],
                                upsample_kernel_size=[[2, 2, 1], [2, 2, 2]], filters=[8, 16, 32]),
                        shape=(2, 1, 16, 24, 6), seed=15),
}
IN_CH = {"basic": 1, "res_ds": 2, "stride0": 1, "aniso": 1, "aniso_basic": 1, "2d_basic": 1, "2d_res_ds": 2, "2d_aniso": 1}
# 2-D networks (SliceInferer needs a product 2-D net): the 3-D engine on one plane -- kernel (1, k, k), stride (1, s, s)
CFGS_2D = {
    "2d_basic": dict(kw=dict(kernel_size=[3, 3, 3, 3], strides=[1, 2, 2, 2], upsample_kernel_size=[2, 2, 2], filters=[16, 32, 32, 64]),
                     shape=(2, 1, 48, 64), seed=21),
    "2d_res_ds": dict(kw=dict(kernel_size=[3, 3, 3], strides=[1, 2, 2], upsample_kernel_size=[2, 2], filters=[8, 16, 32], res_block=True,
                              deep_supervision=True, deep_supr_num=1),
                      shape=(1, 2, 32, 40), seed=22),
    "2d_aniso": dict(kw=dict(kernel_size=[[3, 3], [3, 3], [3, 1]], strides=[[1, 1], [2, 2], [2, 1]], upsample_kernel_size=[[2, 2], [2, 1]],
                             filters=[8, 8, 16], act_name="relu", trans_bias=True),
                     shape=(1, 1, 32, 24), seed=23),
}
CFGS.update(CFGS_2D)
SLICE = dict(roi_size=(32, 32), sw_batch_size=3, spatial_dim=0, overlap=0.5, mode="gaussian")


def slice_volume():
    return torch.rand((1, 1, 5, 48, 40), generator=torch.Generator().manual_seed(802))


def digest(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def build(cls, name):
    """seeded construction -> (eval net with perturbed norm affine / bias parameters, digest of the fresh state_dict)"""
    c = CFGS[name]
    torch.manual_seed(c["seed"])
    net = cls(spatial_dims=2 if name in CFGS_2D else 3, in_channels=IN_CH[name], out_channels=3, **c["kw"])
    init = digest(net.state_dict())
    gen = torch.Generator().manual_seed(500 + c["seed"])
    with torch.no_grad():
        seen = set()
        for k, v in net.state_dict().items():     # the default norm affine / bias values (1 / 0) would hide a swapped or dropped parameter
            if v.data_ptr() in seen:
                continue
            seen.add(v.data_ptr())
            if ".norm" in k and k.endswith("weight"):
                v.copy_(1.0 + 0.2 * torch.randn(v.shape, generator=gen))
            elif k.endswith("bias"):
                v.copy_(0.1 * torch.randn(v.shape, generator=gen))
    return net.eval(), init


def inputs(name):
    return torch.rand(CFGS[name]["shape"], generator=torch.Generator().manual_seed(700 + CFGS[name]["seed"]))


def case_dynunet_vs_reference(device, names=tuple(n for n in CFGS if n not in CFGS_2D), golden="dynunet.npz"):
    """DynUNet against the real reference's output (tests/golden/make_golden_dynunet.py): state_dict keys, the same weights from the
    same seed, logits within 1e-4, identical argmax outside near-ties."""
    from monai_amd.networks.nets import DynUNet

    g = np.load(os.path.join(GOLDEN, golden))
    out = {}
    for name in names:
        net, init = build(DynUNet, name)
        assert list(net.state_dict().keys()) == list(g[f"{name}_keys"]), name
        assert init == str(g[f"{name}_init_sha256"]), f"{name}: same seed must give the reference's weights"
        y = net.to(device)(inputs(name).to(device)).cpu()
        exp = torch.from_numpy(g[f"{name}_out"])
        assert y.shape == exp.shape, (name, y.shape, exp.shape)
        err = (y.double() - exp.double()).abs().max().item()
        assert err < LOGIT_TOL, (name, err)
        top2 = exp.topk(2, dim=1).values
        mism = y.argmax(1) != exp.argmax(1)
        assert not mism.any() or float((top2[:, 0] - top2[:, 1])[mism].max()) < 2 * LOGIT_TOL, name
        out[name] = err
    return out


SW = dict(roi_size=(32, 32, 32), sw_batch_size=2, overlap=0.5, mode="gaussian")


def sw_volume():
    return torch.rand((1, 1, 48, 48, 40), generator=torch.Generator().manual_seed(801))


def case_dynunet_sliding_window(device):
    """SlidingWindowInferer over DynUNet ("basic" and the half-resolution "stride0" net) against the reference inferer + reference net."""
    from monai_amd.inferers import SlidingWindowInferer
    from monai_amd.networks.nets import DynUNet

    g = np.load(os.path.join(GOLDEN, "dynunet.npz"))
    out = {}
    for name in ("basic", "stride0"):
        net, _ = build(DynUNet, name)
        y = SlidingWindowInferer(**SW)(sw_volume().to(device), net.to(device)).cpu()
        exp = torch.from_numpy(g[f"{name}_sw_out"])
        assert y.shape == exp.shape, (name, y.shape, exp.shape)
        out[name] = (y.double() - exp.double()).abs().max().item()
        assert out[name] < LOGIT_TOL, (name, out[name])
    return out


def case_dynunet_2d_vs_reference(device):
    """2-D DynUNet (the 3-D engine on one plane) and SliceInferer over it, against the real reference (tests/golden/make_golden_dynunet2d.py):
    state_dict keys, same seed => same weights, logits within 1e-4; SliceInferer(spatial_dim=0) over a 5-slice volume."""
    from monai_amd.inferers import SliceInferer
    from monai_amd.networks.nets import DynUNet

    out = case_dynunet_vs_reference(device, names=tuple(CFGS_2D), golden="dynunet2d.npz")
    g = np.load(os.path.join(GOLDEN, "dynunet2d.npz"))
    net, _ = build(DynUNet, "2d_basic")
    y = SliceInferer(**SLICE)(slice_volume().to(device), net.to(device)).cpu()
    exp = torch.from_numpy(g["2d_basic_slice_out"])
    assert y.shape == exp.shape, (y.shape, exp.shape)
    out["slice_inferer"] = (y.double() - exp.double()).abs().max().item()
    assert out["slice_inferer"] < LOGIT_TOL, out
    return out


def case_dynunet_api(device):
    import pytest

    from monai_amd.networks.nets import DynUNet

    with pytest.raises(ValueError):
        DynUNet(3, 1, 2, [3, 3], [1, 2], [2])
    with pytest.raises(ValueError):
        DynUNet(3, 1, 2, K3, [1, 2, 2, 2], [2, 2, 2], filters=[8, 16])
    with pytest.raises(ValueError):
        DynUNet(3, 1, 2, K3, [1, 2, 2, 2], [2, 2, 2], deep_supervision=True, deep_supr_num=3)
    with pytest.raises(NotImplementedError):
        DynUNet(3, 1, 2, [3, [3, 3, 5], 3], [1, 2, 2], [2, 2])
    with pytest.raises(NotImplementedError):
        DynUNet(3, 1, 2, [3, 3, 3], [1, 2, [2, 2, 1]], [2, 2])        # upsample kernels must equal the strides
    with pytest.raises(NotImplementedError):
        DynUNet(1, 1, 2, K3, [1, 2, 2, 2], [2, 2, 2])
    with pytest.raises(ValueError):
        DynUNet(2, 1, 2, [[3, 3, 3]] * 3, [1, 2, 2], [2, 2])             # 3-sequences for a 2-D net
    net = DynUNet(3, 1, 2, [3, 3, 3], [1, 2, 2], [2, 2], filters=[8, 8, 8])
    with pytest.raises(NotImplementedError):
        net(torch.zeros(1, 1, 8, 8, 8, device=device))           # training mode: inference engine only
    with pytest.raises(NotImplementedError):
        net.eval().to(device)(torch.zeros(1, 1, 10, 8, 8, device=device))


def case_dynunet_wide_concat(device, cin=288, cout=32, dims=(3, 8, 12)):
    """the engine's convolution of a concat with more input channels than the split-precision kernel keeps records for (nnU-Net's 512-channel 12^3 level): evaluated as
    two halves of the input channels, the second ADDED onto the first with the statistics of the sum -- against ATen in float64, records included"""
    import torch.nn as nn
    import torch.nn.functional as F

    from monai_amd import ops
    from monai_amd.networks.nets import DynUNet

    gen = torch.Generator().manual_seed(97)
    net = DynUNet(3, 1, 2, [3, 3, 3], [1, 2, 2], [2, 2], filters=[8, 8, 8]).eval().to(device)
    conv = nn.Conv3d(cin, cout, 3, padding=1, bias=False)
    norm = nn.InstanceNorm3d(cout, affine=True)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) / np.sqrt(27.0 * cin))
        norm.weight.copy_(1.0 + 0.2 * torch.randn(cout, generator=gen))
        norm.bias.copy_(0.1 * torch.randn(cout, generator=gen))
    conv, norm = conv.to(device), norm.to(device)
    n = 2
    x = torch.randn((n, cin) + tuple(dims), generator=gen)
    rec = torch.zeros(n, cin, 4)
    rec[:, :, 0] = torch.rand(n, cin, generator=gen) + 0.5
    rec[:, :, 1] = torch.randn(n, cin, generator=gen) * 0.3
    rec[:, :, 2] = 0.01
    a, b = rec[:, :, 0][:, :, None, None, None].double(), rec[:, :, 1][:, :, None, None, None].double()
    y = x.double() * a + b
    y = torch.where(y > 0, y, y * 0.01)
    rec[:, :, 3] = y.abs().amax(dim=(2, 3, 4)).float() * 1.5
    exp = F.conv3d(y, conv.weight.detach().cpu().double(), None, padding=1)
    calls = []
    orig = ops.conv3d_k3

    def spy(cfg, *args, **kw):
        calls.append((cfg, bool(kw.get("accumulate", False))))
        return orig(cfg, *args, **kw)

    ops.conv3d_k3 = spy
    try:
        with torch.no_grad():
            out, nrm = net._conv_norm(conv, norm, x.to(device), rec.to(device), (1, 1, 1), 0.01)
    finally:
        ops.conv3d_k3 = orig
    h2 = ops.conv3d_k3_h2_config()
    assert calls == [(h2, False), (h2, True)], calls
    got = out.cpu().double()
    err = (got - exp).abs().max().item()
    assert err < 2e-5 * max(1.0, exp.abs().max().item()), err
    mean, var = got.mean(dim=(2, 3, 4)), got.var(dim=(2, 3, 4), unbiased=False)
    alpha = norm.weight.detach().cpu().double()[None] / torch.sqrt(var + norm.eps)
    r = nrm.cpu().double()
    assert (r[:, :, 0] - alpha).abs().max().item() < 1e-5 * alpha.abs().max().item() + 1e-6
    assert (r[:, :, 1] - (norm.bias.detach().cpu().double()[None] - mean * alpha)).abs().max().item() < 2e-5
    return err


def case_nets_window_vs_oracle(device, edge=96, filters=(32, 64, 128, 256)):
    """DynUNet (nnU-Net filters) and SegResNet on one BASELINE-sized window (edge^3) against the CPU oracle (oracle/dynunet.py, itself pinned
    to the reference's goldens): the large-plane convolution configurations that the 32^3 goldens do not reach."""
    from monai_amd.networks.nets import DynUNet, SegResNet
    from oracle import dynunet as od

    out = {}
    x = torch.rand((1, 1, edge, edge, edge), generator=torch.Generator().manual_seed(4096))
    torch.manual_seed(41)
    net = DynUNet(3, 1, 5, [3] * len(filters), [1] + [2] * (len(filters) - 1), [2] * (len(filters) - 1), filters=list(filters)).eval()
    with torch.no_grad():
        exp = od.dynunet_forward(net.state_dict(), x, [1] + [2] * (len(filters) - 1))
    y = net.to(device)(x.to(device)).cpu()
    out["dynunet"] = (y.double() - exp.double()).abs().max().item()
    torch.manual_seed(42)
    seg = SegResNet(init_filters=filters[0] // 2, in_channels=1, out_channels=5, blocks_down=(1, 2, 2), blocks_up=(1, 1)).eval()
    with torch.no_grad():
        exp = od.segresnet_forward(seg.state_dict(), x, (1, 2, 2), (1, 1))
    y = seg.to(device)(x.to(device)).cpu()
    out["segresnet"] = (y.double() - exp.double()).abs().max().item()
    assert max(out.values()) < LOGIT_TOL, out
    return out
