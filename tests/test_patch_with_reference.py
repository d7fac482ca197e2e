"""Drop-in check against the real MONAI (only where /root/reference exists, i.e. the build container): after
``monai_amd.patch.install()`` the reference's own bundle machinery resolves ``"_target_": "SlidingWindowInferer"`` /
``"BasicUNet"`` / ``"Spacingd"`` to the MI355X classes, reference checkpoints load into our BasicUNet, and uninstall
restores everything."""
import os
import sys

import pytest
import torch

REF = "/root/reference"
pytestmark = [pytest.mark.fallthrough, pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "monai")), reason="reference MONAI not available here")]


@pytest.fixture()
def monai_ref():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import monai

    yield monai
    import monai_amd.patch as patch

    patch.uninstall()
    sys.path.remove(REF)


def test_bundle_targets_resolve_to_amd_classes(monai_ref):
    import monai_amd.patch as patch
    from monai.bundle import ConfigParser
    from monai_amd.inferers.inferer import SlidingWindowInferer as Ours

    ref_cls = monai_ref.inferers.SlidingWindowInferer
    done = patch.install()
    assert "monai.inferers.inferer.SlidingWindowInferer" in done and "monai._C" in done
    assert monai_ref.inferers.SlidingWindowInferer is Ours and monai_ref.inferers.inferer.SlidingWindowInferer is Ours
    cfg = {
        "inferer": {"_target_": "SlidingWindowInferer", "roi_size": [96, 96, 96], "sw_batch_size": 4, "overlap": 0.5, "mode": "gaussian"},
        "network": {"_target_": "BasicUNet", "spatial_dims": 3, "in_channels": 1, "out_channels": 5},
        "pre": {"_target_": "Spacingd", "keys": ["image"], "pixdim": [1.0, 1.0, 1.0], "mode": "bilinear"},
        "smooth": {"_target_": "GaussianSmoothd", "keys": ["image"], "sigma": 1.0},
        "dyn": {"_target_": "DynUNet", "spatial_dims": 3, "in_channels": 1, "out_channels": 2, "kernel_size": [3, 3, 3], "strides": [1, 2, 2],
                "upsample_kernel_size": [2, 2]},
        "seg": {"_target_": "SegResNet", "spatial_dims": 3, "in_channels": 4, "out_channels": 3},
        "crop": {"_target_": "CropForegroundd", "keys": ["image"], "source_key": "image"},
        "norm": {"_target_": "NormalizeIntensityd", "keys": ["image"], "nonzero": True, "channel_wise": True},
    }
    parser = ConfigParser(cfg)
    assert type(parser.get_parsed_content("inferer")).__module__ == "monai.inferers.inferer"
    assert isinstance(parser.get_parsed_content("inferer"), Ours)
    from monai.inferers import Inferer

    assert isinstance(parser.get_parsed_content("inferer"), Inferer)        # virtual subclass of the reference's ABC
    from monai_amd.networks.nets.basic_unet import BasicUNet as OurNet
    from monai_amd.transforms.intensity.dictionary import GaussianSmoothd as OurSmooth
    from monai_amd.transforms.spatial.dictionary import Spacingd as OurSpacingd

    assert isinstance(parser.get_parsed_content("network"), OurNet)
    assert isinstance(parser.get_parsed_content("pre"), OurSpacingd)
    assert isinstance(parser.get_parsed_content("smooth"), OurSmooth)
    for key in ("dyn", "seg", "crop", "norm"):
        assert type(parser.get_parsed_content(key)).__module__.startswith("monai."), key            # rebound in place ...
        assert sys.modules[type(parser.get_parsed_content(key)).__init__.__module__].__name__.startswith("monai_amd."), key    # ... to the MI355X class
    import monai._C as native

    assert native.BoundType.__members__["reflect"] == 2 and hasattr(native, "grid_pull")
    for fn in ("grid_pull", "grid_push", "grid_count", "grid_grad"):       # the differentiable wrappers and their backends
        import monai.networks.layers as ref_layers
        from monai_amd.networks import layers as our_layers

        assert getattr(ref_layers, fn) is getattr(our_layers, fn), fn
        assert hasattr(native, fn) and hasattr(native, fn + "_backward")
    patch.uninstall()
    assert monai_ref.inferers.SlidingWindowInferer is ref_cls


def test_reference_checkpoint_loads_into_amd_basic_unet(monai_ref):
    from monai.networks.nets import BasicUNet as RefNet

    from monai_amd.networks.nets.basic_unet import BasicUNet as OurNet

    torch.manual_seed(3)
    ref = RefNet(spatial_dims=3, in_channels=1, out_channels=5)
    ours = OurNet(spatial_dims=3, in_channels=1, out_channels=5)
    missing, unexpected = ours.load_state_dict(ref.state_dict(), strict=True)
    assert not missing and not unexpected
    assert list(ours.state_dict().keys()) == list(ref.state_dict().keys())
    torch.manual_seed(3)
    ours2 = OurNet(spatial_dims=3, in_channels=1, out_channels=5)   # same seed -> same initial weights as the reference
    for (k, a), (_, b) in zip(ref.state_dict().items(), ours2.state_dict().items()):
        assert torch.equal(a, b), k


@pytest.mark.parametrize("which", ["UNet", "UNETR"])
def test_reference_checkpoint_loads_into_amd_unet_and_unetr(monai_ref, which):
    import monai_amd.patch as patch
    from monai.bundle import ConfigParser

    if which == "UNet":
        from monai.networks.nets import UNet as RefNet
        from monai_amd.networks.nets.unet import UNet as OurNet
        kw = dict(spatial_dims=3, in_channels=1, out_channels=2, channels=(16, 32, 64, 128, 256), strides=(2, 2, 2, 2), num_res_units=2)
    else:
        from monai.networks.nets import UNETR as RefNet
        from monai_amd.networks.nets.unetr import UNETR as OurNet
        kw = dict(in_channels=1, out_channels=3, img_size=(32, 32, 32), feature_size=16, hidden_size=128, mlp_dim=256, num_heads=2)
    ref = RefNet(**kw)
    ours = OurNet(**kw)
    missing, unexpected = ours.load_state_dict(ref.state_dict(), strict=True)
    assert not missing and not unexpected
    assert list(ours.state_dict().keys()) == list(ref.state_dict().keys())
    patch.install()
    net = ConfigParser({"network": {"_target_": which, **{k: list(v) if isinstance(v, tuple) else v for k, v in kw.items()}}}).get_parsed_content("network")
    assert isinstance(net, OurNet)
    patch.uninstall()


def test_monai_compose_and_invertd_drive_the_amd_transforms(monai_ref, emu):
    """An unchanged bundle's pre-processing ``Compose`` and its ``Invertd(transform="@preprocessing")`` post-processing run on the MI355X classes
    after ``patch.install()`` (virtual ``MapTransform`` / ``InvertibleTransform`` registration): forward result, inverted prediction and its
    affine equal the pure-reference run of the same chain."""
    import numpy as np

    import monai_amd.patch as patch
    import pipeline_ct_case as pc
    from monai.data import MetaTensor

    def run(ns, nearest=False):
        pre = ns.Compose([
            ns.ScaleIntensityRanged(keys=["image"], a_min=-175.0, a_max=250.0, b_min=0.0, b_max=1.0, clip=True),
            ns.CropForegroundd(keys=["image"], source_key="image", margin=2),
            # nearest: spacing ratios without exact .5 source coordinates (a tie may legitimately resolve either way)
            ns.Spacingd(keys=["image"], pixdim=(1.37, 1.11, 1.83) if nearest else (1.5, 1.5, 2.0), mode="bilinear"),
            ns.DivisiblePadd(keys=["image"], k=8),
            ns.Flipd(keys=["image"], spatial_axis=0),
        ])
        post = ns.Compose([ns.Invertd(keys="pred", transform=pre, orig_keys="image", nearest_interp=nearest, to_tensor=True)])
        d = pre({"image": MetaTensor(pc.volume(), affine=pc.AFFINE)})
        x = d["image"]
        d["pred"] = MetaTensor(torch.sin(x.as_tensor() * 3.0), meta=dict(x.meta), applied_operations=[])     # a stand-in prediction
        return x, post(d)["pred"]

    import monai.transforms as T

    x_ref, inv_ref = run(T)
    xn_ref, invn_ref = run(T, nearest=True)
    patch.install()
    from monai_amd.transforms.spatial.dictionary import Spacingd as OurSpacingd

    assert T.Spacingd is OurSpacingd and isinstance(OurSpacingd(keys=["image"], pixdim=(1.0, 1.0, 1.0)), T.MapTransform)
    x_our, inv_our = run(T)
    xn_our, invn_our = run(T, nearest=True)       # Invertd's default: the recorded interpolation modes rewritten to "nearest" (TraceKeys.NONE align_corners)
    patch.uninstall()
    assert float((xn_our.as_tensor() - xn_ref.as_tensor()).abs().max()) < 2e-6
    assert float((torch.as_tensor(invn_our) - torch.as_tensor(invn_ref)).abs().max()) < 2e-6
    assert tuple(x_our.shape) == tuple(x_ref.shape) and tuple(inv_our.shape) == tuple(inv_ref.shape) == (1, 48, 56, 40)
    assert float((x_our.as_tensor() - x_ref.as_tensor()).abs().max()) < 2e-6
    assert float((torch.as_tensor(inv_our) - torch.as_tensor(inv_ref)).abs().max()) < 2e-6
    np.testing.assert_allclose(np.asarray(inv_our.affine), np.asarray(inv_ref.affine), rtol=0, atol=1e-9)


def test_spleen_shaped_bundle_config_runs_on_the_amd_classes(monai_ref, emu):
    """A bundle ``inference.json`` in the shape of MONAI's CT segmentation bundles (Orientationd / Spacingd / ScaleIntensityRanged / CropForegroundd ->
    UNet(norm="batch") under SlidingWindowInferer -> Activationsd / Invertd / AsDiscreted), parsed by the reference's own ``ConfigParser``: every
    ``_target_`` resolves to the MI355X class after ``patch.install()`` and the result equals the reference classes' run of the same chain (the
    volume is already RAS, so the reference chain -- which cannot import nibabel here -- simply omits the no-op Orientationd)."""
    import numpy as np

    import monai_amd.patch as patch
    import pipeline_ct_case as pc
    from monai.bundle import ConfigParser
    from monai.data import MetaTensor

    def config(with_orientation):
        pre = [
            {"_target_": "Spacingd", "keys": "image", "pixdim": [1.5, 1.5, 2.0], "mode": "bilinear"},
            {"_target_": "ScaleIntensityRanged", "keys": "image", "a_min": -57, "a_max": 164, "b_min": 0.0, "b_max": 1.0, "clip": True},
            {"_target_": "CropForegroundd", "keys": "image", "source_key": "image", "k_divisible": 16},
        ]
        if with_orientation:
            pre.insert(0, {"_target_": "Orientationd", "keys": "image", "axcodes": "RAS"})
        return {
            "network_def": {"_target_": "UNet", "spatial_dims": 3, "in_channels": 1, "out_channels": 2, "channels": [16, 32, 64], "strides": [2, 2],
                            "num_res_units": 2, "norm": "batch"},
            "preprocessing": {"_target_": "Compose", "transforms": pre},
            "inferer": {"_target_": "SlidingWindowInferer", "roi_size": [32, 32, 32], "sw_batch_size": 2, "overlap": 0.5},
            "postprocessing": {"_target_": "Compose", "transforms": [
                {"_target_": "Activationsd", "keys": "pred", "softmax": True},
                {"_target_": "Invertd", "keys": "pred", "transform": "@preprocessing", "orig_keys": "image", "nearest_interp": False, "to_tensor": True},
                {"_target_": "AsDiscreted", "keys": "pred", "argmax": True},
            ]},
        }

    def run(with_orientation):
        parser = ConfigParser(config(with_orientation))
        torch.manual_seed(21)
        net = parser.get_parsed_content("network_def").eval()
        with torch.no_grad():
            for k, v in net.state_dict().items():          # non-trivial batch-norm statistics
                if k.endswith("running_var"):
                    v.fill_(0.8)
                elif k.endswith("running_mean"):
                    v.fill_(0.05)
        pre, post, inferer = (parser.get_parsed_content(k) for k in ("preprocessing", "postprocessing", "inferer"))
        d = pre({"image": MetaTensor(pc.volume(), affine=pc.AFFINE)})
        with torch.no_grad():
            logits = inferer(d["image"][None], net)
        d["pred"] = MetaTensor(logits[0], meta=dict(d["image"].meta), applied_operations=[])
        out = post(d)
        return parser, logits, out["pred"]

    _, logits_ref, label_ref = run(False)
    patch.install()
    parser, logits_our, label_our = run(True)
    names = {k: type(parser.get_parsed_content(k)).__module__ for k in ("network_def", "inferer")}
    pre_mods = [type(t).__module__ for t in parser.get_parsed_content("preprocessing").transforms]
    patch.uninstall()
    from monai_amd.networks.nets.unet import UNet as OurUNet

    assert isinstance(parser.get_parsed_content("network_def"), OurUNet) and names["inferer"] == "monai.inferers.inferer"
    assert all(m.startswith("monai.transforms.") for m in pre_mods)            # rebound in place: ComponentLocator found them by short name
    assert tuple(label_our.shape) == tuple(label_ref.shape) == (1, 48, 56, 40)
    assert float((logits_our - logits_ref).abs().max()) < 1e-4
    mism = torch.as_tensor(label_our) != torch.as_tensor(label_ref)
    assert int(mism.sum()) <= 8, int(mism.sum())       # argmax after a resampling inverse: ties at interpolated probabilities ~0.5 only


def test_dataset_loader_decollate_invertd_flow(monai_ref, emu):
    """The evaluator-shaped data flow of a bundle -- ``Dataset(transform=pre)`` -> ``DataLoader`` (MetaTensor collate) -> network on the batch ->
    ``decollate_batch`` -> per-item ``Invertd`` + ``AsDiscreted`` -- over the patched classes: the records the MI355X transforms push on
    ``applied_operations`` survive MONAI's collate / decollate, and every label map equals the pure-reference run."""
    import monai_amd.patch as patch
    import pipeline_ct_case as pc
    from monai.data import DataLoader, Dataset, MetaTensor, decollate_batch

    def run(ns):
        pre = ns.Compose([
            ns.ScaleIntensityRanged(keys=["image"], a_min=-175.0, a_max=250.0, b_min=0.0, b_max=1.0, clip=True),
            ns.Spacingd(keys=["image"], pixdim=(1.37, 1.11, 1.83), mode="bilinear"),
            ns.CenterSpatialCropd(keys=["image"], roi_size=(24, 24, 24)),
            ns.SpatialPadd(keys=["image"], spatial_size=(32, 32, 32)),
        ])
        post = ns.Compose([ns.Invertd(keys="pred", transform=pre, orig_keys="image", nearest_interp=True, to_tensor=True),
                           ns.AsDiscreted(keys="pred", threshold=0.5)])
        items = [{"image": MetaTensor(pc.volume() + 10.0 * i, affine=pc.AFFINE)} for i in range(2)]
        batch = next(iter(DataLoader(Dataset(items, transform=pre), batch_size=2, num_workers=0)))
        batch["pred"] = torch.sigmoid(batch["image"] * 4.0 - 2.0)            # a stand-in network output on the batch
        return batch["image"], [post(i)["pred"] for i in decollate_batch(batch)]

    import monai.transforms as T

    x_ref, p_ref = run(T)
    patch.install()
    x_our, p_our = run(T)
    patch.uninstall()
    assert tuple(x_our.shape) == tuple(x_ref.shape) == (2, 1, 32, 32, 32)
    assert float((torch.as_tensor(x_our) - torch.as_tensor(x_ref)).abs().max()) < 2e-6
    for a, b in zip(p_our, p_ref):
        assert tuple(a.shape) == tuple(b.shape) == (1, 48, 56, 40) and int((torch.as_tensor(a) != torch.as_tensor(b)).sum()) == 0


def test_reference_checkpoints_load_into_amd_dynunet_and_segresnet(monai_ref, emu):
    """``state_dict``s of the REAL DynUNet (deep supervision, residual blocks: includes the re-registered ``skip_layers.*`` entries) and SegResNet load
    with ``strict=True``; the loaded nets reproduce the reference's outputs."""
    import dynunet_cases as dc
    import segresnet_cases as sc
    from monai.networks.nets import DynUNet as RefDyn
    from monai.networks.nets import SegResNet as RefSeg

    from monai_amd.networks.nets import DynUNet, SegResNet

    for ref_cls, our_cls, cases, name, in_kw in ((RefDyn, DynUNet, dc, "res_ds", dict(spatial_dims=3, in_channels=2, out_channels=3)),
                                                 (RefSeg, SegResNet, sc, "f16", dict(spatial_dims=3))):
        torch.manual_seed(99)
        ref = ref_cls(**in_kw, **cases.CFGS[name]["kw"]).eval()
        torch.manual_seed(1234)                        # different initial weights: everything must come from the checkpoint
        ours = our_cls(**in_kw, **cases.CFGS[name]["kw"]).eval()
        missing, unexpected = ours.load_state_dict(ref.state_dict(), strict=True)
        assert not missing and not unexpected
        x = cases.inputs(name)
        with torch.no_grad():
            assert float((ours(x) - ref(x)).abs().max()) < 1e-4, name


def test_reference_compose_lazy_runs_the_fused_resampling(monai_ref, emu):
    """SURVEY 8f-2 through the reference's own machinery: after install() `monai.transforms.Compose(..., lazy=True)` sees the MI355X
    transforms as LazyTrait instances, lets them record their affines, and executes the pending group through the rebound
    `monai.transforms.lazy.utils.resample` = ONE `mh_affine_resample_f32` launch per image.  Results equal the pure-reference lazy run
    (tests/golden/lazy.npz)."""
    import numpy as np

    import lazy_cases as lc
    import monai_amd.patch as patch

    patch.install()
    from monai.data import MetaTensor
    from monai.transforms import Compose, CropForegroundd, Flipd, SpatialPadd, Spacingd
    from monai.transforms.traits import LazyTrait

    assert getattr(Spacingd, "_mh_is_product", False) and isinstance(Spacingd("image", pixdim=(1, 1, 1)), LazyTrait)
    g = np.load(lc.GOLDEN)
    aff = torch.as_tensor(g["affine"])
    chains = {
        "a": [Flipd(lc.KEYS, spatial_axis=[0, 1]), Spacingd(lc.KEYS, pixdim=(1.0, 1.0, 1.0), mode=("bilinear", "nearest"))],
        "b": [Flipd(lc.KEYS, spatial_axis=[0, 1]), Spacingd(lc.KEYS, pixdim=(1.1, 0.7, 1.3), mode=("bilinear", "nearest")),
              CropForegroundd(lc.KEYS, source_key="image", margin=2), SpatialPadd(lc.KEYS, spatial_size=(48, 48, 48))],
    }
    for name, ts in chains.items():
        data = {k: MetaTensor(torch.as_tensor(g[k]), affine=aff.clone()) for k in lc.KEYS}
        with lc._Count() as c:
            out = Compose(ts, lazy=True)(data)
        assert c.n == 2, (name, c.n)                      # one interpolating launch per image, whatever the chain length
        for k in lc.KEYS:
            exp = torch.as_tensor(g[f"{name}_lazy_{k}"])
            assert out[k].shape == exp.shape and float((out[k].as_tensor() - exp).abs().max()) < 2e-5, (name, k)
            assert np.allclose(out[k].affine.numpy(), g[f"{name}_lazy_{k}_affine"], atol=1e-9)
            assert not out[k].pending_operations
    patch.uninstall()
    from monai.transforms.lazy import utils as lu

    assert lu.resample.__module__ == "monai.transforms.lazy.utils"          # the reference's own function is back


@pytest.mark.parametrize("kw", [
    dict(proj_type="perceptron", qkv_bias=True, res_block=False, hidden_size=128, mlp_dim=256, num_heads=4, img_size=(32, 32, 32)),        # head dim 32
    dict(conv_block=False, hidden_size=128, mlp_dim=192, num_heads=2, img_size=(32, 48, 32)),                                                # head dim 64, 12 tokens
])
def test_unetr_option_surface_matches_the_reference(monai_ref, emu, kw):
    """VERDICT r2 missing #2: UNETR configurations beyond the default one -- perceptron patch projection, qkv bias, plain (non-residual) blocks, projection
    up-sampling without convolution blocks, other head dimensions -- load the reference's state_dict strictly and reproduce its logits."""
    from monai.networks.nets import UNETR as RefNet

    from monai_amd.networks.nets.unetr import UNETR as OurNet

    torch.manual_seed(12)
    ref = RefNet(in_channels=1, out_channels=3, feature_size=16, **kw).eval()
    ours = OurNet(in_channels=1, out_channels=3, feature_size=16, **kw).eval()
    missing, unexpected = ours.load_state_dict(ref.state_dict(), strict=True)
    assert not missing and not unexpected and list(ours.state_dict().keys()) == list(ref.state_dict().keys())
    x = torch.rand((1, 1) + tuple(kw["img_size"]), generator=torch.Generator().manual_seed(13))
    with torch.no_grad():
        exp = ref(x)
    got = ours(x)
    err = (got - exp).abs().max().item()
    assert err < 1e-4, (kw, err)


@pytest.mark.parametrize("kw,shape", [
    (dict(norm=("batch", {"eps": 1e-4}), act=("prelu", {"init": 0.2})), (2, 1, 32, 32, 32)),
    (dict(norm=("group", {"num_groups": 4}), upsample="nontrainable", act="relu"), (1, 1, 32, 48, 32)),
    (dict(norm="instance", upsample="nontrainable"), (1, 2, 40, 32, 36)),          # odd extents at the lower levels: UpCat's replicate padding after the interpolation
    (dict(upsample="pixelshuffle", act=("leakyrelu", {"negative_slope": 0.2})), (1, 1, 16, 32, 16)),      # SubpixelUpsample: k3 conv to 8 x the channels + shuffle + pad / average pool
])
def test_basic_unet_option_surface_matches_the_reference(monai_ref, emu, kw, shape):
    """VERDICT r2 missing #3: BasicUNet beyond instance norm + LeakyReLU + deconv -- BatchNorm (evaluated with its running statistics), GroupNorm, PReLU / ReLU,
    upsample="nontrainable" / "pixelshuffle" -- strict state_dict load and logits of the real reference."""
    from monai.networks.nets import BasicUNet as RefNet

    from monai_amd.networks.nets.basic_unet import BasicUNet as OurNet

    features = (16, 16, 32, 32, 64, 16)
    torch.manual_seed(21)
    ref = RefNet(spatial_dims=3, in_channels=shape[1], out_channels=3, features=features, **kw).eval()
    gen = torch.Generator().manual_seed(22)
    with torch.no_grad():
        for m in ref.modules():           # trained-looking normalisation state
            if isinstance(m, torch.nn.BatchNorm3d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.3)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)
            if isinstance(m, (torch.nn.BatchNorm3d, torch.nn.GroupNorm)) and m.weight is not None:
                m.weight.copy_(torch.rand(m.weight.shape, generator=gen) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.2)
    ours = OurNet(spatial_dims=3, in_channels=shape[1], out_channels=3, features=features, **kw).eval()
    missing, unexpected = ours.load_state_dict(ref.state_dict(), strict=True)
    assert not missing and not unexpected and list(ours.state_dict().keys()) == list(ref.state_dict().keys())
    x = torch.rand(shape, generator=gen)
    with torch.no_grad():
        exp = ref(x)
    got = ours(x)
    err = (got - exp).abs().max().item()
    assert err < 1e-4 * max(1.0, exp.abs().max().item()), (kw, err)


def test_spacing_recompute_affine_matches_the_reference(monai_ref, emu):
    """VERDICT r2 missing #5: Spacing(recompute_affine=True) (monai/transforms/spatial/array.py:538-542)"""
    import numpy as np
    from monai.data import MetaTensor as RefMeta
    from monai.transforms import Spacing as RefSpacing

    from monai_amd.data import MetaTensor
    from monai_amd.transforms import Spacing

    aff = np.diag([0.8, 1.3, 2.1, 1.0])
    aff[:3, 3] = [5.0, -3.0, 2.0]
    x = torch.rand((1, 17, 22, 13), generator=torch.Generator().manual_seed(31))
    for kw in (dict(pixdim=(1.0, 1.0, 1.0)), dict(pixdim=(1.7, 0.9, 1.2), diagonal=True)):
        exp = RefSpacing(recompute_affine=True, **kw)(RefMeta(x.clone(), affine=aff))
        got = Spacing(recompute_affine=True, **kw)(MetaTensor(x.clone(), affine=aff))
        assert tuple(got.shape) == tuple(exp.shape)
        np.testing.assert_allclose(got.affine.numpy(), exp.affine.numpy(), atol=1e-12)
        assert (got.as_tensor() - exp.as_tensor()).abs().max().item() < 2e-6
