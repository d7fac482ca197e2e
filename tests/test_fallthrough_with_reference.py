"""SURVEY.md 8b, boundary B3: after ``monai_amd.patch.install()`` no call that is valid with the unpatched reference may fail.
The argument matrices of the reference's own tests -- tests/networks/nets/test_basic_unet.py:23-84 (1-D / 2-D / 3-D, every
up-sampling mode), tests/inferers/test_sliding_window_inference.py (device / cpu cases, positional call :280-300),
tests/networks/nets/test_unet.py / test_segresnet.py / test_dynunet.py style configurations, Resample spline orders, lazy=True --
run over the PATCHED names on CPU tensors (no GPU here: every call is outside the HIP path and has to fall through to the
displaced reference objects, monai_amd/_fallback.py), and are compared with the unpatched reference's results.
Only where /root/reference exists (the build container)."""
import os
import sys
import warnings

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = [pytest.mark.fallthrough, pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "monai")), reason="reference MONAI not available here")]

CASES_1D = [[{"spatial_dims": 1, "in_channels": 5, "out_channels": 8, **({"upsample": m} if m else {})}, (10, 5, 33), (10, 8, 33)]
            for m in ["pixelshuffle", "nontrainable", "deconv", None]]
CASES_2D = [[{"spatial_dims": 2, "in_channels": 2, "out_channels": 3, "features": (12, 12, 13, 14, 15, 16), "upsample": m}, (2, 2, d1, d2), (2, 3, d1, d2)]
            for m in ["pixelshuffle", "nontrainable", "deconv"] for d1 in range(33, 64, 14) for d2 in range(63, 33, -21)]
CASES_3D = [
    [{"spatial_dims": 3, "in_channels": 1, "out_channels": 2, "features": (16, 20, 21, 22, 23, 11), "upsample": "pixelshuffle"}, (2, 1, 33, 34, 35), (2, 2, 33, 34, 35)],
    [{"spatial_dims": 3, "in_channels": 2, "out_channels": 7, "features": (14, 15, 16, 17, 18, 11), "upsample": "deconv"}, (3, 2, 33, 37, 34), (3, 7, 33, 37, 34)],
    [{"spatial_dims": 3, "in_channels": 4, "out_channels": 2, "features": (14, 15, 16, 17, 18, 10), "upsample": "nontrainable"}, (5, 4, 34, 35, 37), (5, 2, 34, 35, 37)],
]


@pytest.fixture()
def patched():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import monai
    import monai_amd.patch as patch

    ref = {"BasicUNet": monai.networks.nets.BasicUNet, "swi": monai.inferers.utils.sliding_window_inference}
    patch.install()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        yield monai, ref
    patch.uninstall()
    sys.path.remove(REF)


@pytest.mark.parametrize("kwargs,in_shape,out_shape", CASES_1D + CASES_2D[::3] + CASES_3D)
def test_basic_unet_argument_matrix(patched, kwargs, in_shape, out_shape):
    """tests/networks/nets/test_basic_unet.py:86-95 (`test_shape`) over the patched name, plus equality with the displaced class."""
    monai, ref = patched
    from monai.networks import eval_mode
    from monai.networks.nets import BasicUNet          # the MI355X class after install()

    assert getattr(BasicUNet, "_mh_is_product", False)
    torch.manual_seed(3)
    net = BasicUNet(**kwargs)
    torch.manual_seed(3)
    ref_net = ref["BasicUNet"](**kwargs)
    x = torch.randn(in_shape)
    with eval_mode(net), eval_mode(ref_net):
        y, y_ref = net(x), ref_net(x)
    assert tuple(y.shape) == out_shape
    assert torch.equal(y, y_ref)                         # same seed -> same initial weights -> the same reference arithmetic


def test_training_step_through_the_shared_parameters(patched):
    monai, _ = patched
    from monai.networks.nets import BasicUNet, SegResNet, UNet

    for net in (BasicUNet(spatial_dims=3, in_channels=1, out_channels=2, features=(8, 8, 16, 16, 32, 8)),
                UNet(spatial_dims=3, in_channels=1, out_channels=2, channels=(4, 8, 16), strides=(2, 2), num_res_units=1),
                SegResNet(spatial_dims=3, init_filters=8, in_channels=1, out_channels=2)):
        assert getattr(type(net), "_mh_is_product", False)
        net.train()
        opt = torch.optim.SGD(net.parameters(), lr=0.1)
        before = [p.detach().clone() for p in net.parameters()]
        loss = net(torch.rand(2, 1, 32, 32, 32)).square().mean()
        loss.backward()
        opt.step()
        assert any(not torch.equal(a, b) for a, b in zip(before, net.parameters())), type(net).__name__
        # state_dict still the reference's layout, loadable into a fresh reference module
        fresh = type(net._mh_twin())(*net._mh_ctor[0], **net._mh_ctor[1])
        fresh.load_state_dict(net.state_dict(), strict=True)


def test_sliding_window_cpu_cases(patched):
    """tests/inferers/test_sliding_window_inference.py: cpu tensors, sw_device / device arguments, the fully positional call (:280-300),
    SlidingWindowInfererAdapt, SliceInferer with a 2-D network -- valid reference calls, outside the HIP path."""
    monai, ref = patched
    from monai.inferers import SliceInferer, SlidingWindowInferer, SlidingWindowInfererAdapt, sliding_window_inference
    from monai.networks.nets import BasicUNet

    inputs = torch.ones((1, 1, 3, 3))
    t1, t2 = torch.ones(1), torch.ones(1)

    def compute(data, test1, test2):
        return data + test1 + test2

    r = sliding_window_inference(inputs, (5, 5), 10, compute, 0.5, "constant", 1.0, "constant", 0.0, "cpu:0", "cpu:0", False, None, None, None, 0, False, t1, test2=t2)
    np.testing.assert_allclose(r.numpy(), np.ones((1, 1, 3, 3)) + 2.0, rtol=1e-4)
    for cls in (SlidingWindowInferer, SlidingWindowInfererAdapt):
        r = cls((5, 5), 10, overlap=0.5, mode="constant", cval=-1)(inputs, compute, t1, test2=t2)
        np.testing.assert_allclose(r.numpy(), np.ones((1, 1, 3, 3)) + 2.0, rtol=1e-4)
    # random volume, gaussian mode, equality with the displaced function
    torch.manual_seed(0)
    vol = torch.rand(1, 1, 20, 24, 28)
    a = sliding_window_inference(vol, (8, 12, 16), 3, lambda w: w * 3 + 1, overlap=0.5, mode="gaussian")
    b = ref["swi"](vol, (8, 12, 16), 3, lambda w: w * 3 + 1, overlap=0.5, mode="gaussian")
    assert torch.equal(a, b)
    # half / double precision volumes (the reference computes in the input dtype)
    for dt in (torch.float64, torch.float16):
        out = sliding_window_inference(vol.to(dt), (8, 12, 16), 3, lambda w: w + 1, overlap=0.25)
        assert out.dtype == dt
    # a 2-D network over a 3-D CPU volume (tests/inferers/test_slice_inferer.py): CPU tensors are outside the HIP path, the call reaches the reference twin
    net2d = BasicUNet(spatial_dims=2, in_channels=1, out_channels=2, features=(4, 4, 8, 8, 16, 4)).eval()
    with torch.no_grad():
        s = SliceInferer(roi_size=(32, 32), spatial_dim=2, sw_batch_size=4)(torch.rand(1, 1, 32, 32, 6), net2d)
    assert tuple(s.shape) == (1, 2, 32, 32, 6)


def test_transform_options_outside_the_hip_path(patched):
    monai, _ = patched
    from monai.data import MetaTensor
    from monai.transforms import GaussianSmooth, Resample, ScaleIntensityRanged, Spacing, Spacingd
    from monai.transforms.utils import create_grid

    img = MetaTensor(torch.rand(1, 12, 14, 16), affine=torch.diag(torch.tensor([2.0, 2.0, 2.0, 1.0])))
    # numpy / cpu inputs
    assert GaussianSmooth(sigma=1.0)(np.random.rand(1, 8, 8, 8).astype(np.float32)).shape == (1, 8, 8, 8)
    out = Spacing(pixdim=(1.0, 1.0, 1.0), mode="bilinear")(img)
    assert tuple(out.shape) == (1, 23, 27, 31)
    # spline interpolation orders (scipy branch of the reference, monai/transforms/spatial/array.py:2092-2100)
    grid = create_grid((12, 14, 16), backend="torch")
    r = Resample(mode=3, padding_mode="nearest")(img, grid=grid)
    assert tuple(r.shape) == (1, 12, 14, 16)
    # lazy resampling: pending operations recorded, nothing resampled yet (monai/transforms/lazy)
    lz = Spacingd(keys="image", pixdim=(1.0, 1.0, 1.0), lazy=True)({"image": img})["image"]
    assert len(lz.pending_operations) == 1 and tuple(lz.shape) == (1, 12, 14, 16)
    # float64 image
    d = ScaleIntensityRanged(keys="image", a_min=0.0, a_max=1.0, b_min=0.0, b_max=2.0)({"image": img.to(torch.float64)})["image"]
    assert d.dtype == torch.float32 or d.dtype == torch.float64


def test_torchscript_export_uses_the_reference_twin(patched):
    """`torch.jit.script(net)` (bundle `ckpt_export`, the reference's test_script cases): the product net hands TorchScript its reference twin, which
    shares its parameters; the scripted module equals the eager CPU call (itself a fall-through to that twin) bit for bit."""
    from monai.networks.nets import BasicUNet

    for dims, shape in ((2, (2, 1, 32, 32)), (3, (1, 1, 32, 32, 32))):
        net = BasicUNet(spatial_dims=dims, in_channels=1, out_channels=3, features=(4, 4, 8, 8, 16, 4)).eval()
        assert getattr(type(net), "_mh_is_product", False)
        ts = torch.jit.script(net)
        x = torch.rand(shape)
        with torch.no_grad():
            assert torch.equal(ts(x), net(x))
        with torch.no_grad():
            next(net.parameters()).add_(0.5)         # shared parameters: the scripted twin follows the product net's weights
            assert torch.equal(ts(x), net(x))


def test_fused_argmax_and_moved_buffers_fall_through_cleanly(patched):
    """(a) sliding_window_argmax / SlidingWindowInferer.argmax of a CPU volume falls through to the reference WITHOUT handing this
    package's private keyword to the predictor, and still returns the label map; (b) a net whose reference twin already exists and that is then
    converted with .to() (nn.Module._apply REPLACES buffer tensors) keeps its twin on the new tensors."""
    monai, ref = patched
    from monai.inferers import SlidingWindowInferer
    from monai.networks.nets import UNet

    from monai_amd.inferers import sliding_window_argmax

    torch.manual_seed(4)
    pred = torch.nn.Conv3d(1, 3, 3, padding=1).eval()          # a plain torch predictor on the CPU: would raise on an unexpected keyword
    x = torch.rand(1, 1, 20, 20, 20)
    with torch.no_grad():
        blended = ref["swi"](x, (16, 16, 16), 2, pred, overlap=0.5, mode="gaussian")
        labels = sliding_window_argmax(x, (16, 16, 16), 2, pred, overlap=0.5, mode="gaussian")
        labels2 = SlidingWindowInferer((16, 16, 16), 2, overlap=0.5, mode="gaussian").argmax(x, pred, labels_dtype=torch.uint8)
    assert labels.shape == (1, 1, 20, 20, 20) and torch.equal(labels, blended.argmax(1, keepdim=True).float())
    assert labels2.dtype == torch.uint8 and torch.equal(labels2.long(), blended.argmax(1, keepdim=True))
    net = UNet(spatial_dims=3, in_channels=1, out_channels=2, channels=(4, 8), strides=(2,), norm="batch").train()
    y = torch.rand(2, 1, 8, 8, 8)
    net(y)                                                     # training mode: falls through, the twin is built and updates the running statistics
    before = net.model[0].adn.N.running_mean.clone()
    net = net.to(torch.float64)                                # buffers are replaced by _apply
    net(y.double())
    after = net.model[0].adn.N.running_mean
    assert after.dtype == torch.float64 and not torch.equal(after.float(), before), "the twin must update the CONVERTED buffers"


def test_no_monai_keeps_the_explicit_error():
    """Without MONAI on the path nothing can be delegated: the original explicit error is raised."""
    # the process is shared with other tests (pytest-xdist hands a worker tests of several files): the imported reference modules and the path entry are put back
    # afterwards -- a second, fresh `monai` next to classes that were derived from the first one breaks every later isinstance check against the reference
    gone = {k: sys.modules.pop(k) for k in [k for k in sys.modules if k == "monai" or k.startswith("monai.")]}
    had_path = REF in sys.path
    import monai_amd._fallback as fb
    from monai_amd.networks.nets import BasicUNet

    try:
        if had_path:
            sys.path.remove(REF)
        assert fb.reference_object("monai.networks.nets.basic_unet", "BasicUNet") is None
        with pytest.raises(NotImplementedError):
            BasicUNet(spatial_dims=1)
        with pytest.raises(NotImplementedError):
            BasicUNet(spatial_dims=3, upsample="no such mode")
        with pytest.raises(RuntimeError):
            BasicUNet(spatial_dims=3).eval()(torch.rand(1, 1, 32, 32, 32))
    finally:
        for k in [k for k in sys.modules if k == "monai" or k.startswith("monai.")]:
            del sys.modules[k]
        sys.modules.update(gone)
        if had_path and REF not in sys.path:
            sys.path.insert(0, REF)
