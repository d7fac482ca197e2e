"""Inferer API behaviour restated from the reference's unit tests (tests/inferers/test_sliding_window_inference.py,
tests/inferers/test_slice_inferer.py), run through the product path on the emulator build of the kernels."""
import numpy as np
import pytest
import torch

from monai_amd.inferers import SliceInferer, SlidingWindowInferer, SlidingWindowInfererAdapt, sliding_window_inference


class _Pred:  # test_sliding_window_inference.py:164-169
    def __init__(self):
        self.add = 1

    def compute(self, data):
        self.add += 1
        return data + self.add


def test_sigma_tables(emu):
    x = torch.ones((1, 1, 7, 7))
    r = sliding_window_inference(x, (3, 3), 10, _Pred().compute, overlap=0.5, padding_mode="constant", cval=-1, mode="constant", sigma_scale=1.0)
    rows = [3.0, 3.0, 3.3333, 3.6667, 4.3333, 4.5, 5.0]
    np.testing.assert_allclose(r.numpy()[0, 0], np.repeat(np.asarray(rows)[:, None], 7, 1), rtol=1e-4)
    exp = np.array([[3.0] * 7, [3.0] * 7,
                    [3.3271625, 3.3271623, 3.3271623, 3.3271623, 3.3271623, 3.3271623, 3.3271625],
                    [3.6728377] * 7,
                    [4.3271623, 4.3271623, 4.3271627, 4.3271627, 4.3271627, 4.3271623, 4.3271623],
                    [4.513757] * 7,
                    [4.9999995, 5.0, 5.0, 5.0, 5.0, 5.0, 4.9999995]])
    for inf in (
        lambda: sliding_window_inference(x, (3, 3), 10, _Pred().compute, overlap=0.5, padding_mode="constant", cval=-1, mode="gaussian", sigma_scale=1.0),
        lambda: SlidingWindowInferer((3, 3), 10, overlap=0.5, mode="gaussian", sigma_scale=1.0)(x, _Pred().compute),
        lambda: SlidingWindowInferer((3, 3), 10, overlap=0.5, mode="gaussian", sigma_scale=[1.0, 1.0])(x, _Pred().compute),
        lambda: SlidingWindowInferer((3, 3), 10, overlap=0.5, mode="gaussian", sigma_scale=[1.0, 1.0], cache_roi_weight_map=True)(x, _Pred().compute),
    ):
        np.testing.assert_allclose(inf().numpy()[0, 0], exp, rtol=1e-4)


def test_cval_and_default_exact(emu):
    x = torch.ones((1, 1, 3, 3))
    r = sliding_window_inference(x, (5, 5), 10, lambda d: d + d.sum(), overlap=0.5, padding_mode="constant", cval=-1, mode="constant")
    np.testing.assert_allclose(r.numpy(), np.full((1, 1, 3, 3), -6.0), rtol=1e-4)
    x = torch.arange(1 * 3 * 16 * 15 * 7, dtype=torch.float32).reshape(1, 3, 16, 15, 7)
    assert torch.equal(sliding_window_inference(x, (4, 10, 7), 3, lambda d: d + 1, overlap=0.25, mode="constant"), x + 1)


@pytest.mark.parametrize(
    "image,roi,sw,ov,mode",
    [((2, 3, 16), (4,), 3, 0.25, "constant"), ((1, 3, 16, 15, 7), (4, -1, 7), 3, 0.25, "constant"), ((2, 3, 16, 15, 7), (4, -1, 7), 3, 0.25, "gaussian"),
     ((1, 3, 16, 15, 7), (20, 22, 23), 10, 0.25, "constant"), ((2, 3, 15, 7), (2, 6), 1000, 0.25, "constant"), ((1, 3, 16, 7), (80, 50), 7, 0.25, "gaussian"),
     ((1, 3, 16, 15, 7), (20, 22, 23), 10, (0.5, 0.25, 0), "gaussian")],
)
def test_identity_like_cases(emu, image, roi, sw, ov, mode):
    n = int(np.prod(image))
    x = torch.arange(n, dtype=torch.float32).reshape(image) / n
    r = sliding_window_inference(x, roi, sw, lambda d: d + 1, overlap=ov, mode=mode)
    np.testing.assert_allclose(r.numpy(), x.numpy() + 1, rtol=1e-6, atol=1e-6)


def test_multioutput_and_args(emu):
    x = torch.ones((1, 6, 20, 20))

    def compute(d, t1, t2):
        return d + t1, d[:, ::3, ::2, ::2] + t2, d[:, ::2, ::4, ::4] + 3

    t = sliding_window_inference(x, (8, 8), 10, compute, 0.5, "constant", 0.125, "constant", 0.0, None, None, False, None, None, None, -1, False, 1, 2)
    assert [tuple(o.shape) for o in t] == [(1, 6, 20, 20), (1, 2, 10, 10), (1, 3, 5, 5)]
    for o, v in zip(t, (2.0, 3.0, 4.0)):
        np.testing.assert_allclose(o.numpy(), np.full(o.shape, v), rtol=1e-4)
    d = SlidingWindowInferer((8, 8), 10, 0.5)(x, lambda w: dict(zip("abc", compute(w, 1, 2))))
    assert sorted(d.keys()) == ["a", "b", "c"] and tuple(d["c"].shape) == (1, 3, 5, 5)


def test_errors_and_buffer_args(emu):
    x = torch.ones((1, 1, 8, 8))
    with pytest.raises(ValueError):
        sliding_window_inference(x, (4, 4), 2, lambda d: d, overlap=1.0)
    with pytest.raises(ValueError):
        sliding_window_inference(x, (4, 4), 2, lambda d: d, buffer_steps=1, buffer_dim=5)
    with pytest.raises(ValueError):
        SlidingWindowInferer((4, 4), mode="bogus")
    r = SlidingWindowInferer((4, 4), 2, overlap=0.5, buffer_steps=2, buffer_dim=-1)(x, lambda d: 2 * d)   # |x - sw/2| < 1e-3 (test_buffers :74-96)
    assert (x - r / 2).abs().max().item() < 1e-3


def test_slice_inferer_and_adapt(emu):
    # tests/inferers/test_slice_inferer.py: a 2-D predictor over a 3-D volume, each spatial_dim
    x = torch.rand(1, 1, 12, 10, 8)
    for sd, roi in ((0, (10, 8)), (1, (12, 8)), (2, (12, 10))):
        r = SliceInferer(roi_size=roi, spatial_dim=sd, sw_batch_size=3)(x, lambda s: s.dim() == 4 and s * 3.0)
        np.testing.assert_allclose(r.numpy(), 3.0 * x.numpy(), rtol=1e-6)
    with pytest.raises(RuntimeError):
        SliceInferer(roi_size=(4, 4, 4), spatial_dim=0)(x, lambda s: s)
    r = SlidingWindowInfererAdapt((8, 8, 8), 2, overlap=0.25)(x, lambda s: s + 1)
    np.testing.assert_allclose(r.numpy(), x.numpy() + 1, rtol=1e-6)


def test_patch_inferer_vs_reference(emu):
    import patch_cases as pc

    print("cases", pc.case_patch_inferer_vs_reference("cpu"))


def test_gathered_split_and_batched_merge(emu):
    import patch_cases as pc

    pc.case_gathered_split_and_batched_merge("cpu")


def test_patch_inferer_api(emu):
    import patch_cases as pc

    pc.case_patch_inferer_api("cpu")


def test_half_precision_predictor_output_is_accepted(emu):
    """a predictor under torch.autocast hands back half precision: widened to the compute dtype, blended in fp32 (the reference blends it too)"""
    import torch

    from monai_amd.inferers import SlidingWindowInferer

    net = torch.nn.Conv3d(1, 2, 3, padding=1).eval()
    x = torch.rand(1, 1, 24, 24, 24)
    inf = SlidingWindowInferer(roi_size=(16, 16, 16), sw_batch_size=2, overlap=0.5, mode="gaussian")
    with torch.no_grad():
        y = inf(x, net)
        yh = inf(x, lambda w: net(w).to(torch.bfloat16))
    assert yh.dtype == torch.float32 and float((y - yh).abs().max()) < 1e-2
