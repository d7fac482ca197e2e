"""Pins the oracle (oracle/) against (a) the golden vectors produced by the real reference
(tests/golden/make_golden.py) and (b) the known-answer tables of the reference's own unit tests,
restated here with file:line citations (the reference tests need `parameterized`, absent here).
CPU only."""

import hashlib
import itertools
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import sliding_window as osw


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _toy(k_out):
    def f(x):
        return torch.cat([torch.sin(x[:, :1] * (1.0 + 0.37 * k)) + 0.05 * k * x[:, :1] for k in range(k_out)], dim=1)

    return f


def _digest(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


# ---------------------------------------------------------------- host index math
def test_window_starts_match_reference(golden_dir):
    g = _load(golden_dir, "host_math.npz")
    i = 0
    while f"slices_{i}_img" in g:
        img, roi = tuple(g[f"slices_{i}_img"]), tuple(g[f"slices_{i}_roi"])
        ov = float(g[f"slices_{i}_ov"])
        itv = osw.get_scan_interval(img, roi, (ov,) * len(img))
        assert tuple(itv) == tuple(g[f"slices_{i}_interval"])
        starts, _ = osw.dense_patch_starts(img, roi, itv)
        got = np.asarray(list(itertools.product(*starts)), dtype=np.int32)
        np.testing.assert_array_equal(got, g[f"slices_{i}_starts"])
        i += 1
    assert i >= 6


def test_bench_config_has_1000_windows():
    itv = osw.get_scan_interval((512,) * 3, (96,) * 3, (0.5,) * 3)
    starts, _ = osw.dense_patch_starts((512,) * 3, (96,) * 3, itv)
    assert itv == (48, 48, 48)
    assert starts[0] == [0, 48, 96, 144, 192, 240, 288, 336, 384, 416]
    assert len(list(itertools.product(*starts))) == 1000


def test_importance_map_matches_reference(golden_dir):
    g = _load(golden_dir, "host_math.npz")
    i = 0
    while f"imp_{i}_ps" in g:
        ps = tuple(int(v) for v in g[f"imp_{i}_ps"])
        sig = g[f"imp_{i}_sigma"]
        sig = float(sig) if sig.ndim == 0 else tuple(float(s) for s in sig)
        m = osw.compute_importance_map(ps, mode=str(g[f"imp_{i}_mode"]), sigma_scale=sig)
        if f"imp_{i}_map" in g:
            np.testing.assert_array_equal(m.numpy(), g[f"imp_{i}_map"])
        else:
            np.testing.assert_array_equal(m.numpy()[::7, ::5, ::3], g[f"imp_{i}_sample"])
            assert m.double().sum().item() == float(g[f"imp_{i}_sum"])
            assert [m.min().item(), m.max().item()] == list(g[f"imp_{i}_minmax"])
        i += 1
    assert i >= 5


# ---------------------------------------------------------------- blend, bit-exact vs the reference
def test_blend_bitwise_vs_reference(golden_dir):
    g = _load(golden_dir, "blend.npz")
    i = 0
    while f"blend_{i}_shape" in g:
        shape = tuple(int(v) for v in g[f"blend_{i}_shape"])
        torch.manual_seed(int(g[f"blend_{i}_seed"]))
        x = torch.rand(shape)
        y = osw.sliding_window_inference(
            x, tuple(int(v) for v in g[f"blend_{i}_roi"]), int(g[f"blend_{i}_sw"]), _toy(int(g[f"blend_{i}_k"])),
            overlap=float(g[f"blend_{i}_ov"]), mode=str(g[f"blend_{i}_mode"]), padding_mode="constant", cval=-0.5,
        )
        assert y.shape == g[f"blend_{i}_out"].shape
        assert np.array_equal(y.numpy(), g[f"blend_{i}_out"]), f"blend case {i} not bit-identical"
        i += 1
    assert i >= 5


# ---------------------------------------------------------------- reference unit-test tables
class _Pred:
    """stateful predictor of tests/inferers/test_sliding_window_inference.py:164-169"""

    def __init__(self):
        self.add = 1

    def compute(self, data):
        self.add += 1
        return data + self.add


def test_sigma_tables_constant_and_gaussian():
    # /root/reference/tests/inferers/test_sliding_window_inference.py:158-241 (test_sigma)
    x = torch.ones((1, 1, 7, 7))
    r = osw.sliding_window_inference(x, (3, 3), 10, _Pred().compute, overlap=0.5, padding_mode="constant", cval=-1,
                                     mode="constant", sigma_scale=1.0)
    rows = [3.0, 3.0, 3.3333, 3.6667, 4.3333, 4.5, 5.0]
    np.testing.assert_allclose(r.numpy()[0, 0], np.repeat(np.asarray(rows)[:, None], 7, 1), rtol=1e-4)
    r = osw.sliding_window_inference(x, (3, 3), 10, _Pred().compute, overlap=0.5, padding_mode="constant", cval=-1,
                                     mode="gaussian", sigma_scale=1.0)
    exp = np.array(
        [
            [3.0, 3.0, 3.0, 3.0, 3.0, 3.0, 3.0],
            [3.0, 3.0, 3.0, 3.0, 3.0, 3.0, 3.0],
            [3.3271625, 3.3271623, 3.3271623, 3.3271623, 3.3271623, 3.3271623, 3.3271625],
            [3.6728377, 3.6728377, 3.6728377, 3.6728377, 3.6728377, 3.6728377, 3.6728377],
            [4.3271623, 4.3271623, 4.3271627, 4.3271627, 4.3271627, 4.3271623, 4.3271623],
            [4.513757, 4.513757, 4.513757, 4.513757, 4.513757, 4.513757, 4.513757],
            [4.9999995, 5.0, 5.0, 5.0, 5.0, 5.0, 4.9999995],
        ]
    )
    np.testing.assert_allclose(r.numpy()[0, 0], exp, rtol=1e-4)


def test_cval_padding_table():
    # /root/reference/tests/inferers/test_sliding_window_inference.py:243-267 (test_cval): 3x3 ones,
    # roi 5x5 padded with -1: data.sum() = 9 - 16 = -7 -> 1 + (-7) = -6 everywhere after the crop.
    x = torch.ones((1, 1, 3, 3))
    r = osw.sliding_window_inference(x, (5, 5), 10, lambda d: d + d.sum(), overlap=0.5, padding_mode="constant",
                                     cval=-1, mode="constant", sigma_scale=1.0)
    np.testing.assert_allclose(r.numpy(), np.full((1, 1, 3, 3), -6.0), rtol=1e-4)


@pytest.mark.parametrize(
    "image,roi,sw,ov,mode",
    [  # subset of TEST_CASES, tests/inferers/test_sliding_window_inference.py:28-46
        ((2, 3, 16), (4,), 3, 0.25, "constant"),
        ((2, 3, 16, 15, 7, 9), 4, 3, 0.25, "constant"),
        ((1, 3, 16, 15, 7), (4, -1, 7), 3, 0.25, "constant"),
        ((2, 3, 16, 15, 7), (4, -1, 7), 3, 0.25, "constant"),
        ((3, 3, 16, 15, 7), (4, -1, 7), 3, 0.25, "constant"),
        ((2, 3, 16, 15, 7), (4, -1, 7), 3, 0.25, "gaussian"),
        ((1, 3, 16, 15, 7), (4, 10, 7), 3, 0.25, "constant"),
        ((1, 3, 16, 15, 7), (20, 22, 23), 10, 0.25, "constant"),
        ((2, 3, 15, 7), (2, 6), 1000, 0.25, "constant"),
        ((1, 3, 16, 7), (80, 50), 7, 0.25, "gaussian"),
        ((1, 3, 16, 15, 7), (20, 22, 23), 10, 0.5, "gaussian"),
        ((1, 3, 16, 15, 7), (20, 22, 23), 10, (0.5, 0.25, 0), "gaussian"),
    ],
)
def test_identity_like_predictor_cases(image, roi, sw, ov, mode):
    # test_sliding_window_default :98-121 -- predictor x+1 must come back as x+1
    n = int(np.prod(image))
    x = torch.arange(n, dtype=torch.float32).reshape(image) / n
    r = osw.sliding_window_inference(x, roi, sw, lambda d: d + 1, overlap=ov, mode=mode)
    np.testing.assert_allclose(r.numpy(), x.numpy() + 1, rtol=1e-6, atol=1e-6)


def test_default_device_exact_arange():
    # test_default_device :123-141: constant mode on an integer arange must reproduce x+1 EXACTLY
    x = torch.arange(1 * 3 * 16 * 15 * 7, dtype=torch.float32).reshape(1, 3, 16, 15, 7)
    r = osw.sliding_window_inference(x, (4, 10, 7), 3, lambda d: d + 1, overlap=0.25, mode="constant")
    assert torch.equal(r, x + 1)


def test_multioutput_tuple_and_dict():
    # test_multioutput :314-373: outputs at 1x, 2x (upsampled), 1/3x resolution, tuple and dict
    x = torch.ones((1, 6, 20, 20))

    def compute(d):
        return d + 1, d[:, ::3, ::2, ::2] + torch.tensor(2.0), d[:, ::2, ::4, ::4] + torch.tensor(3.0)

    def compute_dict(d):
        a, b, c = compute(d)
        return {1: a, "2": b, "3.0": c} if False else {"1": a, "2": b, "3": c}

    t = osw.sliding_window_inference(x, (8, 8), 10, compute, overlap=0.5, mode="constant")
    assert [tuple(o.shape) for o in t] == [(1, 6, 20, 20), (1, 2, 10, 10), (1, 3, 5, 5)]
    for o, v in zip(t, (2.0, 3.0, 4.0)):
        np.testing.assert_allclose(o.numpy(), np.full(o.shape, v), rtol=1e-4)
    d = osw.sliding_window_inference(x, (8, 8), 10, compute_dict, overlap=0.5, mode="constant")
    assert sorted(d.keys()) == ["1", "2", "3"]
    np.testing.assert_allclose(d["3"].numpy(), np.full((1, 3, 5, 5), 4.0), rtol=1e-4)


# ---------------------------------------------------------------- network, bit-exact vs the reference
def test_state_init_order_matches_reference(golden_dir):
    g0, g5 = _load(golden_dir, "config0.npz"), _load(golden_dir, "net5.npz")
    torch.manual_seed(0)
    assert _digest(oracle.make_basic_unet_state(1, 2)) == str(g0["cfg0_state_sha256"])
    torch.manual_seed(1)
    sd = oracle.make_basic_unet_state(1, 5)
    assert _digest(sd) == str(g5["net5_state_sha256"])
    assert list(sd.keys()) == list(g5["net5_keys"])
    assert [",".join(map(str, v.shape)) for v in sd.values()] == list(g5["net5_shapes"])


def test_basic_unet_forward_bitwise_vs_reference(golden_dir):
    g = _load(golden_dir, "net5.npz")
    torch.manual_seed(1)
    sd = oracle.make_basic_unet_state(1, 5)
    torch.manual_seed(21)
    x = torch.rand(2, 1, 32, 32, 32)
    with torch.no_grad():
        assert np.array_equal(oracle.basic_unet_forward(sd, x).numpy(), g["net5_win32_out"])
    torch.manual_seed(22)
    x = torch.rand(1, 1, 48, 32, 16)
    with torch.no_grad():
        assert np.array_equal(oracle.basic_unet_forward(sd, x).numpy(), g["net5_win48x32x16_out"])


def test_config0_end_to_end_bitwise_vs_reference(golden_dir):
    """BASELINE.json configs[0]: BasicUNet(1->2), rand 64^3, roi 32^3, sw_batch 4, ov .5, gaussian."""
    g = _load(golden_dir, "config0.npz")
    torch.manual_seed(0)
    sd = oracle.make_basic_unet_state(1, 2)
    x = torch.rand(1, 1, 64, 64, 64)
    assert x.double().sum().item() == float(g["cfg0_x_sum"])
    with torch.no_grad():
        y = osw.sliding_window_inference(x, (32, 32, 32), 4, lambda w: oracle.basic_unet_forward(sd, w), overlap=0.5,
                                         mode="gaussian")
    assert np.array_equal(y.numpy(), g["cfg0_out"])


# ------------------------------------------------------------------------------------------ UNet / UNETR oracles
def _sha(sd, skip=()):
    h = hashlib.sha256()
    for k, v in sd.items():
        if k.endswith(tuple(skip)) if skip else False:
            continue
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("name", ["res2", "plain", "mixed"])
def test_unet_oracle_bitwise_vs_reference(golden_dir, name):
    """oracle/unet.py against tests/golden/unet.npz (made by the real monai.networks.nets.UNet): keys, init, logits."""
    from e2e_cases import UNET_CFGS
    from oracle import unet as ou

    g = _load(golden_dir, "unet.npz")
    c = UNET_CFGS[name]
    torch.manual_seed(c["seed"])
    sd = ou.make_unet_state(1, 3, c["channels"], c["strides"], c["num_res_units"])
    assert list(sd.keys()) == list(g[f"{name}_keys"])
    assert _sha(sd) == str(g[f"{name}_init_sha256"])
    for k, v in sd.items():
        if k.endswith("adn.A.weight"):
            v.fill_(0.1 + 0.01 * (len(k) % 7))
    torch.manual_seed(100 + c["seed"])
    x = torch.rand(c["shape"])
    with torch.no_grad():
        y = ou.unet_forward(sd, x, c["channels"], c["strides"], c["num_res_units"])
    assert np.array_equal(y.numpy(), g[f"{name}_out"])


def test_unetr_oracle_small_vs_reference(golden_dir):
    """oracle/unetr.py against tests/golden/unetr.npz (real monai UNETR, 32^3, hidden 128): init digest and logits."""
    from oracle import unetr as our

    g = _load(golden_dir, "unetr.npz")
    torch.manual_seed(2)
    sd = our.make_unetr_state(1, 3, (32, 32, 32), 16, 128, 256)
    assert _sha(sd, skip=("position_embeddings",)) == str(g["small_state_sha256"])
    np.testing.assert_allclose(sd["vit.patch_embedding.position_embeddings"].flatten()[::13].numpy(), g["small_pos_sample"], rtol=2e-6, atol=1e-9)
    torch.manual_seed(32)
    x = torch.rand(2, 1, 32, 32, 32)
    with torch.no_grad():
        y = our.unetr_forward(sd, x, heads=2)
    assert np.abs(y.numpy() - g["small_out"]).max() < 1e-5


def test_basic_unet_oracle_odd_window_bitwise_vs_reference(golden_dir):
    """UpCat's replicate padding of odd extents (basic_unet.py:163-170): oracle vs the real reference, bit for bit."""
    g = _load(golden_dir, "net5_odd.npz")
    torch.manual_seed(1)
    sd = oracle.make_basic_unet_state(1, 5)
    torch.manual_seed(23)
    x = torch.rand(1, 1, 40, 36, 34)
    with torch.no_grad():
        y = oracle.basic_unet_forward(sd, x)
    assert np.array_equal(y.numpy(), g["out"])


# ------------------------------------------------------------------------------------------------ widening rows
@pytest.mark.parametrize("name", ["basic", "res_ds", "stride0", "aniso", "aniso_basic"])
def test_dynunet_oracle_vs_reference(golden_dir, name):
    """oracle/dynunet.py against tests/golden/dynunet.npz (made by the real monai DynUNet); parameters from the product module, whose
    keys / seeded values are themselves pinned to the reference's (tests/dynunet_cases.py)."""
    import dynunet_cases as dc
    from monai_amd.networks.nets import DynUNet
    from oracle import dynunet as od

    g = _load(golden_dir, "dynunet.npz")
    net, init = dc.build(DynUNet, name)
    assert init == str(g[f"{name}_init_sha256"])
    kw = dc.CFGS[name]["kw"]
    strides = kw["strides"]
    slope = 0.0 if kw.get("act_name") == "relu" else 0.01
    with torch.no_grad():
        y = od.dynunet_forward(net.state_dict(), dc.inputs(name), strides, slope, kw.get("res_block", False))
    np.testing.assert_allclose(y.numpy(), g[f"{name}_out"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("name", ["default", "f16", "deconv_inst", "features"])
def test_segresnet_oracle_vs_reference(golden_dir, name):
    import segresnet_cases as sc
    from monai_amd.networks.nets import SegResNet
    from oracle import dynunet as od

    g = _load(golden_dir, "segresnet.npz")
    net, init = sc.build(SegResNet, name)
    assert init == str(g[f"{name}_init_sha256"])
    kw = sc.CFGS[name]["kw"]
    inst = isinstance(kw.get("norm"), tuple) and kw["norm"][0] == "instance"
    act = kw.get("act", ("RELU",))
    with torch.no_grad():
        y = od.segresnet_forward(net.state_dict(), sc.inputs(name), kw.get("blocks_down", (1, 2, 2, 4)), kw.get("blocks_up", (1, 1, 1)),
                                 0 if inst else 8, act[1]["negative_slope"] if act[0] == "leakyrelu" else 0.0,
                                 kw.get("upsample_mode", "nontrainable"), kw.get("use_conv_final", True))
    np.testing.assert_allclose(y.numpy(), g[f"{name}_out"], rtol=0, atol=1e-6)


def test_preproc_oracle_vs_reference(golden_dir):
    """oracle/preproc.py against the real reference's outputs (tests/golden/{preproc,normalize}.npz)"""
    import normalize_cases as nc
    import preproc_cases as pc
    from oracle import preproc as op

    g = _load(golden_dir, "preproc.npz")
    x = pc.ct()
    for name, kw in pc.SCALE_CASES:
        if "dtype" in kw:
            continue
        y = op.scale_intensity_range(x, kw["a_min"], kw["a_max"], kw.get("b_min"), kw.get("b_max"), kw.get("clip", False))
        assert np.array_equal(y.numpy(), g[name], equal_nan=True), name
    for name, make, kw in pc.CROP_CASES:
        if "select_fn" in kw or "channel_indices" in kw or make().dim() != 4:
            continue
        img = make()
        s, e = op.foreground_box(img, kw.get("margin", 0), kw.get("allow_smaller", False), kw.get("k_divisible", 1))
        assert list(s) == list(g[name + "__start"]) and list(e) == list(g[name + "__end"]), name
        assert np.array_equal(op.crop_pad(img, s, e, kw.get("value", 0.0)).numpy(), g[name], equal_nan=True), name
    gn = _load(golden_dir, "normalize.npz")
    for name, kw in nc.NORM_CASES:
        if "subtrahend" in kw or "divisor" in kw:
            continue
        y = op.normalize_intensity(nc.mri(), kw.get("nonzero", False), kw.get("channel_wise", False))
        assert np.array_equal(y.numpy(), gn[name]), name
    t = torch.arange(2 * 3 * 4 * 5, dtype=torch.float32).reshape(2, 3, 4, 5)
    assert torch.equal(op.flip_permute(t, [2, 0, 1], [True, False, True]), torch.flip(t, [1, 3]).permute(0, 3, 1, 2))


def test_synthetic_ct_volume_matches_reference(golden_dir):
    """oracle/synthetic.py restates monai/data/synthetic.py:97-170 (+ rescale_array, monai/transforms/utils.py:229-257): bit-identical
    volumes and label maps for the seeded cases of tests/golden/make_golden_synthetic.py, and the 512^3 benchmark volume of SURVEY.md 8(d)
    config 1 by digest."""
    import hashlib

    from oracle import synthetic

    g = np.load(os.path.join(golden_dir, "synthetic.npz"))
    for name, seed in (("a", 7), ("b", 7), ("c", 0)):
        h, w, d, nobj, rmax, rmin, ncls = (int(v) for v in g[f"{name}_args"])
        img, lab = synthetic.create_test_image_3d(h, w, d, num_objs=nobj, rad_max=rmax, rad_min=rmin, noise_max=float(g[f"{name}_noise"]),
                                                  num_seg_classes=ncls, random_state=np.random.RandomState(seed))
        assert img.dtype == np.float32 and np.array_equal(img, g[f"{name}_img"]), name
        assert np.array_equal(lab, g[f"{name}_lab"].astype(np.int32)), name
    small = synthetic.benchmark_volume(64)          # reduced-size bench / test volumes scale the sphere radii
    assert small.shape == (64, 64, 64) and 0.0 == small.min() and small.max() == 1.0
    vol = synthetic.benchmark_volume(512)
    assert vol.shape == (512, 512, 512) and vol.dtype == np.float32
    for idx, val in zip(g["bench_probe_idx"], g["bench_probe_val"]):
        assert vol[tuple(idx)] == val
    assert hashlib.sha256(np.ascontiguousarray(vol).tobytes()).digest() == bytes(g["bench_sha256"])


def test_oracle_buffered_schedule_bitwise_vs_reference(golden_dir):
    """oracle/sliding_window.py:_buffered against the REAL reference's buffered runs (tests/golden/make_golden_buffered.py): bit-exact"""
    import numpy as np
    import torch

    from oracle import sliding_window as osw

    g = np.load(os.path.join(golden_dir, "buffered.npz"))

    def toy(k_out):
        return lambda x: torch.cat([torch.sin(x[:, :1] * (1.0 + 0.37 * k)) + 0.05 * k * x[:, :1] for k in range(k_out)], dim=1)

    i = 0
    while f"buf_{i}_shape" in g:
        torch.manual_seed(int(g[f"buf_{i}_seed"]))
        x = torch.rand(tuple(int(v) for v in g[f"buf_{i}_shape"]))
        y = osw.sliding_window_inference(x, tuple(int(v) for v in g[f"buf_{i}_roi"]), int(g[f"buf_{i}_sw"]), toy(int(g[f"buf_{i}_k"])), overlap=float(g[f"buf_{i}_ov"]),
                                         mode=str(g[f"buf_{i}_mode"]), padding_mode="constant", cval=-0.5, buffer_steps=int(g[f"buf_{i}_steps"]), buffer_dim=int(g[f"buf_{i}_dim"]))
        assert np.array_equal(y.numpy(), g[f"buf_{i}_out"]), f"buffered case {i}"
        i += 1
    assert i >= 8
