"""Activations / AsDiscrete cases shared by the golden generator (real reference, CPU) and the emulator / MI355X tests."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def logits(seed=0, shape=(5, 6, 7, 8)):
    gen = torch.Generator().manual_seed(1200 + seed)
    x = torch.randn(*shape, generator=gen) * 3.0
    x[:, 0, 0, 0] = 1.5                      # an exact tie over all channels: argmax must pick channel 0
    x[2, 1, 1, 1] = float("nan")             # NaN counts as the maximum
    x[[1, 3], 2, 2, 2] = 7.25                # a two-way tie: the first index wins
    return x


POST_CASES = [
    ("act_softmax", "Activations", {"softmax": True}, {}),
    ("act_sigmoid", "Activations", {"sigmoid": True}, {}),
    ("act_other", "Activations", {"other": torch.tanh}, {}),
    ("act_call_softmax", "Activations", {}, {"softmax": True}),
    ("disc_argmax", "AsDiscrete", {"argmax": True}, {}),
    ("disc_argmax_onehot", "AsDiscrete", {"argmax": True, "to_onehot": 5}, {}),
    ("disc_threshold", "AsDiscrete", {"threshold": 0.25}, {}),
    ("disc_round", "AsDiscrete", {"rounding": "torchrounding"}, {}),
    ("disc_argmax_onehot_thr", "AsDiscrete", {"argmax": True, "to_onehot": 5, "threshold": 0.5}, {}),
    ("disc_call_args", "AsDiscrete", {}, {"argmax": True, "to_onehot": 6}),
]


def run_all(mod_transforms, device):
    out = {}
    x = logits().to(device)
    for name, cls, init, call in POST_CASES:
        tr = getattr(mod_transforms, cls)(**init)
        out[name] = tr(x, **call).cpu().numpy()
    d = {"pred": logits(1).to(device), "other": logits(2).to(device)}
    d = mod_transforms.Activationsd(keys=["pred", "other"], softmax=[True, False], sigmoid=[False, True])(d)
    d = mod_transforms.AsDiscreted(keys=["pred", "other"], argmax=[True, False], to_onehot=[5, None], threshold=[None, 0.5])(d)
    out["dict_pred"] = d["pred"].cpu().numpy()
    out["dict_other"] = d["other"].cpu().numpy()
    return out


EXACT = ("disc_", "dict_")


def case_post_transforms_vs_reference(device):
    """Activations / AsDiscrete (+ the dictionary versions) against the real reference transforms
    (tests/golden/make_golden_post.py): discrete outputs exact (ties -> first index, NaN -> maximal), softmax / sigmoid
    within 1e-6."""
    import monai_amd.transforms as ours

    g = np.load(os.path.join(GOLDEN, "post_transforms.npz"))
    got = run_all(ours, device)
    assert set(got) == set(g.files)
    for name, y in got.items():
        exp = g[name]
        assert y.shape == exp.shape and y.dtype == np.float32, (name, y.shape, exp.shape, y.dtype)
        if name.startswith(EXACT):
            np.testing.assert_array_equal(y, exp, err_msg=name)
        else:
            np.testing.assert_allclose(y, exp, rtol=0, atol=1e-6, err_msg=name, equal_nan=True)
    return len(got)


def case_post_transforms_api(device):
    import pytest

    from monai_amd.transforms import Activations, AsDiscrete, AsDiscreted

    x = logits().to(device)
    with pytest.raises(ValueError):
        Activations()(x, sigmoid=True, softmax=True)
    with pytest.raises(TypeError):
        Activations(other=3)
    with pytest.raises(ValueError):
        AsDiscrete(to_onehot=True)
    with pytest.raises(ValueError):
        AsDiscrete()(x, to_onehot=2.0)
    with pytest.raises(AssertionError):
        AsDiscrete(to_onehot=5)(x)                      # five channels: not a label map
    with pytest.raises(ValueError):
        AsDiscrete(rounding="floor")(x)
    with pytest.raises(NotImplementedError):
        AsDiscrete(argmax=True, dim=1)(x)
    with pytest.raises(KeyError):
        AsDiscreted(keys=["missing"], argmax=True)({"pred": x})
    assert AsDiscreted(keys=["missing"], argmax=True, allow_missing_keys=True)({"pred": x})["pred"] is x
