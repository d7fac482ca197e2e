"""Resampling transforms on the CPU through the SIMT-emulator build of the kernels, against the real reference's
outputs; plus the oracle's resampling restatement pinned bit-for-bit against the same golden vectors."""
import os

import numpy as np
import torch

import transform_cases as tc


def test_oracle_affine_transform_bitwise_vs_reference(golden_dir):
    from oracle import resample as orz

    g = np.load(os.path.join(golden_dir, "resample.npz"))
    src, theta = torch.from_numpy(g["at_src"]), torch.from_numpy(g["at_theta"])
    for k in range(int(g["at_n"])):
        normalized, rev, ac, pad, nearest = (int(v) for v in g[f"at_{k}_cfg"])
        th = theta.clone()
        if normalized:
            th[:3, :3] = torch.eye(3) + 0.1 * (theta[:3, :3] - torch.eye(3))
            th[:3, 3] = theta[:3, 3] * 0.1
        y = orz.affine_transform(src, th, spatial_size=(7, 12, 10), normalized=bool(normalized), mode="nearest" if nearest else "bilinear",
                                 padding_mode=tc.PADS[pad], align_corners=bool(ac), reverse_indexing=bool(rev))
        assert np.array_equal(y.numpy(), g[f"at_{k}_out"]), k
    y = orz.affine_transform(src, theta, normalized=False, zero_centered=True, align_corners=False)
    assert np.array_equal(y.numpy(), g["at_zc_out"])


def test_spacing_reference_tables(emu):
    tc.case_spacing_reference_tables("cpu")


def test_spacing_3d_all_modes(emu):
    print("worst bilinear error", tc.case_spacing_3d("cpu"))


def test_spacingd_two_keys_and_inverse(emu):
    tc.case_spacingd("cpu")


def test_affine_transform_flags(emu):
    tc.case_affine_transform("cpu")
