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


def test_grid_pull_vs_reference_build(emu):
    print("worst grid_pull error", tc.case_grid_pull_vs_reference_build("cpu"))


def test_grid_pull_reference_golden_rows(emu):
    tc.case_grid_pull_reference_golden_rows("cpu")


def test_pushpull_vs_reference_build(emu):
    print("worst scatter error", tc.case_pushpull_vs_reference_build("cpu"))


def test_grid_pull_reference_rows_all_orders(emu):
    tc.case_grid_pull_reference_rows_all_orders("cpu")


def test_grid_functions_autograd(emu):
    print(tc.case_grid_functions_autograd("cpu"))


def test_resample_dense_grid(emu):
    tc.case_resample_dense_grid("cpu")


def test_grid_pull_live_against_oracle_ref(emu):
    """When oracle/_ref (the reference's own C++ resampler, built by oracle/build_ref.py) is present, compare live."""
    import pytest

    from monai_amd import _C
    from oracle import build_ref

    ref = build_ref.load()
    if ref is None:
        pytest.skip("oracle/_ref not built (python oracle/build_ref.py in the build container)")
    torch.manual_seed(9)
    inp = torch.randn(1, 2, 6, 7, 8)
    grid = (torch.rand(1, 5, 4, 6, 3) * 2.4 - 0.7) * torch.tensor([6.0, 7.0, 8.0])
    for b in (0, 1, 2, 3, 4, 5, 7):
        for interp in (0, 1):
            exp = ref.grid_pull(inp, grid, [ref.BoundType(b)], [ref.InterpolationType(interp)], True)
            got = _C.grid_pull(inp, grid, [_C.BoundType(b)], [_C.InterpolationType(interp)], True)
            assert (got - exp).abs().max().item() < 2e-5, (b, interp)


def test_gaussian_1d_tables():
    tc.case_gaussian_1d_tables()


def test_gaussian_smooth(emu):
    tc.case_gaussian_smooth("cpu")


def test_separable_fast_path_equals_general(emu):
    tc.case_separable_vs_general("cpu")


def test_general_rows_kernel_equals_linear_index_kernel(emu):
    tc.case_general_rows_vs_linear("cpu")


def test_gaussian_z_chunks(emu):
    tc.case_gaussian_z_chunks("cpu")


def test_gaussian_rowvec_equals_tile(emu):
    tc.case_gaussian_rowvec_equals_tile("cpu")


def test_resample_compiled_vs_reference(emu):
    print("worst error", tc.case_resample_compiled_vs_reference("cpu"))


def test_warp_vs_reference(emu):
    print("worst error by build mode", tc.case_warp_vs_reference("cpu"))


def test_pushpull_tiny_extents_wide_coordinates(emu):
    print("cases", tc.case_pushpull_tiny_extents_wide_coordinates("cpu"))


def test_post_transforms_vs_reference(emu):
    import post_cases as pc

    print("arrays", pc.case_post_transforms_vs_reference("cpu"))


def test_post_transforms_api(emu):
    import post_cases as pc

    pc.case_post_transforms_api("cpu")


def test_lazy_resampling_vs_reference(emu):
    import lazy_cases as lc

    print("launches", lc.case_lazy_chains_vs_reference("cpu"))
    assert lc.case_lazy_orientation_spacing_fused("cpu")


def test_preproc_vs_reference(emu):
    import preproc_cases as pc

    print("arrays", pc.case_preproc_vs_reference("cpu"))
    print("boxes", pc.case_bbox_large("cpu"))


def test_preproc_api(emu):
    import preproc_cases as pc

    pc.case_preproc_api("cpu")


def test_orientation_reference_tables(emu):
    import orientation_cases as oc

    print("rows", oc.case_orientation_reference_tables("cpu"))
    oc.case_orientation_api("cpu")


def test_orientation_kernel_and_inverse(emu):
    import orientation_cases as oc

    print("axis codes", oc.case_orientation_kernel_and_inverse("cpu"))


def test_normalize_intensity_vs_reference(emu):
    import normalize_cases as nc

    print("worst relative error", nc.case_normalize_vs_reference("cpu"))
    nc.case_normalize_api("cpu")


def test_preproc_properties_small(emu):
    import preproc_cases as pc

    pc.case_preproc_full_size("cpu", 48)      # the -m gpu run does this at 512^3


def test_croppad_family_vs_reference(emu):
    import croppad_cases as cc

    print("arrays", cc.case_croppad_vs_reference("cpu"))
    cc.case_croppad_api("cpu")


def test_flip_rotate90_vs_reference(emu):
    import flip_cases as fc

    print("arrays", fc.case_flip_rotate_vs_reference("cpu"))
    fc.case_flip_rotate_api("cpu")


def test_scale_intensity_vs_reference(emu):
    import normalize_cases as nc

    print("arrays", nc.case_scale_intensity_vs_reference("cpu"))


def test_conventions_pinned_by_the_reference_suites(emu):
    """what tests/test_reference_suites_emu.py established against the reference's own tests, restated without the reference"""
    import lazy_cases as lc

    tc.case_reference_argument_conventions("cpu")
    assert lc.case_axis_only_resample_conventions("cpu")
