"""-m gpu: resampling transforms (Spacing/Spacingd/SpatialResample/AffineTransform/Resample/grid_pull) on the MI355X
against the real reference's outputs (tests/golden), the reference's compiled C++ resampler (oracle/_ref, when the
prebuilt .so travelled with the snapshot) and the CPU oracle at a larger size."""
import numpy as np
import pytest
import torch

import transform_cases as tc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_spacing_reference_tables():
    tc.case_spacing_reference_tables(DEV)


def test_spacing_3d_all_modes():
    print("worst bilinear error", tc.case_spacing_3d(DEV))


def test_spacingd_two_keys_and_inverse():
    tc.case_spacingd(DEV)


def test_affine_transform_flags():
    tc.case_affine_transform(DEV)


def test_grid_pull_vs_reference_build():
    print("worst grid_pull error", tc.case_grid_pull_vs_reference_build(DEV))


def test_grid_pull_reference_golden_rows():
    tc.case_grid_pull_reference_golden_rows(DEV)


def test_pushpull_vs_reference_build():
    print("worst scatter error", tc.case_pushpull_vs_reference_build(DEV))


def test_grid_pull_reference_rows_all_orders():
    tc.case_grid_pull_reference_rows_all_orders(DEV)


def test_grid_functions_autograd():
    print(tc.case_grid_functions_autograd(DEV))


def test_resample_dense_grid():
    tc.case_resample_dense_grid(DEV)


def test_grid_pull_live_against_oracle_ref():
    from monai_amd import _C
    from oracle import build_ref

    ref = build_ref.load()
    if ref is None:
        pytest.skip("oracle/_ref not present in this snapshot")
    torch.manual_seed(9)
    inp = torch.randn(2, 3, 24, 28, 20)
    grid = (torch.rand(2, 18, 22, 26, 3) * 2.4 - 0.7) * torch.tensor([24.0, 28.0, 20.0])
    for b in (0, 1, 2, 3, 4, 5, 7):
        for interp in (0, 1):
            exp = ref.grid_pull(inp, grid, [ref.BoundType(b)], [ref.InterpolationType(interp)], True)
            got = _C.grid_pull(inp.to(DEV), grid.to(DEV), [_C.BoundType(b)], [_C.InterpolationType(interp)], True)
            assert (got.cpu() - exp).abs().max().item() < 5e-5, (b, interp)


def test_spacing_config4_shape_vs_oracle():
    """BASELINE.json configs[4] geometry at 1/4 size: affine diag(0.8, 0.8, 1.6) -> pixdim 1: (128^3 -> 103x103x204), fp64
    and fp32 interpolation against the CPU oracle (torch affine_grid + grid_sample)."""
    from monai_amd.data import MetaTensor
    from monai_amd.transforms import Spacing
    from oracle import resample as orz

    torch.manual_seed(0)
    x = torch.rand(1, 128, 128, 128)
    aff = np.diag([0.8, 0.8, 1.6, 1.0])
    for dt, tol in ((np.float64, tc.TOL_F64), (np.float32, tc.TOL_F32)):
        y = Spacing(pixdim=(1.0, 1.0, 1.0), mode="bilinear", padding_mode="border", dtype=dt)(MetaTensor(x.to(DEV), affine=aff))
        assert tuple(y.shape) == (1, 103, 103, 204)  # round(127 * s) + 1 per axis
        xform = np.linalg.solve(aff, np.asarray(y.affine))
        ref = orz.spatial_resample_eager(x, xform, (103, 103, 204), "bilinear", "border", False, torch.float64 if dt is np.float64 else torch.float32)
        assert (y.cpu() - ref).abs().max().item() < tol


def test_gaussian_smooth():
    tc.case_gaussian_smooth(DEV)


def test_gaussian_smooth_128_vs_torch():
    """sigma = 1 (9 taps) on 2 x 128^3 against the reference operator sequence on the CPU (F.pad + depthwise F.conv3d)."""
    import torch.nn.functional as F

    from monai_amd.networks.layers import gaussian_1d
    from monai_amd.transforms import GaussianSmooth

    torch.manual_seed(3)
    x = torch.rand(2, 128, 128, 128)
    y = GaussianSmooth(sigma=1.0)(x.to(DEV)).cpu()
    k = gaussian_1d(1.0)
    ref = x[None]
    for ax in range(3):
        shape = [1, 1, 1]
        shape[ax] = -1
        w = k.reshape(shape)[None, None].repeat(2, 1, 1, 1, 1)
        pad = [0, 0, 0, 0, 0, 0]
        pad[2 * (2 - ax)] = pad[2 * (2 - ax) + 1] = 4
        ref = F.conv3d(F.pad(ref, pad), w, groups=2)
    assert (y - ref[0]).abs().max().item() < 1e-5


def test_separable_fast_path_equals_general():
    tc.case_separable_vs_general(DEV)


def test_general_rows_kernel_equals_linear_index_kernel():
    tc.case_general_rows_vs_linear(DEV)


def test_gaussian_rowvec_equals_tile():
    tc.case_gaussian_rowvec_equals_tile(DEV)


def test_gaussian_z_chunks():
    tc.case_gaussian_z_chunks(DEV)


def test_resample_compiled_vs_reference():
    print("worst error", tc.case_resample_compiled_vs_reference(DEV))


def test_warp_vs_reference():
    print("worst error by build mode", tc.case_warp_vs_reference(DEV))


def test_pushpull_tiny_extents_wide_coordinates():
    print("cases", tc.case_pushpull_tiny_extents_wide_coordinates(DEV))


def test_post_transforms_vs_reference():
    import post_cases as pc

    print("arrays", pc.case_post_transforms_vs_reference(DEV))
    pc.case_post_transforms_api(DEV)


def test_lazy_resampling_vs_reference():
    """SURVEY 8f-2 on the MI355X: the chains of tests/golden/lazy.npz (real reference, Compose(lazy=True)) with one fused launch per image"""
    import lazy_cases as lc

    print("launches", lc.case_lazy_chains_vs_reference("cuda"))
    assert lc.case_lazy_orientation_spacing_fused("cuda")


def test_conventions_pinned_by_the_reference_suites():
    """what tests/test_reference_suites_emu.py established against the reference's own tests, restated without the reference"""
    import lazy_cases as lc

    tc.case_reference_argument_conventions(DEV)
    assert lc.case_axis_only_resample_conventions(DEV)
