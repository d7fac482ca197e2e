"""-m gpu: the widening rows of SURVEY.md 8(f) added after the last GPU session of round 1 -- the same cases the emulator
tests run (tests/test_transforms_emu.py, tests/test_e2e_emu.py), here through the real .so on the MI355X.  Kept in a file that
sorts after the others so that `pytest -x` reaches every test that has already run on the hardware first."""
import os

import pytest

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_preproc_vs_reference():
    import preproc_cases as pc

    print("arrays", pc.case_preproc_vs_reference(DEV))
    print("boxes", pc.case_bbox_large(DEV))


def test_preproc_api():
    import preproc_cases as pc

    pc.case_preproc_api(DEV)


def test_dynunet_vs_reference():
    import dynunet_cases as dc

    print("max |dlogit|", dc.case_dynunet_vs_reference(DEV))
    dc.case_dynunet_api(DEV)


def test_dynunet_concat_wider_than_the_record_table():
    import dynunet_cases as dc

    print("max |d|", dc.case_dynunet_wide_concat(DEV))
    print("max |d|", dc.case_dynunet_wide_concat(DEV, cin=512, cout=256, dims=(12, 12, 12)))


def test_dynunet_2d_and_slice_inferer_vs_reference():
    """SURVEY 8 row a9 on the MI355X: a product 2-D network (DynUNet on the one-plane 3-D engine) under SliceInferer, against the real reference"""
    import dynunet_cases as dc

    print("max |dlogit|", dc.case_dynunet_2d_vs_reference(DEV))


def test_dynunet_sliding_window_vs_reference():
    import dynunet_cases as dc

    print("max |dlogit|", dc.case_dynunet_sliding_window(DEV))


def test_loaded_extension_runs_on_the_device():
    """boundary B2: an extension built by ``load_module`` (hipcc on the box) launches on the device"""
    import ctypes
    import os

    import torch

    from monai_amd._extensions import loader

    loader.EXTENSION_DIRS.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ext_sample"))
    try:
        m = loader.load_module("axpb", defines={"AXPB_SCALE": 2})
    finally:
        loader.EXTENSION_DIRS.pop()
    x = torch.arange(1000, dtype=torch.float32, device="cuda")
    y = torch.empty_like(x)
    m.axpb_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
    assert m.axpb_f32(x.data_ptr(), y.data_ptr(), x.numel(), 0.5, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), 2 * x.cpu() + 0.5)


def test_orientation():
    import orientation_cases as oc

    print("rows", oc.case_orientation_reference_tables(DEV))
    print("axis codes", oc.case_orientation_kernel_and_inverse(DEV))
    oc.case_orientation_api(DEV)


def test_segresnet_vs_reference():
    import segresnet_cases as sc

    print("max |dlogit|", sc.case_segresnet_vs_reference(DEV))
    print("sliding window", sc.case_segresnet_sliding_window(DEV))
    sc.case_segresnet_api(DEV)


def test_ct_bundle_pipeline_vs_reference():
    import pipeline_ct_case as pc

    print(pc.case_ct_pipeline_vs_reference(DEV))


def test_normalize_intensity_and_mri_pipeline():
    import normalize_cases as nc

    print("worst relative error", nc.case_normalize_vs_reference(DEV))
    nc.case_normalize_api(DEV)
    print(nc.case_mri_pipeline_vs_reference(DEV))
    print("ScaleIntensity arrays", nc.case_scale_intensity_vs_reference(DEV))


def test_unet_batch_norm_vs_reference():
    import e2e_cases as ec

    print(ec.case_unet_vs_golden(DEV, names=("batch",)))


def test_unet_activations_and_adn_orderings_vs_reference():
    """UNet(act=RELU / LEAKYRELU, adn_ordering="NAD" / "AN" / "A" / "ADN" with batch norm) on the HIP path, against the reference's own logits"""
    import e2e_cases as ec

    print(ec.case_unet_vs_golden(DEV, names=tuple(ec.UNET_VARIANTS), golden="unet_variants.npz"))


def test_preproc_properties_at_512():
    import preproc_cases as pc

    pc.case_preproc_full_size(DEV, 512)


def test_dynunet_segresnet_96_window_vs_oracle():
    import dynunet_cases as dc

    print("max |dlogit|", dc.case_nets_window_vs_oracle(DEV, 96, (32, 64, 128, 256)))


def test_croppad_family_vs_reference():
    import croppad_cases as cc

    print("arrays", cc.case_croppad_vs_reference(DEV))
    cc.case_croppad_api(DEV)


def test_flip_rotate90_vs_reference():
    import flip_cases as fc

    print("arrays", fc.case_flip_rotate_vs_reference(DEV))
    fc.case_flip_rotate_api(DEV)


def test_swin_unetr_vs_reference():
    """SwinUNETR (windowed attention kernel + the shared conv engine) against the real reference's logits"""
    import swin_cases as sc

    print(sc.case_swin_unetr_vs_golden("cuda"))


def test_basic_unet_on_the_exact_fp32_families_and_under_autocast():
    """GPU twins of emulator cases that `pytest -m 'not gpu'` only runs with MONAI_AMD_HEAVY_EMU=1: the golden BasicUNet windows with every eligible
    convolution on the in-plane Winograd family and on the direct fp32 tiles (config.CONV_ALGO), and a UNETR called inside torch.autocast (an evaluator
    with amp=True): the engine keeps computing in fp32 and returns the same logits."""
    import torch

    import e2e_cases as ec
    from monai_amd import config
    from monai_amd.networks.nets import UNETR

    saved = config.CONV_ALGO
    try:
        for algo in ("wino2d", "direct", "fp32"):
            config.CONV_ALGO = algo
            print(algo, ec.case_net_single_window_vs_golden(DEV))
    finally:
        config.CONV_ALGO = saved
    torch.manual_seed(5)
    net = UNETR(in_channels=1, out_channels=2, img_size=(32, 32, 32), feature_size=8, hidden_size=128, mlp_dim=256, num_heads=2).eval().to(DEV)
    x = torch.rand(1, 1, 32, 32, 32).to(DEV)
    y = net(x)
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        ya = net(x)
    assert ya.dtype == torch.float32 and torch.equal(y, ya)
