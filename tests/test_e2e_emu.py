"""End-to-end product path (SlidingWindowInferer -> BasicUNet engine -> blend) on the CPU through the SIMT
emulator build of the kernels, checked against the golden outputs of the real reference."""
import pytest

import e2e_cases as ec


def test_blend_only_bitwise_vs_reference(emu):
    ec.case_blend_only_vs_golden("cpu")


def test_net_single_window_vs_reference(emu):
    r, r2 = ec.case_net_single_window_vs_golden("cpu")
    print(r, r2)


def test_net_single_window_fp16_split_precision_vs_reference(emu, monkeypatch):
    """The product default: 3x3x3 convolutions on the fp16 matrix cores in two-piece split precision -- the SAME golden logits of the
    real reference and the same tolerance as the exact-fp32 kernels."""
    monkeypatch.delenv("MONAI_AMD_CONV_ALGO", raising=False)
    from monai_amd import ops

    assert ops.conv3d_k3_select(32, 32, 32, 32, 32, bounded=True) == ops.conv3d_k3_h2w_config() and ops.conv3d_k3_select(64, 64, 16, 16, 16, bounded=True) == ops.conv3d_k3_h2_config()
    print(ec.case_net_single_window_vs_golden("cpu", second_window=False))      # the split-precision kernel is 10x slower to emulate: one golden batch


def test_sliding_window_net5_vs_reference(emu):
    print(ec.case_sliding_window_net5_vs_golden("cpu"))


def test_unetr_small_vs_reference(emu):
    print(ec.case_unetr_small_vs_golden("cpu"))


def test_unet_vs_reference(emu):
    """UNet with residual units, plain, and with a stride-1 level."""
    print(ec.case_unet_vs_golden("cpu"))


def test_unet_activations_and_adn_orderings_vs_reference(emu):
    """UNet with ReLU / LeakyReLU, `adn_ordering` "NAD" / "AN" (activation before the normalisation) / "A" (no normalisation) / "ADN" + batch norm: the reference's own
    logits (tests/golden/unet_variants.npz), state_dict keys in the reference's order"""
    import e2e_cases

    print(ec.case_unet_vs_golden("cpu", names=tuple(e2e_cases.UNET_VARIANTS), golden="unet_variants.npz"))


@pytest.mark.heavy_emu
def test_basic_unet_with_inplane_winograd(emu, monkeypatch):
    """The whole BasicUNet window path with every eligible 3x3x3 conv on the in-plane Winograd configuration."""
    monkeypatch.setenv("MONAI_AMD_CONV_ALGO", "wino2d")
    print(ec.case_net_single_window_vs_golden("cpu"))


def test_basic_unet_odd_window_vs_reference(emu):
    print(ec.case_net_odd_window_vs_golden("cpu"))


def test_process_fn_bitwise_vs_reference(emu):
    ec.case_process_fn_vs_golden("cpu")


def test_buffered_schedule_bitwise_vs_reference(emu):
    """row a7: `buffer_steps` / `buffer_dim` reproduce the summation order of the reference's buffered schedule"""
    assert ec.case_buffered_blend_vs_golden("cpu") >= 8


def test_slabwise_equals_whole(emu):
    print(ec.case_slabwise_equals_whole("cpu"))


@pytest.mark.heavy_emu
def test_swin_unetr_vs_reference(emu):
    import swin_cases as sc

    print(sc.case_swin_unetr_vs_golden("cpu", names=("a",)))


@pytest.mark.heavy_emu
def test_fused_argmax_epilogue(emu):
    assert ec.case_fused_argmax_epilogue("cpu")


def test_bundle_shaped_pipeline_vs_reference(emu):
    import pipeline_case as pl

    print(pl.case_pipeline_vs_reference("cpu"))


def test_conv_engine_splits_couts_16_mod_32(emu):
    print(ec.case_conv_cout_16_mod_32_split("cpu"))
    print(ec.case_conv_cout_16_mod_32_split("cpu", cin=16, cout=80, dims=(3, 8, 24), n=1))


def test_dynunet_vs_reference(emu):
    import dynunet_cases as dc

    print("max |dlogit|", dc.case_dynunet_vs_reference("cpu"))
    dc.case_dynunet_api("cpu")


def test_dynunet_concat_wider_than_the_record_table(emu):
    import dynunet_cases as dc
    from monai_amd import config

    with config.conv_algo_scope("auto"):
        print("max |d|", dc.case_dynunet_wide_concat("cpu"))


@pytest.mark.heavy_emu
def test_strided_convolutions_of_dynunet_and_segresnet_on_matrix_cores(emu):
    """the reference goldens with the down-sampling convolutions on the split-precision stride-2 kernel (the `emu` fixture pins the exact-fp32 family otherwise)"""
    import dynunet_cases as dc
    import segresnet_cases as sc
    from monai_amd import config, ops

    calls, orig = [], ops.conv3d_k3s2

    def spy(x, *a, **k):
        calls.append(tuple(x.shape))
        return orig(x, *a, **k)

    packs, orig_pack = [], ops.deconv_k2s2_h2_packed

    def spy_pack(w):
        packs.append(tuple(w.shape))
        return orig_pack(w)

    ops.conv3d_k3s2, ops.deconv_k2s2_h2_packed = spy, spy_pack
    try:
        with config.conv_algo_scope("auto"):
            print("dynunet max |dlogit|", dc.case_dynunet_vs_reference("cpu", names=("basic",)))
            assert calls == [(2, 16, 32, 32, 32)], calls          # smaller planes stay on the direct kernel (ops.conv3d_k3s2_selected)
            assert (64, 48, 2, 2, 2) in packs and (32, 16, 2, 2, 2) in packs, packs      # its transposed convolutions ran on the matrix cores (kernels/deconv_h2.h)
            calls.clear()
            print("segresnet max |dlogit|", sc.case_segresnet_vs_reference("cpu", names=("f16",)))
            assert len(calls) == 2, calls
    finally:
        ops.conv3d_k3s2, ops.deconv_k2s2_h2_packed = orig, orig_pack


@pytest.mark.heavy_emu
def test_dynunet_2d_and_slice_inferer_vs_reference(emu):
    """a product 2-D network (DynUNet on the one-plane 3-D engine) under SliceInferer, against the real reference"""
    import dynunet_cases as dc

    print("max |dlogit|", dc.case_dynunet_2d_vs_reference("cpu"))


@pytest.mark.heavy_emu
def test_dynunet_sliding_window_vs_reference(emu):
    import dynunet_cases as dc

    print("max |dlogit|", dc.case_dynunet_sliding_window("cpu"))


def test_segresnet_vs_reference(emu):
    import segresnet_cases as sc

    print("max |dlogit|", sc.case_segresnet_vs_reference("cpu"))
    print("sliding window", sc.case_segresnet_sliding_window("cpu"))
    sc.case_segresnet_api("cpu")


@pytest.mark.heavy_emu
def test_ct_bundle_pipeline_vs_reference(emu):
    import pipeline_ct_case as pc

    print(pc.case_ct_pipeline_vs_reference("cpu"))


@pytest.mark.heavy_emu
def test_mri_bundle_pipeline_vs_reference(emu):
    import normalize_cases as nc

    print(nc.case_mri_pipeline_vs_reference("cpu"))


def test_dynunet_segresnet_window_vs_oracle(emu):
    import dynunet_cases as dc

    print("max |dlogit|", dc.case_nets_window_vs_oracle("cpu", 48, (16, 32, 64)))     # the -m gpu run does this at 96^3 with nnU-Net filters


@pytest.mark.heavy_emu
def test_unetr_under_autocast_stays_fp32(emu):
    """an evaluator with amp=True calls the network inside torch.autocast: the engine keeps computing in fp32 and returns the same logits"""
    import torch

    from monai_amd.networks.nets import UNETR

    torch.manual_seed(5)
    net = UNETR(in_channels=1, out_channels=2, img_size=(32, 32, 32), feature_size=8, hidden_size=128, mlp_dim=256, num_heads=2).eval()
    x = torch.rand(1, 1, 32, 32, 32)
    y = net(x)
    with torch.autocast(device_type="cpu", dtype=torch.bfloat16):
        ya = net(x)
    assert ya.dtype == torch.float32 and torch.equal(y, ya)


def test_dropout_arguments_are_inference_inert(emu):
    """bundles configure nets with dropout; Dropout holds no parameters and the engines only run in eval mode: same weights, same logits"""
    import torch

    from monai_amd.networks.nets import BasicUNet, DynUNet, UNet

    x = torch.rand(1, 1, 16, 16, 16)
    for make, none in ((lambda d: UNet(3, 1, 2, (8, 16), (2,), num_res_units=1, dropout=d), 0.0),
                       (lambda d: BasicUNet(3, 1, 2, features=(8, 8, 16, 16, 16, 8), dropout=d), 0.0),
                       (lambda d: DynUNet(3, 1, 2, [3, 3, 3], [1, 2, 2], [2, 2], filters=[8, 8, 8], dropout=d), None)):
        torch.manual_seed(8)
        a = make(none).eval()
        torch.manual_seed(8)
        b = make(0.3).eval()
        assert list(a.state_dict().keys()) == list(b.state_dict().keys())
        assert torch.equal(a(x), b(x))


def test_narrow_and_host_inputs(emu):
    ec.case_narrow_and_host_inputs("cpu")


def test_basic_unet_2d_and_slice_inferer_vs_reference(emu):
    """BasicUNet(spatial_dims=2) on the one-plane engine and SliceInferer over it, against the real reference"""
    print("max |dlogit|", ec.case_basic_unet_2d_vs_reference("cpu"))


def test_nets_with_trained_like_affine_spreads(emu):
    """gamma in +-[1e-3, 1e3], |beta| to ~1e3, a raw-CT-valued window: split-precision path vs oracle at 1e-4 of the logit scale, and vs the fp32 kernels"""
    print(ec.case_nets_with_spread_affine("cpu", window=(16, 16, 16), nets=("dynunet_res", "segresnet")))      # BasicUNet (32^3 at least): the -m gpu twin


@pytest.mark.heavy_emu
def test_net_nonfinite_inputs_like_the_reference(emu):
    ec.case_net_nonfinite_inputs("cpu", features=(32, 32, 32, 32, 32, 32))


def test_mosaic_layout_equals_window_major(emu):
    ec.case_mosaic_layout_equals_window_major("cpu", cases=(((1, 1, 40, 56, 32), 0.5, "gaussian"),), features=(8, 8, 8, 16, 16, 8))      # 2 x 3 x 1 windows, clipped last ones


@pytest.mark.heavy_emu          # four minutes of emulated split-precision kernels; the composite kernel's own cases (test_kernels_emu.py) and the -m gpu twin stay
def test_upcat_fused_vs_two_layers_and_reference(emu):
    """UpCat without its up-sampled intermediate (kernels/upconv_h2.h) inside BasicUNet: golden logits of the real reference + the engine's two-layer path"""
    print(ec.case_net_upcat_fused_vs_two_layers("cpu"))


def test_basic_unet_pixelshuffle_vs_reference(emu):
    """BasicUNet(upsample="pixelshuffle") on the HIP path against the real reference's golden logits (the odd-extent input and the two-dimensional net; the GPU twin runs all three)"""
    print(ec.case_basic_unet_pixelshuffle_vs_golden("cpu", which=("odd", "2d")))


@pytest.mark.heavy_emu          # minutes of emulated split-precision kernels; the -m gpu twin runs every round
def test_conv_halves_vs_one_launch_and_reference(emu):
    """UpCat's convolution over a 64-channel concatenation as two 32-channel launches of the Winograd split-precision kernel (BasicUNet._conv_halves)"""
    print(ec.case_net_conv_halves_vs_one_launch("cpu"))


def test_buffered_schedule_with_callbacks_bitwise_vs_reference(emu):
    """SURVEY 8a row a7 with the rest of its call surface: process_fn / with_coord / tuple and dict outputs under buffer_steps"""
    assert ec.case_buffered_calls_vs_golden("cpu") == 6


@pytest.mark.heavy_emu
def test_pooling_epilogue_leaves_the_logits_bitwise(emu):
    assert ec.case_net_pool_fused_bitwise("cpu")


@pytest.mark.heavy_emu
def test_swin_attention_from_table_and_regions_bitwise(emu):
    """round 5: window attention with bias / mask evaluated in the kernel == the S x S table form through the whole SwinUNETR, bit for bit"""
    import swin_cases as sc

    print(sc.case_swin_rel_attention_bitwise("cpu"))


@pytest.mark.heavy_emu
def test_swin_block_moves_folded_into_kernels_bitwise(emu):
    import swin_cases as sc

    assert sc.case_swin_fused_moves_bitwise("cpu")
