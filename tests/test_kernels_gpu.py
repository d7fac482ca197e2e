"""-m gpu: the kernel parity cases of tests/kernel_cases.py on the real MI355X, through the C ABI of the in-tree
libmonai_amd.so (the same cases run on the CPU against the emulator build in tests/test_kernels_emu.py)."""
import os

import pytest
import torch

import kernel_cases as kc
from monai_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu_and_native_lib():
    if not torch.cuda.is_available():
        pytest.fail("the -m gpu tests need an MI355X")
    from monai_amd import _lib

    assert _lib.lib().path.endswith("libmonai_amd.so")  # the HIP build, loaded from the package tree


def test_window_extract():
    kc.case_window_extract(DEV)
    kc.case_window_extract(DEV, img=(1, 13, 10, 11), roi=(5, 4, 3), overlap=0.25)
    kc.case_window_extract(DEV, img=(1, 128, 120, 112), roi=(96, 96, 96), overlap=0.5)


@pytest.mark.parametrize("mode", ["gaussian", "constant"])
def test_sw_blend_bitwise(mode):
    kc.case_sw_blend(DEV, mode=mode)
    kc.case_sw_blend(DEV, img=(11, 13, 10), roi=(4, 5, 3), overlap=0.25, k=3, mode=mode)
    kc.case_sw_blend(DEV, img=(1, 20, 24), roi=(1, 8, 8), overlap=0.75, k=11, mode=mode)
    kc.case_sw_blend(DEV, img=(144, 112, 160), roi=(96, 96, 96), overlap=0.5, k=5, mode=mode)  # bench-shaped windows


@pytest.mark.parametrize("g", ["1", "2", "4", "8"])
def test_sw_blend_batch_sizes(g, monkeypatch):
    monkeypatch.setenv("MONAI_AMD_BLEND_G", g)
    monkeypatch.setenv("MONAI_AMD_BLEND_NT", "0" if g == "2" else "1")
    kc.case_sw_blend(DEV)
    kc.case_sw_blend(DEV, img=(20, 40, 52), roi=(8, 16, 16), overlap=0.5)
    kc.case_sw_blend(DEV, img=(144, 112, 160), roi=(96, 96, 96), overlap=0.5, k=5)


def test_sw_blend_special_cases(monkeypatch):
    kc.case_sw_blend_special_values(DEV)
    kc.case_sw_blend_irregular(DEV)
    kc.case_sw_blend_many_windows(DEV)
    kc.case_sw_blend_many_windows(DEV, slices=700)
    monkeypatch.setenv("MONAI_AMD_BLEND_LEGACY", "1")
    kc.case_sw_blend(DEV)


def test_conv_direct():
    kc.case_conv3d(DEV, 0, 2, 1, 32, (6, 7, 9), with_nrm=False, fused_stats=False)
    kc.case_conv3d(DEV, 0, 1, 5, 7, (4, 5, 6), fused_stats=False)
    kc.case_conv3d(DEV, 0, 2, 1, 32, (32, 48, 96), with_nrm=False, fused_stats=False)


@pytest.mark.parametrize(
    "cfg,cin,cout,dims",
    [
        (1, 8, 32, (4, 8, 32)), (1, 16, 32, (5, 6, 40)), (1, 64, 32, (16, 16, 96)),
        (2, 8, 32, (4, 8, 16)), (2, 32, 32, (6, 10, 24)), (2, 64, 32, (16, 48, 48)),
        (3, 8, 64, (2, 8, 8)), (3, 64, 64, (3, 12, 12)), (3, 128, 64, (24, 24, 24)),
        (4, 4, 128, (2, 4, 4)), (4, 256, 128, (12, 12, 12)),
        (5, 2, 256, (2, 4, 4)), (5, 256, 256, (6, 6, 6)),
        (6, 8, 32, (4, 8, 8)), (6, 32, 32, (5, 9, 7)),
        (7, 1, 32, (4, 8, 32)), (7, 3, 32, (5, 5, 33)), (7, 1, 32, (32, 32, 96)),
        (8, 8, 32, (4, 8, 32)), (8, 6, 32, (5, 6, 40)), (8, 64, 32, (16, 16, 96)),
        (9, 8, 32, (6, 10, 24)), (9, 64, 32, (16, 48, 48)), (1, 5, 32, (4, 4, 32)),
        (10, 6, 32, (6, 10, 24)), (10, 64, 32, (16, 48, 48)), (11, 8, 64, (3, 12, 12)), (12, 6, 64, (2, 8, 8)),
        (12, 128, 64, (24, 24, 24)), (13, 4, 128, (3, 6, 6)), (13, 256, 256, (6, 6, 6)), (14, 4, 32, (5, 9, 7)),
    ],
)
def test_conv_mfma_configs(cfg, cin, cout, dims):
    kc.case_conv3d(DEV, cfg, 2, cin, cout, dims, fused_stats=True, tol=5e-5)


def test_conv_mfma_separate_stats_and_select():
    assert kc.case_conv3d(DEV, None, 1, 8, 32, (4, 8, 32), fused_stats=False) >= 1


def test_conv_into_channel_slice():
    kc.case_conv3d_into_channel_slice(DEV)


def test_pool_deconv_1x1_stats():
    kc.case_maxpool(DEV)
    kc.case_maxpool(DEV, dims=(4, 6, 7))
    kc.case_maxpool(DEV, n=2, c=32, dims=(32, 32, 96))
    kc.case_deconv(DEV)
    kc.case_deconv(DEV, n=2, cin=32, cout=32, dims=(8, 16, 48))
    kc.case_conv1x1(DEV)
    kc.case_conv1x1_stats(DEV)
    kc.case_conv1x1(DEV, n=1, cin=13, cout=27, dims=(3, 5, 7))     # 16 + 8 + 3 output channels, ragged channel batch, scalar path
    kc.case_conv1x1(DEV, n=1, cin=6, cout=16, dims=(2, 4, 8))
    kc.case_conv1x1(DEV, cin=7, cout=11, dims=(3, 5, 7))
    kc.case_instnorm_stats(DEV)
    kc.case_instnorm_stats(DEV, n=2, c=32, dims=(32, 48, 96))


def test_attention_and_add_act():
    kc.case_attention(DEV, b=1, s=8, heads=2)
    kc.case_attention(DEV, b=2, s=45, heads=1)
    print("attention max err", kc.case_attention(DEV, b=3, s=216, heads=12))
    print("attention max err, 512 tokens (UNETR img_size 128^3: the reference's docstring example)", kc.case_attention(DEV, b=2, s=512, heads=12))
    kc.case_attention(DEV, b=1, s=1000, heads=3)
    kc.case_attention(DEV, b=2, s=70, heads=2, hd=32)
    kc.case_attention(DEV, b=1, s=343, heads=4, hd=96)
    kc.case_attention(DEV, b=1, s=130, heads=2, hd=128)
    kc.case_add_act(DEV)


def test_conv_mfma_cout_padding():
    kc.case_conv3d(DEV, 7, 2, 1, 16, (4, 8, 32), tol=5e-5)
    kc.case_conv3d(DEV, 10, 1, 32, 16, (6, 10, 24), tol=5e-5)
    kc.case_conv3d(DEV, 13, 1, 8, 48, (3, 6, 6), tol=5e-5)


def test_strided_conv_and_deconv_k3():
    kc.case_strided_conv_and_deconv_k3(DEV)


WINO2D_CASES = [(8, 16, (4, 16, 16), 2), (16, 32, (6, 8, 24), 1), (8, 16, (30, 4, 8), 1), (24, 16, (3, 18, 16), 1),
                (32, 32, (48, 48, 48), 2), (64, 32, (96, 96, 96), 1), (128, 64, (24, 24, 24), 1)]
@pytest.mark.parametrize("cin,cout,dims,n", WINO2D_CASES)
def test_conv3d_wino2d(cin, cout, dims, n):
    """In-plane Winograd F(2x2, 3x3) + direct z taps, z-streaming: chunk halos, ragged regions, several cout groups."""
    from monai_amd import ops

    cfg = ops.conv3d_k3_num_configs()
    assert ops.conv3d_k3_accepts(cfg, cin, cout)
    kc.case_conv3d(DEV, cfg, n, cin, cout, dims, fused_stats=True, tol=5e-5)
    kc.case_conv3d(DEV, cfg, 1, cin, cout, dims, with_nrm=False, fused_stats=False, tol=5e-5)


# (6, 8, 24), (3, 18, 20), (2, 24, 56), (24, 24, 24) take the 8 x 32 region shape, the others 16 x 16
H2_CASES = [(16, 32, (4, 16, 16), 2), (32, 48, (5, 16, 16), 2), (48, 80, (3, 8, 24), 1),      # 48 / 80 couts: a half-filled last cout group (round 4)
            (32, 64, (6, 8, 24), 1), (16, 32, (30, 4, 8), 1), (48, 32, (3, 18, 20), 1), (256, 32, (2, 8, 12), 1), (64, 32, (2, 24, 56), 1),
            (128, 128, (24, 24, 24), 2)]

# 16-couts groups (two z-taps per matrix instruction): one group / two / three groups, resident and streamed weight slabs, both region shapes, ragged regions, two z-chunks
H2C_CASES = [(16, 16, (4, 16, 16), 2), (32, 16, (5, 16, 16), 1), (48, 32, (3, 8, 24), 1), (16, 48, (6, 8, 24), 1), (32, 16, (3, 18, 20), 1), (64, 16, (2, 24, 56), 1),
             (16, 16, (24, 8, 24), 1), (16, 16, (30, 4, 8), 1), (32, 16, (96, 96, 96), 2), (16, 16, (96, 96, 96), 1)]
@pytest.mark.parametrize("cin,cout,dims,n", H2C_CASES)
def test_conv3d_split_precision_16_couts(cin, cout, dims, n):
    """the direct split-precision kernel with output channel groups of 16 (conv3d_h2.h, C16): a completed plane is the sum of three partial planes held in different
    column halves -- the SAME tolerance as the 32-couts form and the fp32 tiles; what conv3d_k3_select returns for bounded 16-couts layers"""
    from monai_amd import ops

    cfg = ops.conv3d_k3_h2c_config()
    assert cfg > ops.conv3d_k3_num_configs() and ops.conv3d_k3_accepts(cfg, cin, cout) and not ops.conv3d_k3_accepts(cfg, 16, 24)
    kc.case_conv3d(DEV, cfg, n, cin, cout, dims, fused_stats=True)
    kc.case_conv3d(DEV, cfg, 1, cin, cout, dims, with_nrm=False, fused_stats=False)
    kc.case_conv3d_accumulate(DEV, n, cin, cout, dims, cfg=cfg)      # out += conv + bias, statistics of the sum (round 6: the Y set starts from the old values)
    if cout == 16 and dims[1] >= 8 and dims[2] >= 8:
        assert ops.conv3d_k3_select(cin, cout, *dims, bounded=True, algo=0) == cfg and ops.conv3d_k3_select(cin, cout, *dims, bounded=False, algo=0) != cfg and ops.conv3d_k3_select(cin, cout, *dims, bounded=True, algo=4) != cfg      # auto: yes; unbounded input or the exact-fp32 family: no

@pytest.mark.parametrize("cin,cout,dims,n", H2_CASES)
def test_conv3d_fp16_split_precision(cin, cout, dims, n):
    """z-streaming direct convolution on the fp16 matrix cores, two fp16 pieces per operand and three exact piece products per
    multiply with fp32 accumulation (conv3d_h2.h) -- held to the SAME tolerance as the fp32 kernels: chunk halos, ragged regions,
    several cout groups and channel chunks, fused statistics."""
    from monai_amd import ops

    cfg = ops.conv3d_k3_h2_config()
    assert cfg > ops.conv3d_k3_num_configs() and ops.conv3d_k3_accepts(cfg, cin, cout)
    kc.case_conv3d(DEV, cfg, n, cin, cout, dims, fused_stats=True)
    kc.case_conv3d(DEV, cfg, 1, cin, cout, dims, with_nrm=False, fused_stats=False)


# in-plane Winograd in front of the split product (conv3d_wino_h2.h): one region / borders on every side, two cout groups / four z-chunks of one region / the headline's two levels
H2W_CASES = [(32, 32, (3, 4, 16), 2), (32, 64, (5, 8, 32), 2), (32, 32, (48, 4, 16), 1), (32, 32, (96, 96, 96), 2), (32, 32, (48, 48, 48), 1)]
@pytest.mark.parametrize("cin,cout,dims,n", H2W_CASES)
def test_conv3d_h2w_winograd_split_precision(cin, cout, dims, n):
    """F(2x2, 3x3) in the plane, transformed weights in registers, split-precision products of the transformed operands -- held to the SAME tolerance as the direct
    split kernel and the fp32 tiles; then the accumulating and (even extents) the pooling forms"""
    from monai_amd import ops

    cfg = ops.conv3d_k3_h2w_config()
    assert cfg > ops.conv3d_k3_num_configs() and ops.conv3d_k3_accepts(cfg, cin, cout) and ops.conv3d_k3_h2w_fits(*dims)
    kc.case_conv3d(DEV, cfg, n, cin, cout, dims, fused_stats=True)
    kc.case_conv3d_accumulate(DEV, n, cin, cout, dims, cfg=cfg)
    if dims[0] % 2 == 0:
        kc.case_conv3d_pool(DEV, n, cin, cout, dims, cfg=cfg)


LINEAR_CASES = [(128, 64, 16, False, False), (200, 144, 48, False, True), (66, 40, 36, True, False), (256, 192, 64, True, True), (13824, 2304, 768, False, False), (1728, 768, 3072, False, True)]
@pytest.mark.parametrize("m,n,k,gelu,res", LINEAR_CASES)
def test_linear_fp16_split_precision(m, n, k, gelu, res):
    """nn.Linear (+ GELU, + residual): ragged rows / columns / K (multiples of 4), bias on and off"""
    kc.case_linear(DEV, m, n, k, gelu=gelu, residual=res)
    kc.case_linear(DEV, m, n, k, gelu=gelu, residual=res, bias=False)


@pytest.mark.parametrize("m,k", [(7, 48), (130, 768), (5, 100), (33, 96), (70, 24), (9, 192), (6, 256), (3, 50), (4, 260)])
def test_layernorm(m, k):
    kc.case_layernorm(DEV, m, k)


def test_sw_blend_mosaic_layout():
    kc.case_sw_blend_mosaic(DEV)


def test_h2_input_scaling():
    """the split-precision convolution at input magnitudes 1e-20 ... 1e20 (incl. > 65504): accuracy of the exact-fp32 tiles"""
    kc.case_h2_input_scaling(DEV)
    kc.case_h2_input_scaling(DEV, cin=64, cout=64, dims=(24, 24, 24))


def test_h2_nonfinite_and_missing_bounds():
    kc.case_h2_nonfinite_and_missing_bounds(DEV)
    kc.case_h2_nonfinite_and_missing_bounds(DEV, cin=48, cout=32, dims=(6, 20, 24))


def test_bound_producers():
    kc.case_bound_producers(DEV)


def test_conv_one_input_channel():
    """kernels/conv3d_c1.h on the MI355X: partial tiles, both cout group widths, z-chunks, a full-size window"""
    cfg = ops.conv3d_k3_c1_config()
    assert ops.conv3d_k3_select(1, 32, 96, 96, 96) == cfg
    kc.case_conv3d(DEV, cfg, 2, 1, 32, (5, 40, 36), with_nrm=False, fused_stats=True)
    kc.case_conv3d(DEV, cfg, 1, 1, 24, (3, 8, 8), with_nrm=True, fused_stats=True)
    kc.case_conv3d(DEV, cfg, 1, 1, 16, (50, 4, 8), with_nrm=False, fused_stats=True)
    kc.case_conv3d(DEV, cfg, 1, 1, 16, (2, 33, 4), with_nrm=False, fused_stats=False)
    kc.case_conv3d(DEV, cfg, 2, 1, 32, (96, 96, 96), with_nrm=False, fused_stats=True)


@pytest.mark.parametrize("s,hd", [(343, 16), (343, 32), (64, 16), (27, 8), (200, 8)])
def test_window_attention(s, hd):
    """SwinUNETR's WindowAttention core: head dims 16 / 32 on the split-precision matrix-core kernel (round 4), 8 on the VALU kernel"""
    kc.case_window_attention(DEV, bw=2 if DEV == "cpu" else 6, s=s, heads=2, hd=hd)


@pytest.mark.parametrize("n,cin,cout,dims,scale", [(2, 96, 48, (6, 8, 44), 1.0), (1, 40, 80, (5, 4, 60), 1e4), (2, 24, 5, (3, 4, 8), 1e-6), (1, 32, 64, (2, 2, 257 * 4), 1.0), (2, 96, 48, (48, 48, 48), 1.0), (1, 512, 256, (12, 12, 12), 1.0), (3, 768, 384, (6, 6, 6), 1.0), (1, 528, 33, (2, 3, 4), 1.0)])
def test_conv1x1_all_couts_from_one_read(n, cin, cout, dims, scale):
    """the split-precision 1x1x1 convolution (kernels/conv1x1_h2.h): values, statistics, input magnitudes 1e-6 ... 1e4"""
    print(kc.case_conv1x1_h2(DEV, n, cin, cout, dims, scale))


@pytest.mark.parametrize("cin,cout,dims", [(16, 5, (6, 8, 44)), (48, 3, (2, 5, 12)), (7, 8, (3, 4, 8))])
def test_residual_join_inside_the_output_convolution(cin, cout, dims):
    assert kc.case_conv1x1_sum2(DEV, 2, cin, cout, dims)


@pytest.mark.parametrize("m_src,m_out,k,n", [(150, 210, 48, 96), (64, 64, 192, 64), (37, 50, 384, 40), (20000, 20480, 96, 288), (3000, 4000, 768, 768)])
def test_layernorm_gather_linear_scatter(m_src, m_out, k, n):
    """SwinTransformerBlock's copies folded into the gathering LayerNorm and the projection's scattering epilogue (round 5)"""
    assert kc.case_layernorm_gather_linear_scatter(DEV, m_src, m_out, k, n)


@pytest.mark.parametrize("ws,n,hd,masked", [((7, 7, 7), None, 16, True), ((7, 7, 7), 216, 32, True), ((7, 7, 7), None, 32, False), ((3, 4, 5), None, 16, True)])
def test_window_attention_rel(ws, n, hd, masked):
    """bias from the relative-position table + mask from region ids inside the kernel (round 5) == the S x S table form bit for bit; 216 of 343 = a clamped 6^3 window"""
    kc.case_window_attention_rel(DEV, bw=6, ws=ws, n=n, heads=2, hd=hd, masked=masked)


# (n, up channels, couts, coarse extents): one whole tile / ragged tiles in y and x with two cout groups / two z-chunks / one plane pair with x tiles of 16 + 16 + 4
UPCONV_CASES = [(1, 32, 32, (3, 8, 16)), (2, 16, 64, (5, 9, 20)), (1, 32, 32, (25, 4, 4)), (1, 8, 32, (2, 17, 36)), (2, 32, 32, (48, 48, 48))]
@pytest.mark.parametrize("n,cup,cout,ldims", UPCONV_CASES)
def test_upcat_composite_transposed_convolution(n, cup, cout, ldims):
    """conv3(cat([x_e, deconv2(x)]))'s up half as one transposed convolution k4 s2 p1 of x added in place (kernels/upconv_h2.h) == the two-layer evaluation in float64,
    with and without the deconvolution's bias, statistics of the sum included"""
    kc.case_upconv_k4s2(DEV, n, cup, cout, ldims)
    kc.case_upconv_k4s2(DEV, 1, cup, cout, ldims, with_bias=False, fused_stats=False)


# (n, cin, cout, dims): 64 of 256 rows / the 6^3 level, two cout groups per workgroup, two chunks / odd extents (scalar stores), three chunks / a shape the z-marching kernel takes / the benchmark networks' 6^3 layers
SMALL_VOLUME_CASES = [(2, 16, 32, (4, 4, 4)), (2, 32, 64, (6, 6, 6)), (1, 48, 32, (3, 5, 7)), (2, 16, 64, (2, 8, 8)), (64, 256, 256, (6, 6, 6)), (8, 320, 320, (6, 6, 6)), (4, 768, 384, (6, 6, 6))]
@pytest.mark.parametrize("n,cin,cout,dims", SMALL_VOLUME_CASES)
def test_small_volume_convolution_on_matrix_cores(n, cin, cout, dims):
    """Conv3d k3 p1 with one sample's whole volume as the workgroup's tile (kernels/conv3d_vol_h2.h) == ATen in float64, statistics, selection and poisoning included"""
    kc.case_conv3d_k3_small_volume(DEV, n, cin, cout, dims)
    if n <= 2:
        kc.case_conv3d_k3_small_volume(DEV, 1, cin, cout, dims, with_bias=False, fused_stats=False)


# (n, cin, cout, dims): a volume that ends inside a wave / five k-steps (two weight chunks), 16 output channels / two groups of 32 / the decoder shapes
DECONV_H2_CASES = [(2, 16, 32, (3, 5, 7)), (2, 80, 16, (2, 4, 9)), (1, 32, 64, (4, 4, 8)), (2, 64, 32, (48, 48, 48)), (2, 320, 256, (6, 6, 6))]
@pytest.mark.parametrize("n,cin,cout,dims", DECONV_H2_CASES)
def test_transposed_convolution_on_matrix_cores(n, cin, cout, dims):
    """ConvTranspose3d k2 s2 as one split-precision GEMM with (cout, parity) rows (kernels/deconv_h2.h) == ATen in float64, bounds and poisoning included"""
    kc.case_deconv_k2s2_h2(DEV, n, cin, cout, dims)
    kc.case_deconv_k2s2_h2(DEV, 1, cin, cout, dims, with_bias=False)


# (n, cin, cout, dims): one tile, vector stores / two cout groups per workgroup, odd output widths (scalar stores), ragged tiles / two z-chunks with a run-in plane /
# three channel chunks, 17 x 25 outputs in two tiles / DynUNet's layers at full size
S2_CASES = [(1, 16, 32, (4, 8, 8)), (2, 32, 64, (6, 10, 12)), (1, 16, 32, (36, 4, 6)), (1, 48, 96, (4, 34, 50)), (2, 32, 64, (96, 96, 96)), (2, 64, 128, (48, 48, 48)), (3, 256, 320, (12, 12, 12))]
@pytest.mark.parametrize("n,cin,cout,dims", S2_CASES)
def test_strided_convolution_on_matrix_cores(n, cin, cout, dims):
    """Conv3d k3 s2 p1 as 8 dense sub-convolutions over the input's parity phases on the fp16 matrix cores (kernels/conv3d_s2_h2.h) == ATen in float64, statistics included"""
    kc.case_conv3d_k3s2(DEV, n, cin, cout, dims)
    if n == 1:
        kc.case_conv3d_k3s2(DEV, 1, cin, cout, dims, with_bias=False, fused_stats=False)


def test_strided_convolution_bounds_and_poison():
    kc.case_conv3d_k3s2_poison_and_scale(DEV)


# (n, cin, cout, dims): resident slabs / streamed slabs with ragged 16 x 16 regions and two cout groups / the 8 x 32 region shape in two z-chunks / 8 x 32 regions, four chunks of channels
ACC_CASES = [(2, 32, 32, (4, 16, 16)), (1, 48, 64, (3, 18, 20)), (1, 16, 32, (24, 8, 24)), (1, 64, 32, (2, 24, 56)), (2, 32, 32, (96, 96, 96))]
@pytest.mark.parametrize("n,cin,cout,dims", ACC_CASES)
def test_conv3d_split_precision_accumulating(n, cin, cout, dims):
    """out += conv3x3x3(act(x)) + bias with the statistics of the sum (conv3d_h2.h, ACC): the second half of the UpCat path"""
    kc.case_conv3d_accumulate(DEV, n, cin, cout, dims)


# (n, cin, cout, dims): resident slabs / streamed slabs with ragged 16 x 16 regions and two cout groups / two z-chunks
POOL_CASES = [(2, 32, 32, (4, 16, 16)), (1, 48, 64, (6, 18, 36)), (1, 16, 32, (24, 16, 32)), (2, 32, 32, (96, 96, 96)), (2, 64, 64, (48, 48, 48))]
@pytest.mark.parametrize("n,cin,cout,dims", POOL_CASES)
def test_conv3d_split_precision_pooling_epilogue(n, cin, cout, dims):
    """MaxPool3d(2) out of the producing convolution's epilogue (conv3d_h2.h, POOL): raw maxima / minima, bitwise; the convolution itself untouched"""
    kc.case_conv3d_pool(DEV, n, cin, cout, dims)
