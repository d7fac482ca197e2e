"""CPU checks of the kernel LOGIC: the unmodified HIP sources compiled against the SIMT emulator
(tests/emu) and compared with torch-CPU restatements of the reference operators.  These do not replace the
-m gpu parity tests (tests/test_kernels_gpu.py runs the same cases on the MI355X); they exist because the
build container has no GPU and every index/LDS/barrier bug caught here saves a GPU round trip."""
import os

import pytest

import kernel_cases as kc


def test_window_extract(emu):
    kc.case_window_extract("cpu")
    kc.case_window_extract("cpu", img=(1, 13, 10, 11), roi=(5, 4, 3), overlap=0.25)  # unaligned -> scalar path


@pytest.mark.parametrize("mode", ["gaussian", "constant"])
def test_sw_blend_bitwise(emu, mode):
    kc.case_sw_blend("cpu", mode=mode)
    kc.case_sw_blend("cpu", img=(11, 13, 10), roi=(4, 5, 3), overlap=0.25, k=3, mode=mode)   # scalar path, ragged
    kc.case_sw_blend("cpu", img=(1, 20, 24), roi=(1, 8, 8), overlap=0.75, k=11, mode=mode)   # 2-D, K > 8, 4x overlap


@pytest.mark.parametrize("g", ["1", "2", "8"])
def test_sw_blend_batch_sizes(emu, g, monkeypatch):
    """every window-batch size of the regular-grid blend gives the same bits (K = 5 carries all of them)"""
    monkeypatch.setenv("MONAI_AMD_BLEND_G", g)
    monkeypatch.setenv("MONAI_AMD_BLEND_NT", "0" if g == "2" else "1")
    kc.case_sw_blend("cpu")
    kc.case_sw_blend("cpu", img=(20, 40, 52), roi=(8, 16, 16), overlap=0.5)       # clipped last window: 3 covering windows per axis


def test_sw_blend_special_cases(emu, monkeypatch):
    kc.case_sw_blend_special_values("cpu")
    kc.case_sw_blend_irregular("cpu")
    kc.case_sw_blend_many_windows("cpu")
    monkeypatch.setenv("MONAI_AMD_BLEND_LEGACY", "1")      # the table kernels on a regular grid: same bits
    kc.case_sw_blend("cpu")


def test_conv_direct(emu):
    assert kc.case_conv3d("cpu", 0, 2, 1, 32, (6, 7, 9), with_nrm=False, fused_stats=False) == 0
    kc.case_conv3d("cpu", 0, 1, 5, 7, (4, 5, 6), fused_stats=False)


@pytest.mark.parametrize(
    "cfg,cin,cout,dims",
    [
        (1, 8, 32, (4, 8, 32)),     # exact tiles of the 96^3-level configuration
        (1, 16, 32, (5, 6, 40)),    # ragged in every axis -> masked stores / masked statistics
        (2, 8, 32, (4, 8, 16)),
        (2, 8, 32, (6, 10, 24)),
        (3, 8, 64, (2, 8, 8)),
        (3, 16, 64, (3, 12, 12)),
        (4, 4, 128, (2, 4, 4)),
        (4, 8, 128, (3, 6, 6)),
        (5, 2, 256, (2, 4, 4)),
        (5, 4, 256, (3, 3, 3)),
        (6, 8, 32, (4, 8, 8)),
        (6, 8, 32, (5, 9, 7)),
        (7, 1, 32, (4, 8, 32)),     # first layer: Cin = 1 zero-padded to the 2-channel chunk
        (7, 3, 32, (5, 5, 33)),     # odd Cin, ragged
        (8, 8, 32, (4, 8, 32)),
        (8, 6, 32, (5, 6, 40)),     # Cin padded 6 -> 8
        (9, 8, 32, (6, 10, 24)),
        (1, 5, 32, (4, 4, 32)),     # Cin padded 5 -> 8 in the 8-channel configuration
        (10, 6, 32, (6, 10, 24)),
        (11, 8, 64, (3, 12, 12)),
        (12, 6, 64, (2, 8, 8)),
        (13, 4, 128, (3, 6, 6)),
        (14, 4, 32, (5, 9, 7)),
    ],
)
def test_conv_mfma_configs(emu, cfg, cin, cout, dims):
    kc.case_conv3d("cpu", cfg, 2, cin, cout, dims, fused_stats=True)


def test_conv_mfma_separate_stats_and_select(emu):
    cfg = kc.case_conv3d("cpu", None, 1, 8, 32, (4, 8, 32), fused_stats=False)
    assert cfg >= 1
    from monai_amd import ops

    # exact-fp32 family (the emulator fixture's default, MONAI_AMD_CONV_ALGO=fp32): the large planes on the in-plane Winograd configuration
    # (the highest fp32 id), the rest on direct implicit-GEMM tiles
    wino2d = ops.conv3d_k3_num_configs()
    c1 = ops.conv3d_k3_c1_config()           # one input channel: the packed-VALU kernel (exact fp32), a direct tile when W % 4 != 0
    assert ops.conv3d_k3_select(1, 32, 96, 96, 96) == c1 and ops.conv3d_k3_select(1, 32, 95, 95, 95) == 7
    assert ops.conv3d_k3_select(32, 32, 96, 96, 96) == wino2d
    assert ops.conv3d_k3_select(64, 32, 96, 96, 96) == wino2d
    assert ops.conv3d_k3_select(32, 32, 48, 48, 48) == wino2d
    assert ops.conv3d_k3_select(32, 64, 24, 24, 24) == 12
    assert ops.conv3d_k3_select(64, 128, 12, 12, 12) == 13
    assert ops.conv3d_k3_select(128, 256, 6, 6, 6) == 13
    # the arithmetic family is an ARGUMENT of the C entry point (the library reads no environment).  MH_ALGO_AUTO: fp16 two-piece split
    # precision wherever the shape fits AND the input carries magnitude bounds, the exact-fp32 kernels everywhere else
    AUTO, DIRECT, WINO2D, H2, FP32 = 0, 1, 2, 3, 4
    h2 = ops.conv3d_k3_h2_config()
    shapes = ((32, 32, 96, 96, 96), (64, 32, 96, 96, 96), (32, 64, 48, 48, 48), (64, 128, 12, 12, 12))
    assert ops.conv3d_k3_select(1, 32, 96, 96, 96, algo=AUTO) == c1
    h2w = ops.conv3d_k3_h2w_config()      # round 6: 32 input channels on planes of whole 4 x 16 regions go behind the in-plane Winograd transform
    assert [ops.conv3d_k3_select(*a, bounded=True, algo=AUTO) for a in shapes] == [h2w, h2, h2w, h2]
    assert [ops.conv3d_k3_select(*a, bounded=True, algo=H2) for a in shapes] == [h2] * 4
    for a in shapes:       # no bounds, no split precision -- under AUTO and even when the family is asked for by name
        assert ops.conv3d_k3_select(*a, bounded=False, algo=AUTO) == ops.conv3d_k3_select(*a, bounded=True, algo=FP32) <= wino2d
        assert 1 <= ops.conv3d_k3_select(*a, bounded=False, algo=H2) < wino2d
    assert ops.conv3d_k3_select(256, 128, 12, 12, 12, bounded=True, algo=AUTO) == h2 and not ops.conv3d_k3_accepts(h2, 272, 32)
    # W % 4 != 0 and a plane below 8 x 8: not the z-marching split kernel -- since round 5 the small-volume one (the whole 6^3 volume as a tile), the fp32 tiles without bounds
    assert ops.conv3d_k3_select(128, 256, 6, 6, 6, bounded=True, algo=AUTO) == ops.conv3d_k3_h2v_config() == ops.conv3d_k3_select(128, 256, 6, 6, 6, bounded=True, algo=H2)
    assert ops.conv3d_k3_select(128, 256, 6, 6, 6, bounded=False, algo=AUTO) == 13 == ops.conv3d_k3_select(128, 256, 6, 6, 6, bounded=True, algo=FP32)
    # the split-precision kernel addresses its 32 output planes with 31-bit byte offsets: beyond D*H*W = 2^24 voxels the selector returns the fp32 kernels
    assert ops.conv3d_k3_select(32, 32, 255, 256, 256, bounded=True, algo=AUTO) == h2w and ops.conv3d_k3_select(32, 32, 255, 256, 256, bounded=True, algo=H2) == h2
    assert 1 <= ops.conv3d_k3_select(32, 32, 256, 256, 256, bounded=True, algo=AUTO) <= wino2d
    assert 1 <= ops.conv3d_k3_select(32, 32, 64, 512, 512, bounded=True, algo=H2) <= wino2d
    # region shapes of the split-precision kernel (16 x 16 | 8 x 32, whichever covers the plane with fewer): the statistics record count follows
    assert ops.conv3d_k3_stat_tiles(h2, 24, 24, 24) == 3 * 2 and ops.conv3d_k3_stat_tiles(h2, 96, 96, 96) == 36 and ops.conv3d_k3_stat_tiles(h2, 48, 48, 48) == 9 * 2
    assert 1 <= ops.conv3d_k3_select(32, 32, 96, 96, 96, bounded=True, algo=DIRECT) < wino2d
    assert ops.conv3d_k3_select(32, 32, 24, 24, 24, bounded=True, algo=WINO2D) == wino2d
    with pytest.raises(RuntimeError):
        ops.conv3d_k3_select(32, 32, 24, 24, 24, algo=9)
    # host-side switch: monai_amd.config.CONV_ALGO / the MONAI_AMD_CONV_ALGO environment variable, read by the PYTHON side at call time
    from monai_amd import config

    saved = config.CONV_ALGO
    try:
        config.CONV_ALGO = "auto"
        assert ops.conv3d_k3_select(32, 32, 96, 96, 96, bounded=True) == h2w and ops.conv3d_k3_select(32, 32, 96, 96, 96) == wino2d
        config.CONV_ALGO = "h2"          # the direct split-precision kernel by name
        assert ops.conv3d_k3_select(32, 32, 96, 96, 96, bounded=True) == h2
        config.CONV_ALGO = "fp32"
        assert ops.conv3d_k3_select(32, 32, 96, 96, 96, bounded=True) == wino2d
        config.CONV_ALGO = "no-such-family"
        with pytest.raises(ValueError):
            ops.conv3d_k3_select(32, 32, 96, 96, 96)
    finally:
        config.CONV_ALGO = saved


def test_conv_into_channel_slice(emu):
    kc.case_conv3d_into_channel_slice("cpu")


def test_pool_deconv_1x1_stats(emu):
    kc.case_maxpool("cpu")
    kc.case_maxpool("cpu", dims=(4, 6, 7))  # odd W -> scalar path, floor
    kc.case_deconv("cpu")
    kc.case_deconv("cpu", n=1, cin=12, cout=16, dims=(3, 4, 6))     # full 8-channel groups, ragged input-channel batch
    kc.case_conv1x1("cpu")
    kc.case_conv1x1_stats("cpu")
    kc.case_conv1x1("cpu", n=1, cin=13, cout=27, dims=(3, 5, 7))     # 16 + 8 + 3 output channels, ragged channel batch, scalar path
    kc.case_conv1x1("cpu", n=1, cin=6, cout=16, dims=(2, 4, 8))
    kc.case_conv1x1("cpu", cin=7, cout=11, dims=(3, 5, 7))
    kc.case_instnorm_stats("cpu")


def test_attention_and_add_act(emu):
    kc.case_attention("cpu", b=1, s=8, heads=2)       # one key tile, mostly masked
    kc.case_attention("cpu", b=2, s=45, heads=1)      # two tiles, ragged
    kc.case_attention("cpu", b=1, s=216, heads=1)     # ViT-B/16 on 96^3
    kc.case_attention("cpu", b=1, s=300, heads=2)     # beyond the old 224-token limit: three query blocks of one workgroup each, ten key tiles
    kc.case_attention("cpu", b=1, s=70, heads=1, hd=32)
    kc.case_attention("cpu", b=1, s=40, heads=1, hd=96)
    kc.case_attention("cpu", b=1, s=33, heads=1, hd=128)
    kc.case_add_act("cpu")


def test_conv_mfma_cout_padding(emu):
    kc.case_conv3d("cpu", 7, 2, 1, 16, (4, 8, 32))    # UNETR feature_size 16: Cout padded 16 -> 32
    kc.case_conv3d("cpu", 10, 1, 32, 16, (6, 10, 24))
    kc.case_conv3d("cpu", 13, 1, 8, 48, (3, 6, 6))    # 48 -> 128
    from monai_amd import ops

    assert ops.conv3d_k3_select(32, 16, 96, 96, 96) >= 1 and ops.conv3d_k3_select(32, 5, 96, 96, 96) >= 1


def test_strided_conv_and_deconv_k3(emu):
    kc.case_strided_conv_and_deconv_k3("cpu")


WINO2D_CASES = [(8, 16, (4, 16, 16), 2), (16, 32, (6, 8, 24), 1), (8, 16, (30, 4, 8), 1), (24, 16, (3, 18, 16), 1)]
@pytest.mark.parametrize("cin,cout,dims,n", WINO2D_CASES)
def test_conv3d_wino2d(emu, cin, cout, dims, n):
    """In-plane Winograd F(2x2, 3x3) + direct z taps, z-streaming: chunk halos, ragged regions, several cout groups."""
    from monai_amd import ops

    cfg = ops.conv3d_k3_num_configs()
    assert ops.conv3d_k3_accepts(cfg, cin, cout)
    kc.case_conv3d("cpu", cfg, n, cin, cout, dims, fused_stats=True)
    kc.case_conv3d("cpu", cfg, 1, cin, cout, dims, with_nrm=False, fused_stats=False)


# (6, 8, 24), (3, 18, 20), (2, 24, 56), (24, 8, 24) take the 8 x 32 region shape (one / three ragged / 3 x 2 regions / one region in two z-chunks), the others 16 x 16
H2_CASES = [(16, 32, (4, 16, 16), 2), (32, 48, (5, 16, 16), 2), (48, 80, (3, 8, 24), 1),      # 48 / 80 couts: a half-filled last cout group (round 4)
            (32, 64, (6, 8, 24), 1), (16, 32, (30, 4, 8), 1), (48, 32, (3, 18, 20), 1), (256, 32, (2, 8, 12), 1), (64, 32, (2, 24, 56), 1),
            (32, 32, (24, 8, 24), 1)]
# 16-couts groups (two z-taps per matrix instruction): one group / two / three groups, resident and streamed weight slabs, both region shapes, ragged regions, two z-chunks
H2C_CASES = [(16, 16, (4, 16, 16), 2), (32, 16, (5, 16, 16), 1), (48, 32, (3, 8, 24), 1), (16, 48, (6, 8, 24), 1), (32, 16, (3, 18, 20), 1), (64, 16, (2, 24, 56), 1),
             (16, 16, (24, 8, 24), 1), (16, 16, (30, 4, 8), 1)]
@pytest.mark.parametrize("cin,cout,dims,n", H2C_CASES)
def test_conv3d_split_precision_16_couts(emu, cin, cout, dims, n):
    """the direct split-precision kernel with output channel groups of 16 (conv3d_h2.h, C16): a completed plane is the sum of three partial planes held in different
    column halves -- the SAME tolerance as the 32-couts form and the fp32 tiles; what conv3d_k3_select returns for bounded 16-couts layers"""
    from monai_amd import ops

    cfg = ops.conv3d_k3_h2c_config()
    assert cfg > ops.conv3d_k3_num_configs() and ops.conv3d_k3_accepts(cfg, cin, cout) and not ops.conv3d_k3_accepts(cfg, 16, 24)
    kc.case_conv3d("cpu", cfg, n, cin, cout, dims, fused_stats=True)
    kc.case_conv3d("cpu", cfg, 1, cin, cout, dims, with_nrm=False, fused_stats=False)
    kc.case_conv3d_accumulate("cpu", n, cin, cout, dims, cfg=cfg)      # out += conv + bias, statistics of the sum (round 6: the Y set starts from the old values)
    if cout == 16 and dims[1] >= 8 and dims[2] >= 8:
        assert ops.conv3d_k3_select(cin, cout, *dims, bounded=True, algo=0) == cfg and ops.conv3d_k3_select(cin, cout, *dims, bounded=False, algo=0) != cfg and ops.conv3d_k3_select(cin, cout, *dims, bounded=True, algo=4) != cfg      # auto: yes; unbounded input or the exact-fp32 family: no

@pytest.mark.parametrize("cin,cout,dims,n", H2_CASES)
def test_conv3d_fp16_split_precision(emu, cin, cout, dims, n):
    """z-streaming direct convolution on the fp16 matrix cores, two fp16 pieces per operand and three exact piece products per
    multiply with fp32 accumulation (conv3d_h2.h) -- held to the SAME tolerance as the fp32 kernels: chunk halos, ragged regions,
    several cout groups and channel chunks, fused statistics."""
    from monai_amd import ops

    cfg = ops.conv3d_k3_h2_config()
    assert cfg > ops.conv3d_k3_num_configs() and ops.conv3d_k3_accepts(cfg, cin, cout)
    kc.case_conv3d("cpu", cfg, n, cin, cout, dims, fused_stats=True)
    kc.case_conv3d("cpu", cfg, 1, cin, cout, dims, with_nrm=False, fused_stats=False)


def test_sw_blend_mosaic_layout(emu):
    kc.case_sw_blend_mosaic("cpu")


def test_h2_input_scaling(emu):
    kc.case_h2_input_scaling("cpu")


def test_h2_nonfinite_and_missing_bounds(emu):
    kc.case_h2_nonfinite_and_missing_bounds("cpu")


def test_bound_producers(emu):
    kc.case_bound_producers("cpu")


LINEAR_CASES = [(128, 64, 16, False, False), (200, 144, 48, False, True), (66, 40, 36, True, False), (256, 192, 64, True, True)]
@pytest.mark.parametrize("m,n,k,gelu,res", LINEAR_CASES)
def test_linear_fp16_split_precision(emu, m, n, k, gelu, res):
    """nn.Linear (+ GELU, + residual): ragged rows / columns / K (multiples of 4), bias on and off"""
    kc.case_linear("cpu", m, n, k, gelu=gelu, residual=res)
    kc.case_linear("cpu", m, n, k, gelu=gelu, residual=res, bias=False)


@pytest.mark.parametrize("m,k", [(7, 48), (130, 768), (5, 100), (33, 96), (70, 24), (9, 192), (6, 256), (3, 50), (4, 260)])
def test_layernorm(emu, m, k):
    kc.case_layernorm("cpu", m, k)


def test_conv_one_input_channel(emu):
    """kernels/conv3d_c1.h: the first layer (Cin = 1) on packed fp32 vector arithmetic, statistics from the epilogue"""
    from monai_amd import ops

    cfg = ops.conv3d_k3_c1_config()
    assert cfg > ops.conv3d_k3_num_configs() and ops.conv3d_k3_accepts(cfg, 1, 32) and not ops.conv3d_k3_accepts(cfg, 2, 32)
    assert ops.conv3d_k3_select(1, 32, 96, 96, 96) == cfg and ops.conv3d_k3_select(1, 32, 9, 9, 9) != cfg       # W % 4
    assert 1 <= ops.conv3d_k3_select(1, 32, 96, 96, 96, algo=1) <= ops.conv3d_k3_num_configs()      # MH_ALGO_DIRECT: a matrix-core tile
    kc.case_conv3d("cpu", cfg, 2, 1, 32, (5, 40, 36), with_nrm=False, fused_stats=True)     # partial tiles in x and y, 16 couts per thread
    kc.case_conv3d("cpu", cfg, 1, 1, 24, (3, 8, 8), with_nrm=True, fused_stats=True)        # 8 couts per thread, deferred norm on the input
    kc.case_conv3d("cpu", cfg, 1, 1, 16, (50, 4, 8), with_nrm=False, fused_stats=True)      # two z-chunks (25 planes each)
    kc.case_conv3d("cpu", cfg, 1, 1, 16, (2, 33, 4), with_nrm=False, fused_stats=False)     # no statistics; a second tile row of one line


@pytest.mark.parametrize("s,hd", [(343, 16), (343, 32), (64, 16), (27, 8), (200, 8)])
def test_window_attention(emu, s, hd):
    """SwinUNETR's WindowAttention core: head dims 16 / 32 on the split-precision matrix-core kernel (round 4), 8 on the VALU kernel"""
    kc.case_window_attention("cpu", bw=2 if "cpu" == "cpu" else 6, s=s, heads=2, hd=hd)


@pytest.mark.parametrize("n,cin,cout,dims,scale", [(2, 96, 48, (6, 8, 44), 1.0), (1, 40, 80, (5, 4, 60), 1e4), (2, 24, 5, (3, 4, 8), 1e-6), (1, 32, 64, (2, 2, 257 * 4), 1.0), (1, 528, 33, (2, 3, 4), 1.0)])
def test_conv1x1_all_couts_from_one_read(emu, n, cin, cout, dims, scale):
    """the split-precision 1x1x1 convolution (kernels/conv1x1_h2.h): values, statistics, input magnitudes 1e-6 ... 1e4"""
    print(kc.case_conv1x1_h2("cpu", n, cin, cout, dims, scale))


@pytest.mark.parametrize("cin,cout,dims", [(16, 5, (6, 8, 44)), (48, 3, (2, 5, 12)), (7, 8, (3, 4, 8))])
def test_residual_join_inside_the_output_convolution(emu, cin, cout, dims):
    assert kc.case_conv1x1_sum2("cpu", 2, cin, cout, dims)


@pytest.mark.parametrize("m_src,m_out,k,n", [(150, 210, 48, 96), (64, 64, 192, 64), (37, 50, 384, 40)])
def test_layernorm_gather_linear_scatter(emu, m_src, m_out, k, n):
    """SwinTransformerBlock's copies folded into the gathering LayerNorm and the projection's scattering epilogue (round 5)"""
    assert kc.case_layernorm_gather_linear_scatter("cpu", m_src, m_out, k, n)


@pytest.mark.parametrize("ws,n,hd,masked", [((7, 7, 7), None, 16, True), ((7, 7, 7), 216, 32, True), pytest.param((7, 7, 7), None, 32, False, marks=pytest.mark.heavy_emu), ((3, 4, 5), None, 16, True)])
def test_window_attention_rel(emu, ws, n, hd, masked):
    """bias from the relative-position table + mask from region ids inside the kernel (round 5) == the S x S table form bit for bit; 216 of 343 = a clamped 6^3 window"""
    kc.case_window_attention_rel("cpu", bw=2, ws=ws, n=n, heads=2, hd=hd, masked=masked)


# (n, up channels, couts, coarse extents): one whole tile / ragged tiles in y and x with two cout groups / two z-chunks / one plane pair with x tiles of 16 + 16 + 4
UPCONV_CASES = [(1, 32, 32, (3, 8, 16)), pytest.param(2, 16, 64, (5, 9, 20), marks=pytest.mark.heavy_emu), (1, 32, 32, (13, 4, 4)), (1, 8, 32, (2, 17, 36))]
@pytest.mark.parametrize("n,cup,cout,ldims", UPCONV_CASES)
def test_upcat_composite_transposed_convolution(emu, n, cup, cout, ldims):
    """conv3(cat([x_e, deconv2(x)]))'s up half as one transposed convolution k4 s2 p1 of x added in place (kernels/upconv_h2.h) == the two-layer evaluation in float64,
    with and without the deconvolution's bias, statistics of the sum included"""
    kc.case_upconv_k4s2("cpu", n, cup, cout, ldims)
    kc.case_upconv_k4s2("cpu", 1, cup, cout, ldims, with_bias=False, fused_stats=False)


# (n, cin, cout, dims): 64 of 256 rows / the 6^3 level, two cout groups per workgroup, two chunks / odd extents (scalar stores), three chunks / a shape the z-marching kernel takes
SMALL_VOLUME_CASES = [(2, 16, 32, (4, 4, 4)), (2, 32, 64, (6, 6, 6)), (1, 48, 32, (3, 5, 7)), (2, 16, 64, (2, 8, 8)), (1, 768, 32, (4, 4, 4))]
@pytest.mark.parametrize("n,cin,cout,dims", SMALL_VOLUME_CASES)
def test_small_volume_convolution_on_matrix_cores(emu, n, cin, cout, dims):
    """Conv3d k3 p1 with one sample's whole volume as the workgroup's tile (kernels/conv3d_vol_h2.h) == ATen in float64, statistics, selection and poisoning included"""
    kc.case_conv3d_k3_small_volume("cpu", n, cin, cout, dims)
    if n <= 2:
        kc.case_conv3d_k3_small_volume("cpu", 1, cin, cout, dims, with_bias=False, fused_stats=False)


# (n, cin, cout, dims): a volume that ends inside a wave / five k-steps (two weight chunks), 16 output channels / two groups of 32
DECONV_H2_CASES = [(2, 16, 32, (3, 5, 7)), (2, 80, 16, (2, 4, 9)), (1, 32, 64, (4, 4, 8))]
@pytest.mark.parametrize("n,cin,cout,dims", DECONV_H2_CASES)
def test_transposed_convolution_on_matrix_cores(emu, n, cin, cout, dims):
    """ConvTranspose3d k2 s2 as one split-precision GEMM with (cout, parity) rows (kernels/deconv_h2.h) == ATen in float64, bounds and poisoning included"""
    kc.case_deconv_k2s2_h2("cpu", n, cin, cout, dims)
    kc.case_deconv_k2s2_h2("cpu", 1, cin, cout, dims, with_bias=False)


# (n, cin, cout, dims): one tile, vector stores / two cout groups per workgroup, odd output widths (scalar stores), ragged tiles / two z-chunks with a run-in plane /
# three channel chunks, 17 x 25 outputs in two tiles
S2_CASES = [(1, 16, 32, (4, 8, 8)), (2, 32, 64, (6, 10, 12)), (1, 16, 32, (36, 4, 6)), pytest.param(1, 48, 96, (4, 34, 50), marks=pytest.mark.heavy_emu)]
@pytest.mark.parametrize("n,cin,cout,dims", S2_CASES)
def test_strided_convolution_on_matrix_cores(emu, n, cin, cout, dims):
    """Conv3d k3 s2 p1 as 8 dense sub-convolutions over the input's parity phases on the fp16 matrix cores (kernels/conv3d_s2_h2.h) == ATen in float64, statistics included"""
    kc.case_conv3d_k3s2("cpu", n, cin, cout, dims)
    if n == 1:
        kc.case_conv3d_k3s2("cpu", 1, cin, cout, dims, with_bias=False, fused_stats=False)


def test_strided_convolution_bounds_and_poison(emu):
    kc.case_conv3d_k3s2_poison_and_scale("cpu")


# (n, cin, cout, dims): resident slabs / streamed slabs with ragged 16 x 16 regions and two cout groups / the 8 x 32 region shape in two z-chunks / 8 x 32 regions, four chunks of channels
ACC_CASES = [(2, 32, 32, (4, 16, 16)), (1, 48, 64, (3, 18, 20)), (1, 16, 32, (24, 8, 24)), pytest.param(1, 64, 32, (2, 24, 56), marks=pytest.mark.heavy_emu)]
@pytest.mark.parametrize("n,cin,cout,dims", ACC_CASES)
def test_conv3d_split_precision_accumulating(emu, n, cin, cout, dims):
    """out += conv3x3x3(act(x)) + bias with the statistics of the sum (conv3d_h2.h, ACC): the second half of the UpCat path"""
    kc.case_conv3d_accumulate("cpu", n, cin, cout, dims)


# (n, cin, cout, dims): resident slabs / streamed slabs with ragged 16 x 16 regions and two cout groups / two z-chunks
POOL_CASES = [(2, 32, 32, (4, 16, 16)), pytest.param(1, 48, 64, (6, 18, 36), marks=pytest.mark.heavy_emu), (1, 16, 32, (24, 16, 32))]
@pytest.mark.parametrize("n,cin,cout,dims", POOL_CASES)
def test_conv3d_split_precision_pooling_epilogue(emu, n, cin, cout, dims):
    """MaxPool3d(2) out of the producing convolution's epilogue (conv3d_h2.h, POOL): raw maxima / minima, bitwise; the convolution itself untouched"""
    kc.case_conv3d_pool("cpu", n, cin, cout, dims)


# in-plane Winograd in front of the split product (conv3d_wino_h2.h): one region; 2 x 2 regions with the volume's borders on every side; two z-chunks are the GPU twin's
@pytest.mark.parametrize("n,cin,cout,dims", [(1, 32, 32, (3, 4, 16)), (2, 32, 64, (5, 8, 32))])
def test_conv_h2w_winograd_split(emu, n, cin, cout, dims):
    from monai_amd import ops

    cfg = ops.conv3d_k3_h2w_config()
    assert cfg > ops.conv3d_k3_num_configs() and ops.conv3d_k3_accepts(cfg, cin, cout) and not ops.conv3d_k3_accepts(cfg, 16, 32) and not ops.conv3d_k3_accepts(cfg, 64, 32)
    assert ops.conv3d_k3_h2w_fits(*dims) and not ops.conv3d_k3_h2w_fits(4, 6, 16) and not ops.conv3d_k3_h2w_fits(4, 8, 24)
    kc.case_conv3d("cpu", cfg, n, cin, cout, dims, fused_stats=True)
    if n == 1:      # the accumulating and the pooling forms
        kc.case_conv3d_accumulate("cpu", 1, cin, cout, (3, 4, 32), cfg=cfg)
        kc.case_conv3d_pool("cpu", 1, cin, cout, (4, 8, 16), cfg=cfg)
