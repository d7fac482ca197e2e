"""-m "not gpu": the C-ABI library builds for gfx950, loads on a GPU-less host and exports every entry point that
include/monai_amd.h declares -- and the Python binding table (monai_amd/_lib.py:SIGNATURES) names exactly those."""
import ctypes
import os
import re

from monai_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "monai_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return sorted(set(re.findall(r"\b(mh_[a-z0-9_]+)\s*\(", text)))


def _ensure_built():
    if not os.path.isfile(_lib.LIB_PATH):      # a fresh checkout: the .so is git-ignored; hipcc cross-compiles without a GPU
        from monai_amd import build

        build.build()


def test_library_exports_every_declared_symbol():
    _ensure_built()
    names = _declared()
    assert len(names) >= 26 and "mh_sw_blend_f32" in names and "mh_conv3d_k3_f32" in names
    assert os.path.isfile(_lib.LIB_PATH), "build the extension first: python -m monai_amd.build (or __graft_entry__.build())"
    dll = ctypes.CDLL(_lib.LIB_PATH)          # loading needs no GPU; no compute entry point is called here
    missing = [n for n in names if not hasattr(dll, n)]
    assert not missing, missing
    dll.mh_version.restype = ctypes.c_int
    assert dll.mh_version() >= 100
    dll.mh_last_error.restype = ctypes.c_char_p
    assert isinstance(dll.mh_last_error(), bytes)


def test_binding_table_matches_the_header():
    assert sorted(_lib.SIGNATURES) == _declared()


def test_host_side_queries_need_no_gpu():
    _ensure_built()
    dll = ctypes.CDLL(_lib.LIB_PATH)
    dll.mh_conv3d_k3_select.restype = ctypes.c_int
    dll.mh_conv3d_k3_num_configs.restype = ctypes.c_int
    n = dll.mh_conv3d_k3_num_configs()
    dll.mh_conv3d_k3_c1_config.restype = ctypes.c_int
    AUTO, FP32 = 0, 4      # MH_ALGO_*: the arithmetic family is an argument; the library reads no environment variable
    assert dll.mh_conv3d_k3_select(AUTO, 0, 1, 32, 96, 96, 96) == dll.mh_conv3d_k3_c1_config() > n      # first layer: the one-input-channel kernel
    assert 1 <= dll.mh_conv3d_k3_select(AUTO, 0, 1, 32, 9, 9, 9) <= n                                    # W % 4 != 0: an fp32 matrix-core tile
    dll.mh_conv3d_k3_h2_config.restype = ctypes.c_int
    dll.mh_conv3d_k3_h2w_config.restype = ctypes.c_int
    dll.mh_conv3d_k3_h2w_fits.restype = ctypes.c_int
    h2w = dll.mh_conv3d_k3_h2w_config()
    assert h2w > n and h2w != dll.mh_conv3d_k3_h2_config()
    # bounded input: fp16 two-piece split precision -- 32 input channels on planes of whole 4 x 16 regions: behind the in-plane Winograd transform (round 6); otherwise direct
    assert dll.mh_conv3d_k3_select(AUTO, 1, 32, 32, 96, 96, 96) == h2w and dll.mh_conv3d_k3_select(AUTO, 1, 32, 64, 48, 48, 48) == h2w
    assert dll.mh_conv3d_k3_select(AUTO, 1, 64, 32, 96, 96, 96) == dll.mh_conv3d_k3_h2_config() and dll.mh_conv3d_k3_select(AUTO, 1, 32, 32, 24, 24, 24) == dll.mh_conv3d_k3_h2_config()
    assert dll.mh_conv3d_k3_h2w_fits(96, 96, 96) == 1 and dll.mh_conv3d_k3_h2w_fits(24, 24, 24) == 0 and dll.mh_conv3d_k3_h2w_fits(8, 4, 16) == 1
    assert dll.mh_conv3d_k3_accepts(h2w, 32, 96) == 1 and dll.mh_conv3d_k3_accepts(h2w, 16, 32) == 0 and dll.mh_conv3d_k3_accepts(h2w, 32, 48) == 0
    assert dll.mh_conv3d_k3_packed_floats(h2w, 32, 64) == 2 * 49152 + 4 and dll.mh_conv3d_k3_stat_tiles(h2w, 96, 96, 96) == 144 and dll.mh_conv3d_k3_pool_accepts(h2w, 32, 32, 96, 96, 96) == 1
    assert dll.mh_conv3d_k3_select(AUTO, 0, 32, 32, 96, 96, 96) == n      # no magnitude bounds: exact fp32 (large planes: the in-plane Winograd configuration)
    saved = os.environ.get("MONAI_AMD_CONV_ALGO")
    os.environ["MONAI_AMD_CONV_ALGO"] = "h2"                              # ... and nothing in the environment changes that
    try:
        assert dll.mh_conv3d_k3_select(AUTO, 0, 32, 32, 96, 96, 96) == n
        assert dll.mh_conv3d_k3_select(FP32, 1, 32, 32, 96, 96, 96) == n
        assert dll.mh_conv3d_k3_select(FP32, 1, 32, 32, 12, 12, 12) < n       # small planes: a direct fp32 tile
    finally:
        os.environ.pop("MONAI_AMD_CONV_ALGO", None)
        if saved is not None:
            os.environ["MONAI_AMD_CONV_ALGO"] = saved
    assert dll.mh_instnorm_stat_tiles(96, 96, 96) > 0
    # round 5: the small-volume split-precision configuration (bounded inputs of 64 .. 256-voxel volumes the z-marching kernel does not take), the stride-2 and the
    # transposed k2 s2 matrix-core kernels' shape rules and buffer sizes -- host arithmetic, no device
    dll.mh_conv3d_k3_h2v_config.restype = ctypes.c_int
    h2v = dll.mh_conv3d_k3_h2v_config()
    assert h2v > n and h2v != dll.mh_conv3d_k3_h2_config()
    assert dll.mh_conv3d_k3_select(AUTO, 1, 128, 256, 6, 6, 6) == h2v and dll.mh_conv3d_k3_select(AUTO, 1, 768, 384, 6, 6, 6) == h2v
    assert dll.mh_conv3d_k3_select(AUTO, 0, 128, 256, 6, 6, 6) < n and dll.mh_conv3d_k3_select(FP32, 1, 128, 256, 6, 6, 6) < n
    assert dll.mh_conv3d_k3_select(AUTO, 1, 128, 256, 3, 3, 3) < n and dll.mh_conv3d_k3_select(AUTO, 1, 784, 256, 6, 6, 6) < n        # 27 voxels / more than 768 channels
    assert dll.mh_conv3d_k3_stat_tiles(h2v, 6, 6, 6) == 1
    dll.mh_conv3d_k3_packed_floats.restype = ctypes.c_int64
    assert dll.mh_conv3d_k3_packed_floats(h2v, 128, 256) == 128 * 256 * 27 + 4
    assert dll.mh_conv3d_k3s2_accepts(32, 64, 96, 96, 96) == 1 and dll.mh_conv3d_k3s2_accepts(32, 64, 95, 96, 96) == 0 and dll.mh_conv3d_k3s2_accepts(24, 64, 96, 96, 96) == 0
    assert dll.mh_conv3d_k3s2_accepts(32, 48, 96, 96, 96) == 0 and dll.mh_conv3d_k3s2_accepts(32, 64, 512, 512, 512) == 0      # Cout % 32; 32-bit offsets
    dll.mh_conv3d_k3s2_workspace_floats.restype = ctypes.c_int64
    dll.mh_conv3d_k3s2_packed_floats.restype = ctypes.c_int64
    assert dll.mh_conv3d_k3s2_workspace_floats(3, 32, 8, 8, 8) == 3 * 32 * 512 + 4 and dll.mh_conv3d_k3s2_packed_floats(32, 64) == 32 * 64 * 27 + 4
    assert dll.mh_conv3d_k3s2_stat_tiles(96, 96, 96) == 36 and dll.mh_conv3d_k3s2_stat_tiles(24, 24, 24) == 1
    assert dll.mh_deconv_k2s2_h2_accepts(64, 32, 48, 48, 48) == 1 and dll.mh_deconv_k2s2_h2_accepts(64, 16, 4, 4, 4) == 1 and dll.mh_deconv_k2s2_h2_accepts(60, 32, 4, 4, 4) == 0
    dll.mh_deconv_k2s2_h2_packed_floats.restype = ctypes.c_int64
    assert dll.mh_deconv_k2s2_h2_packed_floats(64, 32) == 64 * 32 * 8 + 4


def test_argument_checks_of_the_round_6_entries_need_no_gpu():
    """mh_pixelshuffle_f32 refuses shapes that are not [N][C * fz * 4][D][H][W] -> [N][C][fz D][2 H][2 W] and mh_conv3d_k3_accumulate_f32 names its configurations (the
    split-precision ones incl. the 16-cout form) BEFORE anything is launched: MH_ERR_ARG / MH_ERR_UNSUPPORTED with a message, on a box without a GPU"""
    _ensure_built()
    dll = ctypes.CDLL(_lib.LIB_PATH)
    dll.mh_last_error.restype = ctypes.c_char_p

    def view(c, d, h, w):
        t = _lib.MhTensor5()
        t.data, t.n_stride, t.nrm, t.nrm_n_stride = 0x10000, c * d * h * w, None, 0
        t.N, t.C, t.D, t.H, t.W = 1, c, d, h, w
        return t

    a = view(16, 2, 2, 2)
    assert dll.mh_pixelshuffle_f32(ctypes.byref(a), ctypes.byref(view(2, 4, 4, 5)), 2, 1, None) == -1 and b"pixelshuffle" in dll.mh_last_error()
    assert dll.mh_pixelshuffle_f32(ctypes.byref(a), ctypes.byref(view(4, 4, 4, 4)), 2, 1, None) == -1          # 16 channels are 2 x 8, not 4 x 8
    assert dll.mh_pixelshuffle_f32(ctypes.byref(a), ctypes.byref(view(2, 4, 4, 4)), 3, 1, None) == -1 and b"fz" in dll.mh_last_error()
    assert dll.mh_pixelshuffle_f32(ctypes.byref(a), ctypes.byref(view(4, 2, 4, 4)), 2, 1, None) == -1          # the one-plane form's shape under fz = 2
    stats = (ctypes.c_float * 4)()
    x = view(16, 4, 8, 8)
    dll.mh_conv3d_k3_h2c_config.restype = ctypes.c_int
    assert dll.mh_conv3d_k3_accumulate_f32(1, ctypes.byref(x), stats, None, ctypes.byref(x), stats, None) == -3 and b"accumulating form" in dll.mh_last_error()      # an fp32 tile
    assert dll.mh_conv3d_k3_accumulate_f32(dll.mh_conv3d_k3_h2c_config(), ctypes.byref(x), stats, None, ctypes.byref(x), None, None) == -3                              # no statistics


def test_library_reads_no_environment():
    """"no global mutable state" -- the shipped library does not even import getenv (development knobs exist only in the
    -DMH_DEV_KNOBS build, libmonai_amd_dev.so, which the product never loads)"""
    import subprocess

    _ensure_built()
    syms = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in syms, [ln for ln in syms.splitlines() if "getenv" in ln]
