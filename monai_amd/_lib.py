"""ctypes binding of ``libmonai_amd.so`` (C ABI declared in ``include/monai_amd.h``).

The library is the HIP/gfx950 build produced by ``python -m monai_amd.build`` (or
``__graft_entry__.build()``) and is loaded from the package directory.  There is NO fallback: if the
shared object is missing or a tensor is not on a ROCm device the call raises ``RuntimeError`` -- the
same contract as the reference's ``monai._C`` ("Not compiled with GPU support",
monai/csrc/resample/pushpull.h:95).
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

from ._fallback import UnsupportedOnDevice

_HERE = os.path.dirname(os.path.abspath(__file__))
# MONAI_AMD_LIB: another build of the same C ABI (the measurement flavour with A/B knobs, `python -m monai_amd.build --dev`; tools/ only)
LIB_PATH = os.environ.get("MONAI_AMD_LIB") or os.path.join(_HERE, "csrc", "libmonai_amd.so")


class MhTensor5(C.Structure):
    """``mh_tensor5`` of include/monai_amd.h"""

    _fields_ = [
        ("data", C.c_void_p),
        ("n_stride", C.c_int64),
        ("nrm", C.c_void_p),
        ("nrm_n_stride", C.c_int64),
        ("N", C.c_int32),
        ("C", C.c_int32),
        ("D", C.c_int32),
        ("H", C.c_int32),
        ("W", C.c_int32),
    ]


_I, _L, _P, _F = C.c_int, C.c_int64, C.c_void_p, C.c_float
_T = C.POINTER(MhTensor5)
_IA = C.POINTER(C.c_int32)

# name -> (restype, argtypes); kept in one table so tests can check it against the header
SIGNATURES = {
    "mh_version": (_I, []),
    "mh_last_error": (C.c_char_p, []),
    "mh_window_extract_f32": (_I, [_P, _I, _I, _I, _I, _IA, _I, _IA, _I, _IA, _I, _I, _I, _I, _I, _I, _P, _P]),
    "mh_patch_accumulate_f32": (_I, [_P, _P, _P] + [_I] * 10 + [_P]),
    "mh_patch_accumulate_batch_f32": (_I, [_P, _P, _P, _I, _IA] + [_I] * 7 + [_P]),
    "mh_avg_finalize_f32": (_I, [_P, _P, C.c_int64, _P]),
    "mh_pointwise_f32": (_I, [_I, _P, _P, _L, _F, _P]),
    "mh_channel_reduce_f32": (_I, [_I, _P, _P, _I, _L, _P]),
    "mh_onehot_f32": (_I, [_P, _P, _I, _L, _P]),
    "mh_scale_intensity_range_f32": (_I, [_P, _P, _L, _F, _F, _I, _F, _F, _I, _F, _I, _F, _P]),
    "mh_foreground_bbox_workspace_ints": (_I, [_I, _I]),
    "mh_foreground_bbox_f32": (_I, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "mh_crop_pad_f32": (_I, [_P, _P] + [_I] * 10 + [_F, _P]),
    "mh_sw_blend_f32": (_I, [_P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _I, _IA, _I, _IA, _I, _IA, _I, _I, _P]),
    "mh_sw_blend_buffered_f32": (_I, [_P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _I, _IA, _I, _IA, _I, _IA, _I, _I, _I, _I, _P]),
    "mh_sw_mosaic_class_counts": (_I, [_I, _I, _IA]),
    "mh_sw_blend_mosaic_f32": (_I, [_P, C.POINTER(C.c_int64), _I, _I, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _IA, _I, _IA, _I, _IA, _I, _P]),
    "mh_sw_blend_argmax_f32": (_I, [_P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _IA, _I, _IA, _I, _IA, _I, _I, _P]),
    "mh_conv3d_k3_select": (_I, [_I, _I, _I, _I, _I, _I, _I]),
    "mh_conv3d_k3_h2_config": (_I, []),
    "mh_conv3d_k3_h2c_config": (_I, []),
    "mh_conv3d_k3_c1_config": (_I, []),
    "mh_conv3d_k3_h2v_config": (_I, []),
    "mh_conv3d_k3_h2w_config": (_I, []),
    "mh_conv3d_k3_h2w_fits": (_I, [_I, _I, _I]),
    "mh_conv3d_k3_num_configs": (_I, []),
    "mh_conv3d_k3_accepts": (_I, [_I, _I, _I]),
    "mh_conv3d_k3_packed_floats": (_L, [_I, _I, _I]),
    "mh_conv3d_k3_pack_f32": (_I, [_I, _P, _I, _I, _P, _P]),
    "mh_conv3d_k3_stat_tiles": (_I, [_I, _I, _I, _I]),
    "mh_conv3d_k3_f32": (_I, [_I, _T, _P, _P, _T, _P, _P]),
    "mh_conv3d_k3_accumulate_f32": (_I, [_I, _T, _P, _P, _T, _P, _P]),
    "mh_conv3d_k3_pool_accepts": (_I, [_I, _I, _I, _I, _I, _I]),
    "mh_conv3d_k3_pool_f32": (_I, [_I, _T, _P, _P, _T, _P, _P, _P, _L, _P]),
    "mh_pool_select_f32": (_I, [_P, _P, _P, _L, _I, _I, _L, _L, _P]),
    "mh_instnorm_stat_tiles": (_I, [_I, _I, _I]),
    "mh_instnorm_stats_f32": (_I, [_T, _P, _P]),
    "mh_instnorm_finalize_f32": (_I, [_P, _I, _I, _I, _P, _P, _F, _F, _P, _L, _P]),
    "mh_groupnorm_finalize_f32": (_I, [_P, _I, _I, _I, _I, _P, _P, _F, _F, _P, _L, _P]),
    "mh_flip_permute_f32": (_I, [_P, _P, _I, _IA, _IA, _IA, _P]),
    "mh_normalize_stats_workspace_doubles": (_L, [_I, _L]),
    "mh_normalize_stats_f32": (_I, [_P, _I, _L, _I, _P, _P, _P]),
    "mh_normalize_apply_f32": (_I, [_P, _P, _I, _L, _I, _P, _P]),
    "mh_minmax_workspace_floats": (_L, [_I, _L]),
    "mh_minmax_f32": (_I, [_P, _I, _L, _P, _P, _P]),
    "mh_minmax_scale_f32": (_I, [_P, _P, _I, _L, _P, _I, _F, _F, _I, _F, _P]),
    "mh_nrm_identity_f32": (_I, [_P, _I, _I, _L, _P]),
    "mh_maxpool2_f32": (_I, [_T, _T, _P]),
    "mh_deconv_k2s2_f32": (_I, [_T, _P, _P, _T, _P]),
    "mh_upconv_k4s2_accepts": (_I, [_I, _I, _I, _I, _I]),
    "mh_upconv_k4s2_packed_floats": (_L, [_I, _I]),
    "mh_upconv_k4s2_pack_f32": (_I, [_P, _I, _I, _P, _P]),
    "mh_upconv_k4s2_stat_tiles": (_I, [_I, _I, _I]),
    "mh_upconv_k4s2_f32": (_I, [_T, _P, _P, _T, _I, _P, _P]),
    "mh_deconv_k2s2_h2_accepts": (_I, [_I, _I, _I, _I, _I]),
    "mh_deconv_k2s2_h2_packed_floats": (_L, [_I, _I]),
    "mh_deconv_k2s2_h2_pack_f32": (_I, [_P, _I, _I, _P, _P]),
    "mh_deconv_k2s2_h2_f32": (_I, [_T, _P, _P, _T, _P]),
    "mh_conv3d_k3s2_accepts": (_I, [_I, _I, _I, _I, _I]),
    "mh_conv3d_k3s2_packed_floats": (_L, [_I, _I]),
    "mh_conv3d_k3s2_workspace_floats": (_L, [_I, _I, _I, _I, _I]),
    "mh_conv3d_k3s2_stat_tiles": (_I, [_I, _I, _I]),
    "mh_conv3d_k3s2_pack_f32": (_I, [_P, _I, _I, _P, _P]),
    "mh_conv3d_k3s2_f32": (_I, [_T, _P, _P, _T, _P, _P, _I, _P]),
    "mh_conv1x1_f32": (_I, [_T, _P, _P, _T, _P]),
    "mh_conv1x1_stat_tiles": (_I, [_I, _I, _I]),
    "mh_conv1x1_stats_f32": (_I, [_T, _P, _P, _T, _P, _P]),
    "mh_conv1x1_h2_accepts": (_I, [_I, _I, _I, _I, _I]),
    "mh_conv1x1_h2_packed_floats": (_L, [_I, _I]),
    "mh_conv1x1_h2_pack_f32": (_I, [_P, _I, _I, _P, _P]),
    "mh_conv1x1_h2_f32": (_I, [_T, _P, _P, _T, _P, _P]),
    "mh_conv1x1_sum2_accepts": (_I, [_I, _I, _I, _I]),
    "mh_conv1x1_sum2_f32": (_I, [_T, _T, _F, _P, _P, _T, _P]),
    "mh_conv1x1_windows_f32": (_I, [_T, _P, _P, _P, _I, C.POINTER(C.c_int64), _P]),
    "mh_conv3d_k3_strided_f32": (_I, [_T, _P, _P, _T, _I, _P]),
    "mh_conv3d_k3_strided3_f32": (_I, [_T, _P, _P, _T, _I, _I, _I, _P]),
    "mh_deconv_ks_f32": (_I, [_T, _P, _P, _T, _I, _I, _I, _P]),
    "mh_deconv_k3_f32": (_I, [_T, _P, _P, _T, _I, _P]),
    "mh_add_act_f32": (_I, [_T, _T, _F, _T, _P]),
    "mh_pad_replicate_f32": (_I, [_T, _T, _P]),
    "mh_pixelshuffle_f32": (_I, [_T, _T, _I, _I, _P]),
    "mh_attention_f32": (_I, [_P, _P, _I, _I, _I, _I, _F, _P]),
    "mh_window_attention_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P]),
    "mh_window_attention_rel_accepts": (_I, [_I, _I, _I]),
    "mh_window_attention_rel_f32": (_I, [_P, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "mh_linear_packed_floats": (_L, [_I, _I]),
    "mh_linear_pack_f32": (_I, [_P, _I, _I, _P, _P]),
    "mh_linear_f32": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _I, _P]),
    "mh_linear_tile_f32": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P]),
    "mh_layernorm_f32": (_I, [_P, _P, _P, _F, _P, _L, _I, _P]),
    "mh_layernorm_gather_accepts": (_I, [_I]),
    "mh_layernorm_gather_f32": (_I, [_P, _P, _P, _F, _P, _L, _I, _P, _P]),
    "mh_linear_scatter_f32": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _I, _P, _P]),
    "mh_affine_resample_workspace_bytes": (_L, [_I, _I, _I]),
    "mh_affine_resample_f32": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _I, C.POINTER(C.c_double), _I, _I, _I, _I, _P, _P]),
    "mh_separable_filter3d_f32": (_I, [_P, _P, _I, _I, _I, _I, C.POINTER(C.c_float), _I, C.POINTER(C.c_float), _I, C.POINTER(C.c_float), _I, _P]),
    "mh_grid_pull": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _IA, _IA, _I, _P]),
    "mh_pushpull": (_I, [_P, _P, _P, _P, _P] + [_I] * 10 + [_IA, _IA] + [_I] * 7 + [_P]),
    "mh_grid_resample_f32": (_I, [_P, _I, _I, _I, _I, _P, _I, C.POINTER(C.c_double), C.POINTER(C.c_double), _P, _I, _I, _I, _I, _I, _I, _I, _P]),
}


class KernelRejected(RuntimeError):
    """An entry point declined a VALID call it has no kernel for (MH_ERR_UNSUPPORTED: a size / shape rule) without launching: callers with an alternative
    path catch THIS, not RuntimeError.  Bad arguments (MH_ERR_ARG: null pointers, mismatched shapes, misalignment -- caller bugs), launch failures and HIP
    errors stay plain RuntimeErrors and must surface."""


class Library:
    """A loaded ``libmonai_amd`` with typed entry points and error translation."""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise RuntimeError(
                f"monai_amd: HIP extension not built ({path} missing). Run `python -m monai_amd.build` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback."
            )
        self.path = path
        self._dll = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self._dll, name)  # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
            setattr(self, "_" + name, fn)

    def last_error(self) -> str:
        return self._mh_last_error().decode("utf-8", "replace")

    def check(self, rc: int) -> None:
        if rc == -3:            # MH_ERR_UNSUPPORTED: a valid call the entry point has no kernel for; nothing was launched
            raise KernelRejected(f"monai_amd: {self.last_error()} (code {rc})")
        if rc != 0:             # MH_ERR_ARG (a caller bug), MH_ERR_LAUNCH and anything else: a real failure, never something to fall back from
            raise RuntimeError(f"monai_amd: {self.last_error()} (code {rc})")

    def call(self, name: str, *args) -> None:
        self.check(getattr(self, "_" + name)(*args))

    def query(self, name: str, *args) -> int:
        rc = getattr(self, "_" + name)(*args)
        if rc < 0:
            raise RuntimeError(f"monai_amd: {self.last_error()} (code {rc})")
        return int(rc)


_LIB: Optional[Library] = None


def lib() -> Library:
    """The process-wide library handle (loaded on first use; raises if the extension is missing)."""
    global _LIB
    if _LIB is None:
        _LIB = Library(LIB_PATH)
    return _LIB


def require_device(*tensors: torch.Tensor, dtypes=(torch.float32,)) -> None:
    """Every tensor handed to a kernel must live on a ROCm device, be fp32 and be contiguous enough for
    the view it is used as.  CPU tensors are an error, never a silent fallback."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise UnsupportedOnDevice(
                "monai_amd: this path runs only on an MI355X (ROCm) device tensor; got a CPU tensor. "
                "There is no CPU fallback in the product path."
            )
        if t.dtype not in dtypes:
            raise UnsupportedOnDevice(f"monai_amd: dtype {t.dtype} is not accepted on this path (expected one of {dtypes})")


def stream_ptr(t: torch.Tensor) -> C.c_void_p:
    """Raw hipStream_t of torch's current stream on t's device (0 = default stream for host stand-ins)."""
    if t.is_cuda:
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return C.c_void_p(0)


def ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def int_array(values: Sequence[int]):
    return (C.c_int32 * len(values))(*[int(v) for v in values])


def tensor5(t: torch.Tensor, nrm: Optional[torch.Tensor] = None) -> MhTensor5:
    """View descriptor of a 5-D NCDHW fp32 tensor whose (C, D, H, W) block is dense; the batch stride is
    free, so channel slices of a wider buffer (``buf[:, 32:64]``) are valid views.  ``nrm`` is the matching
    slice of a ``[N, Ctot, 4]`` parameter tensor."""
    if t.dim() != 5:
        raise RuntimeError(f"monai_amd: expected a 5-D NCDHW tensor, got shape {tuple(t.shape)}")
    n, c, d, h, w = t.shape
    st = t.stride()
    if st[4] != 1 or st[3] != w or st[2] != h * w or st[1] != d * h * w:
        raise RuntimeError(f"monai_amd: tensor view is not dense in (C,D,H,W): strides {st}")
    m = MhTensor5()
    m.data, m.n_stride = t.data_ptr(), (st[0] if n > 1 else max(st[0], c * d * h * w))
    m.N, m.C, m.D, m.H, m.W = n, c, d, h, w
    if nrm is not None:
        if nrm.dim() != 3 or nrm.shape[0] != n or nrm.shape[1] != c or nrm.shape[2] != 4 or nrm.stride(2) != 1 or nrm.stride(1) != 4:
            raise RuntimeError(f"monai_amd: nrm must be a [N,C,4] slice, got {tuple(nrm.shape)} strides {nrm.stride()}")
        m.nrm, m.nrm_n_stride = nrm.data_ptr(), (nrm.stride(0) if n > 1 else max(nrm.stride(0), 4 * c))
    else:
        m.nrm, m.nrm_n_stride = None, 0
    return m
