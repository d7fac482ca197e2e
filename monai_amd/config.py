"""Build flags mirroring monai/config/deviceconfig.py:29-34.

``HAS_EXT``: the native module is available (here: the gfx950 library behind ``monai_amd._C``).  ``USE_COMPILED``: route
resampling transforms through the native ``grid_pull`` instead of ``F.grid_sample`` -- in the reference
``HAS_EXT and os.getenv("BUILD_MONAI", "0") == "1"``; the same rule here.  Transforms read the attribute at call time, so
it can be switched at run time (``monai_amd.config.USE_COMPILED = True``)."""
import os

HAS_EXT = True
USE_COMPILED = HAS_EXT and os.getenv("BUILD_MONAI", "0") == "1"


# ---- arithmetic family of the 3x3x3 convolutions (include/monai_amd.h: MH_ALGO_*) ----------------------------------------------
# The C library takes the family as an ARGUMENT of mh_conv3d_k3_select and keeps no state; this is the host-side switch.
# `CONV_ALGO` (or the environment variable MONAI_AMD_CONV_ALGO, read at call time while CONV_ALGO is None):
#   "auto"   fp16 two-piece split precision (fp32-equivalent) for inputs that carry magnitude bounds, exact fp32 otherwise
#   "fp32"   exact-fp32 kernels only (matrix-core tiles, in-plane Winograd, the one-channel kernel)
#   "direct" / "wino2d" / "h2"   pin one family (measurements): "h2" = the direct split-precision kernel only
CONV_ALGOS = {"auto": 0, "direct": 1, "wino2d": 2, "h2": 3, "fp32": 4}
CONV_ALGO = None


def conv_algo() -> int:
    name = CONV_ALGO if CONV_ALGO is not None else os.environ.get("MONAI_AMD_CONV_ALGO", "auto")
    try:
        return CONV_ALGOS[str(name).lower()]
    except KeyError:
        raise ValueError(f"monai_amd: unknown convolution family {name!r} (one of {sorted(CONV_ALGOS)})") from None


class conv_algo_scope:
    """`with config.conv_algo_scope("fp32"): ...` -- the family for the calls inside the block, the previous setting restored on exit (also when the body raises)."""

    def __init__(self, name: str):
        if str(name).lower() not in CONV_ALGOS:
            raise ValueError(f"monai_amd: unknown convolution family {name!r} (one of {sorted(CONV_ALGOS)})")
        self.name = str(name).lower()

    def __enter__(self):
        global CONV_ALGO
        self._saved, CONV_ALGO = CONV_ALGO, self.name
        return self

    def __exit__(self, *exc):
        global CONV_ALGO
        CONV_ALGO = self._saved
        return False


# ---- UpCat without its up-sampled intermediate ---------------------------------------------------------------------------------
# BasicUNet's decoder levels whose shapes the composite kernel takes (csrc/kernels/upconv_h2.h) evaluate conv3(cat([x_e, deconv2(x)])) as conv3[:, skip](x_e) + convT4(x):
# fp32-equivalent like the split-precision convolution, 1e-6 from the two-layer evaluation.  False (or MONAI_AMD_UPCAT_FUSED=0 while None) keeps the two layers.
UPCAT_FUSED = None


def upcat_fused() -> bool:
    v = UPCAT_FUSED if UPCAT_FUSED is not None else os.environ.get("MONAI_AMD_UPCAT_FUSED", "1")
    return str(v).lower() not in ("0", "false", "off", "no")


# Order of the two halves on the split-precision families: "conv_first" = the skip half is written by the convolution (plain form, no statistics), the composite term adds itself
# in place and leaves the statistics of the sum; "term_first" = the composite term is written, the convolution's accumulating form adds the skip half.  Same two addends either way
# (the sum of two floats does not depend on their order: identical raw tensors); the statistics come from the other kernel's tiles (1e-7 relative).  Since the composite kernel
# carries its epilogue inside the matrix phase (round 6) conv_first is the faster one at the headline's top level: profiles/r06_upcat_order_ab.txt.
UPCAT_ORDER = None


def upcat_order() -> str:
    v = str(UPCAT_ORDER if UPCAT_ORDER is not None else os.environ.get("MONAI_AMD_UPCAT_ORDER", "conv_first")).lower()
    if v not in ("conv_first", "term_first"):
        raise ValueError(f"monai_amd: unknown UpCat order {v!r} (conv_first or term_first)")
    return v


# ---- UpCat's convolution over the concatenation as two 32-channel launches ----------------------------------------------------------
# Decoder levels the composite kernel does not take (64 input channels at the 48^3 level): conv(cat([x_e, x_0])) = conv[:, :32](x_e) + conv[:, 32:](x_0) on the Winograd
# split-precision kernel (plain form, then accumulating form with the statistics of the sum) where the 64-channel whole would run on the direct kernel.
# False (or MONAI_AMD_CONV_HALVES=0 while None) keeps the one launch.
CONV_HALVES = None


def conv_halves() -> bool:
    v = CONV_HALVES if CONV_HALVES is not None else os.environ.get("MONAI_AMD_CONV_HALVES", "1")
    return str(v).lower() not in ("0", "false", "off", "no")


# ---- residual joins inside the producing convolution ----------------------------------------------------------------------------------
# SegResNet's ResBlock x + conv2(...): where conv2 runs on a split-precision configuration its accumulating form adds itself to x in place and leaves the statistics of the sum
# (no pass for the addition, none for the next block's norm statistics).  False (or MONAI_AMD_RESIDUAL_ACC=0 while None) keeps the addition pass.
RESIDUAL_ACC = None


def residual_accumulate() -> bool:
    v = RESIDUAL_ACC if RESIDUAL_ACC is not None else os.environ.get("MONAI_AMD_RESIDUAL_ACC", "1")
    return str(v).lower() not in ("0", "false", "off", "no")


# ---- MaxPool3d(2) inside the producing convolution ------------------------------------------------------------------------------
# BasicUNet's encoder: the split-precision convolution in front of a pooling leaves the pooled tensor itself (csrc/kernels/conv3d_h2.h, POOL) -- bit-identical logits.
# False (or MONAI_AMD_POOL_FUSED=0 while None) keeps the pooling pass.
POOL_FUSED = None


def pool_fused() -> bool:
    v = POOL_FUSED if POOL_FUSED is not None else os.environ.get("MONAI_AMD_POOL_FUSED", "1")
    return str(v).lower() not in ("0", "false", "off", "no")


# ---- stride-2 3x3x3 convolutions on the fp16 matrix cores --------------------------------------------------------------------------
# DynUNet / SegResNet / UNet down-sampling convolutions whose shapes the split-precision stride-2 kernel takes (csrc/kernels/conv3d_s2_h2.h) leave the vector-ALU
# kernel -- fp32-equivalent like the stride-1 split-precision convolution, only for inputs with magnitude bounds and only in the "auto" / "h2" families.
# False (or MONAI_AMD_STRIDED_H2=0 while None) keeps the direct fp32 kernel.
STRIDED_H2 = None


def strided_h2() -> bool:
    v = STRIDED_H2 if STRIDED_H2 is not None else os.environ.get("MONAI_AMD_STRIDED_H2", "1")
    return str(v).lower() not in ("0", "false", "off", "no") and conv_algo() in (CONV_ALGOS["auto"], CONV_ALGOS["h2"])


# the stride-2 kernel's two forms: "0" = phase-split pass + GEMM, "1" = conversion inside the GEMM's staging, "auto" (default) = by layer shape (ops.conv3d_k3s2_fused)
STRIDED_H2_FUSED = None


def strided_h2_fused() -> str:
    v = str(STRIDED_H2_FUSED if STRIDED_H2_FUSED is not None else os.environ.get("MONAI_AMD_STRIDED_H2_FUSED", "auto")).lower()
    return "0" if v in ("0", "false", "off", "no") else ("1" if v in ("1", "true", "on", "yes") else "auto")


# ---- 3x3x3 convolutions of small volumes on the fp16 matrix cores -------------------------------------------------------------------
# Levels whose whole volume has at most 256 voxels (the 6^3 level of a 96^3 window) run on the split-precision kernel with one sample's volume as the workgroup's tile
# (csrc/kernels/conv3d_vol_h2.h) instead of the exact-fp32 matrix tiles.  False (or MONAI_AMD_SMALL_VOLUME_H2=0 while None) keeps the fp32 tiles there.
SMALL_VOLUME_H2 = None


def small_volume_h2() -> bool:
    v = SMALL_VOLUME_H2 if SMALL_VOLUME_H2 is not None else os.environ.get("MONAI_AMD_SMALL_VOLUME_H2", "1")
    return str(v).lower() not in ("0", "false", "off", "no")


# ---- ConvTranspose3d k2 s2 on the fp16 matrix cores -------------------------------------------------------------------------------
# The up-sampling transposed convolutions (BasicUNet's lower decoder levels, DynUNet, UNETR) whose input carries magnitude bounds run as one split-precision GEMM with
# (cout, parity) rows (csrc/kernels/deconv_h2.h) instead of the vector-ALU kernel -- fp32-equivalent, families "auto" / "h2" only.
# False (or MONAI_AMD_DECONV_H2=0 while None) keeps the direct fp32 kernel.
DECONV_H2 = None


def deconv_h2() -> bool:
    v = DECONV_H2 if DECONV_H2 is not None else os.environ.get("MONAI_AMD_DECONV_H2", "1")
    return str(v).lower() not in ("0", "false", "off", "no") and conv_algo() in (CONV_ALGOS["auto"], CONV_ALGOS["h2"])


# ---- SwinTransformerBlock without its copies -------------------------------------------------------------------------------------
# SwinUNETR: norm1 + pad + roll + window_partition as one gathering LayerNorm and window_reverse + roll back + crop + the shortcut sum in the projection's epilogue
# (csrc/kernels/dense.h: layernorm_vec_kernel's src_row, the linear kernels' rowmap) -- the same values, moved once instead of six times.  False (or
# MONAI_AMD_SWIN_FUSED_MOVES=0 while None) keeps the separate passes.
SWIN_FUSED_MOVES = None


def swin_fused_moves() -> bool:
    v = SWIN_FUSED_MOVES if SWIN_FUSED_MOVES is not None else os.environ.get("MONAI_AMD_SWIN_FUSED_MOVES", "1")
    return str(v).lower() not in ("0", "false", "off", "no")


# Window attention with the relative position bias / shift mask evaluated inside the kernel (csrc/kernels/attention.h, REL) instead of read from S x S tables -- bit-identical;
# False (or MONAI_AMD_SWIN_REL_ATTENTION=0 while None) keeps the table form.
SWIN_REL_ATTENTION = None


def swin_rel_attention() -> bool:
    v = SWIN_REL_ATTENTION if SWIN_REL_ATTENTION is not None else os.environ.get("MONAI_AMD_SWIN_REL_ATTENTION", "1")
    return str(v).lower() not in ("0", "false", "off", "no")
