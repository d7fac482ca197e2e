"""UNETR on the MI355X kernels -- drop-in for ``monai.networks.nets.UNETR`` (monai/networks/nets/unetr.py:30-213).

Same constructor signature, module tree and ``state_dict`` keys/shapes as the reference (including the unused
``cross_attn`` / ``norm_cross_attn`` parameters every ``TransformerBlock`` carries, transformerblock.py:83-91) and the same
parameter-initialisation order, so reference checkpoints load unchanged and the same seed gives the same weights.

Inference engine:
  * ViT-B encoder: the self-attention core (QK^T, softmax, PV -- selfattention.py:189-212) is the hand-written fp32-MFMA kernel
    ``mh_attention_f32`` working straight on the qkv projection's output; the dense projections (patch embedding, qkv, out_proj,
    MLP) are plain library GEMMs (``F.linear`` -> hipBLASLt), LayerNorm / GELU / residual adds are torch element-wise ops;
  * conv decoder (UnetrBasicBlock / UnetrPrUpBlock / UnetrUpBlock with UnetResBlock): the same fp32-MFMA 3x3x3 conv with fused
    InstanceNorm statistics, deferred normalise+LeakyReLU(0.01) on load, transposed-conv and 1x1 kernels as BasicUNet, plus
    ``mh_add_act_f32`` for the residual join; concat buffers are written in place (no ``torch.cat``).
"""

from __future__ import annotations

import math
from collections.abc import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import _lib, _prof, ops
from ...utils.misc import ensure_tuple_rep

__all__ = ["UNETR"]


def _trunc_normal_(t: torch.Tensor, mean=0.0, std=1.0, a=-2.0, b=2.0) -> torch.Tensor:
    """inverse-CDF truncated normal (the initialiser the reference uses for the position embedding)"""
    cdf = lambda v: (1.0 + math.erf(v / math.sqrt(2.0))) / 2.0  # noqa: E731
    lo, hi = cdf((a - mean) / std), cdf((b - mean) / std)
    with torch.no_grad():
        t.uniform_(2 * lo - 1, 2 * hi - 1).erfinv_().mul_(std * math.sqrt(2.0)).add_(mean).clamp_(min=a, max=b)
    return t


# --------------------------------------------------------------------------- parameter containers (reference names)
class _PatchEmbedding(nn.Module):
    """PatchEmbeddingBlock (blocks/patchembedding.py:45-139): proj_type "conv" = Conv3d(k16, s16); "perceptron" = Sequential(Rearrange, Linear) over
    (p1 p2 p3 c)-ordered patches -- the Linear sits at ``patch_embeddings.1`` like the reference's, the parameter-free Rearrange is a placeholder"""

    def __init__(self, in_channels, hidden_size, n_patches, proj_type="conv"):
        super().__init__()
        if proj_type == "perceptron":
            self.patch_embeddings = nn.Sequential(nn.Identity(), nn.Linear(in_channels * 4096, hidden_size))
        else:
            self.patch_embeddings = nn.Conv3d(in_channels, hidden_size, kernel_size=16, stride=16)
        self.position_embeddings = nn.Parameter(torch.zeros(1, n_patches, hidden_size))
        _trunc_normal_(self.position_embeddings, mean=0.0, std=0.02, a=-2.0, b=2.0)
        if proj_type == "perceptron":      # patchembedding.py:115-121: every Linear of the block gets the truncated-normal initialisation
            _trunc_normal_(self.patch_embeddings[1].weight, mean=0.0, std=0.02, a=-2.0, b=2.0)
            nn.init.constant_(self.patch_embeddings[1].bias, 0)


class _MLP(nn.Module):
    def __init__(self, hidden, mlp_dim):
        super().__init__()
        self.linear1 = nn.Linear(hidden, mlp_dim)
        self.linear2 = nn.Linear(mlp_dim, hidden)


class _SA(nn.Module):
    def __init__(self, hidden, qkv_bias):
        super().__init__()
        self.out_proj = nn.Linear(hidden, hidden)
        self.qkv = nn.Linear(hidden, hidden * 3, bias=qkv_bias)


class _CrossAttn(nn.Module):   # instantiated by the reference's TransformerBlock, never used by ViT
    def __init__(self, hidden, qkv_bias):
        super().__init__()
        self.out_proj = nn.Linear(hidden, hidden)
        self.to_q = nn.Linear(hidden, hidden, bias=qkv_bias)
        self.to_k = nn.Linear(hidden, hidden, bias=qkv_bias)
        self.to_v = nn.Linear(hidden, hidden, bias=qkv_bias)


class _TransformerBlock(nn.Module):
    def __init__(self, hidden, mlp_dim, qkv_bias):
        super().__init__()
        self.mlp = _MLP(hidden, mlp_dim)
        self.norm1 = nn.LayerNorm(hidden)
        self.attn = _SA(hidden, qkv_bias)
        self.norm2 = nn.LayerNorm(hidden)
        self.norm_cross_attn = nn.LayerNorm(hidden)
        self.cross_attn = _CrossAttn(hidden, qkv_bias)


class _ViT(nn.Module):
    def __init__(self, in_channels, hidden, mlp_dim, num_layers, n_patches, qkv_bias, proj_type="conv"):
        super().__init__()
        self.patch_embedding = _PatchEmbedding(in_channels, hidden, n_patches, proj_type)
        self.blocks = nn.ModuleList([_TransformerBlock(hidden, mlp_dim, qkv_bias) for _ in range(num_layers)])
        self.norm = nn.LayerNorm(hidden)


class _Conv(nn.Module):
    """``Convolution(conv_only=True)``: the parameter lives at ``<name>.conv.weight``"""

    def __init__(self, cin, cout, k, transposed=False, bias=False):
        super().__init__()
        if transposed:
            self.conv = nn.ConvTranspose3d(cin, cout, kernel_size=2, stride=2, bias=bias)
        else:
            self.conv = nn.Conv3d(cin, cout, kernel_size=k, padding=k // 2, bias=bias)


class _ResBlock(nn.Module):
    """UnetResBlock (res=True) or UnetBasicBlock (res=False: the same two convolutions without the shortcut) -- dynunet_block.py:25-164"""

    def __init__(self, cin, cout, res=True):
        super().__init__()
        self.res = res
        self.conv1 = _Conv(cin, cout, 3)
        self.conv2 = _Conv(cout, cout, 3)
        if res and cin != cout:
            self.conv3 = _Conv(cin, cout, 1)


class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, res=True):
        super().__init__()
        self.layer = _ResBlock(cin, cout, res)


class _PrUpBlock(nn.Module):
    """UnetrPrUpBlock (unetr_block.py:106-209): conv_block=False keeps only the transposed convolutions (``blocks.i`` is the bare layer)"""

    def __init__(self, cin, cout, num_layer, conv_block=True, res=True):
        super().__init__()
        self.transp_conv_init = _Conv(cin, cout, 2, transposed=True)
        if conv_block:
            self.blocks = nn.ModuleList([nn.Sequential(_Conv(cout, cout, 2, transposed=True), _ResBlock(cout, cout, res)) for _ in range(num_layer)])
        else:
            self.blocks = nn.ModuleList([_Conv(cout, cout, 2, transposed=True) for _ in range(num_layer)])


class _UpBlock(nn.Module):
    def __init__(self, cin, cout, res=True):
        super().__init__()
        self.transp_conv = _Conv(cin, cout, 2, transposed=True)
        self.conv_block = _ResBlock(cout + cout, cout, res)


class _OutBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = _Conv(cin, cout, 1, bias=True)


# --------------------------------------------------------------------------- the module
class UNETR(nn.Module):
    def __init__(
        self,
        in_channels: int,
        out_channels: int,
        img_size: Sequence[int] | int,
        feature_size: int = 16,
        hidden_size: int = 768,
        mlp_dim: int = 3072,
        num_heads: int = 12,
        proj_type: str = "conv",
        norm_name: tuple | str = "instance",
        conv_block: bool = True,
        res_block: bool = True,
        dropout_rate: float = 0.0,
        spatial_dims: int = 3,
        qkv_bias: bool = False,
        save_attn: bool = False,
    ) -> None:
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise ValueError("dropout_rate should be between 0 and 1.")
        if hidden_size % num_heads != 0:
            raise ValueError("hidden_size should be divisible by num_heads.")
        if proj_type not in ("conv", "perceptron"):
            raise ValueError(f"proj_type should be one of ('conv', 'perceptron'), got {proj_type}.")
        if spatial_dims != 3 or save_attn:       # dropout_rate: inference-inert
            raise NotImplementedError("monai_amd.UNETR: 3-D networks without attention-matrix export are on the HIP path")
        norm = norm_name if isinstance(norm_name, str) else norm_name[0]
        if str(norm).lower() != "instance":
            raise NotImplementedError("monai_amd.UNETR: only norm_name='instance' is on the HIP path")
        if hidden_size // num_heads not in (32, 64, 96, 128):
            raise NotImplementedError("monai_amd.UNETR: the attention kernel is built for head dimensions 32, 64, 96 and 128")
        self.num_layers = 12
        img_size = ensure_tuple_rep(img_size, spatial_dims)
        self.img_size = tuple(int(v) for v in img_size)
        self.patch_size = (16,) * spatial_dims
        self.feat_size = tuple(d // 16 for d in self.img_size)
        n_patches = 1
        for f in self.feat_size:
            n_patches *= f
        self.hidden_size, self.num_heads, self.in_channels, self.out_channels = hidden_size, num_heads, in_channels, out_channels
        self.feature_size = fs = feature_size
        self.features = (2 * fs,)   # used by the inferer to size its window batch

        self.proj_type = proj_type
        self.vit = _ViT(in_channels, hidden_size, mlp_dim, self.num_layers, n_patches, qkv_bias, proj_type)
        self.encoder1 = _BasicBlock(in_channels, fs, res_block)
        self.encoder2 = _PrUpBlock(hidden_size, fs * 2, 2, conv_block, res_block)
        self.encoder3 = _PrUpBlock(hidden_size, fs * 4, 1, conv_block, res_block)
        self.encoder4 = _PrUpBlock(hidden_size, fs * 8, 0, conv_block, res_block)
        self.decoder5 = _UpBlock(hidden_size, fs * 8, res_block)
        self.decoder4 = _UpBlock(fs * 8, fs * 4, res_block)
        self.decoder3 = _UpBlock(fs * 4, fs * 2, res_block)
        self.decoder2 = _UpBlock(fs * 2, fs, res_block)
        self.out = _OutBlock(fs, out_channels)
        self._packed: dict = {}
        self._stats = None

    # ---- helpers -----------------------------------------------------------------------------------
    def _packed_weight(self, conv: nn.Conv3d, cfg: int, part=None) -> torch.Tensor:
        """the layer's weights packed for configuration `cfg` (cached; re-packed when the parameter changed); `part` = (lo, hi): output channels lo .. hi - 1 only"""
        w = conv.weight
        key = (w.data_ptr(), w._version, str(w.device), cfg, part)
        slot = id(conv) if part is None else (id(conv), part)
        hit = self._packed.get(slot)
        if hit is None or hit[0] != key:
            hit = (key, ops.conv3d_k3_pack(cfg, w if part is None else w[part[0]:part[1]]))
            self._packed[slot] = hit
        return hit[1]

    def _packed_cin(self, conv: nn.Conv3d, cfg: int, c0: int, c1: int) -> torch.Tensor:
        """packed weights of INPUT channels c0 .. c1 - 1 (a convolution evaluated in two halves of its input channels)"""
        w = conv.weight
        key = (w.data_ptr(), w._version, str(w.device), cfg)
        slot = (id(conv), "cin", c0, c1)
        hit = self._packed.get(slot)
        if hit is None or hit[0] != key:
            hit = (key, ops.conv3d_k3_pack(cfg, w[:, c0:c1].contiguous()))
            self._packed[slot] = hit
        return hit[1]

    def _stats_buf(self, floats: int, device) -> torch.Tensor:
        if self._stats is None or self._stats.numel() < floats or self._stats.device != device:
            self._stats = torch.empty(floats, dtype=torch.float32, device=device)
        return self._stats

    def _conv3_in(self, conv: nn.Conv3d, x, x_nrm, slope: float):
        """3x3x3 conv (no bias) + InstanceNorm(no affine) statistics -> (raw output, its {alpha, beta, slope} record)."""
        n, cin, d, h, w = x.shape
        cout = conv.weight.shape[0]
        # every record of this engine carries a magnitude bound (instnorm_finalize writes one; plain tensors come with `nrm_identity` records that their
        # producers -- add_act, the transposed convolutions -- folded max |value| into): what the split-precision kernel scales its input by
        cfg = ops.conv3d_k3_select(cin, cout, d, h, w, bounded=x_nrm is not None)
        out = torch.empty((n, cout, d, h, w), dtype=torch.float32, device=x.device)
        nrm = torch.empty((n, cout, 4), dtype=torch.float32, device=x.device)
        h2, h2c = ops.conv3d_k3_h2_config(), ops.conv3d_k3_h2c_config()
        if cfg == h2 and cout % 32 == 16 and ops.conv3d_k3_accepts(h2c, cin, 16):
            # 48, 80, ... couts on the split-precision kernel (SwinUNETR(48)'s full-resolution levels): the last 16 as a group of their own in the 16-couts form (6
            # matrix instructions per tap) instead of a half-filled group of 32 (9): two launches into channel slices, two statistics sets (same tile count)
            tiles = ops.conv3d_k3_stat_tiles(h2, d, h, w)
            stats = self._stats_buf(n * cout * tiles * 3, x.device)
            lo = 0
            for part_cfg, hi in ((h2, cout - 16), (h2c, cout)):
                st = stats[n * lo * tiles * 3:n * hi * tiles * 3]
                with _prof.span(f"conv3d_k3/cfg{part_cfg}", 2.0 * 27 * cin * (hi - lo) * d * h * w * n):
                    ops.conv3d_k3(part_cfg, x, x_nrm, self._packed_weight(conv, part_cfg, (lo, hi)), None, out[:, lo:hi], st)
                ops.instnorm_finalize(st, tiles, n, hi - lo, None, None, 1e-5, slope, nrm[:, lo:hi])
                lo = hi
            return out, nrm
        flops = 2.0 * 27 * cin * cout * d * h * w * n
        half = (cin // 32) * 16
        if (cfg not in (h2, h2c) and x_nrm is not None and cin > 256 and cin % 16 == 0 and cout % 32 == 0
                and ops.conv3d_k3_select(half, cout, d, h, w, bounded=True) == h2 and ops.conv3d_k3_select(cin - half, cout, d, h, w, bounded=True) == h2):
            # more input channels than the split-precision kernel keeps records for (SwinUNETR(48)'s 384-channel concat at 12^3): the convolution is linear in its input
            # channels -- one half written, the other half added onto it by the accumulating form, which leaves the statistics of the sum
            tiles = ops.conv3d_k3_stat_tiles(h2, d, h, w)
            stats = self._stats_buf(n * cout * tiles * 3, x.device)
            with _prof.span(f"conv3d_k3/cfg{h2}", flops):
                ops.conv3d_k3(h2, x[:, :half], x_nrm[:, :half], self._packed_cin(conv, h2, 0, half), None, out, None)
                ops.conv3d_k3(h2, x[:, half:], x_nrm[:, half:], self._packed_cin(conv, h2, half, cin), None, out, stats, accumulate=True)
            ops.instnorm_finalize(stats, tiles, n, cout, None, None, 1e-5, slope, nrm)
            return out, nrm
        tiles = ops.conv3d_k3_stat_tiles(cfg, d, h, w)
        if tiles:
            stats = self._stats_buf(n * cout * tiles * 3, x.device)
            with _prof.span(f"conv3d_k3/cfg{cfg}", flops):
                ops.conv3d_k3(cfg, x, x_nrm, self._packed_weight(conv, cfg), None, out, stats)
        else:
            ops.conv3d_k3(cfg, x, x_nrm, self._packed_weight(conv, cfg), None, out, None)
            tiles = ops.instnorm_stat_tiles(d, h, w)
            stats = self._stats_buf(n * cout * tiles * 3, x.device)
            ops.instnorm_stats(out, stats)
        ops.instnorm_finalize(stats, tiles, n, cout, None, None, 1e-5, slope, nrm)
        return out, nrm

    @staticmethod
    def _records(t: torch.Tensor) -> torch.Tensor:
        """fresh identity records for a plain tensor that is about to be written (its producers leave the magnitude bounds in them)"""
        return ops.nrm_identity(torch.empty((t.shape[0], t.shape[1], 4), dtype=torch.float32, device=t.device))

    def _res_block(self, blk: _ResBlock, x, x_nrm, out, out_nrm, head=None):
        """UnetResBlock (dynunet_block.py:96-111) of a plain (already activated) tensor `x` (+ its identity records, or None) into `out`.
        head = (output convolution, logits): the block's result feeds only that 1x1x1 convolution (UnetOutBlock) -- where the fused kernel takes the shape the join is
        evaluated inside it and `out` is never written (returns the logits); otherwise the join into `out`, then the convolution."""
        c1, n1 = self._conv3_in(blk.conv1.conv, x, x_nrm, 0.01)     # conv1 -> norm1 -> lrelu, applied on load by conv2
        if not getattr(blk, "res", True):       # UnetBasicBlock: lrelu(norm2(conv2(.))), no shortcut -- materialised into `out`
            c2, n2 = self._conv3_in(blk.conv2.conv, c1, n1, 0.01)
            out = torch.empty_like(c2) if out is None else out
            ops.add_act(c2, n2, None, None, 1.0, out, out_nrm)
            return self._head(head, out)
        c2, n2 = self._conv3_in(blk.conv2.conv, c1, n1, 1.0)        # conv2 -> norm2 (no activation before the add)
        if hasattr(blk, "conv3"):
            w3 = blk.conv3.conv.weight
            n, cout = x.shape[0], w3.shape[0]
            r = torch.empty((n, cout) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
            tiles = ops.conv1x1_stat_tiles(*x.shape[2:])          # norm3's statistics come out of the shortcut convolution itself
            stats = self._stats_buf(n * cout * tiles * 3, x.device)
            if x_nrm is not None and cout > 16 and ops.conv1x1_h2_wanted(x.shape[1], cout, *x.shape[2:]):
                # more than one group of 16 output channels: all of them from ONE read of x on the matrix cores (kernels/conv1x1_h2.h; x's records carry the bounds)
                key = (w3.data_ptr(), w3._version, str(w3.device))
                hit = self._packed.get(("1x1", id(blk.conv3.conv)))
                if hit is None or hit[0] != key:
                    hit = (key, ops.conv1x1_h2_pack(w3.view(cout, -1)))
                    self._packed[("1x1", id(blk.conv3.conv))] = hit
                ops.conv1x1_h2(x, x_nrm, hit[1], None, r, stats)
            else:
                ops.conv1x1(x, None, w3.view(cout, -1), None, r, stats)
            n3 = torch.empty((n, cout, 4), dtype=torch.float32, device=x.device)
            ops.instnorm_finalize(stats, tiles, n, cout, None, None, 1e-5, 1.0, n3)
            res, res_nrm = r, n3
        else:
            res, res_nrm = x, None
        if head is not None and ops.conv1x1_sum2_accepts(head[0].weight.shape[0], *c2.shape[2:]):
            oc, logits = head
            return ops.conv1x1_sum2(c2, n2, res, res_nrm, 0.01, oc.weight.view(oc.weight.shape[0], -1), oc.bias, logits)
        out = torch.empty_like(c2) if out is None else out          # (a caller with a head may leave the block's own output to this fallback)
        ops.add_act(c2, n2, res, res_nrm, 0.01, out, out_nrm)
        return self._head(head, out)

    @staticmethod
    def _head(head, t):
        if head is None:
            return t
        oc, logits = head
        return ops.conv1x1(t, None, oc.weight.view(oc.weight.shape[0], -1), oc.bias, logits)

    @staticmethod
    def _tconv(conv: nn.ConvTranspose3d, x, out, out_nrm=None, x_nrm=None):
        """x_nrm: identity records of the plain tensor x that carry its magnitude bounds (left by its producer, or `_bounded`): the transposed convolution then runs on
        the matrix cores in split precision (csrc/kernels/deconv_h2.h)"""
        return ops.deconv_k2s2(x, x_nrm, conv.weight, conv.bias, out, out_nrm, bounded=x_nrm is not None)

    @staticmethod
    def _bounded(t: torch.Tensor) -> torch.Tensor:
        """identity records {1, 0, 1, max |t| per (n, c)} of a plain tensor nobody left bounds for (the ViT's hidden states: 768 x 6^3 values per window -- two small passes)"""
        rec = torch.empty((t.shape[0], t.shape[1], 4), dtype=torch.float32, device=t.device)
        rec[:, :, 0] = 1.0
        rec[:, :, 1] = 0.0
        rec[:, :, 2] = 1.0
        rec[:, :, 3] = t.abs().amax(dim=(2, 3, 4)).clamp_min(1.17549435e-38)       # NaN / inf stay: a poisoned bound, the sample's output is NaN (conv3d_h2.h)
        return rec

    def _new(self, like, c, scale=1):
        n = like.shape[0]
        sp = tuple(int(v * scale) for v in like.shape[2:])
        return torch.empty((n, c) + sp, dtype=torch.float32, device=like.device)

    # ---- ViT ---------------------------------------------------------------------------------------
    def _lin(self, x, weight, bias, residual=None, gelu=False, scatter=None):
        """nn.Linear (+ GELU) (+ residual) as ONE launch of the split-precision GEMM kernel (csrc/kernels/dense.h); the packed weight
        is cached per parameter (re-packed when the parameter is updated in place or moved).  scatter: int32 row map -- row m of the result lands in (and takes its
        residual from) row scatter[m] of a tensor shaped like `residual` (ops.linear_scatter)."""
        key = (weight.data_ptr(), weight._version, str(weight.device))
        hit = self._packed.get(("lin", id(weight)))
        if hit is None or hit[0] != key:
            hit = (key, ops.linear_pack(weight.reshape(weight.shape[0], -1)))
            self._packed[("lin", id(weight))] = hit
        m = x.numel() // x.shape[-1]
        with _prof.span("linear", 2.0 * m * weight.shape[0] * x.shape[-1]):
            if scatter is not None:
                return ops.linear_scatter(x, hit[1], weight.shape[0], bias, residual, scatter, gelu=gelu)
            return ops.linear(x, hit[1], weight.shape[0], bias, residual, gelu=gelu)

    def _vit(self, x_in):
        """ViT (monai/networks/nets/vit.py:27-142): patch embedding, 12 x TransformerBlock (transformerblock.py:88-105: x + attn(norm1(x)),
        x + mlp(norm2(x))), final norm -- LayerNorm, the four linear maps of a block (bias / GELU / residual fused into their epilogues)
        and the attention are HIP kernels (no library GEMM; the patch gather in front is a strided copy)."""
        pe = self.vit.patch_embedding
        b = x_in.shape[0]
        fz, fy, fx = self.feat_size
        c = x_in.shape[1]
        if self.proj_type == "perceptron":      # Rearrange("b c (h p1) (w p2) (d p3) -> b (h w d) (p1 p2 p3 c)") + Linear
            patches = x_in.reshape(b, c, fz, 16, fy, 16, fx, 16).permute(0, 2, 4, 6, 3, 5, 7, 1).reshape(b, fz * fy * fx, c * 4096)
            proj = pe.patch_embeddings[1]
        else:                                   # Conv3d(k16, s16) = the same Linear over (c p1 p2 p3)-ordered patches
            patches = x_in.reshape(b, c, fz, 16, fy, 16, fx, 16).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(b, fz * fy * fx, c * 4096)
            proj = pe.patch_embeddings
        pos = pe.position_embeddings.expand(b, -1, -1).contiguous()
        t = self._lin(patches, proj.weight, proj.bias, residual=pos)
        hidden = []
        hdim = self.hidden_size // self.num_heads
        scale = hdim ** -0.5
        for blk in self.vit.blocks:
            qkv = self._lin(ops.layernorm(t, blk.norm1.weight, blk.norm1.bias, 1e-5), blk.attn.qkv.weight, blk.attn.qkv.bias)
            with _prof.span("attention", 4.0 * qkv.shape[1] ** 2 * hdim * self.num_heads * b):
                a = ops.attention(qkv, self.num_heads, scale, hdim)
            t = self._lin(a, blk.attn.out_proj.weight, blk.attn.out_proj.bias, residual=t)
            m = self._lin(ops.layernorm(t, blk.norm2.weight, blk.norm2.bias, 1e-5), blk.mlp.linear1.weight, blk.mlp.linear1.bias, gelu=True)
            t = self._lin(m, blk.mlp.linear2.weight, blk.mlp.linear2.bias, residual=t)
            hidden.append(t)
        return ops.layernorm(t, self.vit.norm.weight, self.vit.norm.bias, 1e-5), hidden

    def _proj_feat(self, t):
        return t.view(t.size(0), *self.feat_size, self.hidden_size).permute(0, 4, 1, 2, 3).contiguous()

    # ---- forward -----------------------------------------------------------------------------------
    def forward(self, x_in: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and x_in.requires_grad:
            raise NotImplementedError("monai_amd.UNETR: gradients w.r.t. the input are not on the (inference-only) HIP path")
        out = torch.empty((x_in.shape[0], self.out_channels) + tuple(x_in.shape[2:]), dtype=torch.float32, device=x_in.device)
        return self.forward_into(x_in, out)

    @torch.no_grad()
    def forward_into(self, x_in: torch.Tensor, logits: torch.Tensor) -> torch.Tensor:
        _lib.require_device(x_in, logits)
        if self.training:
            raise NotImplementedError("monai_amd.UNETR: training mode (autograd) is not on the HIP path -- the engine is inference-only; with MONAI installed the call falls through to the reference module, which shares these parameters")
        if tuple(x_in.shape[2:]) != self.img_size or x_in.shape[1] != self.in_channels:
            raise RuntimeError(f"monai_amd.UNETR: expected input (B,{self.in_channels},{self.img_size}), got {tuple(x_in.shape)}")
        x_in = x_in.contiguous()
        fs = self.feature_size
        # an evaluator with amp=True calls the network under torch.autocast: the library GEMMs of the ViT would come back in half precision
        # and the fp32 attention kernel would (rightly) refuse them.  This engine computes in fp32 throughout (>= the reference's precision).
        with torch.autocast(device_type=x_in.device.type, enabled=False):
            x, hs = self._vit(x_in)

        # decoder concat buffers: [upsampled | skip], each with identity records its producers fold their magnitude bounds into
        cat2 = self._new(x_in, 2 * fs)                       # decoder2 @ full resolution
        cat2_nrm = self._records(cat2)
        self._res_block(self.encoder1.layer, x_in, None, cat2[:, fs:], cat2_nrm[:, fs:])

        def prup(blk: _PrUpBlock, t, dst, dst_nrm):
            # every tensor of the chain travels with identity records its producer folds max |value| into: the next transposed convolution reads its bounds there
            cout = blk.transp_conv_init.conv.weight.shape[1]
            direct = len(blk.blocks) == 0
            cur = dst if direct else self._new(t, cout, 2)
            cur_nrm = dst_nrm if direct else self._records(cur)
            self._tconv(blk.transp_conv_init.conv, t, cur, cur_nrm, self._bounded(t))
            for i, seq in enumerate(blk.blocks):
                last = i == len(blk.blocks) - 1
                if not isinstance(seq, nn.Sequential):        # conv_block=False: the bare transposed convolution
                    nxt = dst if last else self._new(cur, cout, 2)
                    nxt_nrm = dst_nrm if last else self._records(nxt)
                    self._tconv(seq.conv, cur, nxt, nxt_nrm, cur_nrm)
                    cur, cur_nrm = nxt, nxt_nrm
                    continue
                up = self._new(cur, cout, 2)
                up_nrm = self._records(up)
                self._tconv(seq[0].conv, cur, up, up_nrm, cur_nrm)
                nxt = dst if last else self._new(up, cout)
                nxt_nrm = dst_nrm if last else self._records(nxt)
                self._res_block(seq[1], up, up_nrm, nxt, nxt_nrm)
                cur, cur_nrm = nxt, nxt_nrm
            return cur

        p2 = self._proj_feat(hs[3])
        cat3 = self._new(p2, 4 * fs, 8)                      # decoder3 @ 1/2 resolution
        cat3_nrm = self._records(cat3)
        prup(self.encoder2, p2, cat3[:, 2 * fs:], cat3_nrm[:, 2 * fs:])
        p3 = self._proj_feat(hs[6])
        cat4 = self._new(p3, 8 * fs, 4)                      # decoder4 @ 1/4
        cat4_nrm = self._records(cat4)
        prup(self.encoder3, p3, cat4[:, 4 * fs:], cat4_nrm[:, 4 * fs:])
        p4 = self._proj_feat(hs[9])
        cat5 = self._new(p4, 16 * fs, 2)                     # decoder5 @ 1/8
        cat5_nrm = self._records(cat5)
        prup(self.encoder4, p4, cat5[:, 8 * fs:], cat5_nrm[:, 8 * fs:])

        def up(blk: _UpBlock, inp, inp_nrm, cat, cat_nrm, cout, dst, head=None):
            self._tconv(blk.transp_conv.conv, inp, cat[:, :cout], cat_nrm[:, :cout], inp_nrm)
            dst_nrm = None if dst is None else self._records(dst)          # the block's join leaves the bounds the next level's transposed convolution scales by
            return self._res_block(blk.conv_block, cat, cat_nrm, dst, dst_nrm, head=head), dst_nrm

        xf = self._proj_feat(x)
        dec3, dec3_nrm = up(self.decoder5, xf, self._bounded(xf), cat5, cat5_nrm, 8 * fs, self._new(cat5, 8 * fs))
        dec2, dec2_nrm = up(self.decoder4, dec3, dec3_nrm, cat4, cat4_nrm, 4 * fs, self._new(cat4, 4 * fs))
        dec1, dec1_nrm = up(self.decoder3, dec2, dec2_nrm, cat3, cat3_nrm, 2 * fs, self._new(cat3, 2 * fs))
        up(self.decoder2, dec1, dec1_nrm, cat2, cat2_nrm, fs, None, head=(self.out.conv.conv, logits))      # + UnetOutBlock
        return logits
