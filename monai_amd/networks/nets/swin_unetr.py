"""SwinUNETR on the MI355X kernels -- drop-in for ``monai.networks.nets.SwinUNETR`` (monai/networks/nets/swin_unetr.py:45-345).

Same constructor signature, module tree and ``state_dict`` keys / shapes / buffers as the reference (``swinViT.patch_embed.proj``,
``swinViT.layers{1..4}.0.blocks.{i}.{norm1, attn.{relative_position_bias_table, relative_position_index, qkv, proj}, norm2,
mlp.{linear1, linear2}}``, ``...downsample.{reduction, norm}``, ``encoder{1,2,3,4,10}.layer.*``, ``decoder{5..1}.{transp_conv,
conv_block}.*``, ``out.conv.conv``) and the same parameter-creation order, so reference checkpoints load unchanged and the same
seed gives the same weights.

Inference engine:
  * Swin transformer (SwinTransformer :927-1069, BasicLayer :790-924, SwinTransformerBlock :544-697, PatchMerging :700-771):
    the attention core of every block -- (q * scale) k^T + relative position bias + shifted-window mask, softmax, @ v -- is the HIP
    kernel ``mh_window_attention_f32`` working straight on the qkv projection's output of all windows; the dense projections
    (patch embedding, qkv, proj, MLP with GELU and its residual, patch-merging reduction) and every LayerNorm are the HIP kernels of
    ``csrc/kernels/dense.h`` (``mh_linear_f32`` on the fp16 matrix cores in two-piece split precision, ``mh_layernorm_f32``); the
    cyclic shift, the window partition / reverse and the attention residual are torch data movement on the device;
  * conv part (UnetrBasicBlock / UnetrUpBlock with UnetResBlock, UnetOutBlock): the engine of ``monai_amd.networks.nets.unetr`` --
    fp32-MFMA 3x3x3 convolutions with fused InstanceNorm statistics, deferred normalise + LeakyReLU(0.01) on load, transposed-conv
    and 1x1 kernels, in-place concat buffers.
Configurations outside this (2-D, ``use_v2``, ``patch_norm``, other norms, custom down-sampling modules, head dims other than
8 / 16 / 32) raise ``NotImplementedError`` -- with MONAI installed they fall through to the reference class.
"""

from __future__ import annotations

from collections.abc import Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import _lib, _prof, config, ops
from ...utils.misc import ensure_tuple_rep
from .unetr import UNETR, _BasicBlock, _OutBlock, _trunc_normal_, _UpBlock

__all__ = ["SwinUNETR"]

_EXACT_ALGOS = tuple(config.CONV_ALGOS[k] for k in ("fp32", "direct", "wino2d"))     # the families ops.window_attention keeps on the exact-fp32 kernel


# --------------------------------------------------------------------------- parameter containers (reference names)
class _PatchEmbed(nn.Module):
    def __init__(self, patch_size, in_chans, embed_dim):
        super().__init__()
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.linear1 = nn.Linear(dim, hidden)
        self.linear2 = nn.Linear(hidden, dim)


class _WindowAttention(nn.Module):
    def __init__(self, dim, num_heads, window_size, qkv_bias):
        super().__init__()
        ws = window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws[0] - 1) * (2 * ws[1] - 1) * (2 * ws[2] - 1), num_heads))
        coords = torch.stack(torch.meshgrid(torch.arange(ws[0]), torch.arange(ws[1]), torch.arange(ws[2]), indexing="ij"))
        cf = torch.flatten(coords, 1)
        rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += ws[0] - 1
        rel[:, :, 1] += ws[1] - 1
        rel[:, :, 2] += ws[2] - 1
        rel[:, :, 0] *= (2 * ws[1] - 1) * (2 * ws[2] - 1)
        rel[:, :, 1] *= 2 * ws[2] - 1
        self.register_buffer("relative_position_index", rel.sum(-1))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        _trunc_normal_(self.relative_position_bias_table, std=0.02)
        self.num_heads, self.scale = num_heads, (dim // num_heads) ** -0.5


class _SwinBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size, shift_size, mlp_ratio, qkv_bias):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _WindowAttention(dim, num_heads, window_size, qkv_bias)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.window_size, self.shift_size = tuple(window_size), tuple(shift_size)


class _PatchMerging(nn.Module):
    def __init__(self, dim, v2: bool):
        super().__init__()
        self.reduction = nn.Linear(8 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(8 * dim)
        self.v2 = v2


class _BasicLayer(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size, mlp_ratio, qkv_bias, merging_v2: bool):
        super().__init__()
        shift = tuple(i // 2 for i in window_size)
        self.blocks = nn.ModuleList([_SwinBlock(dim, num_heads, window_size, (0, 0, 0) if i % 2 == 0 else shift, mlp_ratio, qkv_bias)
                                     for i in range(depth)])
        self.downsample = _PatchMerging(dim, merging_v2)
        self.window_size, self.shift_size = tuple(window_size), shift


class _SwinViT(nn.Module):
    def __init__(self, in_chans, embed_dim, window_size, patch_size, depths, num_heads, mlp_ratio, qkv_bias, merging_v2):
        super().__init__()
        self.patch_embed = _PatchEmbed(patch_size, in_chans, embed_dim)
        for i in range(4):
            setattr(self, f"layers{i + 1}", nn.ModuleList())
        for i in range(len(depths)):
            layer = _BasicLayer(int(embed_dim * 2 ** i), depths[i], num_heads[i], window_size, mlp_ratio, qkv_bias, merging_v2)
            getattr(self, f"layers{i + 1}").append(layer)


def _get_window_size(x_size, window_size, shift_size):
    """swin_unetr.py:412-438: a window never exceeds the feature map (and then is not shifted)"""
    ws, ss = list(window_size), list(shift_size)
    for i in range(len(x_size)):
        if x_size[i] <= window_size[i]:
            ws[i], ss[i] = x_size[i], 0
    return tuple(ws), tuple(ss)


def _window_partition(x, ws):
    b, d, h, w, c = x.shape
    x = x.view(b, d // ws[0], ws[0], h // ws[1], ws[1], w // ws[2], ws[2], c)
    return x.permute(0, 1, 3, 5, 2, 4, 6, 7).contiguous().view(-1, ws[0] * ws[1] * ws[2], c)


def _window_reverse(windows, ws, dims):
    b, d, h, w = dims
    x = windows.view(b, d // ws[0], h // ws[1], w // ws[2], ws[0], ws[1], ws[2], -1)
    return x.permute(0, 1, 4, 2, 5, 3, 6, 7).contiguous().view(b, d, h, w, -1)


def _window_rows(b, dims, ws, ss, device):
    """int32 [b * nW * S]: the row of x.reshape(-1, C) (x channel-last [b, d, h, w, C]) that every (window, token) of the padded, cyclically shifted, window-partitioned
    volume holds, -1 for the padding -- SwinTransformerBlock.forward's pad / roll(-shift) / window_partition (swin_unetr.py:628-648) as an index map; its inverse use
    (scatter through the same map) is window_reverse / roll(+shift) / crop (:660-670)"""
    d, h, w = dims
    dp, hp, wp = (-(-d // ws[0]) * ws[0], -(-h // ws[1]) * ws[1], -(-w // ws[2]) * ws[2])
    if b * d * h * w >= 2 ** 31:
        raise RuntimeError("monai_amd.SwinUNETR: more than 2^31 tokens in one launch")
    jz = (torch.arange(dp, device=device) + ss[0]) % dp            # rolled position i holds padded index (i + shift) mod extent
    jy = (torch.arange(hp, device=device) + ss[1]) % hp
    jx = (torch.arange(wp, device=device) + ss[2]) % wp
    row = (jz[:, None, None] * h + jy[None, :, None]) * w + jx[None, None, :]
    inside = (jz < d)[:, None, None] & (jy < h)[None, :, None] & (jx < w)[None, None, :]
    row = torch.where(inside, row, torch.full_like(row, -1))
    row = row.view(dp // ws[0], ws[0], hp // ws[1], ws[1], wp // ws[2], ws[2]).permute(0, 2, 4, 1, 3, 5).reshape(1, -1)
    off = (torch.arange(b, device=device) * (d * h * w))[:, None]
    return torch.where(row >= 0, row + off, row).reshape(-1).to(torch.int32).contiguous()


def _region_ids(dims, ws, ss, device):
    """swin_unetr.py:774-797: the region ids (0 .. 26) of the cyclically shifted volume, window-partitioned: [nW, S]"""
    d, h, w = dims
    img = torch.zeros((1, d, h, w, 1), device=device)
    cnt = 0
    for sd in (slice(-ws[0]), slice(-ws[0], -ss[0]), slice(-ss[0], None)):
        for sh in (slice(-ws[1]), slice(-ws[1], -ss[1]), slice(-ss[1], None)):
            for sw in (slice(-ws[2]), slice(-ws[2], -ss[2]), slice(-ss[2], None)):
                img[:, sd, sh, sw, :] = cnt
                cnt += 1
    return _window_partition(img, ws).squeeze(-1)


def _compute_mask(dims, ws, ss, device):
    """swin_unetr.py:774-812: region ids -> [nW, S, S] additive mask (0 where two tokens share a region, -100 elsewhere)"""
    mw = _region_ids(dims, ws, ss, device)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0).contiguous()


class SwinUNETR(UNETR):
    # UNETR supplies the conv-engine helpers (_conv3_in, _res_block, _tconv, _new, _packed_weight, _stats_buf); its constructor is NOT run
    def __init__(
        self,
        in_channels: int,
        out_channels: int,
        patch_size: int = 2,
        depths: Sequence[int] = (2, 2, 2, 2),
        num_heads: Sequence[int] = (3, 6, 12, 24),
        window_size: Sequence[int] | int = 7,
        qkv_bias: bool = True,
        mlp_ratio: float = 4.0,
        feature_size: int = 24,
        norm_name: tuple | str = "instance",
        drop_rate: float = 0.0,
        attn_drop_rate: float = 0.0,
        dropout_path_rate: float = 0.0,
        normalize: bool = True,
        norm_layer=nn.LayerNorm,
        patch_norm: bool = False,
        use_checkpoint: bool = False,
        spatial_dims: int = 3,
        downsample="merging",
        use_v2: bool = False,
    ) -> None:
        nn.Module.__init__(self)
        if spatial_dims not in (2, 3):
            raise ValueError("spatial dimension should be 2 or 3.")
        for v, nm in ((drop_rate, "dropout rate"), (attn_drop_rate, "attention dropout rate"), (dropout_path_rate, "drop path rate")):
            if not (0 <= v <= 1):
                raise ValueError(f"{nm} should be between 0 and 1.")
        if feature_size % 12 != 0:
            raise ValueError("feature_size should be divisible by 12.")
        norm = norm_name if isinstance(norm_name, str) else norm_name[0]
        if (spatial_dims != 3 or use_v2 or patch_norm or norm_layer is not nn.LayerNorm or str(norm).lower() != "instance"
                or not isinstance(downsample, str) or downsample not in ("merging", "mergingv2") or len(depths) != 4 or len(num_heads) != 4
                or int(patch_size) != 2):
            raise NotImplementedError("monai_amd.SwinUNETR: 3-D, patch_size 2, four stages, LayerNorm / instance norm, 'merging' | 'mergingv2' "
                                      "down-sampling without use_v2 / patch_norm are on the HIP path")
        window_size = ensure_tuple_rep(window_size, 3)
        for i in range(4):
            dim = feature_size * 2 ** i
            if dim % num_heads[i] or dim // num_heads[i] not in (8, 16, 32):
                raise NotImplementedError(f"monai_amd.SwinUNETR: head dim {dim / num_heads[i]} of stage {i + 1} is not on the HIP path (8, 16, 32 are)")
        if window_size[0] * window_size[1] * window_size[2] > 352:
            raise NotImplementedError("monai_amd.SwinUNETR: windows of more than 352 tokens exceed the LDS-resident attention limit")
        self.patch_size, self.normalize = int(patch_size), normalize
        self.in_channels, self.out_channels, self.feature_size = in_channels, out_channels, feature_size
        self.features = (feature_size,)                   # used by the inferer to size its window batch
        fs = feature_size
        self.swinViT = _SwinViT(in_channels, fs, window_size, (2, 2, 2), depths, num_heads, mlp_ratio, qkv_bias, downsample == "mergingv2")
        self.encoder1 = _BasicBlock(in_channels, fs)
        self.encoder2 = _BasicBlock(fs, fs)
        self.encoder3 = _BasicBlock(2 * fs, 2 * fs)
        self.encoder4 = _BasicBlock(4 * fs, 4 * fs)
        self.encoder10 = _BasicBlock(16 * fs, 16 * fs)
        self.decoder5 = _UpBlock(16 * fs, 8 * fs)
        self.decoder4 = _UpBlock(8 * fs, 4 * fs)
        self.decoder3 = _UpBlock(4 * fs, 2 * fs)
        self.decoder2 = _UpBlock(2 * fs, fs)
        self.decoder1 = _UpBlock(fs, fs)
        self.out = _OutBlock(fs, out_channels)
        self._packed: dict = {}
        self._stats = None
        self._bias_cache: dict = {}
        self._mask_cache: dict = {}
        self._rel_cache: dict = {}
        self._rows_cache: dict = {}

    # ---- Swin transformer ----------------------------------------------------------------------------------
    def _bias_t(self, attn: _WindowAttention, n: int) -> torch.Tensor:
        """relative position bias of the first n x n index entries (swin_unetr.py:524-528), per head, key-major for the kernel"""
        t = attn.relative_position_bias_table
        key = (t.data_ptr(), t._version, str(t.device), n)
        hit = self._bias_cache.get(id(attn))
        if hit is None or hit[0] != key:
            idx = attn.relative_position_index[:n, :n].reshape(-1)
            bias = t[idx].reshape(n, n, -1).permute(2, 1, 0).contiguous()       # [head][key][query]
            hit = (key, bias)
            self._bias_cache[id(attn)] = hit
        return hit[1]

    def _mask(self, dims, ws, ss, device, regions: bool = False) -> torch.Tensor:
        """the shift mask of a (feature map, window, shift): [nW, S, S] additive table, or (regions) the int32 [nW, S] region ids it is the pairwise comparison of"""
        key = (tuple(dims), tuple(ws), tuple(ss), str(device), regions)
        m = self._mask_cache.get(key)
        if m is None:
            if len(self._mask_cache) > 32:
                self._mask_cache.clear()
            m = _region_ids(dims, ws, ss, device).to(torch.int32).contiguous() if regions else _compute_mask(dims, ws, ss, device)
            self._mask_cache[key] = m
        return m

    def _rel(self, attn: _WindowAttention, n: int):
        """(table, coord, off) when the first n x n entries of the layer's relative_position_index are what swin_unetr.py:492-519 builds -- linear in the
        token coordinates, index[q][k] = coord[q] - coord[k] + off with coord[t] = index[t][0], off = index[0][0] (checked once per layer and n, on the buffer
        itself: a state_dict may have replaced it) -- and the kernel that evaluates the bias from the table takes the shape; else None (the S x S table form)."""
        t, idx = attn.relative_position_bias_table, attn.relative_position_index
        key = (idx.data_ptr(), idx._version, str(idx.device), n, t.shape[0])
        hit = self._rel_cache.get(id(attn))
        if hit is None or hit[0] != key:
            sub = idx[:n, :n].to(torch.int64)
            coord = sub[:, 0].contiguous()
            off = int(sub[0, 0])
            linear = bool(torch.equal(sub, coord[:, None] - coord[None, :] + off)) and int(sub.min()) >= 0 and int(sub.max()) < t.shape[0]
            hit = (key, (coord.to(torch.int32).contiguous(), off) if linear else None)
            self._rel_cache[id(attn)] = hit
        return hit[1]

    def _rows(self, b, dims, ws, ss, device) -> torch.Tensor:
        key = (b, tuple(dims), tuple(ws), tuple(ss), str(device))
        m = self._rows_cache.get(key)
        if m is None:
            if len(self._rows_cache) > 24:
                self._rows_cache.clear()
            m = self._rows_cache[key] = _window_rows(b, dims, ws, ss, device)
        return m

    def _block(self, blk: _SwinBlock, x):
        """SwinTransformerBlock.forward (swin_unetr.py:598-697) on channel-last x [B, d, h, w, C]"""
        b, d, h, w, c = x.shape
        ws, ss = _get_window_size((d, h, w), blk.window_size, blk.shift_size)
        pd, ph, pw = (ws[0] - d % ws[0]) % ws[0], (ws[1] - h % ws[1]) % ws[1], (ws[2] - w % ws[2]) % ws[2]
        dp, hp, wp = d + pd, h + ph, w + pw
        shifted = any(s > 0 for s in ss)
        a = blk.attn
        if config.swin_fused_moves() and ops.layernorm_gather_accepts(c):
            return self._block_mapped(blk, x, ws, ss, (dp, hp, wp), shifted)
        y = ops.layernorm(x, blk.norm1.weight, blk.norm1.bias, 1e-5)
        if pd or ph or pw:
            y = F.pad(y, (0, 0, 0, pw, 0, ph, 0, pd))
        n, hd = ws[0] * ws[1] * ws[2], c // a.num_heads
        if shifted:
            y = torch.roll(y, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3))
        win = _window_partition(y, ws)                                        # [BW, S, C]
        qkv = self._lin(win, a.qkv.weight, a.qkv.bias)
        att = self._attention(a, qkv, n, hd, (dp, hp, wp), ws, ss, shifted)
        att = self._lin(att, a.proj.weight, a.proj.bias)
        y = _window_reverse(att, ws, (b, dp, hp, wp))
        if shifted:
            y = torch.roll(y, shifts=ss, dims=(1, 2, 3))
        if pd or ph or pw:
            y = y[:, :d, :h, :w, :].contiguous()
        x = (x + y).contiguous()
        m = self._lin(ops.layernorm(x, blk.norm2.weight, blk.norm2.bias, 1e-5), blk.mlp.linear1.weight, blk.mlp.linear1.bias, gelu=True)
        return self._lin(m, blk.mlp.linear2.weight, blk.mlp.linear2.bias, residual=x)

    def _attention(self, a: _WindowAttention, qkv, n, hd, padded_dims, ws, ss, shifted):
        """the window attention of one block: bias and mask evaluated from the table / the region ids where the kernel takes the shape (split precision, head dims
        16 / 32) and the exact-fp32 family is not pinned; the S x S table form otherwise"""
        t = a.relative_position_bias_table
        rel = None
        if (config.swin_rel_attention() and config.conv_algo() not in _EXACT_ALGOS and t.dtype == torch.float32 and t.is_contiguous()
                and ops.window_attention_rel_accepts(n, hd, t.shape[0])):
            rel = self._rel(a, n)
        mask = self._mask(padded_dims, ws, ss, qkv.device, regions=rel is not None) if shifted else None
        with _prof.span("window_attention", 4.0 * n * n * hd * a.num_heads * qkv.shape[0]):
            if rel is not None:
                return ops.window_attention_rel(qkv, a.num_heads, a.scale, t, rel[0], rel[1], mask)
            return ops.window_attention(qkv, a.num_heads, a.scale, self._bias_t(a, n), mask)

    def _block_mapped(self, blk: _SwinBlock, x, ws, ss, padded_dims, shifted):
        """the block with its data movement folded into the kernels on either side of the attention (round 5): norm1 + pad + roll + window_partition = one gathering
        LayerNorm, window_reverse + roll back + crop + the shortcut sum = the projection's epilogue, both through one row map (`_window_rows`) -- the reference's
        six copies of the hidden state per block (swin_unetr.py:628-672) are not made"""
        b, d, h, w, c = x.shape
        a = blk.attn
        n, hd = ws[0] * ws[1] * ws[2], c // a.num_heads
        rows = self._rows(b, (d, h, w), ws, ss, x.device)
        win = ops.layernorm_gather(x, blk.norm1.weight, blk.norm1.bias, 1e-5, rows).view(-1, n, c)
        qkv = self._lin(win, a.qkv.weight, a.qkv.bias)
        att = self._attention(a, qkv, n, hd, padded_dims, ws, ss, shifted)
        x = self._lin(att, a.proj.weight, a.proj.bias, residual=x, scatter=rows)
        m = self._lin(ops.layernorm(x, blk.norm2.weight, blk.norm2.bias, 1e-5), blk.mlp.linear1.weight, blk.mlp.linear1.bias, gelu=True)
        return self._lin(m, blk.mlp.linear2.weight, blk.mlp.linear2.bias, residual=x)

    def _merge(self, pm: _PatchMerging, x):
        """PatchMerging / PatchMergingV2 (swin_unetr.py:700-771): 2x2x2 neighbours -> channels, LayerNorm, Linear 8C -> 2C"""
        b, d, h, w, c = x.shape
        if d % 2 or h % 2 or w % 2:
            x = F.pad(x, (0, 0, 0, w % 2, 0, h % 2, 0, d % 2))
        if pm.v2:
            order = [(i, j, k) for i in range(2) for j in range(2) for k in range(2)]
        else:
            order = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1)]
        x = torch.cat([x[:, i::2, j::2, k::2, :] for i, j, k in order], -1)
        x = ops.layernorm(x, pm.norm.weight, pm.norm.bias, 1e-5)
        return self._lin(x, pm.reduction.weight, None)

    def _swin(self, x_in):
        """SwinTransformer.forward (swin_unetr.py:1047-1069) -> the five hidden states, channel-first"""
        pe = self.swinViT.patch_embed.proj
        b, cin, d, h, w = x_in.shape
        p = x_in.reshape(b, cin, d // 2, 2, h // 2, 2, w // 2, 2).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(b, d // 2, h // 2, w // 2, cin * 8)
        x = self._lin(p, pe.weight, pe.bias)          # channel-last [B, d/2, h/2, w/2, C0]

        def proj_out(t):
            t = ops.layernorm(t, None, None, 1e-5) if self.normalize else t
            return t.permute(0, 4, 1, 2, 3).contiguous()

        outs = [proj_out(x)]
        for i in range(4):
            layer = getattr(self.swinViT, f"layers{i + 1}")[0]
            for blk in layer.blocks:
                x = self._block(blk, x)
            x = self._merge(layer.downsample, x)
            outs.append(proj_out(x))
        return outs

    # ---- forward ---------------------------------------------------------------------------------------------
    def forward(self, x_in: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and x_in.requires_grad:
            raise NotImplementedError("monai_amd.SwinUNETR: gradients w.r.t. the input are not on the (inference-only) HIP path")
        out = torch.empty((x_in.shape[0], self.out_channels) + tuple(x_in.shape[2:]), dtype=torch.float32, device=x_in.device)
        return self.forward_into(x_in, out)

    @torch.no_grad()
    def forward_into(self, x_in: torch.Tensor, logits: torch.Tensor) -> torch.Tensor:
        _lib.require_device(x_in, logits)
        if self.training:
            raise NotImplementedError("monai_amd.SwinUNETR: training mode (autograd) is not on the HIP path -- the engine is inference-only; with MONAI "
                                      "installed the call falls through to the reference module, which shares these parameters")
        if x_in.dim() != 5 or x_in.shape[1] != self.in_channels:
            raise RuntimeError(f"monai_amd.SwinUNETR: expected input (B,{self.in_channels},D,H,W), got {tuple(x_in.shape)}")
        rem = [i + 2 for i, v in enumerate(x_in.shape[2:]) if int(v) % (self.patch_size ** 5)]
        if rem:          # _check_input_size, swin_unetr.py:316-324
            raise ValueError(f"spatial dimensions {rem} of input image (spatial shape: {tuple(x_in.shape[2:])}) must be divisible by {self.patch_size}**5.")
        x_in = x_in.contiguous()
        fs = self.feature_size
        with torch.autocast(device_type=x_in.device.type, enabled=False):          # fp32 throughout, also under an evaluator's amp=True
            hs = self._swin(x_in)

        def new(like, c, scale=1):
            return torch.empty((like.shape[0], c) + tuple(int(v * scale) for v in like.shape[2:]), dtype=torch.float32, device=like.device)

        # decoder concat buffers [upsampled | skip]; the encoders write their halves in place.  The buffers come with identity records that the
        # producers (residual joins, transposed convolutions) fold max |value| into: the magnitude bounds the split-precision convolution scales by.
        # The Swin hidden states are LayerNorm outputs without an affine map when `normalize` (proj_out, swin_unetr.py:1039-1045): every voxel's C channels have
        # mean 0 and sum of squares <= C, so |value| <= sqrt(C) -- a rigorous bound, which puts the convolutions that read them on the split-precision kernels
        # (round 5; before, and still with normalize=False, they arrive without bounds and run on the exact-fp32 kernels).
        def hidden_records(t):
            if not self.normalize:
                return None
            rec = torch.tensor([1.0, 0.0, 1.0, float(np.sqrt(t.shape[1])) * (1.0 + 1e-6)], dtype=torch.float32, device=t.device)
            return rec.expand(t.shape[0], t.shape[1], 4).contiguous()

        cat1 = new(x_in, 2 * fs)                                     # decoder1 @ full resolution
        cat1_nrm = self._records(cat1)
        self._res_block(self.encoder1.layer, x_in, None, cat1[:, fs:], cat1_nrm[:, fs:])
        cat2 = new(hs[0], 2 * fs)                                    # decoder2 @ 1/2
        cat2_nrm = self._records(cat2)
        self._res_block(self.encoder2.layer, hs[0], hidden_records(hs[0]), cat2[:, fs:], cat2_nrm[:, fs:])
        cat3 = new(hs[1], 4 * fs)                                    # decoder3 @ 1/4
        cat3_nrm = self._records(cat3)
        self._res_block(self.encoder3.layer, hs[1], hidden_records(hs[1]), cat3[:, 2 * fs:], cat3_nrm[:, 2 * fs:])
        cat4 = new(hs[2], 8 * fs)                                    # decoder4 @ 1/8
        cat4_nrm = self._records(cat4)
        self._res_block(self.encoder4.layer, hs[2], hidden_records(hs[2]), cat4[:, 4 * fs:], cat4_nrm[:, 4 * fs:])
        cat5 = new(hs[3], 16 * fs)                                   # decoder5 @ 1/16: the skip is the hidden state itself
        cat5[:, 8 * fs:].copy_(hs[3])
        cat5_nrm = None
        if self.normalize:
            cat5_nrm = self._records(cat5)
            cat5_nrm[:, 8 * fs:] = hidden_records(hs[3])
        dec4 = new(hs[4], 16 * fs)
        dec4_nrm = self._records(dec4)          # the join leaves the bounds the transposed convolution of decoder5 scales by (csrc/kernels/deconv_h2.h)
        self._res_block(self.encoder10.layer, hs[4], hidden_records(hs[4]), dec4, dec4_nrm)

        def up(blk: _UpBlock, inp, inp_nrm, cat, cat_nrm, cout, dst, head=None):
            self._tconv(blk.transp_conv.conv, inp, cat[:, :cout], None if cat_nrm is None else cat_nrm[:, :cout], inp_nrm)
            dst_nrm = None if dst is None else self._records(dst)
            return self._res_block(blk.conv_block, cat, cat_nrm, dst, dst_nrm, head=head), dst_nrm

        dec3, dec3_nrm = up(self.decoder5, dec4, dec4_nrm, cat5, cat5_nrm, 8 * fs, new(cat5, 8 * fs))
        dec2, dec2_nrm = up(self.decoder4, dec3, dec3_nrm, cat4, cat4_nrm, 4 * fs, new(cat4, 4 * fs))
        dec1, dec1_nrm = up(self.decoder3, dec2, dec2_nrm, cat3, cat3_nrm, 2 * fs, new(cat3, 2 * fs))
        dec0, dec0_nrm = up(self.decoder2, dec1, dec1_nrm, cat2, cat2_nrm, fs, new(cat2, fs))
        up(self.decoder1, dec0, dec0_nrm, cat1, cat1_nrm, fs, None, head=(self.out.conv.conv, logits))      # + UnetOutBlock
        return logits
