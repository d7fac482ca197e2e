"""``DynUNet`` (the nnU-Net architecture) on the MI355X kernels -- drop-in for ``monai.networks.nets.DynUNet``
(monai/networks/nets/dynunet.py:32-418; blocks: monai/networks/blocks/dynunet_block.py:25-328).

Same constructor signature, the same module tree (``input_block`` / ``downsamples`` / ``bottleneck`` / ``upsamples`` /
``output_block`` / ``deep_supervision_heads`` and the recursive ``skip_layers`` that re-registers the same blocks) and therefore
the same ``state_dict`` keys / shapes, and the same order of random draws at construction (default Conv initialisers, the
``torch.rand(1)`` of ``self.heads``, then ``kaiming_normal_(a=0.01)`` in ``apply`` order): reference checkpoints load unchanged
and the same seed gives the same weights.

Inference engine (the same convolution / normalisation engine as BasicUNet / UNet / UNETR): every 3x3x3 conv
output is stored raw with its InstanceNorm (affine) + LeakyReLU folded into a per-(n, c) {alpha, beta, slope} record that the
consumer applies on load; stride-1 convs run on the fp32-MFMA tiles with fused statistics, stride-2 convs on the direct
kernel, the k2s2 transposed convs and the 1x1 head on their own kernels; the encoder output of each level is materialised
exactly once, straight into the skip half of that level's concat buffer (``torch.cat`` never runs).
On the HIP path: 3-D, and 2-D as ONE PLANE of the 3-D engine (kernel (1, k, k), stride (1, s, s); the parameters live in the reference's 2-D modules, so
the state_dict is the 2-D net's -- what SliceInferer drives); kernel extents 1 / 3 and strides 1 / 2 PER AXIS (anisotropic nnU-Net plans such as kernel (1, 3, 3), stride (1, 2, 2): an
extent-1 axis runs as a 3-tap kernel with zero outer taps, per-axis strides on the direct kernel, kernel == stride transposed convs on a gather
kernel), upsample kernels equal to the strides, instance norm, (leaky) ReLU,
``res_block`` False or True, dropout layers present in the module tree and inference-inert; deep-supervision heads are parameters only (they feed the training loss, the
inference output does not depend on them)."""

from __future__ import annotations

import threading

from collections.abc import Sequence

import torch
import torch.nn as nn

from ... import _lib, _prof, config, ops

__all__ = ["DynUNet", "DynUnet", "Dynunet"]


def _triple(v, what: str, allowed, dims: int = 3) -> tuple:
    """an int or a `dims`-sequence -> (z, y, x) ints, each one of `allowed`; a 2-D network is the 3-D engine on one plane: its z entry is 1
    (kernel extent 1, stride 1, upsample factor 1 -- the anisotropic plans of the 3-D path)"""
    if isinstance(v, (list, tuple)):
        if len(v) != dims:
            raise ValueError(f"length of {what} should be the same as spatial_dims.")
        t = tuple(int(a) for a in v)
    else:
        t = (int(v),) * dims
    if dims == 2:
        t = (1,) + t
    if any(a not in allowed for a in t):
        raise NotImplementedError(f"monai_amd.DynUNet: {what} {t} is not on the HIP path (per axis: {sorted(allowed)})")
    return t


def _out_size(size, stride):
    return tuple((int(v) - 1) // s + 1 for v, s in zip(size, stride))


# --------------------------------------------------------------------------- parameter containers (reference names)
def _dropout_module(dropout):
    """The ``D`` of the reference's ``ADN`` (blocks/acti_norm.py:69-101, layers/factories.py `Dropout`) for 3-D: a probability, a name or
    ``(name, kwargs)``.  Parameter-free and the identity in eval mode -- the inference schedule never calls it; it is in the tree because
    the reference's is (code that walks ``net.modules()``, tests/networks/nets/test_dynunet.py:120-124)."""
    if isinstance(dropout, (int, float)):
        name, args = "dropout", {"p": float(dropout)}
    elif isinstance(dropout, str):
        name, args = dropout, {}
    else:
        name, args = dropout[0], (dropout[1] if len(dropout) > 1 else {})
    kinds = {"dropout": nn.Dropout3d, "alphadropout": nn.AlphaDropout}
    if not isinstance(name, str) or name.lower() not in kinds:
        raise NotImplementedError(f"monai_amd.DynUNet: dropout layer {name!r} is not known to the HIP path")
    return kinds[name.lower()](**args)


class _ADN(nn.Module):
    def __init__(self, dropout):
        super().__init__()
        self.D = _dropout_module(dropout)


class _Conv(nn.Module):
    """``get_conv_layer(..., act=None, norm=None)``: a ``Convolution`` whose children are ``conv`` and, with dropout, ``adn.D``"""

    # construction context, per THREAD (set by DynUNet.__init__ while it builds its blocks; concurrent constructions must not see each other's):
    # `dropout` = the ADN's dropout argument or None; `dims` = 2 builds the reference's 2-D modules (Conv2d ...: the state_dict of a 2-D net), the engine
    # reads them as one-plane 3-D
    _build = threading.local()

    def __init__(self, cin, cout, k, stride=(1, 1, 1), transposed=False, bias=False):
        super().__init__()
        k = (k,) * 3 if isinstance(k, int) else tuple(k)
        stride = tuple(stride)
        pad = tuple((a - b + 1) // 2 for a, b in zip(k, stride))       # get_padding, dynunet_block.py:304-315: (k - s + 1) / 2 per axis, truncated
        if getattr(_Conv._build, "dims", 3) == 2:
            if transposed:
                self.conv = nn.ConvTranspose2d(cin, cout, kernel_size=k[1:], stride=stride[1:], bias=bias)
            else:
                self.conv = nn.Conv2d(cin, cout, kernel_size=k[1:], stride=stride[1:], padding=pad[1:], bias=bias)
        elif transposed:
            self.conv = nn.ConvTranspose3d(cin, cout, kernel_size=k, stride=stride, bias=bias)
        else:
            self.conv = nn.Conv3d(cin, cout, kernel_size=k, stride=stride, padding=pad, bias=bias)
        if getattr(_Conv._build, "dropout", None) is not None:
            self.adn = _ADN(_Conv._build.dropout)


class _Block(nn.Module):
    """UnetBasicBlock (dynunet_block.py:114-166) / UnetResBlock (:25-111): same children in the same order"""

    def __init__(self, cin, cout, kernel, stride, affine, slope, res):
        super().__init__()
        self.stride, self.res = tuple(stride), bool(res)
        self.conv1 = _Conv(cin, cout, kernel, stride)
        self.conv2 = _Conv(cout, cout, kernel)
        self.lrelu = nn.LeakyReLU(slope, inplace=True) if slope != 0.0 else nn.ReLU(inplace=True)
        norm_t = nn.InstanceNorm2d if getattr(_Conv._build, "dims", 3) == 2 else nn.InstanceNorm3d
        self.norm1 = norm_t(cout, affine=affine)
        self.norm2 = norm_t(cout, affine=affine)
        if res and (cin != cout or any(a != 1 for a in stride)):
            self.conv3 = _Conv(cin, cout, 1, stride)
            self.norm3 = norm_t(cout, affine=affine)


class _UpBlock(nn.Module):
    """UnetUpBlock (dynunet_block.py:169-229)"""

    def __init__(self, cin, cout, kernel, up, affine, slope, trans_bias):
        super().__init__()
        self.up = tuple(up)
        self.transp_conv = _Conv(cin, cout, up, up, transposed=True, bias=trans_bias)
        self.conv_block = _Block(cout + cout, cout, kernel, (1, 1, 1), affine, slope, False)


class _OutBlock(nn.Module):
    """UnetOutBlock (dynunet_block.py:232-253)"""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = _Conv(cin, cout, 1, bias=True)


class _SkipLayer(nn.Module):
    """DynUNetSkipLayer (dynunet.py:32-65): parameter-free, re-registers the blocks under ``skip_layers.*``"""

    def __init__(self, index, downsample, upsample, next_layer, super_head=None):
        super().__init__()
        self.downsample, self.next_layer, self.upsample, self.super_head, self.index = downsample, next_layer, upsample, super_head, index


# --------------------------------------------------------------------------- the module
class DynUNet(nn.Module):
    def __init__(
        self,
        spatial_dims: int,
        in_channels: int,
        out_channels: int,
        kernel_size: Sequence,
        strides: Sequence,
        upsample_kernel_size: Sequence,
        filters: Sequence[int] | None = None,
        dropout=None,
        norm_name: tuple | str = ("INSTANCE", {"affine": True}),
        act_name: tuple | str = ("leakyrelu", {"inplace": True, "negative_slope": 0.01}),
        deep_supervision: bool = False,
        deep_supr_num: int = 1,
        res_block: bool = False,
        trans_bias: bool = False,
    ) -> None:
        super().__init__()
        self.spatial_dims, self.in_channels, self.out_channels = spatial_dims, in_channels, out_channels
        self.kernel_size, self.strides, self.upsample_kernel_size = kernel_size, strides, upsample_kernel_size
        self.norm_name, self.act_name, self.dropout, self.trans_bias = norm_name, act_name, dropout, trans_bias
        # dynunet.py:213-231 (checked before any parameter exists here; the reference checks after building)
        if len(kernel_size) != len(strides) or len(kernel_size) < 3:
            raise ValueError("length of kernel_size and strides should be the same, and no less than 3.")
        if spatial_dims not in (2, 3):
            raise NotImplementedError("monai_amd.DynUNet: spatial_dims 2 and 3 are on the HIP path")
        ks = [_triple(k, f"kernel_size in block {i}", {1, 3}, spatial_dims) for i, k in enumerate(kernel_size)]
        ss = [_triple(s, f"stride in block {i}", {1, 2}, spatial_dims) for i, s in enumerate(strides)]
        us = [_triple(u, "upsample_kernel_size", {1, 2}, spatial_dims) for u in upsample_kernel_size]
        if len(us) != len(ss) - 1 or any(u != s for u, s in zip(us, ss[1:])):
            raise NotImplementedError("monai_amd.DynUNet: upsample_kernel_size must equal strides[1:] on the HIP path")
        # dropout: the reference puts a Dropout module into every conv layer's ADN (dynunet_block.py:256-301); they are in this tree too
        # (parameter-free: checkpoints load unchanged) and inert -- the engine runs in eval mode only, training falls through
        if dropout is not None:
            _dropout_module(dropout)          # an unknown layer name: NotImplementedError -> the reference's class
        nname, nargs = (norm_name, {}) if isinstance(norm_name, str) else (norm_name[0], norm_name[1] if len(norm_name) > 1 else {})
        if str(nname).lower() != "instance":
            raise NotImplementedError("monai_amd.DynUNet: only instance norm is on the HIP path")
        affine = bool(nargs.get("affine", False))
        aname, aargs = (act_name, {}) if isinstance(act_name, str) else (act_name[0], act_name[1] if len(act_name) > 1 else {})
        if str(aname).lower() == "leakyrelu":
            slope = float(aargs.get("negative_slope", 0.01))
        elif str(aname).lower() == "relu":
            slope = 0.0
        else:
            raise NotImplementedError("monai_amd.DynUNet: only (leaky) ReLU is on the HIP path")
        self._strides, self._slope = ss, slope
        if filters is not None:
            if len(filters) < len(strides):
                raise ValueError("length of filters should be no less than the length of strides.")
            self.filters = list(filters[: len(strides)])
        else:
            self.filters = [min(2 ** (5 + i), 320) for i in range(len(strides))]
        f = self.filters
        self.features = (f[0],)    # used by the inferer to size its window batch
        self.window_sized_output = ss[0] == (1, 1, 1)     # the inferer writes straight into its logits buffer only then

        def block(cin, cout, k, s):
            return _Block(cin, cout, k, s, affine, slope, res_block)

        # construction order = the reference's (dynunet.py:154-166): it fixes the random stream of the default initialisers
        _Conv._build.dropout, _Conv._build.dims = dropout, spatial_dims
        try:
            self.input_block = block(in_channels, f[0], ks[0], ss[0])
            self.downsamples = nn.ModuleList([block(i, o, k, s) for i, o, k, s in zip(f[:-2], f[1:-1], ks[1:-1], ss[1:-1])])
            self.bottleneck = block(f[-2], f[-1], ks[-1], ss[-1])
            self.upsamples = nn.ModuleList([_UpBlock(i, o, k, u, affine, slope, trans_bias)
                                            for i, o, k, u in zip(f[1:][::-1], f[:-1][::-1], ks[1:][::-1], us[::-1])])
            self.output_block = _OutBlock(f[0], out_channels)
            self.deep_supervision, self.deep_supr_num = deep_supervision, deep_supr_num
            self.heads = [torch.rand(1)] * deep_supr_num            # one draw from the global generator, as the reference
            if deep_supervision:
                self.deep_supervision_heads = nn.ModuleList([_OutBlock(f[i + 1], out_channels) for i in range(deep_supr_num)])
                if deep_supr_num >= len(strides) - 1:
                    raise ValueError("deep_supr_num should be less than the number of up sample layers.")
                if deep_supr_num < 1:
                    raise ValueError("deep_supr_num should be larger than 0.")
        finally:
            _Conv._build.dropout, _Conv._build.dims = None, 3
        self.apply(self.initialize_weights)

        def create_skips(index, downs, ups, heads):
            if len(downs) != len(ups):
                raise ValueError(f"{len(downs)} != {len(ups)}")
            if len(downs) == 0:
                return self.bottleneck
            if heads is None:
                return _SkipLayer(index, downs[0], ups[0], create_skips(index + 1, downs[1:], ups[1:], None))
            head, rest = None, heads
            if index > 0:
                head, rest = (heads[0], heads[1:]) if len(heads) > 0 else (None, nn.ModuleList())
            return _SkipLayer(index, downs[0], ups[0], create_skips(index + 1, downs[1:], ups[1:], rest), super_head=head)

        self.skip_layers = create_skips(0, [self.input_block] + list(self.downsamples), self.upsamples[::-1],
                                        self.deep_supervision_heads if deep_supervision else None)
        self._packed: dict = {}
        self._stats = None

    @staticmethod
    def initialize_weights(module):
        """dynunet.py:412-417"""
        if isinstance(module, (nn.Conv3d, nn.Conv2d, nn.ConvTranspose3d, nn.ConvTranspose2d)):
            module.weight = nn.init.kaiming_normal_(module.weight, a=0.01)
            if module.bias is not None:
                module.bias = nn.init.constant_(module.bias, 0)

    # ---- helpers -----------------------------------------------------------------------------------
    @staticmethod
    def _w5(conv) -> torch.Tensor:
        """the layer's weight as the engine sees it: 5-D; a 2-D layer's [O, I, kh, kw] is the one-plane kernel [O, I, 1, kh, kw] (a view)"""
        w = conv.weight
        return w if w.dim() == 5 else w.unsqueeze(2)

    def _packed_weight(self, conv: nn.Conv3d, cfg: int) -> torch.Tensor:
        w = self._w5(conv)
        key = (w.data_ptr(), conv.weight._version, str(w.device))
        hit = self._packed.get((id(conv), cfg))
        if hit is None or hit[0] != key:
            if tuple(w.shape[2:]) != (3, 3, 3):
                # a kernel extent of 1 along an axis = a 3-tap kernel whose outer taps are zero: with padding 1 the centre tap sits on the
                # same sample (s * o) as the reference's padding-0 extent-1 kernel.  Exact; the zero taps cost matrix time, not accuracy.
                w3 = torch.zeros(w.shape[:2] + (3, 3, 3), dtype=w.dtype, device=w.device)
                sl = tuple(slice(0, 3) if k == 3 else slice(1, 2) for k in w.shape[2:])
                w3[(slice(None), slice(None)) + sl] = w
                w = w3
            hit = (key, ops.conv3d_k3_pack(cfg, w))
            self._packed[(id(conv), cfg)] = hit
        return hit[1]

    def _packed_slice(self, conv: nn.Conv3d, cfg: int, c0: int, c1: int) -> torch.Tensor:
        """packed weights of input channels c0 .. c1 - 1 of a 3x3x3 convolution (a convolution evaluated in two halves of its input channels)"""
        w = conv.weight
        key = (w.data_ptr(), w._version, str(w.device))
        hit = self._packed.get((id(conv), cfg, c0, c1))
        if hit is None or hit[0] != key:
            hit = (key, ops.conv3d_k3_pack(cfg, w[:, c0:c1].contiguous()))
            self._packed[(id(conv), cfg, c0, c1)] = hit
        return hit[1]

    def _packed_s2(self, conv: nn.Conv3d) -> torch.Tensor:
        """the stride-2 split-precision kernel's tap matrices of a [Cout, Cin, 3, 3, 3] weight (once per parameter version)"""
        w = conv.weight
        key = (w.data_ptr(), w._version, str(w.device))
        hit = self._packed.get((id(conv), "s2"))
        if hit is None or hit[0] != key:
            hit = (key, ops.conv3d_k3s2_pack(w))
            self._packed[(id(conv), "s2")] = hit
        return hit[1]

    def _workspace(self, floats: int, device) -> torch.Tensor:
        """scratch of the stride-2 kernel (the phase-split fp16 pieces of its input): one buffer, grown to the largest layer"""
        ws = getattr(self, "_ws", None)
        if ws is None or ws.numel() < floats or ws.device != device:
            self._ws = ws = torch.empty(floats, dtype=torch.float32, device=device)
        return ws

    def _stats_buf(self, floats: int, device) -> torch.Tensor:
        if self._stats is None or self._stats.numel() < floats or self._stats.device != device:
            self._stats = torch.empty(floats, dtype=torch.float32, device=device)
        return self._stats

    def _finalize(self, norm: nn.InstanceNorm3d, raw, stats, tiles, slope):
        n, c = raw.shape[:2]
        if not tiles:
            tiles = ops.instnorm_stat_tiles(*raw.shape[2:])
            stats = self._stats_buf(n * c * tiles * 3, raw.device)
            ops.instnorm_stats(raw, stats)
        nrm = torch.empty((n, c, 4), dtype=torch.float32, device=raw.device)
        ops.instnorm_finalize(stats, tiles, n, c, norm.weight, norm.bias, norm.eps, slope, nrm)
        return nrm

    def _conv_norm(self, conv: nn.Conv3d, norm, x, x_nrm, stride, slope: float, out=None):
        """3x3x3 conv (no bias) of the (deferred) input + InstanceNorm statistics -> (raw output, {alpha, beta, slope} record); `out`: where the raw output goes (a channel
        range of a concat buffer: the skip tensor is then never materialised -- its consumers apply the record on load)"""
        n, cin, d, h, w = x.shape
        cout = conv.weight.shape[0]
        sp = _out_size((d, h, w), stride)
        if out is None:
            out = torch.empty((n, cout) + sp, dtype=torch.float32, device=x.device)
        tiles, stats = 0, None
        flops = 2.0 * 27 * cin * cout * sp[0] * sp[1] * sp[2] * n
        if stride == (1, 1, 1) and not (cin <= 8 and cout <= 8):
            # every record of this engine carries a magnitude bound: instnorm_finalize writes one, plain tensors come with `nrm_identity` records their
            # producers (add_act, the transposed convolutions) folded max |value| into -- what the split-precision kernel scales its input by
            cfg = ops.conv3d_k3_select(cin, cout, d, h, w, bounded=x_nrm is not None)
            h2 = ops.conv3d_k3_h2_config()
            half = (cin // 32) * 16
            if (cfg != h2 and x_nrm is not None and tuple(conv.weight.shape[2:]) == (3, 3, 3) and cin > 256 and cin % 16 == 0
                    and ops.conv3d_k3_select(half, cout, d, h, w, bounded=True) == h2 and ops.conv3d_k3_select(cin - half, cout, d, h, w, bounded=True) == h2):
                # more input channels than the split-precision kernel keeps records for (the 512-channel concat of nnU-Net's 12^3 level): the convolution is linear in
                # its input channels -- one half written, the other half added onto it by the accumulating form, which leaves the statistics of the sum
                tiles = ops.conv3d_k3_stat_tiles(h2, d, h, w)
                stats = self._stats_buf(n * cout * tiles * 3, x.device)
                with _prof.span(f"conv3d_k3/cfg{h2}", flops):
                    ops.conv3d_k3(h2, x[:, :half], x_nrm[:, :half], self._packed_slice(conv, h2, 0, half), conv.bias, out, None)
                    ops.conv3d_k3(h2, x[:, half:], x_nrm[:, half:], self._packed_slice(conv, h2, half, cin), None, out, stats, accumulate=True)
                return out, self._finalize(norm, out, stats, tiles, slope)
            hw = ops.conv3d_k3_h2w_config()
            if (cfg == h2 and cin == 64 and x_nrm is not None and tuple(conv.weight.shape[2:]) == (3, 3, 3) and config.conv_halves()
                    and ops.conv3d_k3_select(32, cout, d, h, w, bounded=True) == hw):
                # a 64-channel concatenation at a level whose 32-channel halves the Winograd split-precision kernel takes (the top decoder level: 64 -> 32 @ 96^3): two launches of it
                # (plain form, then accumulating form with the statistics of the sum) cost less than one of the direct kernel -- BasicUNet._conv_halves, config.CONV_HALVES
                tiles = ops.conv3d_k3_stat_tiles(hw, d, h, w)
                stats = self._stats_buf(n * cout * tiles * 3, x.device)
                with _prof.span(f"conv3d_k3/cfg{hw}", flops):
                    ops.conv3d_k3(hw, x[:, :32], x_nrm[:, :32], self._packed_slice(conv, hw, 0, 32), None, out, None)
                    ops.conv3d_k3(hw, x[:, 32:], x_nrm[:, 32:], self._packed_slice(conv, hw, 32, 64), conv.bias, out, stats, accumulate=True)
                return out, self._finalize(norm, out, stats, tiles, slope)
            tiles = ops.conv3d_k3_stat_tiles(cfg, d, h, w)
            stats = self._stats_buf(n * cout * tiles * 3, x.device) if tiles else None
            with _prof.span(f"conv3d_k3/cfg{cfg}", flops):
                ops.conv3d_k3(cfg, x, x_nrm, self._packed_weight(conv, cfg), conv.bias, out, stats)
        elif tuple(conv.weight.shape[2:]) == (3, 3, 3) and ops.conv3d_k3s2_selected(cin, cout, d, h, w, stride, bounded=x_nrm is not None):
            # the down-sampling convolution on the fp16 matrix cores (csrc/kernels/conv3d_s2_h2.h), statistics of its output included
            tiles = ops.conv3d_k3s2_stat_tiles(d, h, w)
            stats = self._stats_buf(n * cout * tiles * 3, x.device)
            with _prof.span("conv3d_k3s2", flops):
                fused = ops.conv3d_k3s2_fused(cin, cout, d * h * w)          # conversion inside the GEMM's staging, or a phase-split pass into the workspace first
                ops.conv3d_k3s2(x, x_nrm, self._packed_s2(conv), conv.bias, out, stats, None if fused else self._workspace(ops.conv3d_k3s2_workspace_floats(n, cin, d, h, w), x.device), fused)
        else:       # other strides, or so few channels that the matrix tiles would mostly pad: the direct kernel at the true width
            ops.conv3d_k3_strided3(x, x_nrm, self._packed_weight(conv, 0), conv.bias, out, stride)
        return out, self._finalize(norm, out, stats, tiles, slope)

    def _basic(self, blk: _Block, x, x_nrm, out=None):
        """UnetBasicBlock: conv1 -> norm1 -> lrelu -> conv2 -> norm2 -> lrelu, the last normalise + activate left to the consumer"""
        c1, n1 = self._conv_norm(blk.conv1.conv, blk.norm1, x, x_nrm, blk.stride, self._slope)
        return self._conv_norm(blk.conv2.conv, blk.norm2, c1, n1, (1, 1, 1), self._slope, out)

    @staticmethod
    def _records(t: torch.Tensor) -> torch.Tensor:
        """fresh identity records for a plain tensor that is about to be written (its producer leaves the magnitude bounds in them)"""
        return ops.nrm_identity(torch.empty((t.shape[0], t.shape[1], 4), dtype=torch.float32, device=t.device))

    def _res(self, blk: _Block, x, x_nrm, dst, dst_nrm):
        """UnetResBlock of a plain tensor (+ its identity records) into `dst` (plain): lrelu(norm2(conv2(lrelu(norm1(conv1 x)))) + shortcut)"""
        c1, n1 = self._conv_norm(blk.conv1.conv, blk.norm1, x, x_nrm, blk.stride, self._slope)
        c2, n2 = self._conv_norm(blk.conv2.conv, blk.norm2, c1, n1, (1, 1, 1), 1.0)
        if not hasattr(blk, "conv3"):
            return ops.add_act(c2, n2, x, None, self._slope, dst, dst_nrm)
        w3 = self._w5(blk.conv3.conv)
        cout = w3.shape[0]
        r = torch.empty_like(c2)
        stats, tiles = None, 0
        if blk.stride == (1, 1, 1):
            tiles = ops.conv1x1_stat_tiles(*r.shape[2:])          # norm3's statistics come out of the shortcut convolution itself
            stats = self._stats_buf(r.shape[0] * cout * tiles * 3, r.device)
            ops.conv1x1(x, None, w3.view(cout, -1), None, r, stats)
        else:       # the strided 1x1 shortcut as the centre tap of the strided 3x3x3 kernel
            ops.conv3d_k3_strided3(x, None, self._packed_weight(blk.conv3.conv, 0), None, r, blk.stride)
        n3 = self._finalize(blk.norm3, r, stats, tiles, 1.0)
        return ops.add_act(c2, n2, r, n3, self._slope, dst, dst_nrm)

    def _encode(self, blk: _Block, x, x_nrm, dst, dst_nrm):
        """an encoder block of a plain tensor, materialised into `dst` (the skip half of a concat buffer)"""
        if blk.res:
            return self._res(blk, x, x_nrm, dst, dst_nrm)
        # a basic block's result stays deferred: its last convolution writes the raw values straight into `dst`, the record of norm2 + lrelu (with its magnitude bound)
        # goes into `dst_nrm` -- both consumers (the next level's first convolution, the decoder's concat convolution) apply records on load.  Until round 5 a pass
        # materialised it (3.5 % of the nnU-Net-shaped network's step).
        c, cn = self._basic(blk, x, x_nrm, dst)
        dst_nrm.copy_(cn)
        return c

    def _level(self, i: int, x, x_nrm, downs, ups):
        """DynUNetSkipLayer.forward at depth i: down -> next level -> transposed conv + [up | skip] concat -> conv block (deferred)"""
        blk, up = downs[i], ups[i]
        cout = blk.conv1.conv.weight.shape[0]
        n = x.shape[0]
        sp = _out_size(x.shape[2:], blk.stride)
        cat = torch.empty((n, 2 * cout) + sp, dtype=torch.float32, device=x.device)       # torch.cat((out, skip), dim=1)
        cat_nrm = self._records(cat)
        skip, skip_nrm = self._encode(blk, x, x_nrm, cat[:, cout:], cat_nrm[:, cout:]), cat_nrm[:, cout:]
        if i + 1 < len(downs):
            t, tn = self._level(i + 1, skip, skip_nrm, downs, ups)
        elif self.bottleneck.res:
            bsp = _out_size(sp, self.bottleneck.stride)
            t = torch.empty((n, self.filters[-1]) + bsp, dtype=torch.float32, device=x.device)
            t, tn = self._res(self.bottleneck, skip, skip_nrm, t, None), None
        else:
            t, tn = self._basic(self.bottleneck, skip, skip_nrm)
        tc = up.transp_conv.conv
        if up.up == (2, 2, 2):
            ops.deconv_k2s2(t, tn, tc.weight, tc.bias, cat[:, :cout], cat_nrm[:, :cout], bounded=tn is not None)      # records of instnorm_finalize: they carry bounds
        else:
            ops.deconv_ks(t, tn, self._w5(tc).contiguous(), tc.bias, cat[:, :cout], up.up, cat_nrm[:, :cout])
        return self._basic(up.conv_block, cat, cat_nrm)

    # ---- forward -----------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and x.requires_grad:
            raise NotImplementedError("monai_amd.DynUNet: gradients w.r.t. the input are not on the (inference-only) HIP path")
        if self.spatial_dims == 2:
            if x.dim() != 4:
                raise NotImplementedError(f"monai_amd.DynUNet: a 2-D network takes (B, C, H, W), got {tuple(x.shape)}")
            sp = _out_size((1,) + tuple(x.shape[2:]), self._strides[0])
            out = torch.empty((x.shape[0], self.out_channels) + sp[1:], dtype=torch.float32, device=x.device)
            return self.forward_into(x, out)
        sp = _out_size(x.shape[2:], self._strides[0])
        out = torch.empty((x.shape[0], self.out_channels) + sp, dtype=torch.float32, device=x.device)
        return self.forward_into(x, out)

    @torch.no_grad()
    def forward_into(self, x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        _lib.require_device(x, out)
        if self.spatial_dims == 2 and x.dim() == 4 and out.dim() == 4:
            self.forward_into(x.unsqueeze(2), out.unsqueeze(2))          # one plane of the 3-D engine (views: no copy)
            return out
        if self.training:
            raise NotImplementedError("monai_amd.DynUNet: training mode (autograd) is not on the HIP path -- the engine is inference-only; with MONAI installed the call falls through to the reference module, which shares these parameters")
        total = [1, 1, 1]
        for st in self._strides:
            total = [a * b for a, b in zip(total, st)]
        if x.dim() != 5 or x.shape[1] != self.in_channels or any(int(v) % t for v, t in zip(x.shape[2:], total)):
            raise NotImplementedError(f"monai_amd.DynUNet: input (B,{self.in_channels},D,H,W) with edges divisible by {tuple(total)} expected, got {tuple(x.shape)}")
        if tuple(out.shape) != (x.shape[0], self.out_channels) + _out_size(x.shape[2:], self._strides[0]):
            raise RuntimeError(f"monai_amd.DynUNet: output buffer of shape {tuple(out.shape)} does not fit input {tuple(x.shape)}")
        downs = [self.input_block] + list(self.downsamples)
        t, tn = self._level(0, x.contiguous(), None, downs, list(self.upsamples[::-1]))
        oc = self.output_block.conv.conv
        ops.conv1x1(t, tn, oc.weight.view(oc.weight.shape[0], -1), oc.bias, out)
        return out


DynUnet = Dynunet = DynUNet
