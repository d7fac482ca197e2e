"""``SegResNet`` on the MI355X kernels -- drop-in for ``monai.networks.nets.SegResNet`` (monai/networks/nets/segresnet.py:31-213;
blocks: monai/networks/blocks/segresnet_block.py:25-100, upsampling: monai/networks/blocks/upsample.py:43-184).

Same constructor signature, module tree (``convInit`` / ``down_layers`` / ``up_layers`` / ``up_samples`` / ``conv_final``), the same
``state_dict`` keys / shapes and the same construction order (so the same seed gives the same weights).

Inference engine (the engine of BasicUNet / UNet / UNETR / DynUNet): the pre-activation residual block
``x + conv2(relu(norm2(conv1(relu(norm1 x)))))`` maps onto the deferred-normalisation convolutions directly -- GroupNorm
statistics come from the per-channel records the convolutions already emit (``mh_groupnorm_finalize_f32`` merges the channels of
a group), the normalise + ReLU is applied by the consuming convolution on load, the residual add is one fused pass.  Stride-2
convolutions run on the direct kernel, the 2x trilinear upsampling on the affine resampler (index = o / 2 - 1 / 4, border
clamp: ``F.interpolate(scale_factor=2, mode="trilinear", align_corners=False)``), ``upsample_mode="deconv"`` on the k2s2
transposed-conv kernel.  On the HIP path: 3-D, group / instance norm, (leaky) ReLU, ``nontrainable`` / ``deconv`` upsampling;
dropout is inference-inert."""

from __future__ import annotations

import torch
import torch.nn as nn

from ... import _lib, _prof, config, ops

__all__ = ["SegResNet"]


# --------------------------------------------------------------------------- parameter containers (reference names)
class _Conv(nn.Module):
    """``Convolution(conv_only=True)``: the parameter lives at ``<name>.conv.weight``"""

    def __init__(self, cin, cout, k=3, stride=1, bias=False):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=bias)


def _norm(kind, args, channels):
    if kind == "group":
        return nn.GroupNorm(num_groups=int(args.get("num_groups", 8)), num_channels=channels, eps=float(args.get("eps", 1e-5)),
                            affine=bool(args.get("affine", True)))
    return nn.InstanceNorm3d(channels, eps=float(args.get("eps", 1e-5)), affine=bool(args.get("affine", False)))


def _act(slope):
    return nn.ReLU(inplace=True) if slope == 0.0 else nn.LeakyReLU(slope, inplace=True)


class _ResBlock(nn.Module):
    """segresnet_block.py:48-100 (children in the reference's registration order)"""

    def __init__(self, channels, kind, args, slope):
        super().__init__()
        self.norm1 = _norm(kind, args, channels)
        self.norm2 = _norm(kind, args, channels)
        self.act = _act(slope)
        self.conv1 = _Conv(channels, channels)
        self.conv2 = _Conv(channels, channels)


class _UpSample(nn.Sequential):
    """``UpSample`` (upsample.py:43-184) in its two parameter layouts: ``deconv`` or the parameter-free ``upsample_non_trainable``"""

    def __init__(self, channels, mode):
        super().__init__()
        if mode == "deconv":
            self.add_module("deconv", nn.ConvTranspose3d(channels, channels, kernel_size=2, stride=2))
        else:
            self.add_module("upsample_non_trainable", nn.Upsample(scale_factor=2, mode="trilinear", align_corners=False))


# --------------------------------------------------------------------------- the module
class SegResNet(nn.Module):
    def __init__(
        self,
        spatial_dims: int = 3,
        init_filters: int = 8,
        in_channels: int = 1,
        out_channels: int = 2,
        dropout_prob: float | None = None,
        act: tuple | str = ("RELU", {"inplace": True}),
        norm: tuple | str = ("GROUP", {"num_groups": 8}),
        norm_name: str = "",
        num_groups: int = 8,
        use_conv_final: bool = True,
        blocks_down: tuple = (1, 2, 2, 4),
        blocks_up: tuple = (1, 1, 1),
        upsample_mode: str = "nontrainable",
    ) -> None:
        super().__init__()
        if spatial_dims not in (2, 3):
            raise ValueError("`spatial_dims` can only be 2 or 3.")
        if spatial_dims != 3:
            raise NotImplementedError("monai_amd.SegResNet: only spatial_dims=3 is on the HIP path")
        self.spatial_dims, self.init_filters, self.in_channels, self.out_channels = spatial_dims, init_filters, in_channels, out_channels
        self.blocks_down, self.blocks_up, self.dropout_prob, self.act = blocks_down, blocks_up, dropout_prob, act
        aname, aargs = (act, {}) if isinstance(act, str) else (act[0], act[1] if len(act) > 1 else {})
        if str(aname).lower() == "relu":
            slope = 0.0
        elif str(aname).lower() == "leakyrelu":
            slope = float(aargs.get("negative_slope", 0.01))
        else:
            raise NotImplementedError("monai_amd.SegResNet: only (leaky) ReLU is on the HIP path")
        self.act_mod = _act(slope)
        if norm_name:
            if norm_name.lower() != "group":
                raise ValueError(f"Deprecating option 'norm_name={norm_name}', please use 'norm' instead.")
            norm = ("group", {"num_groups": num_groups})
        self.norm = norm
        kind, nargs = (norm, {}) if isinstance(norm, str) else (norm[0], norm[1] if len(norm) > 1 else {})
        kind = str(kind).lower()
        if kind not in ("group", "instance"):
            raise NotImplementedError("monai_amd.SegResNet: only group / instance norm are on the HIP path")
        mode = str(getattr(upsample_mode, "value", upsample_mode)).lower()
        if mode not in ("nontrainable", "deconv"):
            raise NotImplementedError("monai_amd.SegResNet: upsample_mode 'nontrainable' / 'deconv' are on the HIP path")
        self.upsample_mode, self.use_conv_final, self._slope = mode, use_conv_final, slope
        if len(blocks_up) != len(blocks_down) - 1:
            raise NotImplementedError("monai_amd.SegResNet: len(blocks_up) must be len(blocks_down) - 1 (every up level needs its skip)")
        f = init_filters
        self.features = (f,)     # used by the inferer to size its window batch

        # construction order = the reference's (segresnet.py:107-113)
        self.convInit = _Conv(in_channels, f)
        self.down_layers = nn.ModuleList()
        for i, item in enumerate(blocks_down):
            c = f * 2 ** i
            pre = _Conv(c // 2, c, stride=2) if i > 0 else nn.Identity()
            self.down_layers.append(nn.Sequential(pre, *[_ResBlock(c, kind, nargs, slope) for _ in range(item)]))
        self.up_layers, self.up_samples = nn.ModuleList(), nn.ModuleList()
        n_up = len(blocks_up)
        for i in range(n_up):
            c = f * 2 ** (n_up - i)
            self.up_layers.append(nn.Sequential(*[_ResBlock(c // 2, kind, nargs, slope) for _ in range(blocks_up[i])]))
            self.up_samples.append(nn.Sequential(_Conv(c, c // 2, k=1), _UpSample(c // 2, mode)))
        self.conv_final = nn.Sequential(_norm(kind, nargs, f), self.act_mod, _Conv(f, out_channels, k=1, bias=True))
        if dropout_prob is not None:
            self.dropout = nn.Dropout3d(dropout_prob)
        self._packed: dict = {}
        self._stats = None

    # ---- helpers -----------------------------------------------------------------------------------
    def _packed_weight(self, conv: nn.Conv3d, cfg: int) -> torch.Tensor:
        w = conv.weight
        key = (w.data_ptr(), w._version, str(w.device))
        hit = self._packed.get((id(conv), cfg))
        if hit is None or hit[0] != key:
            hit = (key, ops.conv3d_k3_pack(cfg, w))
            self._packed[(id(conv), cfg)] = hit
        return hit[1]

    def _stats_buf(self, floats: int, device) -> torch.Tensor:
        # two live sets at most: the record set a convolution writes while its input's has already been finalised
        if self._stats is None or self._stats.numel() < floats or self._stats.device != device:
            self._stats = torch.empty(floats, dtype=torch.float32, device=device)
        return self._stats

    def _record(self, norm, x, stats, tiles):
        """{alpha, beta, slope} of `norm` (+ activation) for the raw tensor x; statistics from `stats` or an own pass over x"""
        n, c = x.shape[:2]
        if not tiles:
            tiles = ops.instnorm_stat_tiles(*x.shape[2:])
            stats = self._stats_buf(n * c * tiles * 3, x.device)
            ops.instnorm_stats(x, stats)
        nrm = torch.empty((n, c, 4), dtype=torch.float32, device=x.device)
        groups = norm.num_groups if isinstance(norm, nn.GroupNorm) else c
        ops.groupnorm_finalize(stats, tiles, n, c, groups, norm.weight, norm.bias, norm.eps, self._slope, nrm)
        return nrm

    def _conv3(self, conv: nn.Conv3d, x, x_nrm, stride: int = 1):
        """3x3x3 conv (no bias) of the (deferred) input -> (raw output, its statistics records or None, tiles)"""
        n, cin, d, h, w = x.shape
        cout = conv.weight.shape[0]
        sp = tuple((v - 1) // stride + 1 for v in (d, h, w))
        out = torch.empty((n, cout) + sp, dtype=torch.float32, device=x.device)
        if stride == 1 and not (cin <= 8 and cout <= 8):
            cfg = ops.conv3d_k3_select(cin, cout, d, h, w, bounded=x_nrm is not None)      # every record here comes from groupnorm_finalize: it carries a magnitude bound
            tiles = ops.conv3d_k3_stat_tiles(cfg, d, h, w)
            stats = self._stats_buf(n * cout * tiles * 3, x.device) if tiles else None
            with _prof.span(f"conv3d_k3/cfg{cfg}", 2.0 * 27 * cin * cout * d * h * w * n):
                ops.conv3d_k3(cfg, x, x_nrm, self._packed_weight(conv, cfg), conv.bias, out, stats)
            return out, stats, tiles
        if ops.conv3d_k3s2_selected(cin, cout, d, h, w, stride, bounded=x_nrm is not None):
            # the down-sampling convolution on the fp16 matrix cores (csrc/kernels/conv3d_s2_h2.h), statistics of its output included
            tiles = ops.conv3d_k3s2_stat_tiles(d, h, w)
            stats = self._stats_buf(n * cout * tiles * 3, x.device)
            with _prof.span("conv3d_k3s2", 2.0 * 27 * cin * cout * sp[0] * sp[1] * sp[2] * n):
                fused = ops.conv3d_k3s2_fused(cin, cout, d * h * w)          # conversion inside the GEMM's staging, or a phase-split pass into the workspace first
                ops.conv3d_k3s2(x, x_nrm, self._packed_s2(conv), conv.bias, out, stats, None if fused else self._workspace(ops.conv3d_k3s2_workspace_floats(n, cin, d, h, w), x.device), fused)
            return out, stats, tiles
        ops.conv3d_k3_strided(x, x_nrm, self._packed_weight(conv, 0), conv.bias, out, stride)
        return out, None, 0

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

    def _res_block(self, blk: _ResBlock, x, stats=None, tiles=0, out_nrm=None):
        """x + conv2(act(norm2(conv1(act(norm1 x))))) for a plain x (whose statistics records may come from its producer) -> (result, its statistics records or None, tiles).
        Where the second convolution's configuration has an accumulating form (the split-precision kernels) the join is that form: conv2 ADDS itself to x in place and leaves
        the statistics of the sum (what the next block's norm1 needs) -- no pass for the addition, none for the statistics; `x` is this engine's own tensor and has no
        other reader.  out_nrm: `nrm_identity` records the join must leave the result's magnitude bounds in (the split-precision down-sampling convolution that reads it
        scales its input by them): that join stays a pass of its own.  Reference: ResBlock.forward, monai/networks/blocks/segresnet_block.py:62-97."""
        n1 = self._record(blk.norm1, x, stats, tiles)
        c1, s1, t1 = self._conv3(blk.conv1.conv, x, n1)
        n2 = self._record(blk.norm2, c1, s1, t1)
        conv2 = blk.conv2.conv
        n, cin, d, h, w = c1.shape
        cout = conv2.weight.shape[0]
        if out_nrm is None and config.residual_accumulate() and not (cin <= 8 and cout <= 8):
            cfg = ops.conv3d_k3_select(cin, cout, d, h, w, bounded=True)
            if cfg in (ops.conv3d_k3_h2_config(), ops.conv3d_k3_h2c_config(), ops.conv3d_k3_h2w_config()):
                tiles2 = ops.conv3d_k3_stat_tiles(cfg, d, h, w)
                stats2 = self._stats_buf(n * cout * tiles2 * 3, x.device)
                with _prof.span(f"conv3d_k3/cfg{cfg}", 2.0 * 27 * cin * cout * d * h * w * n):
                    ops.conv3d_k3(cfg, c1, n2, self._packed_weight(conv2, cfg), conv2.bias, x, stats2, accumulate=True)
                return x, stats2, tiles2
        c2, _, _ = self._conv3(conv2, c1, n2)
        return ops.add_act(c2, None, x, None, 1.0, torch.empty_like(c2), out_nrm), None, 0

    # ---- forward -----------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and x.requires_grad:
            raise NotImplementedError("monai_amd.SegResNet: gradients w.r.t. the input are not on the (inference-only) HIP path")
        c = self.out_channels if self.use_conv_final else self.init_filters
        out = torch.empty((x.shape[0], c) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
        return self.forward_into(x, out)

    @torch.no_grad()
    def forward_into(self, x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        _lib.require_device(x, out)
        if self.training:
            raise NotImplementedError("monai_amd.SegResNet: training mode (autograd) is not on the HIP path -- the engine is inference-only; with MONAI installed the call falls through to the reference module, which shares these parameters")
        total = 2 ** (len(self.blocks_down) - 1)
        if x.dim() != 5 or x.shape[1] != self.in_channels or any(int(v) % total for v in x.shape[2:]):
            raise NotImplementedError(f"monai_amd.SegResNet: input (B,{self.in_channels},D,H,W) with edges divisible by {total} expected, got {tuple(x.shape)}")
        # encode (segresnet.py:170-182); dropout is the identity in eval mode
        t, stats, tiles = self._conv3(self.convInit.conv, x.contiguous(), None)
        down_x = []
        t_nrm = None           # identity records with the magnitude bounds of `t`, left by the last residual join of a level that a strided convolution follows
        for li, layer in enumerate(self.down_layers):
            if not isinstance(layer[0], nn.Identity):
                t, stats, tiles = self._conv3(layer[0].conv, t, t_nrm, stride=2)
                t_nrm = None
            blks = list(layer)[1:]
            for bi, blk in enumerate(blks):
                feeds_down = bi == len(blks) - 1 and li + 1 < len(self.down_layers) and not isinstance(self.down_layers[li + 1][0], nn.Identity)
                if feeds_down:
                    t_nrm = ops.nrm_identity(torch.empty((t.shape[0], t.shape[1], 4), dtype=torch.float32, device=t.device))
                t, stats, tiles = self._res_block(blk, t, stats, tiles, t_nrm if feeds_down else None)
            down_x.append(t)
        down_x.reverse()
        # decode (segresnet.py:184-192): x = up(x) + skip; x = up_layer(x)
        for i, (up, upl) in enumerate(zip(self.up_samples, self.up_layers)):
            w1 = up[0].conv.weight
            cout = w1.shape[0]
            low = torch.empty((t.shape[0], cout) + tuple(t.shape[2:]), dtype=torch.float32, device=t.device)
            ops.conv1x1(t, None, w1.view(cout, -1), None, low)
            n, _, d, h, w = low.shape
            if self.upsample_mode == "deconv":
                dc = up[1].deconv
                hi = ops.deconv_k2s2(low, None, dc.weight, dc.bias, torch.empty((n, cout, 2 * d, 2 * h, 2 * w), dtype=torch.float32, device=t.device))
            else:
                m = [0.5, 0, 0, -0.25, 0, 0.5, 0, -0.25, 0, 0, 0.5, -0.25]
                hi = ops.affine_resample(low.reshape(n * cout, d, h, w), m, (2 * d, 2 * h, 2 * w), "bilinear", "border", False, False)
                hi = hi.reshape(n, cout, 2 * d, 2 * h, 2 * w)
            t = ops.add_act(hi, None, down_x[i + 1], None, 1.0, torch.empty_like(hi))
            stats, tiles = None, 0
            for blk in upl:
                t, stats, tiles = self._res_block(blk, t, stats, tiles)
        if not self.use_conv_final:
            out.copy_(t)
            return out
        oc = self.conv_final[2].conv
        ops.conv1x1(t, self._record(self.conv_final[0], t, stats, tiles), oc.weight.view(oc.weight.shape[0], -1), oc.bias, out)
        return out
