"""MONAI ``UNet`` on the MI355X kernels -- drop-in for ``monai.networks.nets.UNet`` (monai/networks/nets/unet.py:106-298).

Same constructor signature, the same recursive module tree (``model.0`` down layer, ``model.1.submodule`` skip-connected
sub-block, ``model.2`` up layer; ``ResidualUnit`` = ``conv.unit{i}`` + ``residual``) and therefore the same ``state_dict``
keys/shapes and parameter-initialisation order as the reference.

``norm="batch"`` (the spleen / tutorial configuration) is on the path as well: eval-mode BatchNorm3d is a per-channel affine map, i.e. directly
the consumer-side record, with no statistics pass at all.

Inference engine over the C ABI: every conv output is stored raw with its InstanceNorm+PReLU folded into a per-(n, c)
{alpha, beta, slope} record applied by the consumer on load (the PReLU weight is the slope); stride-1 3x3x3 convs run on the
fp32-MFMA tiles, the strided encoder convs and the k3 transposed convs on direct kernels (UNet is 11.8 GFLOP per 96^3 window --
bandwidth, not FLOPs, dominates it); residual adds are one fused kernel; ``torch.cat`` of the skip connection never runs (both
producers write straight into the concat buffer)."""

from __future__ import annotations

import threading
import warnings
from collections.abc import Sequence

import torch
import torch.nn as nn

from ... import _lib, ops

__all__ = ["UNet", "Unet"]


# --------------------------------------------------------------------------- parameter containers (reference names)
_ADN_SPEC = threading.local()      # (activation name, its arguments, ordering) of the network under construction -- per thread, like basic_unet._BUILD


def _parse_act(act):
    """-> (name, arguments) of an activation the deferred-activation records can express: y > 0 ? y : slope * y (blocks/acti_norm.py:69-101 via layers/factories.py)"""
    name, args = (act, {}) if isinstance(act, str) else (act[0], dict(act[1]) if len(act) > 1 else {})
    if not isinstance(name, str):       # an activation given as a class / factory callable (blocks/acti_norm.py accepts both)
        raise NotImplementedError("monai_amd.UNet: the HIP path takes the activation by name")
    name = name.upper()
    if name not in ("PRELU", "RELU", "LEAKYRELU"):
        raise NotImplementedError(f"monai_amd.UNet: activation {act!r} is not on the HIP path (PReLU / ReLU / LeakyReLU are)")
    return name, args


class _ADN(nn.Module):
    """ADN parameter holder: the modules are registered in the order of `adn_ordering` (as the reference's `for item in ordering`), which is also the order of the
    state_dict keys; Dropout(0.0) is a module without parameters (UNet always passes dropout=0.0 -> the reference creates it wherever "D" appears)"""

    def __init__(self, channels: int, affine):
        super().__init__()
        name, args, ordering = getattr(_ADN_SPEC, "v", ("PRELU", {}, "NDA"))
        for item in ordering:
            if item == "N":
                # `affine`: bool -> InstanceNorm3d(affine); ("batch", kwargs) -> BatchNorm3d (evaluated with its running statistics)
                self.N = nn.BatchNorm3d(channels, **affine[1]) if isinstance(affine, tuple) else nn.InstanceNorm3d(channels, affine=affine)
            elif item == "D":
                self.D = nn.Dropout(0.0)
            elif item == "A":
                if name == "PRELU":
                    self.A = nn.PReLU(**{k: v for k, v in args.items() if k in ("num_parameters", "init")})
                elif name == "RELU":
                    self.A = nn.ReLU(inplace=bool(args.get("inplace", False)))
                else:
                    self.A = nn.LeakyReLU(negative_slope=float(args.get("negative_slope", 0.01)), inplace=bool(args.get("inplace", False)))
        # activation BEFORE the normalisation ("AN", "ADN", "DAN", "AND"): the statistics are those of the activated tensor
        self.act_first = "A" in ordering and "N" in ordering and ordering.index("A") < ordering.index("N")


class _Convolution(nn.Module):
    def __init__(self, cin, cout, strides=1, bias=True, conv_only=False, is_transposed=False, affine=False):
        super().__init__()
        self.strides, self.is_transposed = int(strides), is_transposed
        if is_transposed:
            self.conv = nn.ConvTranspose3d(cin, cout, 3, stride=strides, padding=1, output_padding=strides - 1, bias=bias)
        else:
            self.conv = nn.Conv3d(cin, cout, 3, stride=strides, padding=1, bias=bias)
        if not conv_only and getattr(_ADN_SPEC, "v", ("PRELU", {}, "NDA"))[2]:
            self.adn = _ADN(cout, affine)


class _ResidualUnit(nn.Module):
    def __init__(self, cin, cout, strides=1, subunits=2, bias=True, last_conv_only=False, affine=False):
        super().__init__()
        self.strides = int(strides)
        self.conv = nn.Sequential()
        self.residual = nn.Identity()
        sc, ss = cin, strides
        for su in range(max(1, subunits)):
            self.conv.add_module(f"unit{su:d}", _Convolution(sc, cout, ss, bias, conv_only=last_conv_only and su == max(1, subunits) - 1, affine=affine))
            sc, ss = cout, 1
        if strides != 1 or cin != cout:
            k = 3 if strides != 1 else 1
            self.residual = nn.Conv3d(cin, cout, k, strides, 1 if k == 3 else 0, bias=bias)


class _SkipConnection(nn.Module):
    def __init__(self, submodule):
        super().__init__()
        self.submodule = submodule


# --------------------------------------------------------------------------- the module
class UNet(nn.Module):
    def __init__(
        self,
        spatial_dims: int,
        in_channels: int,
        out_channels: int,
        channels: Sequence[int],
        strides: Sequence[int],
        kernel_size: Sequence[int] | int = 3,
        up_kernel_size: Sequence[int] | int = 3,
        num_res_units: int = 0,
        act: tuple | str = "PRELU",
        norm: tuple | str = "INSTANCE",
        dropout: float = 0.0,
        bias: bool = True,
        adn_ordering: str = "NDA",
    ) -> None:
        super().__init__()
        if len(channels) < 2:
            raise ValueError("the length of `channels` should be no less than 2.")
        delta = len(strides) - (len(channels) - 1)
        if delta < 0:
            raise ValueError("the length of `strides` should equal to `len(channels) - 1`.")
        if delta > 0:
            warnings.warn(f"`len(strides) > len(channels) - 1`, the last {delta} values of strides will not be used.")
        if isinstance(kernel_size, Sequence) and len(kernel_size) != spatial_dims:
            raise ValueError("the length of `kernel_size` should equal to `dimensions`.")
        if isinstance(up_kernel_size, Sequence) and len(up_kernel_size) != spatial_dims:
            raise ValueError("the length of `up_kernel_size` should equal to `dimensions`.")
        act_name, act_args = _parse_act(act)
        norm_name, norm_args = (norm, {}) if isinstance(norm, str) else (norm[0], norm[1] if len(norm) > 1 else {})
        ordering = str(adn_ordering).upper()
        if any(ch not in "NDA" for ch in ordering):
            raise ValueError(f"ordering must be a string of {{'A': None, 'D': None, 'N': None}}, got {[ch for ch in ordering if ch not in 'NDA'][0]} in it.")   # acti_norm.py:98-99
        if len(set(ordering)) != len(ordering):
            raise KeyError(f"attribute '{[ch for ch in ordering if ordering.count(ch) > 1][0]}' already exists")      # nn.Module.add_module on the repeated letter
        if (spatial_dims != 3 or kernel_size not in (3, (3, 3, 3), [3, 3, 3]) or up_kernel_size not in (3, (3, 3, 3), [3, 3, 3])
                or str(norm_name).upper() not in ("INSTANCE", "BATCH") or any(int(s) not in (1, 2) for s in strides)):
            raise NotImplementedError("monai_amd.UNet: the HIP path covers 3-D, kernel 3, PReLU / ReLU / LeakyReLU + instance / batch norm in any `adn_ordering`, strides 1/2 "
                                      "(dropout is inference-inert)")
        self.dimensions, self.in_channels, self.out_channels = spatial_dims, in_channels, out_channels
        self.channels, self.strides, self.num_res_units, self.bias = tuple(channels), tuple(int(s) for s in strides), num_res_units, bias
        self.kernel_size, self.up_kernel_size, self.act, self.norm, self.dropout, self.adn_ordering = kernel_size, up_kernel_size, act, norm, dropout, adn_ordering
        affine = bool(norm_args.get("affine", False))
        if str(norm_name).upper() == "BATCH":       # eval-mode BatchNorm is a per-channel affine map: it IS a {alpha, beta} record, no statistics pass
            affine = ("batch", {k: v for k, v in norm_args.items() if k in ("eps", "momentum", "affine", "track_running_stats")})
        self.features = (channels[0],)
        _ADN_SPEC.v = (act_name, act_args, ordering)
        try:
            self._create_model(in_channels, out_channels, num_res_units, bias, affine)
        finally:
            del _ADN_SPEC.v
        self._packed: dict = {}
        self._slopes: dict = {}
        self._stats = None

    def _create_model(self, in_channels, out_channels, num_res_units, bias, affine) -> None:
        def down(cin, cout, s):
            if num_res_units > 0:
                return _ResidualUnit(cin, cout, s, num_res_units, bias, affine=affine)
            return _Convolution(cin, cout, s, bias, affine=affine)

        def up(cin, cout, s, is_top):
            conv = _Convolution(cin, cout, s, bias, conv_only=is_top and num_res_units == 0, is_transposed=True, affine=affine)
            if num_res_units > 0:
                return nn.Sequential(conv, _ResidualUnit(cout, cout, 1, 1, bias, last_conv_only=is_top, affine=affine))
            return conv

        def create(inc, outc, chans, strs, is_top):
            c, s = chans[0], strs[0]
            if len(chans) > 2:
                sub = create(c, c, chans[1:], strs[1:], False)
                upc = c * 2
            else:
                sub = down(c, chans[1], 1)
                upc = c + chans[1]
            d = down(inc, c, s)
            u = up(upc, outc, s, is_top)
            return nn.Sequential(d, _SkipConnection(sub), u)

        self.model = create(in_channels, out_channels, self.channels, self.strides, True)

    # ---- helpers -----------------------------------------------------------------------------------
    def _packed_weight(self, conv: nn.Conv3d, cfg: int) -> torch.Tensor:
        w = conv.weight
        key = (w.data_ptr(), w._version, str(w.device))
        hit = self._packed.get((id(conv), cfg))
        if hit is None or hit[0] != key:
            hit = (key, ops.conv3d_k3_pack(cfg, w))
            self._packed[(id(conv), cfg)] = hit
        return hit[1]

    def _slope(self, adn: _ADN) -> float:
        """the activation of an ADN as the slope of `y > 0 ? y : slope * y`: PReLU's weight, LeakyReLU's negative_slope, 0 for ReLU, 1 (identity) without an activation"""
        act = getattr(adn, "A", None)
        if act is None:
            return 1.0
        if isinstance(act, nn.ReLU):
            return 0.0
        if isinstance(act, nn.LeakyReLU):
            return float(act.negative_slope)
        w = act.weight
        if w.numel() != 1:
            raise NotImplementedError("monai_amd.UNet: per-channel PReLU is not on the HIP path")
        key = (w.data_ptr(), w._version)
        hit = self._slopes.get(id(act))
        if hit is None or hit[0] != key:
            hit = (key, float(w.detach().cpu()))     # one host read per weight version
            self._slopes[id(act)] = hit
        return hit[1]

    def _act_record(self, slope: float, n: int, c: int, device) -> torch.Tensor:
        """[n, c, 4] records {1, 0, slope, 0}: the bare activation (no normalisation in front of it; no magnitude bound)"""
        hit = self._packed.get(("act", slope, n, c, str(device)))
        if hit is None:
            hit = torch.tensor([1.0, 0.0, slope, 0.0], dtype=torch.float32, device=device).repeat(n, c, 1).contiguous()
            self._packed[("act", slope, n, c, str(device))] = hit
        return hit

    def _bn_record(self, bn: nn.BatchNorm3d, slope: float, n: int) -> torch.Tensor:
        """Eval-mode BatchNorm3d + PReLU as the consumer-side record [n, C, 4] = {alpha, beta, slope, 0}: alpha = weight / sqrt(running_var
        + eps), beta = bias - running_mean * alpha -- the x * alpha + beta form of ATen's CPU batch norm.  A parameter fold over C values
        (cached per parameter version), not a pass over activations."""
        if bn.running_mean is None or bn.running_var is None:
            raise NotImplementedError("monai_amd.UNet: BatchNorm without running statistics is not on the (inference) HIP path")
        parts = [bn.running_mean, bn.running_var] + ([bn.weight, bn.bias] if bn.affine else [])
        key = tuple((t.data_ptr(), t._version) for t in parts) + (slope, str(bn.running_mean.device))
        hit = self._packed.get(("bn", id(bn)))
        if hit is None or hit[0] != key:
            invstd = 1.0 / torch.sqrt(bn.running_var.float() + bn.eps)
            alpha = invstd * bn.weight.float() if bn.affine else invstd
            beta = (bn.bias.float() if bn.affine else 0.0) - bn.running_mean.float() * alpha
            tab = torch.stack([alpha, beta, torch.full_like(alpha, slope), torch.zeros_like(alpha)], dim=1).contiguous()
            hit = (key, tab)
            self._packed[("bn", id(bn))] = hit
        return hit[1].unsqueeze(0).expand(n, -1, -1).contiguous()

    def _has_batchnorm(self) -> bool:
        """BatchNorm anywhere in the net, or ADN blocks without a normalisation (their records carry no magnitude bounds: the exact-fp32 convolutions take them) --
        the module tree is walked once, not per convolution launch"""
        hit = self.__dict__.get("_bn_cached")
        if hit is None:
            hit = self.__dict__["_bn_cached"] = any(isinstance(m, nn.BatchNorm3d) or (isinstance(m, _ADN) and not hasattr(m, "N")) for m in self.modules())
        return hit

    def _stats_buf(self, floats: int, device) -> torch.Tensor:
        if self._stats is None or self._stats.numel() < floats or self._stats.device != device:
            self._stats = torch.empty(floats, dtype=torch.float32, device=device)
        return self._stats

    def _conv_unit(self, unit: _Convolution, x, x_nrm):
        """`Convolution`: conv (+bias) of the (deferred) input -> (raw output, {alpha, beta, slope} record or None)."""
        n, cin, d, h, w = x.shape
        conv = unit.conv
        s = unit.strides
        if unit.is_transposed:
            cout = conv.weight.shape[1]
            out = torch.empty((n, cout, d * s, h * s, w * s), dtype=torch.float32, device=x.device)
            ops.deconv_k3(x, x_nrm, conv.weight, conv.bias, out, s)
            stats_tiles = 0
        else:
            cout = conv.weight.shape[0]
            do, ho, wo = (d - 1) // s + 1, (h - 1) // s + 1, (w - 1) // s + 1
            out = torch.empty((n, cout, do, ho, wo), dtype=torch.float32, device=x.device)
            # few channels on both sides (the 5-class top level): the matrix tiles would pad them to 32; the direct kernel runs at
            # the channels' true width
            tiny = s == 1 and cin <= 8 and cout <= 8
            # records written by instnorm_finalize carry magnitude bounds (the split-precision kernel needs them); folded BatchNorm records do not
            bounded = x_nrm is not None and not self._has_batchnorm()
            cfg = ops.conv3d_k3_select(cin, cout, d, h, w, bounded=bounded) if (s == 1 and not tiny) else 0
            wants_stats = hasattr(unit, "adn") and hasattr(unit.adn, "N") and not isinstance(unit.adn.N, nn.BatchNorm3d) and not unit.adn.act_first
            stats_tiles = ops.conv3d_k3_stat_tiles(cfg, d, h, w) if (s == 1 and not tiny and wants_stats) else 0
            if s == 1 and not tiny:
                stats = self._stats_buf(n * cout * stats_tiles * 3, x.device) if stats_tiles else None
                ops.conv3d_k3(cfg, x, x_nrm, self._packed_weight(conv, cfg), conv.bias, out, stats)
            elif s == 2 and bounded and ops.conv3d_k3s2_selected(cin, cout, d, h, w, s, bounded=True):
                # the down-sampling convolution on the fp16 matrix cores (csrc/kernels/conv3d_s2_h2.h), the statistics of its output included (round 6: plain tensors of this
                # engine carry identity records with magnitude bounds, left by the residual joins that write them)
                tiles_ = ops.conv3d_k3s2_stat_tiles(d, h, w)
                stats = self._stats_buf(n * cout * tiles_ * 3, x.device)
                self._conv_s2(conv, x, x_nrm, out, stats)
                stats_tiles = tiles_ if wants_stats else 0
            else:
                ops.conv3d_k3_strided(x, x_nrm, self._packed_weight(conv, 0), conv.bias, out, s)
        if not hasattr(unit, "adn"):
            return out, None
        adn = unit.adn
        slope = self._slope(adn)
        if not hasattr(adn, "N"):                   # "A" / "DA" / "D": the bare activation as a record
            return out, (self._act_record(slope, n, cout, x.device) if slope != 1.0 else None)
        if adn.act_first:                           # "AN...": activate, THEN normalise -- the statistics are those of the activated tensor
            if slope != 1.0:
                out = ops.add_act(out, self._act_record(slope, n, cout, x.device), None, None, 1.0, torch.empty_like(out))
            stats_tiles, slope = 0, 1.0
        if isinstance(adn.N, nn.BatchNorm3d):
            return out, self._bn_record(adn.N, slope, n)
        if not stats_tiles:
            stats_tiles = ops.instnorm_stat_tiles(*out.shape[2:])
            stats = self._stats_buf(n * cout * stats_tiles * 3, x.device)
            ops.instnorm_stats(out, stats)
        nrm = torch.empty((n, cout, 4), dtype=torch.float32, device=x.device)
        inorm = unit.adn.N
        ops.instnorm_finalize(stats, stats_tiles, n, cout, inorm.weight, inorm.bias, inorm.eps, slope, nrm)
        return out, nrm

    def _conv_s2(self, conv: nn.Conv3d, x, x_nrm, out, stats) -> None:
        """3x3x3 stride-2 convolution of a bounded input on the split-precision stride-2 kernel; `stats`: the InstanceNorm statistics of `out`"""
        n, cin, d, h, w = x.shape
        cout = conv.weight.shape[0]
        wt = conv.weight
        key = (wt.data_ptr(), wt._version, str(wt.device))
        hit = self._packed.get((id(conv), "s2"))
        if hit is None or hit[0] != key:
            hit = (key, ops.conv3d_k3s2_pack(wt))
            self._packed[(id(conv), "s2")] = hit
        fused = ops.conv3d_k3s2_fused(cin, cout, d * h * w)          # conversion inside the GEMM's staging, or a phase-split pass into the workspace first
        ws = None
        if not fused:
            need = ops.conv3d_k3s2_workspace_floats(n, cin, d, h, w)
            ws = self.__dict__.get("_ws")
            if ws is None or ws.numel() < need or ws.device != x.device:
                ws = self.__dict__["_ws"] = torch.empty(need, dtype=torch.float32, device=x.device)
        ops.conv3d_k3s2(x, x_nrm, hit[1], conv.bias, out, stats, ws, fused)

    def _residual_unit(self, ru: _ResidualUnit, x, x_nrm, dst, dst_nrm=None):
        """cx + res into `dst` (plain; dst_nrm: its `nrm_identity` records, the join leaves the magnitude bounds in them).  `x` may be deferred (raw + record): both the
        sub-units and the shortcut consume it."""
        n = x.shape[0]
        if isinstance(ru.residual, nn.Identity):
            res, res_nrm = x, x_nrm
        else:
            rc = ru.residual
            cout = rc.weight.shape[0]
            s = ru.strides
            if rc.kernel_size[0] == 3:
                cin, d, h, w = x.shape[1:]
                res = torch.empty((n, cout) + tuple((v - 1) // s + 1 for v in x.shape[2:]), dtype=torch.float32, device=x.device)
                if s == 2 and x_nrm is not None and not self._has_batchnorm() and ops.conv3d_k3s2_selected(cin, cout, d, h, w, s, bounded=True):
                    self._conv_s2(rc, x, x_nrm, res, self._stats_buf(n * cout * ops.conv3d_k3s2_stat_tiles(d, h, w) * 3, x.device))
                else:
                    ops.conv3d_k3_strided(x, x_nrm, self._packed_weight(rc, 0), rc.bias, res, s)
            else:
                res = torch.empty((n, cout) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
                ops.conv1x1(x, x_nrm, rc.weight.view(cout, -1), rc.bias, res)
            res_nrm = None
        cx, cn = x, x_nrm
        for unit in ru.conv:
            cx, cn = self._conv_unit(unit, cx, cn)
        return ops.add_act(cx, cn, res, res_nrm, 1.0, dst, dst_nrm)

    def _down(self, mod, x, dst, x_nrm=None, dst_nrm=None):
        if isinstance(mod, _ResidualUnit):
            return self._residual_unit(mod, x, x_nrm, dst, dst_nrm)
        c, cn = self._conv_unit(mod, x, x_nrm)
        return ops.add_act(c, cn, None, None, 1.0, dst, dst_nrm)          # materialise norm + PReLU

    def _up(self, mod, x, dst):
        if isinstance(mod, nn.Sequential):
            c, cn = self._conv_unit(mod[0], x, None)
            return self._residual_unit(mod[1], c, cn, dst)
        c, cn = self._conv_unit(mod, x, None)
        if cn is None:
            dst.copy_(c)
            return dst
        return ops.add_act(c, cn, None, None, 1.0, dst)

    @staticmethod
    def _out_channels(mod) -> int:
        if isinstance(mod, _ResidualUnit):
            return mod.conv[0].conv.weight.shape[0]
        if isinstance(mod, nn.Sequential):
            return mod[0].conv.weight.shape[1]
        return mod.conv.weight.shape[1] if mod.is_transposed else mod.conv.weight.shape[0]

    def _block(self, seq: nn.Sequential, x, dst, x_nrm=None):
        """x_nrm: identity records of the plain tensor x with its magnitude bounds (None: unknown -- the network's input), what the split-precision kernels scale by"""
        down, skip, up = seq[0], seq[1], seq[2]
        s = down.strides
        n = x.shape[0]
        sp = tuple((v - 1) // s + 1 for v in x.shape[2:])
        cd = self._out_channels(down)
        sub = skip.submodule
        cs = self._out_channels(sub[2]) if isinstance(sub, nn.Sequential) and isinstance(sub[1], _SkipConnection) else self._out_channels(sub)
        cat = torch.empty((n, cd + cs) + sp, dtype=torch.float32, device=x.device)      # SkipConnection: cat([x, sub(x)], 1)
        # the down path's joins leave max |value| in identity records of their results: the next level's strided convolutions run on the matrix cores with them
        d_nrm = None if self._has_batchnorm() else ops.nrm_identity(torch.empty((n, cd, 4), dtype=torch.float32, device=x.device))
        d = self._down(down, x, cat[:, :cd], x_nrm, d_nrm)
        if isinstance(sub, nn.Sequential) and isinstance(sub[1], _SkipConnection):
            self._block(sub, d, cat[:, cd:], d_nrm)
        else:
            self._down(sub, d, cat[:, cd:], d_nrm)
        return self._up(up, cat, dst)

    # ---- forward -----------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and x.requires_grad:
            raise NotImplementedError("monai_amd.UNet: gradients w.r.t. the input are not on the (inference-only) HIP path")
        out = torch.empty((x.shape[0], self.out_channels) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
        return self.forward_into(x, out)

    @torch.no_grad()
    def forward_into(self, x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        _lib.require_device(x, out)
        if self.training:
            raise NotImplementedError("monai_amd.UNet: training mode (autograd) is not on the HIP path -- the engine is inference-only; with MONAI installed the call falls through to the reference module, which shares these parameters")
        total = 1
        for s in self.strides[: len(self.channels) - 1]:
            total *= s
        if x.dim() != 5 or x.shape[1] != self.in_channels or any(int(v) % total for v in x.shape[2:]):
            raise NotImplementedError(f"monai_amd.UNet: input (B,{self.in_channels},D,H,W) with edges divisible by {total} expected, got {tuple(x.shape)}")
        self._block(self.model, x.contiguous(), out)
        return out


Unet = UNet
