"""BasicUNet on the MI355X kernels -- drop-in for ``monai.networks.nets.BasicUNet``.

Reference: monai/networks/nets/basic_unet.py:27-279 (TwoConv / Down / UpCat / BasicUNet), whose blocks are
``Convolution`` = Conv3d(k3, p1) -> InstanceNorm3d(affine) -> Dropout(0) -> LeakyReLU(0.1)
(blocks/convolutions.py:98-171, blocks/acti_norm.py:69-101) and ``UpSample("deconv")`` = ConvTranspose3d(k2, s2)
(blocks/upsample.py:102-116).

Same constructor signature, same module tree and ``state_dict`` keys/shapes (``conv_0.conv_0.conv.weight``,
``conv_0.conv_0.adn.N.weight``, ``upcat_4.upsample.deconv.weight``, ``final_conv.weight`` ...), same parameter
initialisation order -- so reference checkpoints load unchanged and the same seed gives the same weights.
The forward pass is an inference engine over the C ABI (include/monai_amd.h):

  * every conv output is stored RAW once; its InstanceNorm+LeakyReLU is folded into a per-(n, c)
    {alpha, beta, slope} record and applied by the consumer while it loads (no normalisation pass);
  * InstanceNorm statistics come out of the producing conv's epilogue (fp32-MFMA tiles) or one reduction
    pass (direct kernel), merged in fp64;
  * skip connections are written straight into the first channels of the decoder's concat buffer and the
    transposed conv writes the remaining channels: ``torch.cat`` never runs;
  * all buffers are planned once per (batch, window shape) and reused; weights are repacked once per
    kernel configuration.

There is no CPU path: tensors must live on a ROCm device (RuntimeError otherwise).
"""

from __future__ import annotations

import threading
from collections.abc import Sequence
from typing import Optional

import torch
import torch.nn as nn

from ... import _lib, _prof, config, ops

__all__ = ["BasicUNet", "BasicUnet", "Basicunet", "basicunet"]


# --------------------------------------------------------------------------- parameter containers
# A 2-D network (what SliceInferer drives) runs as ONE PLANE of the 3-D engine: the parameters live in the reference's 2-D modules
# (Conv2d, InstanceNorm2d, ConvTranspose2d: the 2-D net's state_dict), the engine reads a [O, I, 3, 3] kernel as the centre z-slice of a 3x3x3 one,
# pools and up-samples in-plane only.  The spatial rank under construction is per-THREAD state (set by BasicUNet.__init__ while it builds its blocks): two
# networks built concurrently -- data-loader workers, a server -- must not see each other's.
_BUILD = threading.local()


def _nd(three, two):
    return two if getattr(_BUILD, "dims", 3) == 2 else three


def _w5(w: torch.Tensor) -> torch.Tensor:
    """a layer's weight as the engine sees it: 5-D (a 2-D layer's [O, I, kh, kw] as the one-plane [O, I, 1, kh, kw] view)"""
    return w if w.dim() == 5 else w.unsqueeze(2)


class _ADN(nn.Module):
    """ADN("NDA") parameter holder (blocks/acti_norm.py:69-101): ``N`` = InstanceNorm (affine or not) / BatchNorm (evaluated with its running statistics) /
    GroupNorm; ``A`` exists as a module only when the activation has parameters (PReLU) -- the parameter-free ones leave no trace in the state_dict"""

    def __init__(self, channels: int, norm: tuple, slope):
        super().__init__()
        kind, args = norm
        if kind == "batch":
            self.N = _nd(nn.BatchNorm3d, nn.BatchNorm2d)(channels, **args)
        elif kind == "group":
            self.N = nn.GroupNorm(num_channels=channels, **args)
        else:
            self.N = _nd(nn.InstanceNorm3d, nn.InstanceNorm2d)(channels, **args)
        if isinstance(slope, dict):                 # ("prelu", args): a learnable slope
            self.A = nn.PReLU(**slope)
            if self.A.weight.numel() != 1:
                raise NotImplementedError("monai_amd.BasicUNet: PReLU with one slope per channel is not on the HIP path yet (num_parameters=1 is)")
        else:
            self.negative_slope_ = float(slope)

    @property
    def negative_slope(self) -> float:
        if not hasattr(self, "A"):
            return self.negative_slope_
        w = self.A.weight                           # one device -> host read per parameter version, not per launch
        key = (w.data_ptr(), w._version, str(w.device))
        hit = getattr(self, "_slope_cache", None)
        if hit is None or hit[0] != key:
            hit = (key, float(w.detach().float().item()))
            object.__setattr__(self, "_slope_cache", hit)
        return hit[1]


class _Convolution(nn.Module):
    """conv + adn parameter holder (reference: blocks/convolutions.py:98-171)"""

    def __init__(self, cin: int, cout: int, bias: bool, norm: tuple, slope):
        super().__init__()
        self.conv = _nd(nn.Conv3d, nn.Conv2d)(cin, cout, kernel_size=3, stride=1, padding=1, bias=bias)
        self.adn = _ADN(cout, norm, slope)


class _TwoConv(nn.Module):
    def __init__(self, cin, cout, **kw):
        super().__init__()
        self.conv_0 = _Convolution(cin, cout, **kw)
        self.conv_1 = _Convolution(cout, cout, **kw)


class _Down(nn.Module):
    def __init__(self, cin, cout, **kw):
        super().__init__()
        self.max_pooling = _nd(nn.MaxPool3d, nn.MaxPool2d)(kernel_size=2)
        self.convs = _TwoConv(cin, cout, **kw)


class _Subpixel(nn.Module):
    """SubpixelUpsample's parameters (blocks/upsample.py:186-288, conv_block="default"): one k3 convolution to cout * 2^dims channels; pixel shuffle and pad + average
    pooling hold none"""

    def __init__(self, cin, cout):
        super().__init__()
        nd = 2 if getattr(_BUILD, "dims", 3) == 2 else 3
        self.conv_block = _nd(nn.Conv3d, nn.Conv2d)(cin, cout * 2 ** nd, kernel_size=3, stride=1, padding=1, bias=True)
        # the reference's ICNR initialisation (monai/networks/utils.py:350-367, Aitken et al. 2017): every group of 2^dims sub-voxel kernels starts as copies of ONE
        # Kaiming-normal kernel, arranged through its transpose / reshape / repeat sequence -- the same draws from the generator, so the same seed gives the same parameters
        oc2, dims = cout, [3] * nd
        k = nn.init.kaiming_normal_(torch.zeros([oc2, cin] + dims))
        k = k.transpose(0, 1).reshape(oc2, cin, -1).repeat(1, 1, 2 ** nd).reshape([cin, cout * 2 ** nd] + dims).transpose(0, 1)
        with torch.no_grad():
            self.conv_block.weight.copy_(k)


class _UpSample(nn.Module):
    """UpSample (blocks/upsample.py:43-184) in the parameter layouts BasicUNet uses: "deconv" = ConvTranspose(k2, s2); "nontrainable" = an optional
    1x1 ``preconv`` (present when the channel count changes, pre_conv="default") + parameter-free linear interpolation x2 with align_corners=True;
    "pixelshuffle" = SubpixelUpsample"""

    def __init__(self, cin, cout, mode="deconv", bias=True):
        super().__init__()
        if mode == "deconv":
            self.deconv = _nd(nn.ConvTranspose3d, nn.ConvTranspose2d)(cin, cout, kernel_size=2, stride=2, bias=bias)
        elif mode == "pixelshuffle":
            self.pixelshuffle = _Subpixel(cin, cout)
        elif cin != cout:
            self.preconv = _nd(nn.Conv3d, nn.Conv2d)(cin, cout, kernel_size=1, bias=bias)


class _UpCat(nn.Module):
    def __init__(self, cin, cat, cout, halves=True, upsample="deconv", **kw):
        super().__init__()
        up = cin // 2 if halves else cin
        self.upsample = _UpSample(cin, up, upsample)
        self.convs = _TwoConv(cat + up, cout, **kw)


def _parse_act(act):
    """-> negative slope (float), or the PReLU constructor arguments (dict)"""
    name, args = (act, {}) if isinstance(act, str) else (act[0], act[1] if len(act) > 1 else {})
    name = str(name).lower()
    if name == "leakyrelu":
        return float(args.get("negative_slope", 0.01))
    if name == "relu":
        return 0.0
    if name == "prelu":
        return {k: v for k, v in args.items() if k in ("num_parameters", "init")}
    raise NotImplementedError(f"monai_amd.BasicUNet: activation {act!r} is not on the HIP path yet (LeakyReLU / ReLU / PReLU are)")


def _parse_norm(norm):
    """-> (kind, constructor arguments): instance / batch / group"""
    name, args = (norm, {}) if isinstance(norm, str) else (norm[0], dict(norm[1]) if len(norm) > 1 else {})
    kind = str(name).lower()
    if kind == "instance":
        return kind, dict(affine=bool(args.get("affine", False)), eps=float(args.get("eps", 1e-5)))
    if kind == "batch":
        return kind, {k: v for k, v in args.items() if k in ("eps", "momentum", "affine", "track_running_stats")}
    if kind == "group":
        if "num_groups" not in args:
            raise TypeError("GroupNorm.__init__() missing 1 required positional argument: 'num_groups'")
        return kind, {k: v for k, v in args.items() if k in ("num_groups", "eps", "affine")}
    raise NotImplementedError(f"monai_amd.BasicUNet: norm {norm!r} is not on the HIP path yet (instance / batch / group are)")


# --------------------------------------------------------------------------- the module
class BasicUNet(nn.Module):
    def __init__(
        self,
        spatial_dims: int = 3,
        in_channels: int = 1,
        out_channels: int = 2,
        features: Sequence[int] = (32, 32, 64, 128, 256, 32),
        act: str | tuple = ("LeakyReLU", {"negative_slope": 0.1, "inplace": True}),
        norm: str | tuple = ("instance", {"affine": True}),
        bias: bool = True,
        dropout: float | tuple = 0.0,
        upsample: str = "deconv",
    ):
        super().__init__()
        if spatial_dims not in (2, 3):
            raise NotImplementedError("monai_amd.BasicUNet: spatial_dims 2 and 3 are on the HIP path")
        if upsample not in ("deconv", "nontrainable", "pixelshuffle"):
            raise NotImplementedError("monai_amd.BasicUNet: upsample='deconv', 'nontrainable' and 'pixelshuffle' are on the HIP path")
        # dropout: accepted and inert -- this is an inference engine (forward refuses training mode) and Dropout holds no parameters, so
        # checkpoints of nets trained with dropout load unchanged
        fea = tuple(features)
        if len(fea) != 6:
            raise ValueError(f"Sequence must have length 6, got length {len(fea)}.")  # ensure_tuple_rep
        print(f"BasicUNet features: {fea}.")  # the reference prints this too (basic_unet.py:239)
        slope = _parse_act(act)
        kw = dict(bias=bias, norm=_parse_norm(norm), slope=slope)
        self.features, self.in_channels, self.out_channels = fea, in_channels, out_channels
        self.spatial_dims, self.upsample = spatial_dims, upsample
        _BUILD.dims = spatial_dims
        try:
            self._build(in_channels, out_channels, fea, kw)
        finally:
            _BUILD.dims = 3
        self._plans: dict = {}      # (N, D, H, W, device) -> _Plan
        self._packed: dict = {}     # (layer name, cfg) -> (version key, packed weights)
        self.fused_stats = True     # take InstanceNorm statistics from the conv epilogue when the tile kernel runs

    def _build(self, in_channels, out_channels, fea, kw):
        self.conv_0 = _TwoConv(in_channels, fea[0], **kw)
        self.down_1 = _Down(fea[0], fea[1], **kw)
        self.down_2 = _Down(fea[1], fea[2], **kw)
        self.down_3 = _Down(fea[2], fea[3], **kw)
        self.down_4 = _Down(fea[3], fea[4], **kw)
        up = self.upsample
        self.upcat_4 = _UpCat(fea[4], fea[3], fea[3], upsample=up, **kw)
        self.upcat_3 = _UpCat(fea[3], fea[2], fea[2], upsample=up, **kw)
        self.upcat_2 = _UpCat(fea[2], fea[1], fea[1], upsample=up, **kw)
        self.upcat_1 = _UpCat(fea[1], fea[0], fea[5], halves=False, upsample=up, **kw)
        self.final_conv = _nd(nn.Conv3d, nn.Conv2d)(fea[5], out_channels, kernel_size=1)

    # ---- weights ---------------------------------------------------------------------------------
    def _packed_weight(self, name: str, conv: nn.Conv3d, cfg: int) -> torch.Tensor:
        w = conv.weight
        key = (w.data_ptr(), w._version, str(w.device))
        hit = self._packed.get((name, cfg))
        if hit is None or hit[0] != key:
            if w.dim() == 4:        # a 2-D kernel = the centre z-slice of a 3x3x3 one whose outer slices are zero (exact: they multiply the zero padding / nothing)
                w3 = torch.zeros(w.shape[:2] + (3, 3, 3), dtype=w.dtype, device=w.device)
                w3[:, :, 1] = w
                w = w3
            hit = (key, ops.conv3d_k3_pack(cfg, w))
            self._packed[(name, cfg)] = hit
        return hit[1]

    # ---- forward ---------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """``(B, in_channels, D, H, W)`` -> raw predictions ``(B, out_channels, D, H, W)`` (basic_unet.py:254-279)."""
        if torch.is_grad_enabled() and x.requires_grad:
            raise NotImplementedError("monai_amd.BasicUNet: gradients w.r.t. the input are not on the (inference-only) HIP path")
        _lib.require_device(x)
        if self.spatial_dims == 2:
            if x.dim() != 4 or x.shape[1] != self.in_channels:
                raise RuntimeError(f"monai_amd.BasicUNet: expected input (B,{self.in_channels},H,W), got {tuple(x.shape)}")
            out = torch.empty((x.shape[0], self.out_channels) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
            return self.forward_into(x, out)
        if x.dim() != 5 or x.shape[1] != self.in_channels:
            raise RuntimeError(f"monai_amd.BasicUNet: expected input (B,{self.in_channels},D,H,W), got {tuple(x.shape)}")
        out = torch.empty((x.shape[0], self.out_channels) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
        return self.forward_into(x, out)

    @torch.no_grad()
    def forward_into(self, x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """Forward writing the logits into `out` (e.g. a slice of the inferer's all-window logits buffer)."""
        _lib.require_device(x, out[0].flat if isinstance(out, tuple) else out)
        if self.spatial_dims == 2 and x.dim() == 4 and not isinstance(out, tuple) and out.dim() == 4:
            self.forward_into(x.unsqueeze(2), out.unsqueeze(2))          # one plane of the 3-D engine (views: no copy)
            return out
        if self.training:
            raise NotImplementedError("monai_amd.BasicUNet: training mode (autograd) is not on the HIP path -- the engine is inference-only; with MONAI installed the call falls through to the reference module, which shares these parameters")
        x = x.contiguous()
        n, _, d, h, w = x.shape
        if self.spatial_dims == 2 and d != 1:
            raise RuntimeError(f"monai_amd.BasicUNet: a 2-D network takes one plane, got {tuple(x.shape)}")
        if min((h, w) if self.spatial_dims == 2 else (d, h, w)) < 16:
            raise RuntimeError(f"monai_amd.BasicUNet: window {d}x{h}x{w} is too small for four 2x poolings")
        key = (n, d, h, w, str(x.device))
        plan = self._plans.get(key)
        if plan is None:
            if len(self._plans) >= 4:
                self._plans.clear()
            plan = self._plans[key] = _Plan(self, n, (d, h, w), x.device)
        plan.run(self, x, out)
        return out


    @torch.no_grad()
    def forward_into_windows(self, x: torch.Tensor, mosaic, w0: int) -> None:
        """Forward of a batch of sliding-window windows whose logits go straight into the inferer's mosaic logits layout (ops.LogitsMosaic): batch
        element i is window w0 + i; the final 1x1 convolution writes each window to its strided place (mh_conv1x1_windows_f32)."""
        self.forward_into(x, (mosaic, int(w0)))


BasicUnet = Basicunet = basicunet = BasicUNet


# --------------------------------------------------------------------------- buffers + schedule
class _Plan:
    """All activation / statistics buffers for one (batch, window shape), allocated once.

    Level l has spatial size dims / 2^l.  ``cat[l]`` is the decoder's concat buffer [N, f_l + up_l, ...]: the
    encoder's skip output lives in channels [0, f_l), the transposed conv writes [f_l, f_l + up_l)."""

    def __init__(self, net: "BasicUNet", n: int, dims, device):
        f = net.features
        self.n, self.dims = n, tuple(dims)
        self.planar = net.spatial_dims == 2            # the z axis (one plane) is neither pooled nor up-sampled
        self.sp = [tuple(v if (self.planar and a == 0) else v >> l for a, v in enumerate(dims)) for l in range(5)]
        up = [f[1], f[2] // 2, f[3] // 2, f[4] // 2]            # channels the deconv adds at level 0..3
        dec_out = [f[5], f[1], f[2], f[3]]                      # output channels of the decoder TwoConv at level 0..3
        e = lambda c, l: torch.empty((n, c) + self.sp[l], dtype=torch.float32, device=device)  # noqa: E731
        nz = lambda c: torch.zeros((n, c, 4), dtype=torch.float32, device=device)              # noqa: E731
        self.cat = [e(f[l] + up[l], l) for l in range(4)]
        self.cat_nrm = [nz(f[l] + up[l]) for l in range(4)]
        for l in range(4):  # channels written by the transposed conv carry no norm / activation
            self.cat_nrm[l][:, f[l]:, 0] = 1.0
            self.cat_nrm[l][:, f[l]:, 2] = 1.0
        self.tmp = [e(max(f[l], dec_out[l]) if l < 4 else f[4], l) for l in range(5)]
        self.tmp_nrm = [nz(max(f[l], dec_out[l]) if l < 4 else f[4]) for l in range(5)]
        self.pool = [None] + [e(f[l - 1], l) for l in range(1, 5)]
        self.pool_min = [None] * 5       # pooling fused into the producing convolution (csrc/kernels/conv3d_h2.h, POOL): the raw minima next to the raw maxima, allocated on first use
        # every record of this engine carries a magnitude bound (include/monai_amd.h: mh_tensor5) -- the finalize kernels write it for the normalised
        # tensors, the pooling kernel hands its input's on, the transposed convolutions fold max |value| into identity records -- so every
        # convolution but the first may run on the split-precision kernel, which scales its input by them
        self.pool_nrm = [None] + [nz(f[l - 1]) for l in range(1, 5)]
        # odd extents: the transposed conv of level l+1 yields 2 * floor(sp[l] / 2); UpCat replicate-pads the far end
        # (basic_unet.py:163-170).  Those levels deconvolve into a dense scratch tensor and pad-copy it into the concat buffer.
        self.odd = [any(v & 1 for a, v in enumerate(self.sp[l]) if not (self.planar and a == 0)) for l in range(4)]
        self.up_scratch = [torch.empty((n, up[l]) + tuple(v if (self.planar and a == 0) else 2 * v for a, v in enumerate(self.sp[l + 1])),
                                       dtype=torch.float32, device=device)
                           if self.odd[l] else None for l in range(4)]
        self.up_scratch_nrm = [nz(up[l]) if self.odd[l] else None for l in range(4)]
        self.x4, self.x4_nrm = e(f[4], 4), nz(f[4])
        self.u = [e(dec_out[l], l) for l in range(4)]
        self.u_nrm = [nz(dec_out[l]) for l in range(4)]
        self.up, self.dec_out = up, dec_out
        self.batchnorm = isinstance(net.conv_0.conv_0.adn.N, (nn.BatchNorm3d, nn.BatchNorm2d))
        self.interp = net.upsample == "nontrainable"
        self.shuffle = net.upsample == "pixelshuffle"
        # "pixelshuffle" up-sampling: the sub-pixel convolution's raw result (8 x the up channels) at the lower resolution
        self.sub = [e((4 if self.planar else 8) * up[l], l + 1) if self.shuffle else None for l in range(4)]
        # "nontrainable" up-sampling: the (optional) 1x1 pre-convolution's result at the lower resolution
        self.low = [e(up[l], l + 1) if self.interp else None for l in range(4)]
        self.stats: Optional[torch.Tensor] = None
        self.device = device

    def _stats_buf(self, floats: int) -> torch.Tensor:
        if self.stats is None or self.stats.numel() < floats:
            self.stats = torch.empty(floats, dtype=torch.float32, device=self.device)
        return self.stats

    def _bn_record(self, net, name: str, bn, slope: float, out_nrm) -> None:
        """Eval-mode BatchNorm + activation as consumer-side records {alpha = weight / sqrt(running_var + eps), beta = bias - running_mean * alpha, slope, 0}:
        a parameter fold over C values (cached per parameter version), no pass over activations, no statistics -- and no magnitude bound (0: none given)."""
        if bn.running_mean is None or bn.running_var is None:
            raise NotImplementedError("monai_amd.BasicUNet: BatchNorm without running statistics is not on the (inference) HIP path")
        parts = [bn.running_mean, bn.running_var] + ([bn.weight, bn.bias] if bn.affine else [])
        key = tuple((t.data_ptr(), t._version) for t in parts) + (slope, str(bn.running_mean.device))
        hit = net._packed.get(("bn", name))
        if hit is None or hit[0] != key:
            invstd = 1.0 / torch.sqrt(bn.running_var.float() + bn.eps)
            alpha = invstd * bn.weight.float() if bn.affine else invstd
            beta = (bn.bias.float() if bn.affine else 0.0) - bn.running_mean.float() * alpha
            hit = (key, torch.stack([alpha, beta, torch.full_like(alpha, slope), torch.zeros_like(alpha)], dim=1).contiguous())
            net._packed[("bn", name)] = hit
        out_nrm.copy_(hit[1][None].expand(out_nrm.shape[0], -1, -1))

    def _conv(self, net, name: str, block: _Convolution, x, x_nrm, out, out_nrm, bounded: bool = True, pool_level: int = 0) -> bool:
        """conv -> raw `out`; the normalisation + activation that follows it -> records `out_nrm` ({alpha, beta, slope, bound}).  `bounded`: the input's
        records carry magnitude bounds (the split-precision convolution needs them; BatchNorm folds and interpolated tensors have none).
        `pool_level` l > 0: MaxPool3d(2) of this output feeds level l -- when the kernel can, its epilogue leaves the pooled tensor in self.pool[l] (raw maxima to be read under
        `out_nrm`; returns True) and no pooling pass runs."""
        n, cout, d, h, w = out.shape
        cin = x.shape[1]
        cfg = ops.conv3d_k3_select(cin, cout, d, h, w, bounded=x_nrm is not None and bounded and not self.batchnorm)
        packed = net._packed_weight(name, block.conv, cfg)
        norm = block.adn.N
        if pool_level and block.adn.negative_slope >= 0.0 and self._poolable(net, cfg, cin, cout, d, h, w, pool_level, x_nrm):
            if self.pool_min[pool_level] is None:
                self.pool_min[pool_level] = torch.empty_like(self.pool[pool_level])
            tiles = ops.conv3d_k3_stat_tiles(cfg, d, h, w)
            stats = self._stats_buf(n * cout * tiles * 3)
            with _prof.span(f"conv3d_k3/cfg{cfg}", 2.0 * 27 * cin * cout * d * h * w * n):
                ops.conv3d_k3_pool(cfg, x, x_nrm, packed, block.conv.bias, out, stats, self.pool[pool_level], self.pool_min[pool_level])
            if isinstance(norm, nn.GroupNorm):
                ops.groupnorm_finalize(stats, tiles, n, cout, norm.num_groups, norm.weight, norm.bias, norm.eps, block.adn.negative_slope, out_nrm)
            else:
                ops.instnorm_finalize(stats, tiles, n, cout, norm.weight, norm.bias, norm.eps, block.adn.negative_slope, out_nrm)
            ops.pool_select(self.pool[pool_level], self.pool_min[pool_level], out_nrm)
            return True
        if self.batchnorm:
            with _prof.span(f"conv3d_k3/cfg{cfg}", 2.0 * 27 * cin * cout * d * h * w * n):
                ops.conv3d_k3(cfg, x, x_nrm, packed, block.conv.bias, out, None)
            self._bn_record(net, name, norm, block.adn.negative_slope, out_nrm)
            return False
        tiles = ops.conv3d_k3_stat_tiles(cfg, d, h, w) if net.fused_stats else 0
        flops = 2.0 * 27 * cin * cout * d * h * w * n
        if tiles:
            stats = self._stats_buf(n * cout * tiles * 3)
            with _prof.span(f"conv3d_k3/cfg{cfg}", flops):
                ops.conv3d_k3(cfg, x, x_nrm, packed, block.conv.bias, out, stats)
        else:
            with _prof.span(f"conv3d_k3/cfg{cfg}", flops):
                ops.conv3d_k3(cfg, x, x_nrm, packed, block.conv.bias, out, None)
            tiles = ops.instnorm_stat_tiles(d, h, w)
            stats = self._stats_buf(n * cout * tiles * 3)
            ops.instnorm_stats(out, stats)
        if isinstance(norm, nn.GroupNorm):
            ops.groupnorm_finalize(stats, tiles, n, cout, norm.num_groups, norm.weight, norm.bias, norm.eps, block.adn.negative_slope, out_nrm)
        else:
            ops.instnorm_finalize(stats, tiles, n, cout, norm.weight, norm.bias, norm.eps, block.adn.negative_slope, out_nrm)
        return False

    def _halves_cfg(self, net, l: int, cout: int, bounded: bool) -> int:
        """the configuration for `_conv_halves` at decoder level l, or -1: the concatenation's convolution is linear in its input channels, and where each 32-channel half
        alone is a shape the Winograd split-precision kernel takes (csrc/kernels/conv3d_wino_h2.h: Cin == 32) while the 64-channel whole is not, two launches of it
        (plain form + accumulating form) cost less than one of the direct kernel: 0.90 + 1.02 ms against 2.54 ms at 64 -> 32 @ 48^3 x 64 windows."""
        f = net.features
        if not config.conv_halves() or self.batchnorm or not net.fused_stats or not bounded or 2 * f[l] != int(self.cat[l].shape[1]):
            return -1
        _, _, d, h, w = self.cat[l].shape
        cfg = ops.conv3d_k3_select(f[l], cout, d, h, w, bounded=True)
        if cfg != ops.conv3d_k3_h2w_config() or ops.conv3d_k3_select(2 * f[l], cout, d, h, w, bounded=True) == cfg:
            return -1
        return cfg

    def _conv_halves(self, net, name: str, block: _Convolution, l: int, cfg: int, out, out_nrm) -> None:
        """conv(cat([x_e, x_0])) = conv[:, :f_l](x_e) + conv[:, f_l:](x_0) + b: the skip half written by the plain form, the up-sampled half added by the accumulating form together
        with the bias and the statistics of the sum.  Reference: UpCat.forward, monai/networks/nets/basic_unet.py:160-178 (torch.cat + Convolution)."""
        f = net.features
        fl = f[l]
        conv = block.conv
        n, cout, d, h, w = out.shape
        key = (conv.weight.data_ptr(), conv.weight._version, str(conv.weight.device))
        hit = net._packed.get(("halves", name, cfg))
        if hit is None or hit[0] != key:
            hit = (key, ops.conv3d_k3_pack(cfg, conv.weight[:, :fl].contiguous()), ops.conv3d_k3_pack(cfg, conv.weight[:, fl:].contiguous()))
            net._packed[("halves", name, cfg)] = hit
        flops = 2.0 * 27 * fl * cout * d * h * w * n
        with _prof.span(f"conv3d_k3/cfg{cfg}", flops):
            ops.conv3d_k3(cfg, self.cat[l][:, :fl], self.cat_nrm[l][:, :fl], hit[1], None, out, None)
        tiles = ops.conv3d_k3_stat_tiles(cfg, d, h, w)
        stats = self._stats_buf(n * cout * tiles * 3)
        with _prof.span(f"conv3d_k3/cfg{cfg}", flops):
            ops.conv3d_k3(cfg, self.cat[l][:, fl:], self.cat_nrm[l][:, fl:], hit[2], conv.bias, out, stats, accumulate=True)
        norm = block.adn.N
        if isinstance(norm, nn.GroupNorm):
            ops.groupnorm_finalize(stats, tiles, n, cout, norm.num_groups, norm.weight, norm.bias, norm.eps, block.adn.negative_slope, out_nrm)
        else:
            ops.instnorm_finalize(stats, tiles, n, cout, norm.weight, norm.bias, norm.eps, block.adn.negative_slope, out_nrm)

    def _poolable(self, net, cfg: int, cin: int, cout: int, d: int, h: int, w: int, level: int, x_nrm) -> bool:
        """the pooling epilogue (csrc/kernels/conv3d_h2.h, POOL): the split-precision kernel with 16 x 16 regions on even extents, a non-negative activation slope (the
        activation must be monotone in the raw value), statistics from the epilogue, a 3-D network"""
        if self.planar or self.batchnorm or not net.fused_stats or x_nrm is None or not config.pool_fused():
            return False
        if tuple(self.sp[level]) != (d // 2, h // 2, w // 2) or tuple(self.pool[level].shape[1:]) != (cout, d // 2, h // 2, w // 2):
            return False
        return ops.conv3d_k3_pool_accepts(cfg, cin, cout, d, h, w)

    def _interpolate(self, up: _UpSample, src, src_nrm, low, dst) -> None:
        """UpSample(mode="nontrainable", interp_mode="linear", align_corners=True) (blocks/upsample.py:118-140): the 1x1 `preconv` when the channel count
        changes, then x2 linear interpolation with source index = o (in - 1) / (out - 1) -- one launch of the affine resampler per tensor"""
        if hasattr(up, "preconv"):
            pc = up.preconv
            ops.conv1x1(src, src_nrm, pc.weight.view(pc.weight.shape[0], -1), pc.bias, low)
        else:
            ops.add_act(src, src_nrm, None, None, 1.0, low)          # materialise the deferred normalisation + activation
        n, c, d, h, w = low.shape
        osz = (d if self.planar else 2 * d, 2 * h, 2 * w)
        a = [1.0 if (self.planar or d == 1) else (d - 1) / (osz[0] - 1), (h - 1) / (osz[1] - 1), (w - 1) / (osz[2] - 1)]
        m = [a[0], 0, 0, 0, 0, a[1], 0, 0, 0, 0, a[2], 0]
        hi = ops.affine_resample(low.reshape(n * c, d, h, w), m, osz, "bilinear", "border", False, False)
        dst.copy_(hi.reshape((n, c) + osz))

    def _subpixel(self, net, name: str, conv: nn.Conv3d, src, src_nrm, sub, dst, dst_nrm) -> None:
        """SubpixelUpsample (blocks/upsample.py:274-288): the k3 convolution of the (deferred) input to 8 x (one plane: 4 x) the up channels -- raw, no normalisation follows it -- then one pass
        that shuffles the sub-voxels into place and applies the pad + average pooling; max |value| goes into the identity records of the result"""
        n, cin, d, h, w = src.shape
        cout = sub.shape[1]
        cfg = ops.conv3d_k3_select(cin, cout, d, h, w, bounded=src_nrm is not None and not self.batchnorm)
        with _prof.span(f"conv3d_k3/cfg{cfg}", 2.0 * 27 * cin * cout * d * h * w * n):
            ops.conv3d_k3(cfg, src, src_nrm, net._packed_weight(name, conv, cfg), conv.bias, sub, None)
        ops.pixelshuffle(sub, dst, 1 if self.planar else 2, True, dst_nrm)

    def _fusable(self, net, l: int, src: torch.Tensor, cout: int) -> bool:
        """UpCat level l without its up-sampled intermediate (csrc/kernels/upconv_h2.h): a k2 s2 transposed convolution feeding an instance- / group-normalised
        convolution at exactly twice the extents, shapes the composite kernel takes, the split-precision family allowed"""
        if self.interp or self.shuffle or self.planar or self.odd[l] or self.batchnorm or not net.fused_stats or not config.upcat_fused():
            return False
        if config.conv_algo() not in (config.CONV_ALGOS["auto"], config.CONV_ALGOS["h2"]):
            return False
        return ops.upconv_k4s2_accepts(int(src.shape[1]), int(cout), *self.sp[l + 1])

    def _upcat_fused(self, net, l: int, upc: "_UpCat", src, src_nrm, out, out_nrm) -> None:
        """conv_0(cat([x_e, deconv(x)])) = conv_0[:, :f_l](x_e) + [conv_0[:, f_l:] o deconv](x): the skip half on the 3x3x3 kernel (raw, bias included, no
        statistics), then the composite transposed convolution k4 s2 p1 of the LOW-resolution tensor added in place together with the statistics of the sum.
        Reference: UpCat.forward, monai/networks/nets/basic_unet.py:160-178."""
        f = net.features
        block, name = upc.convs.conv_0, f"upcat_{l + 1}.convs.conv_0"
        conv, dec = block.conv, upc.upsample.deconv
        n, cout, d, h, w = out.shape
        skip, skip_nrm = self.cat[l][:, : f[l]], self.cat_nrm[l][:, : f[l]]
        parts = [conv.weight, dec.weight] + ([dec.bias] if dec.bias is not None else [])
        key = tuple((p_.data_ptr(), p_._version) for p_ in parts) + (str(conv.weight.device),)
        cfg = ops.conv3d_k3_select(f[l], cout, d, h, w, bounded=True)
        hit = net._packed.get(("upcat", name, cfg))
        if hit is None or hit[0] != key:
            w4, table = ops.upconv_k4s2_weights(dec.weight, dec.bias, conv.weight[:, f[l]:])
            hit = (key, ops.conv3d_k3_pack(cfg, conv.weight[:, : f[l]].contiguous()), ops.upconv_k4s2_pack(w4), table)
            net._packed[("upcat", name, cfg)] = hit
        _, packed_skip, packed_up, table = hit
        up_flops = 2.0 * 8 * int(src.shape[1]) * cout * d * h * w * n
        if cfg in (ops.conv3d_k3_h2_config(), ops.conv3d_k3_h2w_config()) and config.upcat_order() == "term_first":
            # the composite term is WRITTEN, the split-precision convolution of the skip channels adds itself to it and leaves the statistics of the sum
            with _prof.span("upconv_k4s2", up_flops):
                ops.upconv_k4s2(src, src_nrm, packed_up, table, out, accumulate=False)
            tiles = ops.conv3d_k3_stat_tiles(cfg, d, h, w)
            stats = self._stats_buf(n * cout * tiles * 3)
            with _prof.span(f"conv3d_k3/cfg{cfg}", 2.0 * 27 * f[l] * cout * d * h * w * n):
                ops.conv3d_k3(cfg, skip, skip_nrm, packed_skip, conv.bias, out, stats, accumulate=True)
        else:
            # the convolution writes the skip half first (any kernel family), the composite term is added in place together with the statistics of the sum
            with _prof.span(f"conv3d_k3/cfg{cfg}", 2.0 * 27 * f[l] * cout * d * h * w * n):
                ops.conv3d_k3(cfg, skip, skip_nrm, packed_skip, conv.bias, out, None)
            tiles = ops.upconv_k4s2_stat_tiles(*self.sp[l + 1])
            stats = self._stats_buf(n * cout * tiles * 3)
            with _prof.span("upconv_k4s2", up_flops):
                ops.upconv_k4s2(src, src_nrm, packed_up, table, out, accumulate=True, stats=stats)
        norm = block.adn.N
        if isinstance(norm, nn.GroupNorm):
            ops.groupnorm_finalize(stats, tiles, n, cout, norm.num_groups, norm.weight, norm.bias, norm.eps, block.adn.negative_slope, out_nrm)
        else:
            ops.instnorm_finalize(stats, tiles, n, cout, norm.weight, norm.bias, norm.eps, block.adn.negative_slope, out_nrm)

    def run(self, net: "BasicUNet", x: torch.Tensor, logits: torch.Tensor) -> None:
        f = net.features
        downs = [None, net.down_1, net.down_2, net.down_3, net.down_4]
        ups = [net.upcat_1, net.upcat_2, net.upcat_3, net.upcat_4]

        # encoder
        t, tn = self.tmp[0][:, : f[0]], self.tmp_nrm[0][:, : f[0]]
        self._conv(net, "conv_0.conv_0", net.conv_0.conv_0, x, None, t, tn)
        pooled = self._conv(net, "conv_0.conv_1", net.conv_0.conv_1, t, tn, self.cat[0][:, : f[0]], self.cat_nrm[0][:, : f[0]], pool_level=1)
        for l in range(1, 5):
            skip, skip_nrm = self.cat[l - 1][:, : f[l - 1]], self.cat_nrm[l - 1][:, : f[l - 1]]
            if pooled:      # the producing convolution left the raw maxima: read them under the skip tensor's records (act(max raw) == max(act(raw)), bit for bit)
                p_in, p_nrm = self.pool[l], skip_nrm
            else:
                ops.maxpool2(skip, skip_nrm, self.pool[l], self.pool_nrm[l])
                p_in, p_nrm = self.pool[l], self.pool_nrm[l]
            t, tn = self.tmp[l][:, : f[l]], self.tmp_nrm[l][:, : f[l]]
            self._conv(net, f"down_{l}.convs.conv_0", downs[l].convs.conv_0, p_in, p_nrm, t, tn)
            if l < 4:
                o, on = self.cat[l][:, : f[l]], self.cat_nrm[l][:, : f[l]]
            else:
                o, on = self.x4, self.x4_nrm
            pooled = self._conv(net, f"down_{l}.convs.conv_1", downs[l].convs.conv_1, t, tn, o, on, pool_level=l + 1 if l < 4 else 0)

        # decoder
        src, src_nrm = self.x4, self.x4_nrm
        for l in range(3, -1, -1):
            upc = ups[l]
            co = self.dec_out[l]
            t, tn = self.tmp[l][:, :co], self.tmp_nrm[l][:, :co]
            if self._fusable(net, l, src, co):
                self._upcat_fused(net, l, upc, src, src_nrm, t, tn)
            else:
                dst = self.up_scratch[l] if self.odd[l] else self.cat[l][:, f[l]:]
                dst_nrm = ops.nrm_identity(self.up_scratch_nrm[l] if self.odd[l] else self.cat_nrm[l][:, f[l]:])
                if self.interp:
                    self._interpolate(upc.upsample, src, src_nrm, self.low[l], dst)
                elif self.shuffle:
                    self._subpixel(net, f"upcat_{l + 1}.upsample.pixelshuffle.conv_block", upc.upsample.pixelshuffle.conv_block, src, src_nrm, self.sub[l], dst, dst_nrm)
                elif self.planar:     # ConvTranspose2d k2 s2 = the (1, 2, 2) kernel == stride transposed conv of the anisotropic path
                    ops.deconv_ks(src, src_nrm, _w5(upc.upsample.deconv.weight).contiguous(), upc.upsample.deconv.bias, dst, (1, 2, 2), dst_nrm)
                else:
                    ops.deconv_k2s2(src, src_nrm, upc.upsample.deconv.weight, upc.upsample.deconv.bias, dst, dst_nrm, bounded=src_nrm is not None and not self.batchnorm)
                if self.odd[l]:
                    ops.pad_replicate(self.up_scratch[l], self.cat[l][:, f[l]:], self.up_scratch_nrm[l], self.cat_nrm[l][:, f[l]:])
                hcfg = self._halves_cfg(net, l, co, bounded=not self.interp)
                if hcfg >= 0:
                    self._conv_halves(net, f"upcat_{l + 1}.convs.conv_0", upc.convs.conv_0, l, hcfg, t, tn)
                else:
                    self._conv(net, f"upcat_{l + 1}.convs.conv_0", upc.convs.conv_0, self.cat[l], self.cat_nrm[l], t, tn, bounded=not self.interp)
            self._conv(net, f"upcat_{l + 1}.convs.conv_1", upc.convs.conv_1, t, tn, self.u[l], self.u_nrm[l])
            src, src_nrm = self.u[l], self.u_nrm[l]

        fc = net.final_conv
        if isinstance(logits, tuple):      # (mosaic, first window): the inferer's mosaic logits layout
            ops.conv1x1_windows(src, src_nrm, fc.weight.view(fc.weight.shape[0], -1), fc.bias, logits[0], logits[1])
        else:
            ops.conv1x1(src, src_nrm, fc.weight.view(fc.weight.shape[0], -1), fc.bias, logits)
