"""Oracle (test infrastructure): MONAI ``DynUNet`` and ``SegResNet`` forward passes restated with the ATen CPU operators the
reference calls, from a reference-layout ``state_dict``.

Reference followed (paths relative to /root/reference):
  * ``DynUNet.forward`` / ``DynUNetSkipLayer.forward``           monai/networks/nets/dynunet.py:32-65, 268-276
  * ``UnetBasicBlock`` / ``UnetResBlock`` / ``UnetUpBlock`` / ``UnetOutBlock``   monai/networks/blocks/dynunet_block.py:25-253
  * ``get_padding`` / ``get_output_padding``                      monai/networks/blocks/dynunet_block.py:304-328
  * ``SegResNet.encode`` / ``decode`` / ``forward``               monai/networks/nets/segresnet.py:170-198
  * ``ResBlock``                                                  monai/networks/blocks/segresnet_block.py:48-100
  * ``UpSample`` (nontrainable = ``nn.Upsample(trilinear, align_corners=False)``; deconv)   monai/networks/blocks/upsample.py:43-184
3-D, inference (deep-supervision heads and dropout do not act).
"""

from __future__ import annotations

import torch
import torch.nn.functional as F


def _inorm(sd, p, x, eps=1e-5):
    return F.instance_norm(x, weight=sd.get(p + ".weight"), bias=sd.get(p + ".bias"), eps=eps)


def _act(x, slope):
    return F.relu(x) if slope == 0.0 else F.leaky_relu(x, slope)


def _t3(v):
    return (int(v),) * 3 if isinstance(v, int) else tuple(int(a) for a in v)


def _dyn_conv(x, w, stride):
    """get_conv_layer (dynunet_block.py:256-301): padding = (kernel - stride + 1) / 2 per axis, truncated (get_padding :304-315)"""
    stride = _t3(stride)
    return F.conv3d(x, w, None, stride=stride, padding=tuple((k - s + 1) // 2 for k, s in zip(w.shape[2:], stride)))


def _dyn_block(sd, p, x, stride, slope, res):
    """UnetBasicBlock (dynunet_block.py:155-166) / UnetResBlock (:96-111)"""
    out = _dyn_conv(x, sd[p + ".conv1.conv.weight"], stride)
    out = _act(_inorm(sd, p + ".norm1", out), slope)
    out = _dyn_conv(out, sd[p + ".conv2.conv.weight"], 1)
    out = _inorm(sd, p + ".norm2", out)
    if not res:
        return _act(out, slope)
    residual = x
    if p + ".conv3.conv.weight" in sd:
        residual = _inorm(sd, p + ".norm3", _dyn_conv(x, sd[p + ".conv3.conv.weight"], stride))
    return _act(out + residual, slope)


def dynunet_forward(sd, x, strides, slope=0.01, res_block=False):
    """``strides``: one int or (z, y, x) triple per level (``strides[0]`` the input block's, ``strides[-1]`` the bottleneck's); upsample
    kernels = strides[1:]"""
    n_down = len(strides) - 2
    downs = ["input_block"] + [f"downsamples.{i}" for i in range(n_down)]
    ups = [f"upsamples.{n_down - i}" for i in range(n_down + 1)]          # self.upsamples[::-1]

    def level(i, t):
        d = _dyn_block(sd, downs[i], t, strides[i], slope, res_block)
        nxt = level(i + 1, d) if i + 1 < len(downs) else _dyn_block(sd, "bottleneck", d, strides[-1], slope, res_block)
        up = ups[i]
        u = F.conv_transpose3d(nxt, sd[up + ".transp_conv.conv.weight"], sd.get(up + ".transp_conv.conv.bias"), stride=_t3(strides[i + 1]))
        return _dyn_block(sd, up + ".conv_block", torch.cat((u, d), dim=1), 1, slope, False)      # dynunet_block.py:223-229

    out = level(0, x)
    return F.conv3d(out, sd["output_block.conv.conv.weight"], sd["output_block.conv.conv.bias"])


def _seg_norm(sd, p, x, groups):
    if groups:
        return F.group_norm(x, groups, sd.get(p + ".weight"), sd.get(p + ".bias"), 1e-5)
    return _inorm(sd, p, x)


def _seg_res(sd, p, x, groups, slope):
    """segresnet_block.py:85-100"""
    identity = x
    y = F.conv3d(_act(_seg_norm(sd, p + ".norm1", x, groups), slope), sd[p + ".conv1.conv.weight"], None, padding=1)
    y = F.conv3d(_act(_seg_norm(sd, p + ".norm2", y, groups), slope), sd[p + ".conv2.conv.weight"], None, padding=1)
    y += identity
    return y


def segresnet_forward(sd, x, blocks_down=(1, 2, 2, 4), blocks_up=(1, 1, 1), groups=8, slope=0.0, upsample_mode="nontrainable", use_conv_final=True):
    """``groups``: GroupNorm groups, or 0 for instance norm"""
    x = F.conv3d(x, sd["convInit.conv.weight"], None, padding=1)
    down_x = []
    for i, nb in enumerate(blocks_down):
        if i > 0:
            x = F.conv3d(x, sd[f"down_layers.{i}.0.conv.weight"], None, stride=2, padding=1)
        for j in range(nb):
            x = _seg_res(sd, f"down_layers.{i}.{j + 1}", x, groups, slope)
        down_x.append(x)
    down_x.reverse()
    for i, nb in enumerate(blocks_up):
        x = F.conv3d(x, sd[f"up_samples.{i}.0.conv.weight"], None)
        if upsample_mode == "deconv":
            x = F.conv_transpose3d(x, sd[f"up_samples.{i}.1.deconv.weight"], sd[f"up_samples.{i}.1.deconv.bias"], stride=2)
        else:
            x = F.interpolate(x, scale_factor=2.0, mode="trilinear", align_corners=False)
        x = x + down_x[i + 1]
        for j in range(nb):
            x = _seg_res(sd, f"up_layers.{i}.{j}", x, groups, slope)
    if use_conv_final:
        x = _act(_seg_norm(sd, "conv_final.0", x, groups), slope)
        x = F.conv3d(x, sd["conv_final.2.conv.weight"], sd["conv_final.2.conv.bias"])
    return x
