"""Oracle (test infrastructure): BasicUNet forward restated with the ATen CPU operators the
reference modules call.

Reference followed (paths relative to /root/reference):
  * ``BasicUNet.__init__/forward``  monai/networks/nets/basic_unet.py:180-279
  * ``TwoConv`` :27-58, ``Down`` :61-89, ``UpCat`` :92-175 (replicate pad of odd edges :163-170)
  * ``Convolution`` = Conv3d(k3, pad 1, bias) -> ADN("NDA")  monai/networks/blocks/convolutions.py:98-171
  * ``ADN`` = InstanceNorm3d(affine, eps 1e-5) -> Dropout(0) -> LeakyReLU(0.1)  blocks/acti_norm.py:69-101
  * ``UpSample(mode="deconv")`` = ConvTranspose3d(k2, s2)  blocks/upsample.py:102-116

State-dict keys are the reference's, so a reference checkpoint feeds this directly.
"""

from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

EPS = 1e-5


def _conv_block(x, sd, prefix, slope):
    x = F.conv3d(x, sd[prefix + ".conv.weight"], sd.get(prefix + ".conv.bias"), stride=1, padding=1)
    x = F.instance_norm(x, weight=sd[prefix + ".adn.N.weight"], bias=sd[prefix + ".adn.N.bias"], eps=EPS)
    return F.leaky_relu(x, slope)


def _two_conv(x, sd, prefix, slope):
    return _conv_block(_conv_block(x, sd, prefix + ".conv_0", slope), sd, prefix + ".conv_1", slope)


def _down(x, sd, prefix, slope):
    return _two_conv(F.max_pool3d(x, kernel_size=2), sd, prefix + ".convs", slope)


def _upcat(x, x_e, sd, prefix, slope):
    x0 = F.conv_transpose3d(x, sd[prefix + ".upsample.deconv.weight"], sd.get(prefix + ".upsample.deconv.bias"), stride=2)
    pad = [0] * 6
    for i in range(3):
        if x_e.shape[-i - 1] != x0.shape[-i - 1]:
            pad[i * 2 + 1] = 1
    x0 = F.pad(x0, pad, "replicate")
    return _two_conv(torch.cat([x_e, x0], dim=1), sd, prefix + ".convs", slope)


def basic_unet_forward(sd, x: torch.Tensor, negative_slope: float = 0.1) -> torch.Tensor:
    """Forward of the default BasicUNet (3-D, instance norm affine, LeakyReLU, deconv upsampling)."""
    x0 = _two_conv(x, sd, "conv_0", negative_slope)
    x1 = _down(x0, sd, "down_1", negative_slope)
    x2 = _down(x1, sd, "down_2", negative_slope)
    x3 = _down(x2, sd, "down_3", negative_slope)
    x4 = _down(x3, sd, "down_4", negative_slope)
    u4 = _upcat(x4, x3, sd, "upcat_4", negative_slope)
    u3 = _upcat(u4, x2, sd, "upcat_3", negative_slope)
    u2 = _upcat(u3, x1, sd, "upcat_2", negative_slope)
    u1 = _upcat(u2, x0, sd, "upcat_1", negative_slope)
    return F.conv3d(u1, sd["final_conv.weight"], sd.get("final_conv.bias"))


def make_basic_unet_state(in_channels=1, out_channels=2, features=(32, 32, 64, 128, 256, 32)):
    """Default-initialised parameters, drawn in the reference's construction order so that the same
    ``torch.manual_seed`` gives the same weights as ``monai.networks.nets.BasicUNet`` (checked against
    the golden checksum in tests/golden).  Only Conv3d / ConvTranspose3d draw random numbers."""
    f = tuple(features)
    sd = OrderedDict()

    def conv(prefix, cin, cout):
        m = nn.Conv3d(cin, cout, 3, padding=1)
        sd[prefix + ".conv.weight"], sd[prefix + ".conv.bias"] = m.weight.detach(), m.bias.detach()
        sd[prefix + ".adn.N.weight"], sd[prefix + ".adn.N.bias"] = torch.ones(cout), torch.zeros(cout)

    def two(prefix, cin, cout):
        conv(prefix + ".conv_0", cin, cout)
        conv(prefix + ".conv_1", cout, cout)

    def upcat(prefix, cin, cat, cout, halves=True):
        up = cin // 2 if halves else cin
        m = nn.ConvTranspose3d(cin, up, 2, stride=2)
        sd[prefix + ".upsample.deconv.weight"], sd[prefix + ".upsample.deconv.bias"] = m.weight.detach(), m.bias.detach()
        two(prefix + ".convs", cat + up, cout)

    two("conv_0", in_channels, f[0])
    two("down_1.convs", f[0], f[1])
    two("down_2.convs", f[1], f[2])
    two("down_3.convs", f[2], f[3])
    two("down_4.convs", f[3], f[4])
    upcat("upcat_4", f[4], f[3], f[3])
    upcat("upcat_3", f[3], f[2], f[2])
    upcat("upcat_2", f[2], f[1], f[1])
    upcat("upcat_1", f[1], f[0], f[5], halves=False)
    m = nn.Conv3d(f[5], out_channels, 1)
    sd["final_conv.weight"], sd["final_conv.bias"] = m.weight.detach(), m.bias.detach()
    return sd
