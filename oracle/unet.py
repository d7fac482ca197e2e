"""Oracle (test infrastructure): MONAI ``UNet`` forward restated with the ATen CPU operators the reference calls.

Reference followed (paths relative to /root/reference):
  * ``UNet.__init__`` recursion / layer factories   monai/networks/nets/unet.py:106-298
  * ``Convolution`` (conv -> ADN "NDA": InstanceNorm3d, Dropout(0), PReLU)   monai/networks/blocks/convolutions.py:98-171
  * ``ResidualUnit`` (sub-units + strided / 1x1 / identity residual)         monai/networks/blocks/convolutions.py:248-318
  * ``SkipConnection`` (cat([x, sub(x)], 1))                                 monai/networks/layers/simplelayers.py:103-137
3-D, kernel 3, up-kernel 3, PReLU + instance norm, no dropout -- the reference's defaults.
"""

from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


def _conv(sd, p, x, stride, transposed=False, conv_only=False):
    w, b = sd[p + ".conv.weight"], sd.get(p + ".conv.bias")
    if transposed:
        y = F.conv_transpose3d(x, w, b, stride=stride, padding=1, output_padding=stride - 1)
    else:
        y = F.conv3d(x, w, b, stride=stride, padding=1)
    if not conv_only:
        y = F.prelu(F.instance_norm(y, eps=1e-5), sd[p + ".adn.A.weight"])
    return y


def _res_unit(sd, p, x, stride, subunits, last_conv_only=False):
    if p + ".residual.weight" in sd:
        rw = sd[p + ".residual.weight"]
        res = F.conv3d(x, rw, sd.get(p + ".residual.bias"), stride=stride, padding=1 if rw.shape[-1] == 3 else 0)
    else:
        res = x
    cx, s = x, stride
    for su in range(subunits):
        cx = _conv(sd, f"{p}.conv.unit{su}", cx, s, conv_only=last_conv_only and su == subunits - 1)
        s = 1
    return cx + res


def unet_forward(sd, x, channels, strides, num_res_units=0):
    def down(p, t, stride):
        return _res_unit(sd, p, t, stride, num_res_units) if num_res_units > 0 else _conv(sd, p, t, stride)

    def up(p, t, stride, is_top):
        if num_res_units > 0:
            t = _conv(sd, p + ".0", t, stride, transposed=True)
            return _res_unit(sd, p + ".1", t, 1, 1, last_conv_only=is_top)
        return _conv(sd, p, t, stride, transposed=True, conv_only=is_top)

    def block(p, t, chans, strs, is_top):
        d = down(p + ".0", t, strs[0])
        sp = p + ".1.submodule"
        sub = block(sp, d, chans[1:], strs[1:], False) if len(chans) > 2 else down(sp, d, 1)
        return up(p + ".2", torch.cat([d, sub], dim=1), strs[0], is_top)

    return block("model", x, list(channels), list(strides), True)


def make_unet_state(in_channels, out_channels, channels, strides, num_res_units=0):
    """Default-initialised parameters drawn in the reference's construction order (sub-block first, then the down
    layer, then the up layer -- unet.py:176-192), keys as in the reference's state_dict."""
    entries = []  # (key, tensor) in construction order; re-sorted into registration order afterwards

    def conv_mod(p, cin, cout, stride, transposed=False, conv_only=False):
        m = nn.ConvTranspose3d(cin, cout, 3, stride, 1, stride - 1) if transposed else nn.Conv3d(cin, cout, 3, stride, 1)
        entries.append((p + ".conv.weight", m.weight.detach()))
        entries.append((p + ".conv.bias", m.bias.detach()))
        if not conv_only:
            entries.append((p + ".adn.A.weight", torch.full((1,), 0.25)))

    def res_unit(p, cin, cout, stride, subunits, last_conv_only=False):
        sc, ss = cin, stride
        for su in range(subunits):
            conv_mod(f"{p}.conv.unit{su}", sc, cout, ss, conv_only=last_conv_only and su == subunits - 1)
            sc, ss = cout, 1
        if stride != 1 or cin != cout:
            k = 3 if stride != 1 else 1
            m = nn.Conv3d(cin, cout, k, stride, 1 if k == 3 else 0)
            entries.append((p + ".residual.weight", m.weight.detach()))
            entries.append((p + ".residual.bias", m.bias.detach()))

    def down(p, cin, cout, stride):
        if num_res_units > 0:
            res_unit(p, cin, cout, stride, num_res_units)
        else:
            conv_mod(p, cin, cout, stride)

    def up(p, cin, cout, stride, is_top):
        if num_res_units > 0:
            conv_mod(p + ".0", cin, cout, stride, transposed=True)
            res_unit(p + ".1", cout, cout, 1, 1, last_conv_only=is_top)
        else:
            conv_mod(p, cin, cout, stride, transposed=True, conv_only=is_top)

    def block(p, inc, outc, chans, strs, is_top):
        c, s = chans[0], strs[0]
        sp = p + ".1.submodule"
        if len(chans) > 2:
            block(sp, c, c, chans[1:], strs[1:], False)
            upc = c * 2
        else:
            down(sp, c, chans[1], 1)
            upc = c + chans[1]
        down(p + ".0", inc, c, s)
        up(p + ".2", upc, outc, s, is_top)

    block("model", in_channels, out_channels, list(channels), list(strides), True)

    # registration order of nn.Sequential(down, SkipConnection(sub), up): .0 < .1 < .2 at every level
    def order(key):
        return [int(t) if t.isdigit() else t for t in key.replace("unit", "unit.").split(".")]

    def rank(key):
        parts = key.split(".")
        out = []
        for i, t in enumerate(parts):
            if t.isdigit():
                out.append((0, int(t), ""))
            elif t.startswith("unit"):
                out.append((0, int(t[4:]), ""))
            else:
                # within a module: conv before adn before residual; weight before bias
                out.append((1, {"conv": 0, "adn": 1, "residual": 2, "weight": 0, "bias": 1}.get(t, 0), t))
        return out

    _ = order
    return OrderedDict(sorted(entries, key=lambda kv: rank(kv[0])))
