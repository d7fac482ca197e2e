"""Oracle (test infrastructure): UNETR forward restated with the ATen CPU operators the reference modules call.

Reference followed (paths relative to /root/reference):
  * ``UNETR.__init__/forward/proj_feat``     monai/networks/nets/unetr.py:30-213
  * ``ViT.forward``                          monai/networks/nets/vit.py:121-134
  * ``PatchEmbeddingBlock`` (conv proj, learnable pos-emb, trunc_normal init)  monai/networks/blocks/patchembedding.py:92-139
  * ``TransformerBlock.forward``             monai/networks/blocks/transformerblock.py:93-101
  * ``SABlock.forward`` (einsum attention)   monai/networks/blocks/selfattention.py:156-218
  * ``MLPBlock.forward``                     monai/networks/blocks/mlp.py:75-80
  * ``UnetrBasicBlock / UnetrPrUpBlock / UnetrUpBlock``  monai/networks/blocks/unetr_block.py:22-259
  * ``UnetResBlock / UnetOutBlock / get_conv_layer``     monai/networks/blocks/dynunet_block.py:25-328
"""

from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


def _res_block(x, sd, p):
    out = F.conv3d(x, sd[p + ".conv1.conv.weight"], None, padding=1)
    out = F.leaky_relu(F.instance_norm(out, eps=1e-5), 0.01)
    out = F.instance_norm(F.conv3d(out, sd[p + ".conv2.conv.weight"], None, padding=1), eps=1e-5)
    res = x
    if p + ".conv3.conv.weight" in sd:
        res = F.instance_norm(F.conv3d(x, sd[p + ".conv3.conv.weight"], None), eps=1e-5)
    out = out + res
    return F.leaky_relu(out, 0.01)


def _attention(x, sd, p, heads):
    b, s, h = x.shape
    d = h // heads
    qkv = F.linear(x, sd[p + ".qkv.weight"], sd.get(p + ".qkv.bias"))
    t = qkv.reshape(b, s, 3, heads, d).permute(2, 0, 3, 1, 4)          # "b h (qkv l d) -> qkv b l h d"
    q, k, v = t[0], t[1], t[2]
    att = (torch.einsum("blxd,blyd->blxy", q, k) * (d ** -0.5)).softmax(dim=-1)
    y = torch.einsum("bhxy,bhyd->bhxd", att, v).permute(0, 2, 1, 3).reshape(b, s, h)   # "b l h d -> b h (l d)"
    return F.linear(y, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def vit_forward(sd, x, heads, num_layers=12):
    w = sd["vit.patch_embedding.patch_embeddings.weight"]
    t = F.conv3d(x, w, sd["vit.patch_embedding.patch_embeddings.bias"], stride=w.shape[2:])
    t = t.flatten(2).transpose(-1, -2) + sd["vit.patch_embedding.position_embeddings"]
    hidden = []
    for i in range(num_layers):
        p = f"vit.blocks.{i}"
        n = t.shape[-1]
        t = t + _attention(F.layer_norm(t, (n,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5), sd, p + ".attn", heads)
        h = F.layer_norm(t, (n,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
        h = F.linear(F.gelu(F.linear(h, sd[p + ".mlp.linear1.weight"], sd[p + ".mlp.linear1.bias"])), sd[p + ".mlp.linear2.weight"], sd[p + ".mlp.linear2.bias"])
        t = t + h
        hidden.append(t)
    n = t.shape[-1]
    return F.layer_norm(t, (n,), sd["vit.norm.weight"], sd["vit.norm.bias"], 1e-5), hidden


def unetr_forward(sd, x_in, heads=12):
    hidden_size = sd["vit.norm.weight"].shape[0]
    feat = tuple(s // 16 for s in x_in.shape[2:])

    def proj(t):
        return t.view(t.size(0), *feat, hidden_size).permute(0, 4, 1, 2, 3).contiguous()

    def tconv(t, key):
        return F.conv_transpose3d(t, sd[key], None, stride=2)

    x, hs = vit_forward(sd, x_in, heads)
    enc1 = _res_block(x_in, sd, "encoder1.layer")

    def prup(t, p, layers):
        t = tconv(t, p + ".transp_conv_init.conv.weight")
        for i in range(layers):
            t = _res_block(tconv(t, f"{p}.blocks.{i}.0.conv.weight"), sd, f"{p}.blocks.{i}.1")
        return t

    enc2 = prup(proj(hs[3]), "encoder2", 2)
    enc3 = prup(proj(hs[6]), "encoder3", 1)
    enc4 = prup(proj(hs[9]), "encoder4", 0)

    def up(inp, skip, p):
        return _res_block(torch.cat((tconv(inp, p + ".transp_conv.conv.weight"), skip), dim=1), sd, p + ".conv_block")

    dec3 = up(proj(x), enc4, "decoder5")
    dec2 = up(dec3, enc3, "decoder4")
    dec1 = up(dec2, enc2, "decoder3")
    out = up(dec1, enc1, "decoder2")
    return F.conv3d(out, sd["out.conv.conv.weight"], sd["out.conv.conv.bias"])


def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
    """Truncated-normal fill by inverse-CDF sampling (the algorithm of monai/networks/layers/weight_init.py:20-45)."""
    cdf = lambda v: (1.0 + math.erf(v / math.sqrt(2.0))) / 2.0  # noqa: E731
    lo, hi = cdf((a - mean) / std), cdf((b - mean) / std)
    with torch.no_grad():
        t.uniform_(2 * lo - 1, 2 * hi - 1).erfinv_().mul_(std * math.sqrt(2.0)).add_(mean).clamp_(min=a, max=b)
    return t


def make_unetr_state(in_channels=1, out_channels=5, img_size=(96, 96, 96), feature_size=16, hidden_size=768, mlp_dim=3072, num_layers=12):
    """Default-initialised parameters drawn in the reference's construction order (same seed -> same weights; checked
    against the golden checksum)."""
    sd = OrderedDict()
    fs = feature_size
    n_patches = 1
    for s in img_size:
        n_patches *= s // 16

    def put(prefix, m):
        for k, v in m.state_dict().items():
            sd[prefix + "." + k] = v.detach()

    put("vit.patch_embedding.patch_embeddings", nn.Conv3d(in_channels, hidden_size, 16, 16))
    pos = trunc_normal_(torch.zeros(1, n_patches, hidden_size), mean=0.0, std=0.02, a=-2.0, b=2.0)
    keys = list(sd.keys())
    sd["vit.patch_embedding.position_embeddings"] = pos
    for k in keys:  # the reference registers position_embeddings first
        sd.move_to_end(k)
    for i in range(num_layers):
        p = f"vit.blocks.{i}"
        put(p + ".mlp.linear1", nn.Linear(hidden_size, mlp_dim))
        put(p + ".mlp.linear2", nn.Linear(mlp_dim, hidden_size))
        put(p + ".norm1", nn.LayerNorm(hidden_size))
        put(p + ".attn.out_proj", nn.Linear(hidden_size, hidden_size))
        put(p + ".attn.qkv", nn.Linear(hidden_size, hidden_size * 3, bias=False))
        put(p + ".norm2", nn.LayerNorm(hidden_size))
        put(p + ".norm_cross_attn", nn.LayerNorm(hidden_size))
        put(p + ".cross_attn.out_proj", nn.Linear(hidden_size, hidden_size))
        for nm in ("to_q", "to_k", "to_v"):
            put(p + ".cross_attn." + nm, nn.Linear(hidden_size, hidden_size, bias=False))
    put("vit.norm", nn.LayerNorm(hidden_size))

    def conv(prefix, cin, cout, k):
        sd[prefix + ".conv.weight"] = nn.Conv3d(cin, cout, k, padding=k // 2, bias=False).weight.detach()

    def tconv(prefix, cin, cout):
        sd[prefix + ".conv.weight"] = nn.ConvTranspose3d(cin, cout, 2, stride=2, bias=False).weight.detach()

    def res(prefix, cin, cout):
        conv(prefix + ".conv1", cin, cout, 3)
        conv(prefix + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(prefix + ".conv3", cin, cout, 1)

    res("encoder1.layer", in_channels, fs)
    for name, cout, layers in (("encoder2", fs * 2, 2), ("encoder3", fs * 4, 1), ("encoder4", fs * 8, 0)):
        tconv(name + ".transp_conv_init", hidden_size, cout)
        for i in range(layers):
            tconv(f"{name}.blocks.{i}.0", cout, cout)
            res(f"{name}.blocks.{i}.1", cout, cout)
    for name, cin, cout in (("decoder5", hidden_size, fs * 8), ("decoder4", fs * 8, fs * 4), ("decoder3", fs * 4, fs * 2), ("decoder2", fs * 2, fs)):
        tconv(name + ".transp_conv", cin, cout)
        res(name + ".conv_block", cout * 2, cout)
    m = nn.Conv3d(fs, out_channels, 1)
    sd["out.conv.conv.weight"], sd["out.conv.conv.bias"] = m.weight.detach(), m.bias.detach()
    return sd
