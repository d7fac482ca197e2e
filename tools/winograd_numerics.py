"""Numerics study (CPU, build container): BasicUNet window forward with every 3x3x3 conv evaluated by Winograd
F(2x2x2, 3x3x3) in fp32 versus the direct fp32 oracle and an fp64 run -- decides whether a Winograd MFMA kernel can
stay inside the 1e-4 logits bar."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import basic_unet as ob  # noqa: E402

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def winograd_conv3d(x, w, b):
    n, cin, d, h, wd = x.shape
    cout = w.shape[0]
    assert d % 2 == 0 and h % 2 == 0 and wd % 2 == 0
    dt = x.dtype
    bt, g, at = BT.to(dt), G.to(dt), AT.to(dt)
    u = torch.einsum("az,by,cx,oizyx->abcio", g, g, g, w)                       # [4,4,4,cin,cout]
    xp = F.pad(x, (1, 1, 1, 1, 1, 1))
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2).unfold(4, 4, 2)                     # [n,cin,tz,ty,tx,4,4,4]
    out = torch.empty(n, cout, d, h, wd, dtype=dt)
    for zs in range(0, t.shape[2], 4):                                          # slabs of tiles to bound memory
        ts = t[:, :, zs:zs + 4]
        v = torch.einsum("az,by,cx,nipqrzyx->abcnpqri", bt, bt, bt, ts)         # [4,4,4,n,tz,ty,tx,cin]
        m = torch.matmul(v.reshape(64, -1, cin), u.reshape(64, cin, cout)).reshape(4, 4, 4, n, *ts.shape[2:5], cout)
        y = torch.einsum("ea,fb,gc,abcnpqro->nopeqfrg", at, at, at, m)          # [n,cout,tz,2,ty,2,tx,2]
        out[:, :, 2 * zs:2 * zs + 2 * ts.shape[2]] = y.reshape(n, cout, 2 * ts.shape[2], h, wd)
    return out + b.to(dt).view(1, -1, 1, 1, 1)


def main():
    torch.manual_seed(1)
    sd = ob.make_basic_unet_state(1, 5)
    torch.manual_seed(0)
    x = torch.rand(1, 1, int(os.environ.get("EDGE", "96")), 96, 96)
    with torch.no_grad():
        ref32 = ob.basic_unet_forward(sd, x)
        sd64 = {k: v.double() for k, v in sd.items()}
        ref64 = ob.basic_unet_forward(sd64, x.double())
        real = F.conv3d

        def patched(inp, w, b=None, stride=1, padding=0, *a, **k):
            if w.shape[-1] == 3 and stride == 1 and padding == 1:
                return winograd_conv3d(inp, w, b)
            return real(inp, w, b, stride, padding, *a, **k)

        F.conv3d = patched
        try:
            win32 = ob.basic_unet_forward(sd, x)
        finally:
            F.conv3d = real
    print("direct fp32 vs fp64   max abs", (ref32.double() - ref64).abs().max().item())
    print("winograd fp32 vs fp64 max abs", (win32.double() - ref64).abs().max().item())
    print("winograd fp32 vs direct fp32 (the parity bar, 1e-4)", (win32 - ref32).abs().max().item())
    print("argmax mismatches winograd vs direct", (win32.argmax(1) != ref32.argmax(1)).sum().item(), "of", ref32[:, 0].numel())
    print("logit scale", ref32.abs().max().item())


if __name__ == "__main__":
    main()
