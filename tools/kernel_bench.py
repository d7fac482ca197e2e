"""Per-kernel timings on the MI355X for the layer shapes of the BASELINE.json workload (BasicUNet, 96^3 windows,
B windows per launch) -> one JSON document on stdout.  Development aid: decides which kernel to optimise next."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monai_amd import ops  # noqa: E402

dev = torch.device("cuda")
B = int(os.environ.get("KB_BATCH", "25"))


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def nrm(n, c):
    t = torch.zeros(n, c, 4, device=dev)
    t[:, :, 0] = 1.1
    t[:, :, 1] = 0.1
    t[:, :, 2] = 0.1
    return t


res = {"batch": B, "conv": [], "other": []}
layers = [  # (name, cin, cout, edge)
    ("conv_0.conv_0", 1, 32, 96), ("conv_0.conv_1", 32, 32, 96), ("down_1.c0", 32, 32, 48), ("down_1.c1", 32, 32, 48),
    ("down_2.c0", 32, 64, 24), ("down_2.c1", 64, 64, 24), ("down_3.c0", 64, 128, 12), ("down_3.c1", 128, 128, 12),
    ("down_4.c0", 128, 256, 6), ("down_4.c1", 256, 256, 6), ("upcat_4.c0", 256, 128, 12), ("upcat_4.c1", 128, 128, 12),
    ("upcat_3.c0", 128, 64, 24), ("upcat_3.c1", 64, 64, 24), ("upcat_2.c0", 64, 32, 48), ("upcat_2.c1", 32, 32, 48),
    ("upcat_1.c0", 64, 32, 96), ("upcat_1.c1", 32, 32, 96),
]
from monai_amd import _lib  # noqa: E402

ncfg = _lib.lib().query("mh_conv3d_k3_num_configs")
for name, cin, cout, e in layers:
    x = torch.randn(B, cin, e, e, e, device=dev)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    bias = torch.zeros(cout, device=dev)
    out = torch.empty(B, cout, e, e, e, device=dev)
    chosen = ops.conv3d_k3_select(cin, cout, e, e, e)
    xn = nrm(B, cin) if cin > 1 else None
    fl = 2.0 * 27 * cin * cout * e ** 3 * B
    row = {"layer": name, "cin": cin, "cout": cout, "edge": e, "selected": chosen, "cfgs": {}}
    for cfg in range(0, ncfg + 1):
        if not _lib.lib().query("mh_conv3d_k3_accepts", cfg, cin, cout):
            continue
        if cfg == 0 and cin > 1:
            continue  # the direct kernel is only interesting for the first layer
        packed = ops.conv3d_k3_pack(cfg, w)
        tiles = ops.conv3d_k3_stat_tiles(cfg, e, e, e)
        stats = torch.empty(B * cout * max(tiles, 1) * 3, device=dev) if tiles else None
        try:
            ms = timeit(lambda: ops.conv3d_k3(cfg, x, xn, packed, bias, out, stats), iters=3, warm=1)
        except RuntimeError:
            continue      # a Winograd configuration that does not take these extents
        row["cfgs"][cfg] = {"ms": ms, "tflops": fl / ms / 1e9}
    res["conv"].append(row)
    del x, out

e = 96
x = torch.randn(B, 32, e, e, e, device=dev)
xn = nrm(B, 32)
t = ops.instnorm_stat_tiles(e, e, e)
st = torch.empty(B * 32 * t * 3, device=dev)
ms = timeit(lambda: ops.instnorm_stats(x, st))
res["other"].append({"kernel": "instnorm_stats 32ch@96", "ms": ms, "GBps": x.numel() * 4 / ms / 1e6})
po = torch.empty(B, 32, 48, 48, 48, device=dev)
ms = timeit(lambda: ops.maxpool2(x, xn, po))
res["other"].append({"kernel": "maxpool2 32ch@96", "ms": ms, "GBps": (x.numel() + po.numel()) * 4 / ms / 1e6})
lo = torch.empty(B, 5, e, e, e, device=dev)
w1 = torch.randn(5, 32, device=dev)
b1 = torch.zeros(5, device=dev)
ms = timeit(lambda: ops.conv1x1(x, xn, w1, b1, lo))
res["other"].append({"kernel": "conv1x1 32->5@96", "ms": ms, "GBps": (x.numel() + lo.numel()) * 4 / ms / 1e6})
xi = torch.randn(B, 32, 48, 48, 48, device=dev)
wd = torch.randn(32, 32, 2, 2, 2, device=dev) * 0.1
bd = torch.zeros(32, device=dev)
ms = timeit(lambda: ops.deconv_k2s2(xi, xn, wd, bd, x))
res["other"].append({"kernel": "deconv 32->32 48->96", "ms": ms, "GBps": (xi.numel() + x.numel()) * 4 / ms / 1e6})
del x, xi, po, lo

# sliding-window kernels at the full 512^3 / 96^3 / ov .5 configuration
from monai_amd.data.utils import compute_importance_map, window_starts  # noqa: E402

starts = window_starts((512,) * 3, (96,) * 3, (48,) * 3)
vol = torch.rand(1, 512, 512, 512, device=dev)
wb = torch.empty(B, 1, 96, 96, 96, device=dev)
ms = timeit(lambda: ops.window_extract(vol, starts, 0, B, (96,) * 3, wb))
res["other"].append({"kernel": f"window_extract {B}x96^3", "ms": ms, "GBps": 2 * wb.numel() * 4 / ms / 1e6})
logits = torch.randn(1000, 5, 96, 96, 96, device=dev)
imp = compute_importance_map((96,) * 3, "gaussian", 0.125).to(dev)
out = torch.empty(5, 512, 512, 512, device=dev)
ms = timeit(lambda: ops.sw_blend(logits, imp, out, starts, (96,) * 3), iters=5)
res["other"].append({"kernel": "sw_blend 512^3 K=5", "ms": ms, "GBps": (logits.numel() + out.numel()) * 4 / ms / 1e6})
# device copy bandwidth for calibration
a = torch.empty(1 << 30, device=dev)
bb = torch.empty(1 << 30, device=dev)
ms = timeit(lambda: bb.copy_(a))
res["other"].append({"kernel": "torch copy 4 GiB", "ms": ms, "GBps": 2 * a.numel() * 4 / ms / 1e6})
print(json.dumps(res, indent=1))
