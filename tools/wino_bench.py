"""Development aid: time the convolution families (direct fp32 tile, in-plane Winograd, fp16 split precision) on BasicUNet layer shapes.
MH_LIB=<path to a libmonai_amd.so variant> selects the library (default: the in-tree build)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import monai_amd._lib as L  # noqa: E402

if os.environ.get("MH_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["MH_LIB"])
from monai_amd import ops  # noqa: E402

dev = torch.device("cuda")
B = int(os.environ.get("KB_BATCH", "25"))


def timeit(fn, iters=4, warm=1):
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


wino2d = ops.conv3d_k3_num_configs()
rows = []
LAYERS = ((32, 32, 96, 7), (64, 32, 96, 7), (32, 32, 48, 10), (64, 64, 24, 12), (128, 128, 12, 13))
if os.environ.get("WB_FIRST"):
    LAYERS = LAYERS[:1]
for cin, cout, e, direct in LAYERS:
    x = torch.randn(B, cin, e, e, e, device=dev)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    bias = torch.zeros(cout, device=dev)
    out = torch.empty(B, cout, e, e, e, device=dev)
    xn = torch.zeros(B, cin, 4, device=dev)
    xn[:, :, 0] = 1.1
    xn[:, :, 1] = 0.1
    xn[:, :, 2] = 0.1
    xn[:, :, 3] = 8.0          # magnitude bound of the activated randn input (what the split-precision kernel scales by)
    fl = 2.0 * 27 * cin * cout * e ** 3 * B
    row = {"cin": cin, "cout": cout, "edge": e}
    h2 = ops.conv3d_k3_h2_config()
    for name, cfg in (("direct", direct), ("wino2d", wino2d), ("h2", h2)):
        if name == "h2" and os.environ.get("WB_SKIP_H2"):
            continue
        if not ops.conv3d_k3_accepts(cfg, cin, cout):
            continue
        if name == "direct" and os.environ.get("WB_SKIP_DIRECT"):
            continue
        if name == "wino2d" and (e % 8 or e < 16):
            continue
        packed = ops.conv3d_k3_pack(cfg, w)
        tiles = ops.conv3d_k3_stat_tiles(cfg, e, e, e)
        stats = torch.empty(B * cout * tiles * 3, device=dev)
        ms = timeit(lambda: ops.conv3d_k3(cfg, x, xn, packed, bias, out, stats))
        row[name] = {"ms": round(ms, 3), "tflops_direct_equiv": round(fl / ms / 1e9, 1)}
    rows.append(row)
    del x, out
print(json.dumps({"lib": os.environ.get("MH_LIB", "in-tree"), "rows": rows}))
