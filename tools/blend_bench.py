"""Blend kernels on the BASELINE.json shape: 512^3 output, 96^3 windows at overlap 0.5 (1000 windows), 5 classes -- 17.69 GB of logits read once +
2.68 GB written (20.38 GB per launch).  Times the window-major blend (dense and padded window stride), the blend over the mosaic logits
layout (ops.LogitsMosaic: residue classes of non-overlapping windows as dense arrays), the fused argmax epilogues, and the two writers of the logits
(conv1x1 into window-major rows / conv1x1_windows into the mosaic) with HIP events; checks that every blend produces the same bits.  One JSON document.

    python tools/blend_bench.py            (needs an MI355X: ~40 GB of HBM)
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monai_amd import ops  # noqa: E402
from monai_amd.data.utils import compute_importance_map, window_starts  # noqa: E402

dev = torch.device("cuda")
E = int(os.environ.get("BB_EDGE", "512"))
R = int(os.environ.get("BB_ROI", "96"))
K = int(os.environ.get("BB_K", "5"))


def timeit(fn, iters=5, warm=1):
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


starts = window_starts((E,) * 3, (R,) * 3, (R // 2,) * 3)
nwin = len(starts[0]) * len(starts[1]) * len(starts[2])
torch.manual_seed(0)
dense = K * R ** 3
pad = ((17 * 256 - dense * 4) % (1 << 17)) // 4
imp = compute_importance_map((R,) * 3, mode="gaussian", sigma_scale=0.125, device="cpu").to(dev)
out = torch.empty((K, E, E, E), device=dev)
nbytes = 4.0 * (nwin * dense + out.numel())
res = {"edge": E, "roi": R, "classes": K, "windows": nwin, "bytes_per_launch": nbytes, "runs": []}


def run(name, fn, nb=nbytes):
    ms = timeit(fn)
    res["runs"].append({"variant": name, "ms": ms, "GBps": nb / ms / 1e6, "frac_of_8TBps": nb / ms / 1e6 / 8000.0})


flat = torch.empty(nwin * (dense + pad), device=dev)
logits = flat.as_strided((nwin, K, R, R, R), (dense + pad, R ** 3, R * R, R, 1))
for i in range(0, nwin, 50):
    logits[i:i + 50].normal_()
run("window-major, padded window stride (the inferer's buffer under window sharding / fused argmax)", lambda: ops.sw_blend(logits, imp, out, starts, (R,) * 3))
ref = out.clone()
mos = ops.LogitsMosaic(starts, (R,) * 3, K, dev)
for w in range(nwin):
    mos.window_view(w).copy_(logits[w])
from monai_amd.data.utils import importance_map_factors  # noqa: E402

fac = importance_map_factors((R,) * 3, "gaussian", 0.125)
facvec = torch.cat([fac[0], fac[1], fac[2], torch.tensor([fac[3]])]).to(dev)
for rep in range(2):
    out.zero_()
    run(f"mosaic layout (pass {rep})", lambda: ops.sw_blend_mosaic(mos, imp, out))
    res["runs"][-1]["bitwise_equal_to_window_major"] = bool(torch.equal(out, ref))
    out.zero_()
    run(f"mosaic layout, importance map re-formed from its factors (pass {rep})", lambda: ops.sw_blend_mosaic(mos, facvec, out))
    res["runs"][-1]["bitwise_equal_to_window_major"] = bool(torch.equal(out, ref))
    out.zero_()
    run(f"window-major (pass {rep})", lambda: ops.sw_blend(logits, imp, out, starts, (R,) * 3))
    res["runs"][-1]["bitwise_equal_to_window_major"] = bool(torch.equal(out, ref))
if os.environ.get("BB_QUICK"):       # knob sweeps (tools/gpu_runs/blend_ab.sh): the two layouts only
    print(json.dumps(res, indent=1))
    sys.exit(0)
lab_ref = ref.argmax(0)
for dt, nm in ((torch.float32, "float32"), (torch.uint8, "uint8")):
    lab = torch.empty((E, E, E), dtype=dt, device=dev)
    run(f"fused argmax epilogue -> {nm} labels (window-major)", lambda: ops.sw_blend_argmax(logits, imp, lab, starts, (R,) * 3, K),
        nb=4.0 * nwin * dense + lab.numel() * lab.element_size())
    res["runs"][-1]["equal_to_argmax_of_blend"] = bool(torch.equal(lab.long(), lab_ref))
res["unfused_argmax_pass_ms"] = timeit(lambda: ops.channel_reduce("argmax", ref))
del ref, out
# the writers: the final 1x1 convolution of 64 windows (32 -> K channels) into window-major rows / into the mosaic
x = torch.randn((64, 32, R, R, R), device=dev)
nrm = torch.zeros((64, 32, 4), device=dev)
nrm[:, :, 0], nrm[:, :, 2] = 1.1, 0.1
wgt, b = torch.randn((K, 32), device=dev) * 0.1, torch.zeros(K, device=dev)
wb = 4.0 * 64 * (32 + K) * R ** 3
run("writer: conv1x1 -> 64 window-major rows", lambda: ops.conv1x1(x, nrm, wgt, b, logits[128:192]), nb=wb)
run("writer: conv1x1_windows -> 64 windows of the mosaic", lambda: ops.conv1x1_windows(x, nrm, wgt, b, mos, 128), nb=wb)
print(json.dumps(res, indent=1))
