"""Blend kernel variants on the BASELINE.json shape: 512^3 output, 96^3 windows at overlap 0.5 (1000 windows), 5 classes --
17.69 GB of logits read once + 2.68 GB written (20.38 GB per launch).  Times every variant the library
carries (window-batch size G, non-temporal on / off, the round-1 table kernel, the fused argmax epilogues) with HIP events
and checks that they all produce the same bits.  Prints one JSON document.

    python tools/blend_bench.py            (needs an MI355X: ~21 GB of HBM)
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
logits = torch.empty((nwin, K, R, R, R), device=dev)
for i in range(0, nwin, 50):
    logits[i:i + 50].normal_()
imp = compute_importance_map((R,) * 3, mode="gaussian", sigma_scale=0.125, device="cpu").to(dev)
out = torch.empty((K, E, E, E), device=dev)
nbytes = 4.0 * (logits.numel() + out.numel())
res = {"edge": E, "roi": R, "classes": K, "windows": nwin, "bytes_per_launch": nbytes, "runs": []}


def run(name, env, fn=None, nb=nbytes):
    saved = {k: os.environ.get(k) for k in ("MONAI_AMD_BLEND_G", "MONAI_AMD_BLEND_NT", "MONAI_AMD_BLEND_LEGACY")}
    for k in saved:
        os.environ.pop(k, None)
    os.environ.update(env)
    call = fn or (lambda: ops.sw_blend(logits, imp, out, starts, (R,) * 3))
    ms = timeit(call)
    res["runs"].append({"variant": name, "ms": ms, "GBps": nb / ms / 1e6, "frac_of_8TBps": nb / ms / 1e6 / 8000.0})
    for k, v in saved.items():
        os.environ.pop(k, None)
        if v is not None:
            os.environ[k] = v


run("round-1 table kernel (one window at a time)", {"MONAI_AMD_BLEND_LEGACY": "1"})
ref = out.clone()
for g in (1, 2, 4, 8):
    for nt in (0, 1):
        out.zero_()
        run(f"regular grid, G={g}, non-temporal={nt}", {"MONAI_AMD_BLEND_G": str(g), "MONAI_AMD_BLEND_NT": str(nt)})
        res["runs"][-1]["bitwise_equal_to_round1"] = bool(torch.equal(out, ref))
run("library default", {})
res["runs"][-1]["bitwise_equal_to_round1"] = bool(torch.equal(out, ref))
# padded window stride (floats between consecutive windows' logits): HBM channel spread.  Each variant twice, interleaved.
dense = K * R ** 3
padded = {}
for pad in (0, 64, 1088, 16448, ((17 * 256 - dense * 4) % (1 << 17)) // 4):
    ws = dense + pad
    flat = torch.empty(nwin * ws, device=dev)
    view = flat.as_strided(logits.shape, (ws,) + tuple(logits.stride()[1:]))
    view.copy_(logits)
    padded[pad] = (flat, view)
for rep in range(2):
    for pad, (flat, view) in padded.items():
        out.zero_()
        run(f"library default, window stride dense + {pad} floats (pass {rep})", {}, lambda: ops.sw_blend(view, imp, out, starts, (R,) * 3))
        res["runs"][-1]["bitwise_equal_to_round1"] = bool(torch.equal(out, ref))
del padded
lab_ref = ref.argmax(0)
for dt, nm in ((torch.float32, "float32"), (torch.uint8, "uint8")):
    lab = torch.empty((E, E, E), dtype=dt, device=dev)
    run(f"fused argmax epilogue -> {nm} labels", {}, lambda: ops.sw_blend_argmax(logits, imp, lab, starts, (R,) * 3, K),
        nb=4.0 * logits.numel() + lab.numel() * lab.element_size())
    res["runs"][-1]["equal_to_argmax_of_blend"] = bool(torch.equal(lab.long(), lab_ref))
# the unfused post-processing the epilogue replaces: blend (above) + channel argmax over the 2.68 GB volume
ms = timeit(lambda: ops.channel_reduce("argmax", ref))
res["unfused_argmax_pass_ms"] = ms
print(json.dumps(res, indent=1))
