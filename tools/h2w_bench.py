"""A/B on the MI355X: the direct split-precision convolution (conv3d_h2.h, MH_CFG_H2) against the in-plane Winograd split-precision one (conv3d_wino_h2.h, MH_CFG_H2W)
for the 32 -> 32 layers of the headline (96^3 and 48^3, 64 windows per launch): plain, accumulating and pooling forms; times from device events, results compared with
each other and (a few windows) with an fp64 convolution.  Development aid -> one JSON document on stdout."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monai_amd import ops  # noqa: E402

dev = torch.device("cuda")
B = int(os.environ.get("KB_BATCH", "64"))
ITERS = int(os.environ.get("KB_ITERS", "10"))


def timeit(fn, iters=ITERS, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    evs[0].record()
    for i in range(iters):
        fn()
        evs[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(iters))
    return {"median_ms": round(ts[len(ts) // 2], 4), "min_ms": round(ts[0], 4), "max_ms": round(ts[-1], 4)}


def act(x, nrm):
    y = x * nrm[:, :, 0, None, None, None] + nrm[:, :, 1, None, None, None]
    return torch.where(y > 0, y, y * nrm[:, :, 2, None, None, None])


res = {"batch": B, "cases": []}
h2, h2w = ops.conv3d_k3_h2_config(), ops.conv3d_k3_h2w_config()
for e in [int(v) for v in os.environ.get("KB_EDGES", "96,48").split(",")]:
    cin = cout = 32
    g = torch.Generator(device="cpu").manual_seed(5 + e)
    x = torch.randn(B, cin, e, e, e, device=dev)
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g) / (27.0 * cin) ** 0.5).to(dev)
    bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
    nrm = torch.zeros(B, cin, 4, device=dev)
    nrm[:, :, 0] = (torch.rand(B, cin, generator=g) + 0.5).to(dev)
    nrm[:, :, 1] = (torch.randn(B, cin, generator=g) * 0.2).to(dev)
    nrm[:, :, 2] = 0.1
    for n0 in range(0, B, 8):
        nrm[n0:n0 + 8, :, 3] = act(x[n0:n0 + 8], nrm[n0:n0 + 8]).abs().amax(dim=(2, 3, 4))
    row = {"edge": e, "cin": cin, "cout": cout}
    outs = {}
    for name, cfg in (("h2", h2), ("h2w", h2w)):
        packed = ops.conv3d_k3_pack(cfg, w)
        tiles = ops.conv3d_k3_stat_tiles(cfg, e, e, e)
        stats = torch.empty(B, cout, tiles, 3, device=dev)
        out = torch.empty(B, cout, e, e, e, device=dev)
        row[name] = {"stat_tiles": tiles, "plain": timeit(lambda: ops.conv3d_k3(cfg, x, nrm, packed, bias, out, stats))}
        outs[name] = out.clone()
        nrm_o = torch.empty(B, cout, 4, device=dev)
        ops.instnorm_finalize(stats, tiles, B, cout, torch.ones(cout, device=dev), torch.zeros(cout, device=dev), 1e-5, 0.1, nrm_o)
        outs[name + "_nrm"] = nrm_o.clone()
        acc = torch.zeros_like(out)
        row[name]["acc"] = timeit(lambda: ops.conv3d_k3(cfg, x, nrm, packed, bias, acc, stats, accumulate=True))
        if ops.conv3d_k3_pool_accepts(cfg, cin, cout, e, e, e):
            pmx = torch.empty(B, cout, e // 2, e // 2, e // 2, device=dev)
            pmn = torch.empty_like(pmx)
            row[name]["pool"] = timeit(lambda: ops.conv3d_k3_pool(cfg, x, nrm, packed, bias, out, stats, pmx, pmn))
            row[name]["pool_bitwise"] = bool(torch.equal(pmx, F.max_pool3d(out, 2)) and torch.equal(pmn, -F.max_pool3d(-out, 2)) and torch.equal(out, outs[name]))
        del acc
    d = (outs["h2"] - outs["h2w"]).abs()
    row["h2_vs_h2w_max_abs"] = float(d.max())
    row["out_abs_max"] = float(outs["h2"].abs().max())
    row["alpha_rel_diff"] = float(((outs["h2_nrm"][:, :, 0] - outs["h2w_nrm"][:, :, 0]).abs() / outs["h2_nrm"][:, :, 0].abs()).max())
    ref = F.conv3d(act(x[:2].double(), nrm[:2].double()), w.double(), bias.double(), padding=1)
    row["h2_vs_f64_max_abs"] = float((outs["h2"][:2].double() - ref).abs().max())
    row["h2w_vs_f64_max_abs"] = float((outs["h2w"][:2].double() - ref).abs().max())
    row["speedup_plain"] = round(row["h2"]["plain"]["median_ms"] / row["h2w"]["plain"]["median_ms"], 3)
    res["cases"].append(row)
    del x, outs
    torch.cuda.empty_cache()
print(json.dumps(res))
