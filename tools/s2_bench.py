"""Stride-2 3x3x3 convolution per layer: the vector-ALU kernel (mh_conv3d_k3_strided_f32) against the split-precision matrix-core path (mh_conv3d_k3s2_f32 = phase-split pass
+ GEMM), DynUNet's four down-sampling layers at `--windows` windows of 96^3.  Prints one JSON line per layer: ms of each path, fp32-equivalent TFLOP/s of the new one."""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monai_amd import ops  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=64)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--layers", default="32,64,96;64,128,48;128,256,24;256,320,12;16,32,96")
    a = ap.parse_args()
    dev = "cuda:0"
    for spec in a.layers.split(";"):
        cin, cout, edge = (int(v) for v in spec.split(","))
        n = a.windows
        x = torch.randn((n, cin, edge, edge, edge), device=dev)
        nrm = torch.zeros((n, cin, 4), device=dev)
        nrm[:, :, 0] = 1.0
        nrm[:, :, 2] = 0.01
        nrm[:, :, 3] = 8.0
        w = torch.randn((cout, cin, 3, 3, 3), device=dev) / (27.0 * cin) ** 0.5
        o = edge // 2
        out_a = torch.empty((n, cout, o, o, o), device=dev)
        out_b = torch.empty_like(out_a)
        p0 = ops.conv3d_k3_pack(0, w)
        ps = ops.conv3d_k3s2_pack(w)
        tiles = ops.conv3d_k3s2_stat_tiles(edge, edge, edge)
        stats = torch.empty((n, cout, tiles, 3), device=dev)
        ws = torch.empty(ops.conv3d_k3s2_workspace_floats(n, cin, edge, edge, edge), device=dev)
        t_old = timed(lambda: ops.conv3d_k3_strided(x, nrm, p0, None, out_a, 2), a.reps)
        t_new = timed(lambda: ops.conv3d_k3s2(x, nrm, ps, None, out_b, stats, ws, fused=False), a.reps)
        t_fused = timed(lambda: ops.conv3d_k3s2(x, nrm, ps, None, out_b, stats, None, fused=True), a.reps) if cin <= 512 else None
        err = (out_a - out_b).abs().max().item()
        flops = 2.0 * 27 * cin * cout * o ** 3 * n
        print(json.dumps({"layer": f"{cin}->{cout} @ {edge}^3 -> {o}^3 x {n}", "valu_ms": round(t_old, 3), "s2_ms": round(t_new, 3), "s2_fused_ms": None if t_fused is None else round(t_fused, 3), "speedup": round(t_old / t_new, 2),
                          "s2_tflops_fp32eq": round(flops / t_new / 1e9, 1), "stat_tiles": tiles, "max_abs_diff": err}), flush=True)
        del x, out_a, out_b, ws


if __name__ == "__main__":
    main()
