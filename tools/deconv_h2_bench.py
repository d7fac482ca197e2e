"""ConvTranspose3d k2 s2 per decoder shape: the vector-ALU kernel (mh_deconv_k2s2_f32) against the split-precision matrix-core kernel (mh_deconv_k2s2_h2_f32) at `--windows`
windows.  One JSON line per shape: ms of each, the share of 8 TB/s the new one's reads + writes reach."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monai_amd import config, ops  # noqa: E402


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
    ap.add_argument("--layers", default="64,32,48;128,64,24;256,128,12;320,256,6;64,32,24;128,64,12;256,128,6;32,16,48")
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
        w = torch.randn((cin, cout, 2, 2, 2), device=dev) / cin ** 0.5
        o = 2 * edge
        out_a = torch.empty((n, cout, o, o, o), device=dev)
        out_b = torch.empty_like(out_a)
        rec = torch.empty((n, cout, 4), device=dev)
        config.DECONV_H2 = False
        t_old = timed(lambda: ops.deconv_k2s2(x, nrm, w, None, out_a, ops.nrm_identity(rec), bounded=True), a.reps)
        config.DECONV_H2 = True
        t_new = timed(lambda: ops.deconv_k2s2(x, nrm, w, None, out_b, ops.nrm_identity(rec), bounded=True), a.reps)
        err = (out_a - out_b).abs().max().item()
        gb = (x.numel() * (cout // (32 if cout % 32 == 0 else 16)) + out_b.numel()) * 4 / 1e9
        print(json.dumps({"layer": f"{cin}->{cout} @ {edge}^3 -> {o}^3 x {n}", "valu_ms": round(t_old, 3), "h2_ms": round(t_new, 3), "speedup": round(t_old / t_new, 2),
                          "h2_GB": round(gb, 2), "h2_frac_of_8TBs": round(gb / t_new / 8.0, 3), "max_abs_diff": err}), flush=True)
        del x, out_a, out_b


if __name__ == "__main__":
    main()
