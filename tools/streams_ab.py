"""A/B of the window-batch stream lanes (monai_amd.config.SW_STREAMS) on the headline workload, ONE process / one box: for every (streams, windows per launch)
pair the complete 512^3 inference is timed (device-synchronised wall clock, 3 steps after a warm-up) and its output compared bitwise with the first pair's.

    python tools/streams_ab.py [--size 512] [--pairs 1:64,2:64,2:32,3:32,4:16]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--roi", type=int, default=96)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--pairs", default="1:64,2:64,2:32,3:32,4:16,1:64")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import bench
    from monai_amd import config
    from monai_amd.inferers import SlidingWindowInferer

    dev = torch.device("cuda", 0)
    net = bench.build_net("basicunet", args.roi, dev)
    vol = bench.benchmark_volume(args.size).to(dev)
    inferer = SlidingWindowInferer(roi_size=(args.roi,) * 3, sw_batch_size=4, overlap=0.5, mode="gaussian", sigma_scale=0.125)
    first, rows = None, []
    for pair in args.pairs.split(","):
        st, nb = (int(v) for v in pair.split(":"))
        config.SW_STREAMS = st
        os.environ["MONAI_AMD_SW_BATCH"] = str(nb)
        out = inferer(vol, net)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = inferer(vol, net)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / args.steps
        if first is None:
            first = out.clone()
        same = bool(torch.equal(out, first))
        rows.append({"streams": st, "windows_per_launch": nb, "ms_per_step": round(ms, 2), "Mvox_s": round(args.size ** 3 / ms / 1e3, 2), "bitwise_equal_to_first": same})
        print(json.dumps(rows[-1]), flush=True)
        del out
        torch.cuda.empty_cache()
    if args.out:
        with open(args.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
