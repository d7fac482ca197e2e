"""nn.Linear on the MI355X at the shapes of UNETR's ViT-B/16 blocks (M = 216 tokens x 64 windows) for the three workgroup tiles (ops.linear(tile=...)):
ms per launch, fp32-equivalent TFLOP/s (2 M N K) and the fraction of the fp16 matrix peak the three piece products issue."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monai_amd import ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
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


def main():
    m = 216 * int(os.environ.get("LB_WINDOWS", "64"))
    rows = []
    for name, n, k, gelu, res in (("qkv", 2304, 768, False, False), ("out_proj + residual", 768, 768, False, True), ("mlp.linear1 + GELU", 3072, 768, True, False),
                                  ("mlp.linear2 + residual", 768, 3072, False, True)):
        x = torch.randn(m, k, device="cuda")
        w = torch.randn(n, k, device="cuda") / k ** 0.5
        b = torch.randn(n, device="cuda") * 0.1
        r = torch.randn(m, n, device="cuda") if res else None
        packed = ops.linear_pack(w)
        ref = None
        for tile in (64, 128, 256, 0):
            y = ops.linear(x, packed, n, b, r, gelu=gelu, tile=tile)
            if ref is None:
                ref = y
            ms = timeit(lambda: ops.linear(x, packed, n, b, r, gelu=gelu, tile=tile))
            tf = 2.0 * m * n * k / ms / 1e9
            rows.append({"layer": name, "M": m, "N": n, "K": k, "tile": tile or "auto", "ms": round(ms, 4), "fp32_equivalent_tflops": round(tf, 1),
                         "issued_frac_of_fp16_peak": round(3 * tf / 2500.0, 3), "bitwise_equal_to_tile_64": bool(torch.equal(y, ref))})
            print(rows[-1], flush=True)
    print(json.dumps({"rows": rows}))


if __name__ == "__main__":
    main()
