"""ConvTranspose3d k2 s2 on the BasicUNet decoder shapes (64 windows per launch): the one-voxel-per-thread kernel against the matrix-core GEMM form
(MONAI_AMD_DECONV_IMPL=mfma).  Prints one JSON document: ms per launch, output GB/s, TFLOP/s."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monai_amd import ops  # noqa: E402

dev = torch.device("cuda")
res = {"windows": 64, "runs": []}
for cin, cout, e in ((32, 32, 48), (64, 32, 24), (128, 64, 12)):
    x = torch.randn(64, cin, e, e, e, device=dev)
    nrm = torch.tensor([1.0, 0.0, 0.01, 0.0], device=dev).repeat(64, cin, 1).contiguous()
    w = torch.randn(cin, cout, 2, 2, 2, device=dev) / cin ** 0.5
    b = torch.zeros(cout, device=dev)
    out = torch.empty(64, cout, 2 * e, 2 * e, 2 * e, device=dev)
    row = {"cin": cin, "cout": cout, "edge": e}
    ref = None
    for impl in ("scalar", "mfma"):
        os.environ["MONAI_AMD_DECONV_IMPL"] = impl
        for _ in range(2):
            ops.deconv_k2s2(x, nrm, w, b, out)
        torch.cuda.synchronize()
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            ops.deconv_k2s2(x, nrm, w, b, out)
        z.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(z) / 5
        row[impl + "_ms"] = ms
        row[impl + "_out_GBps"] = out.numel() * 4 / ms / 1e6
        row[impl + "_TFLOPs"] = 2.0 * x.numel() * cout * 8 / ms / 1e9
        if ref is None:
            ref = out.clone()
        else:
            row["max_abs_diff"] = float((out - ref).abs().max())
    res["runs"].append(row)
    del x, out, ref
os.environ.pop("MONAI_AMD_DECONV_IMPL", None)
print(json.dumps(res, indent=1))
