"""ConvTranspose3d k2 s2 on the BasicUNet decoder shapes (64 windows per launch), written into the second half of a concat buffer like the engine does and into a
dense tensor: ms per launch and output GB/s for the product kernel and -- with the -DMH_DEV_KNOBS library (MONAI_AMD_LIB) -- its measurement variants
(MONAI_AMD_DECONV_VAR: 1 = 8 couts per thread (round 2), 2 = two voxels per thread with 16-byte stores, 3 = the same with the cout group as the fastest workgroup index).
Prints one JSON document."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monai_amd import ops  # noqa: E402

dev = torch.device("cuda")
VARS = [int(v) for v in os.environ.get("DB_VARS", "0").split(",")]
res = {"windows": 64, "runs": []}
for cin, cout, e in ((32, 32, 48), (64, 32, 24), (128, 64, 12)):
    x = torch.randn(64, cin, e, e, e, device=dev)
    nrm = torch.tensor([1.0, 0.0, 0.01, 8.0], device=dev).repeat(64, cin, 1).contiguous()
    w = torch.randn(cin, cout, 2, 2, 2, device=dev) / cin ** 0.5
    b = torch.zeros(cout, device=dev)
    cat = torch.empty(64, 2 * cout, 2 * e, 2 * e, 2 * e, device=dev)
    dense = torch.empty(64, cout, 2 * e, 2 * e, 2 * e, device=dev)
    rec = torch.zeros(64, cout, 4, device=dev)
    row = {"cin": cin, "cout": cout, "edge": e}
    ref = None
    for var in VARS:
        os.environ["MONAI_AMD_DECONV_VAR"] = str(var)
        for name, out, orec in (("concat_half", cat[:, cout:], None), ("dense", dense, None), ("concat_half_with_bounds", cat[:, cout:], rec)):
            fn = (lambda: ops.deconv_k2s2(x, nrm, w, b, out, ops.nrm_identity(orec))) if orec is not None else (lambda: ops.deconv_k2s2(x, nrm, w, b, out))
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                fn()
            z.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(z) / 5
            row[f"var{var}_{name}_ms"] = ms
            row[f"var{var}_{name}_out_GBps"] = out.numel() * 4 / ms / 1e6
        if ref is None:
            ref = dense.clone()
        else:
            row[f"var{var}_max_abs_diff"] = float((dense - ref).abs().max())
    res["runs"].append(row)
    del x, cat, dense, ref
print(json.dumps(res, indent=1))
