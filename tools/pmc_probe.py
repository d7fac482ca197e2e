"""Launch the HBM-bound kernels a few times each (blend at the full 512^3 configuration, separable/general affine
resample and fused Gaussian on one 512^3 volume, the dominant convolution) so a `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE`
pass can attribute HBM traffic per kernel.

    python tools/pmc_probe.py [--only blend,mosaic,resample,gaussian,conv,upconv] [--conv-batch 64] [--conv-cfgs all|h2|auto]

bench.py runs the `mosaic,conv` sections (h2 only) under rocprofv3 itself: `roofline*.traffic` of the driver's line."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monai_amd import ops  # noqa: E402
from monai_amd.data.utils import compute_importance_map, window_starts  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--only", default="blend,mosaic,resample,gaussian,conv")
ap.add_argument("--conv-batch", type=int, default=64)
ap.add_argument("--conv-cfgs", default="all", choices=("all", "h2", "auto"))
args = ap.parse_args()
only = set(args.only.split(","))

dev = torch.device("cuda")
starts = window_starts((512,) * 3, (96,) * 3, (48,) * 3)
imp = compute_importance_map((96,) * 3, "gaussian", 0.125).to(dev)
if only & {"blend", "mosaic"}:
    out = torch.empty(5, 512, 512, 512, device=dev)
    if "blend" in only:
        logits = torch.randn(1000, 5, 96, 96, 96, device=dev)
        for _ in range(3):
            ops.sw_blend(logits, imp, out, starts, (96,) * 3)
        del logits
    if "mosaic" in only:
        mos = ops.LogitsMosaic(starts, (96,) * 3, 5, dev)          # the same logits volume in the mosaic layout (contents do not matter for the counters)
        mos.flat.normal_()
        for _ in range(3):
            ops.sw_blend_mosaic(mos, imp, out)
        del mos
    del out
if only & {"resample", "gaussian"}:
    vol = torch.rand(1, 512, 512, 512, device=dev)
    if "resample" in only:
        m = np.array([[1.25, 0, 0, 0], [0, 1.25, 0, 0], [0, 0, 0.625, 0]], dtype=np.float64)
        for f64 in (True, False):
            for _ in range(3):
                ops.affine_resample(vol, m.reshape(-1), (410, 410, 819), "bilinear", "border", False, f64)
        mr = m.copy()
        mr[0, 1] = 1e-9  # not axis aligned -> general kernel
        for f64 in (True, False):
            for _ in range(3):
                ops.affine_resample(vol, mr.reshape(-1), (410, 410, 819), "bilinear", "border", False, f64)
    if "gaussian" in only:
        from monai_amd.networks.layers import gaussian_1d  # noqa: E402

        k = gaussian_1d(1.0).numpy()
        for _ in range(3):
            ops.separable_filter3d(vol, [k, k, k])
    torch.cuda.synchronize()
    del vol
if "conv" in only:
    # the dominant conv of the headline workload: 32 -> 32 channels at 96^3, B windows per launch: algorithmic HBM bytes = input + output =
    # 2 x B x 32 x 96^3 x 4 B per launch (14.5 GB at B = 64)
    B = args.conv_batch
    x = torch.randn(B, 32, 96, 96, 96, device=dev)
    w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
    bias = torch.zeros(32, device=dev)
    y = torch.empty(B, 32, 96, 96, 96, device=dev)
    xn = torch.zeros(B, 32, 4, device=dev)
    xn[:, :, 0] = 1.1
    xn[:, :, 1] = 0.1
    xn[:, :, 2] = 0.1
    xn[:, :, 3] = 8.0          # magnitude bound of the activated input (the split-precision kernel scales by it)
    if args.conv_cfgs == "auto":        # what the selector gives the headline's 32 -> 32 layers at 96^3 (the Winograd split kernel since round 6)
        cfgs = (ops.conv3d_k3_select(32, 32, 96, 96, 96, bounded=True),)
    else:
        cfgs = (ops.conv3d_k3_h2_config(),) if args.conv_cfgs == "h2" else (ops.conv3d_k3_h2_config(), ops.conv3d_k3_num_configs(), 7)
    for cfg in cfgs:
        packed = ops.conv3d_k3_pack(cfg, w)
        tiles = ops.conv3d_k3_stat_tiles(cfg, 96, 96, 96)
        stats = torch.empty(B * 32 * tiles * 3, device=dev)
        for _ in range(3):
            ops.conv3d_k3(cfg, x, xn, packed, bias, y, stats)
if "upconv" in only:
    # UpCat's composite transposed convolution at the headline's top decoder level: 32 channels @ 48^3 -> 32 couts @ 96^3, B windows per launch (write-only form)
    B = args.conv_batch
    low = torch.randn(B, 32, 48, 48, 48, device=dev)
    ln = torch.zeros(B, 32, 4, device=dev)
    ln[:, :, 0] = 1.1
    ln[:, :, 1] = 0.1
    ln[:, :, 2] = 0.1
    ln[:, :, 3] = 8.0
    w4, table = ops.upconv_k4s2_weights(torch.randn(32, 32, 2, 2, 2, device=dev) * 0.2, torch.randn(32, device=dev) * 0.1, torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05)
    packed = ops.upconv_k4s2_pack(w4)
    y = torch.empty(B, 32, 96, 96, 96, device=dev)
    if os.environ.get("PMC_UPCONV_FORM", "write") == "rmw":      # the form the headline runs since round 6: added in place, statistics of the sum
        y.normal_()
        ustats = torch.empty(B * 32 * ops.upconv_k4s2_stat_tiles(48, 48, 48) * 3, device=dev)
        for _ in range(3):
            ops.upconv_k4s2(low, ln, packed, table, y, accumulate=True, stats=ustats)
    else:
        for _ in range(3):
            ops.upconv_k4s2(low, ln, packed, table, y, accumulate=False)
torch.cuda.synchronize()
