"""Headline benchmark (BASELINE.json): output voxels/s of 3-D sliding-window segmentation --
512^3 fp32 synthetic volume, 96^3 windows, overlap 0.5 (1000 windows), 5-class BasicUNet, gaussian blend.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one complete ``SlidingWindowInferer(...)(volume, net)`` call: window gather, the BasicUNet forward of
all 1000 windows, (N > 1: RCCL all-gather of the per-window logits,) blend + normalise.  The volume is resident in
HBM before the timed region.  N > 1 shards the windows of the SAME volume over the ranks (strong scaling, config 2
of BASELINE.json).  Rank 0 prints ONE JSON line; `roofline` is measured live with HIP events around the dominant
kernel's launches inside the timed region, `cpu_baseline` times the CPU oracle (a port of the reference path:
torch-CPU ATen ops, bit-identical to the reference -- tests/test_oracle_golden.py) on a bounded sample.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "voxels/s sliding-window 3D seg (512^3 vol, 96^3 win, ov 0.5) at 1/8 GPU"
PEAK_FP32_TFLOPS = 157.3   # MI355X fp32 matrix/vector peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0      # HBM3E spec peak (≈6.3 TB/s achievable)


def cpu_baseline(size: int, roi: int, windows: int, vol=None, net=None):
    """The CPU oracle (a port of the reference path: the same ATen CPU operators, bit-identical to the reference --
    tests/test_oracle_golden.py) on a bounded sample of windows; value = size^3 / (num_windows * mean window time).

    `torch.set_num_threads(os.cpu_count())` (BASELINE.md's plan) oversubscribes oneDNN on a 256-thread host, so a few
    thread counts are probed with one window each and the fastest is used for the timed sample; `cores` reports it."""
    import oracle
    from oracle.sliding_window import dense_patch_starts, get_scan_interval

    torch.manual_seed(1)
    sd = oracle.make_basic_unet_state(1, 5)
    starts, _ = dense_patch_starts((size,) * 3, (roi,) * 3, get_scan_interval((size,) * 3, (roi,) * 3, (0.5,) * 3))
    torch.manual_seed(0)
    x = torch.rand(4, 1, roi, roi, roi)
    sample = None
    if vol is not None:     # time the oracle on REAL windows of the benchmark volume, so its logits can be compared below
        import itertools

        first = list(itertools.islice(itertools.product(*starts), ((windows + 3) // 4) * 4))
        sample = torch.stack([vol[0, :, a:a + roi, b:b + roi, c:c + roi] for a, b, c in first]).cpu()
        x = sample[:4]
    ncpu = os.cpu_count() or 1
    cands = sorted({c for c in (8, 16, 32, 64, 128, ncpu) if c <= ncpu})
    best, best_t = cands[0], float("inf")
    with torch.no_grad():
        for c in cands:
            torch.set_num_threads(c)
            oracle.basic_unet_forward(sd, x[:1])
            t0 = time.perf_counter()
            oracle.basic_unet_forward(sd, x[:1])
            dt = time.perf_counter() - t0
            if dt < best_t:
                best, best_t = c, dt
        torch.set_num_threads(best)
        t0 = time.perf_counter()
        done = 0
        ref = []
        while done < windows:
            xb = x if sample is None else sample[done:done + 4]
            ref.append(oracle.basic_unet_forward(sd, xb))  # sw_batch_size = 4, as in the workload
            done += 4
        dt = time.perf_counter() - t0
    parity = None
    if sample is not None and net is not None:   # the same windows through the HIP path: logits and label maps vs the oracle
        with torch.no_grad():
            got = net(sample.to(vol.device)).cpu()
        exp = torch.cat(ref)
        la, lb = got.argmax(1), exp.argmax(1)
        dice = []
        for k in range(exp.shape[1]):
            a, b = la == k, lb == k
            den = int(a.sum()) + int(b.sum())
            dice.append(1.0 if den == 0 else 2.0 * int((a & b).sum()) / den)
        top2 = exp.topk(2, dim=1).values
        margin = (top2[:, 0] - top2[:, 1])[la != lb]          # the oracle's own top-2 margin where the label maps differ
        parity = {"windows": int(exp.shape[0]), "max_abs_logit_diff": float((got - exp).abs().max()), "tolerance": 1e-4,
                  "argmax_mismatch_voxels": int((la != lb).sum()), "voxels": int(la.numel()), "min_class_dice": min(dice),
                  "max_top2_margin_at_mismatch": float(margin.max()) if margin.numel() else 0.0,
                  "note": "label maps can only differ where the reference's own top-2 logits are closer than the fp32 "
                          "reordering noise of two different conv summation orders (random-init weights: near-ties exist)"}
    nwin = len(starts[0]) * len(starts[1]) * len(starts[2])
    per_win = dt / done
    return {
        "value": size ** 3 / (nwin * per_win),
        "unit": "voxels/s",
        "cores": best,
        "kind": "port",
        "parity_vs_gpu": parity,
        "sample": f"{done} of {nwin} windows ({roi}^3, sw_batch 4) through the CPU oracle's BasicUNet on {best} of {ncpu} host threads "
                  f"(fastest of {cands}): {per_win:.3f} s/window; value = {size}^3 voxels / ({nwin} windows x that); the blend (1-2 % on CPU) is not included",
    }


def pmc_traffic(kernel_key: str):
    """HBM bytes per launch of `kernel_key` from the committed PMC passes (profiles/r01_pmc_hbm_traffic.json: rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 corrections applied -- counters cannot be collected inside this process)."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_hbm_traffic.json")) as f:
            k = json.load(f)["kernels"].get(kernel_key)
        return None if k is None else {"hbm_bytes_per_launch": k["hbm_bytes_per_launch"], "algorithmic_bytes": k["algorithmic_bytes"],
                                       "ratio": k["hbm_bytes_per_launch"] / k["algorithmic_bytes"],
                                       "measured_on": "32->32 ch, 96^3, 64 windows per launch" if "conv" in kernel_key else "the bench configuration",
                                       "source": "profiles/r01_pmc_hbm_traffic.txt"}
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=512, help="volume edge (512 = the BASELINE.json workload)")
    ap.add_argument("--roi", type=int, default=96)
    ap.add_argument("--cpu-windows", type=int, default=12, help="windows timed for cpu_baseline (0 = skip)")
    ap.add_argument("--net", default="basicunet", choices=["basicunet", "unetr", "unet", "dynunet", "segresnet"],
                    help="basicunet = the BASELINE.json metric (configs[1]); unetr = configs[3] (ViT-B/16 UNETR, MFMA attention path); unet = MONAI UNet 16..256, 2 res units (row a11); "
                         "dynunet = nnU-Net-shaped DynUNet (5 levels, 32..320 filters); segresnet = SegResNet(init_filters=16) (SURVEY 8f-4)")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist

    from monai_amd import _prof, parallel
    from monai_amd.inferers import SlidingWindowInferer
    from monai_amd.networks.nets import UNETR, BasicUNet, DynUNet, SegResNet, UNet

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        parallel.enable_window_sharding()

    # weights / volume exactly as SURVEY.md 8(d) config 1 (fallback volume: seeded uniform noise)
    torch.manual_seed(1)
    if args.net == "unetr":
        net = UNETR(in_channels=1, out_channels=5, img_size=(args.roi,) * 3).eval().to(dev)
    elif args.net == "unet":
        net = UNet(spatial_dims=3, in_channels=1, out_channels=5, channels=(16, 32, 64, 128, 256), strides=(2, 2, 2, 2), num_res_units=2).eval().to(dev)
    elif args.net == "dynunet":
        net = DynUNet(spatial_dims=3, in_channels=1, out_channels=5, kernel_size=[3] * 5, strides=[1, 2, 2, 2, 2], upsample_kernel_size=[2] * 4).eval().to(dev)
    elif args.net == "segresnet":
        net = SegResNet(spatial_dims=3, init_filters=16, in_channels=1, out_channels=5).eval().to(dev)
    else:
        net = BasicUNet(spatial_dims=3, in_channels=1, out_channels=5).eval().to(dev)
    torch.manual_seed(0)
    vol = torch.rand(1, 1, args.size, args.size, args.size).to(dev)
    inferer = SlidingWindowInferer(roi_size=(args.roi,) * 3, sw_batch_size=4, overlap=0.5, mode="gaussian", sigma_scale=0.125)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    for _ in range(args.warmup):
        out = inferer(vol, net)
    sync()
    _prof.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = inferer(vol, net)
    sync()
    dt = time.perf_counter() - t0
    spans = _prof.stop()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        voxels = float(args.size) ** 3
        ms = 1e3 * dt / args.steps
        # dominant kernel = the 3x3x3 conv tile configuration with the largest share of the step (the 96^3-level convs)
        convs = {k: v for k, v in spans.items() if k.startswith("conv3d_k3/")}
        roof = None
        if convs:
            key, conv = max(convs.items(), key=lambda kv: kv[1]["ms_total"])
            tf = conv["work"] / (conv["ms_total"] * 1e-3) / 1e12
            cfg_id = int(key.split("/cfg")[1])
            from monai_amd import ops as _ops
            ncfg = _ops.conv3d_k3_num_configs()
            if cfg_id == ncfg:      # in-plane Winograd: 12 instead of 27 multiply-adds per (voxel, cin, cout)
                kname = (f"conv3d_k3_wino2d_kernel (Winograd F(2x2,3x3) in-plane + 3 direct z taps on v_mfma_f32_16x16x4_f32, "
                         f"32|64 -> 32 ch @ {args.roi}^3)")
                pipe = tf / 2.25
            elif cfg_id == ncfg - 1:
                kname, pipe = "conv3d_k3_winograd_kernel (Winograd F(2x2x2,3x3x3) on v_mfma_f32_16x16x4_f32)", tf / 3.375
            else:
                kname = f"conv3d_k3_mfma_kernel (cfg{cfg_id}: direct 3x3x3 implicit GEMM on v_mfma_f32_32x32x2_f32 @ {args.roi}^3)"
                pipe = tf
            roof = {"bound": "mfma", "achieved": tf, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_FP32_TFLOPS,
                    "traffic": None, "kernel": kname,
                    "note": "achieved = ALGORITHMIC flops of the 3x3x3 convolution (2*27*Cin*Cout per voxel) / kernel time; "
                            "mfma_pipe_frac = matrix-core flops actually issued / time / peak",
                    "mfma_pipe_frac": pipe / PEAK_FP32_TFLOPS,
                    "launches": conv["launches"], "ms_avg": conv["ms_avg"], "flops_per_launch": conv["work"] / conv["launches"],
                    "share_of_step": conv["ms_total"] / args.steps / ms}
            td = pmc_traffic("conv3d_k3_wino2d_kernel" if cfg_id == ncfg else "conv3d_k3_mfma_kernel" if cfg_id < ncfg - 1 else "")
            if td:
                roof["traffic"], roof["traffic_detail"] = td["hbm_bytes_per_launch"], td
        blend = spans.get("sw_blend")
        roof_hbm = None
        if blend:
            gbs = blend["work"] / (blend["ms_total"] * 1e-3) / 1e9
            # the practical ceiling on this box: a plain device-to-device copy of the output-sized buffer (read + write streams)
            ca = torch.empty(out.numel(), dtype=torch.float32, device=dev)
            cb = torch.empty_like(ca)
            cb.copy_(ca)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                cb.copy_(ca)
            e1.record()
            torch.cuda.synchronize()
            copy_gbs = 3 * 8.0 * ca.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del ca, cb
            roof_hbm = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                        "device_copy_GBps": copy_gbs, "frac_of_device_copy": gbs / copy_gbs,
                        "traffic": None, "kernel": "sw_blend_kernel<5,4>", "launches": blend["launches"], "ms_avg": blend["ms_avg"],
                        "bytes_per_launch": blend["work"] / blend["launches"]}
            td = pmc_traffic("sw_blend_kernel")
            if td:
                roof_hbm["traffic"], roof_hbm["traffic_detail"] = td["hbm_bytes_per_launch"], td
        conv_all = {k: {"ms_total": v["ms_total"] / args.steps, "tflops": v["work"] / (v["ms_total"] * 1e-3) / 1e12}
                    for k, v in spans.items() if k.startswith("conv3d_k3/")}
        line = {
            "metric": METRIC,
            "value": voxels / (dt / args.steps),
            "unit": "voxels/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{ {'unetr': 'UNETR ViT-B/16', 'unet': 'UNet 16-256 res2', 'basicunet': 'BasicUNet', 'dynunet': 'DynUNet 32-320 (5 levels)', 'segresnet': 'SegResNet f16'}[args.net]} 5-class (default features, seed-1 init), {args.size}^3 fp32 synthetic volume resident in HBM, "
                            f"{args.roi}^3 windows overlap 0.5 gaussian blend, sw_batch_size 4 (engine batches up to 64 windows per launch)",
                "parallelism": "1 GPU" if world == 1 else f"windows sharded over {world} GPUs, RCCL all-gather of logits before the blend",
            },
            "roofline": roof,
            "roofline_hbm": roof_hbm,
            "attention": ({"kernel": "attention_kernel<7> (fp32 MFMA, S=216, 12 heads x 64)", "ms_per_step": spans["attention"]["ms_total"] / args.steps,
                           "tflops": spans["attention"]["work"] / (spans["attention"]["ms_total"] * 1e-3) / 1e12,
                           "frac_of_fp32_mfma_peak": spans["attention"]["work"] / (spans["attention"]["ms_total"] * 1e-3) / 1e12 / PEAK_FP32_TFLOPS}
                          if "attention" in spans else None),
            "conv_ms_per_step": conv_all,
            "checksum": float(out.double().sum().item()),
        }
        if world == 1 and args.cpu_windows > 0 and args.net == "basicunet":
            line["cpu_baseline"] = cpu_baseline(args.size, args.roi, args.cpu_windows, vol, net)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
