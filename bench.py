"""Headline benchmark (BASELINE.json): output voxels/s of 3-D sliding-window segmentation --
512^3 fp32 synthetic CT volume, 96^3 windows, overlap 0.5 (1000 windows), 5-class BasicUNet, gaussian blend.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one complete ``SlidingWindowInferer(...)(volume, net)`` call: window gather, the BasicUNet forward of
all 1000 windows, (N > 1: RCCL all-gather of the per-window logits,) blend + normalise.  The volume -- the reference's own
``create_test_image_3d`` phantom of SURVEY.md 8(d) config 1, restated bit-identically in oracle/synthetic.py -- is resident
in HBM before the timed region.  N > 1 shards the windows of the SAME volume over the ranks (strong scaling, config 2 of
BASELINE.json).  Rank 0 prints ONE JSON line:

  roofline      the dominant kernel (the split-precision 3x3x3 convolution), HIP events around its launches inside the timed
                region: `achieved` = matrix-core flops ISSUED per launch / average launch time, `frac` = achieved / the dense
                MFMA peak of the instruction it issues (a true fraction); the convolution's own flops are `algorithmic_tflops`;
  roofline_hbm  the blend: 20.38 GB of algorithmic traffic / its launch time, against the 8 TB/s spec; `traffic` of both rooflines = HBM bytes per
                launch from FETCH_SIZE / WRITE_SIZE, collected by two rocprofv3 child processes of this run (pmc_inrun; the committed builder pass if that fails);
  cpu_baseline  the CPU oracle (a port of the reference path: the same ATen CPU operators, bit-identical to the reference --
                tests/test_oracle_golden.py) running the COMPLETE inferer (windows, network, blend) on the WHOLE benchmark volume --
                all 1000 windows, the per-window network in worker processes (oracle/parallel_predict.py), the blend in the reference's
                window order -- timed on the host cores; `parity_vs_gpu` is the headline parity rule (oracle/parity.py) over every
                output voxel of the timed steps' result (134 217 728 voxels x 5 logits).  A probe batch prices the leg first: if it would
                exceed --cpu-budget-s (900 s) the largest corner sub-volume that fits is taken and the line says so;
  extra         (N = 1) fp32_exact: the same workload with the exact-fp32 convolution kernels (config CONV_ALGO "fp32") and its parity
                against the same whole-volume reference; config3: UNETR ViT-B/16 on the same volume (attention kernel's MFMA rate, parity
                of the complete inferer on a 27-window corner vs the oracle, the oracle's CPU time per window); config4: Spacing +
                GaussianSmooth on 4 x 512^3 (kernel times against 8 TB/s and against a device copy measured in the same run, parity on
                a 128^3 volume, the reference path's CPU time on one 512^3 volume with the product's full-size result against it).
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
PEAK_F16_TFLOPS = 2500.0   # MI355X dense fp16 / bf16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PF dense, without sparsity)
PEAK_HBM_GBS = 8000.0      # HBM3E spec peak (~6.3 TB/s achievable: the guide; 6.2-7.0 TB/s measured by tools/ubench/hbm_stream.hip)
NETS = {"swinunetr": "SwinUNETR f48", "unetr": "UNETR ViT-B/16", "unet": "UNet 16-256 res2", "basicunet": "BasicUNet", "dynunet": "DynUNet 32-320 (5 levels)", "segresnet": "SegResNet f16"}


def benchmark_volume(size: int) -> torch.Tensor:
    """create_test_image_3d(512, 512, 512, num_objs=40, rad_max=60, rad_min=10, noise_max=0.2,
    num_seg_classes=4, RandomState(0)) -> [1, 1, size, size, size] fp32 in [0, 1] (oracle/synthetic.py, bit-identical to the
    reference's generator; input data, generated on the host before anything is timed)."""
    from oracle import synthetic

    return torch.from_numpy(synthetic.benchmark_volume(size))[None, None]


def sub_volume_extents(size: int, roi: int, windows: int):
    """Extents of a corner sub-volume that the inferer covers with (about) `windows` windows at overlap 0.5: 2 x 2 x 3 for 12."""
    counts = [1, 1, 1]
    step = max(roi // 2, 1)
    most = max(1, (size - roi) // step + 1)
    ax = 2
    while counts[0] * counts[1] * counts[2] < windows and any(c < most for c in counts):
        if counts[ax] < most:
            counts[ax] += 1
        ax = (ax - 1) % 3
    return tuple(min(size, roi + (c - 1) * step) for c in counts)


def _host_layout():
    """(procs, threads): the oracle's per-window network runs in `procs` worker processes of `threads` ATen threads each (oracle/parallel_predict.py) -- one
    oneDNN thread group does not scale beyond ~32 threads (a 256-thread group measured SLOWER than 32 on the GPU boxes of rounds 1-3), several groups do"""
    try:
        ncpu = len(os.sched_getaffinity(0))          # the threads this process may run on (a container's cpuset), not the machine's count
    except (AttributeError, OSError):
        ncpu = os.cpu_count() or 1
    # a CFS quota (cgroup cpu.max) is what the box really gives: the round-4 GPU boxes show 256 threads and a quota of 16 CPUs -- thread groups beyond the quota are
    # throttled as a whole (profiles/r04_cpu_probe*.txt: 1 x 32 threads 1.5 windows/s, 8 x 16 threads 2.6)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    eff = min(ncpu, quota) if quota else ncpu
    env_p, env_t = os.environ.get("MONAI_AMD_BENCH_CPU_PROCS"), os.environ.get("MONAI_AMD_BENCH_CPU_THREADS")
    if quota and quota < ncpu:
        threads, procs = HOST_LAYOUT_UNDER_QUOTA(eff)
    else:
        threads = min(32, eff)
        procs = max(1, min(8, eff // threads))
    if env_t:
        threads = int(env_t)
    if env_p:
        procs = int(env_p)
    return max(1, procs), max(1, threads), ncpu


def HOST_LAYOUT_UNDER_QUOTA(cpus: int):
    """(threads, procs) under a CPU quota of `cpus`: many small workers -- measured on the round-4 boxes (profiles/r04_cpu_probe_v2.txt, quota 16 of 256 threads):
    8 x 2 threads 5.6 windows/s, 8 x 4: 5.4, 4 x 4: 4.2, 2 x 8: 2.5, 1 x 16: 1.4, 1 x 32: 1.5 (a group wider than its share of the quota is throttled as a whole)"""
    return 2, max(1, cpus // 2)


def cpu_baseline(size: int, roi: int, windows: int, vol: torch.Tensor, net, inferer, full_out, budget_s: float, more=None):
    """CPU oracle = port of the reference path (kind "port": the same ATen CPU operators in the same order, pinned bit-for-bit to the real reference by
    tests/test_oracle_golden.py -- the GPU box has no MONAI): the COMPLETE sliding-window inference -- window loop, BasicUNet, importance-weighted blend -- on the host
    cores, and the product's output on the same voxels against it (`parity_vs_gpu`, the headline rule of oracle/parity.py on the BLENDED logits).
    Default: the WHOLE benchmark volume (1000 windows, every output voxel compared).  The per-window network runs in several worker processes
    (oracle/parallel_predict.py), the blend in this process in the reference's window order.  A probe batch prices the run first: when the whole volume would
    exceed `budget_s` seconds the largest corner sub-volume that fits is taken instead (and the line says so)."""
    import oracle
    from oracle import parallel_predict as pp
    from oracle.sliding_window import dense_patch_starts, get_scan_interval

    torch.manual_seed(1)
    sd = oracle.make_basic_unet_state(1, 5, **({"features": tuple(net.features)} if tuple(net.features) != (32, 32, 64, 128, 256, 32) else {}))
    fstarts, _ = dense_patch_starts((size,) * 3, (roi,) * 3, get_scan_interval((size,) * 3, (roi,) * 3, (0.5,) * 3))
    nfull = len(fstarts[0]) * len(fstarts[1]) * len(fstarts[2])
    procs, threads, ncpu = _host_layout()
    vol_cpu = vol.cpu()
    # probe: one batch of 4 windows on one thread group
    torch.set_num_threads(threads)
    rr0 = min(roi, size)
    probe = torch.cat([vol_cpu[:, :, :rr0, :rr0, :rr0]] * 4)
    with torch.no_grad():
        oracle.basic_unet_forward(sd, probe[:1])
        t0 = time.perf_counter()
        oracle.basic_unet_forward(sd, probe)
        t_win = (time.perf_counter() - t0) / 4
    eff = 0.6 if procs > 1 else 1.0                      # what several groups sharing the memory system keep of their stand-alone rate (measured value is reported below)
    fit = int(budget_s * procs * eff / max(t_win, 1e-6))
    full = windows >= nfull and fit >= nfull
    if full:
        ext = (size,) * 3
    else:
        ext = sub_volume_extents(size, roi, max(1, min(windows, fit, nfull)))
    sub = vol if full else vol[:, :, : ext[0], : ext[1], : ext[2]].contiguous()
    sub_cpu = vol_cpu if full else sub.cpu()
    rr = tuple(min(roi, e) for e in ext)
    starts, _ = dense_patch_starts(ext, rr, get_scan_interval(ext, rr, (0.5,) * 3))
    nsub = len(starts[0]) * len(starts[1]) * len(starts[2])
    with torch.no_grad():
        t0 = time.perf_counter()
        if procs > 1:
            with pp.PoolPredictor(sub_cpu, rr, 4, (0.5,) * 3, pp.basic_unet_factory, (sd,), procs=procs, threads=threads) as pred:
                ref = oracle.sliding_window_inference(sub_cpu, rr, 4, pred, overlap=0.5, mode="gaussian", sigma_scale=0.125)
        else:
            ref = oracle.sliding_window_inference(sub_cpu, rr, 4, lambda w: oracle.basic_unet_forward(sd, w), overlap=0.5, mode="gaussian", sigma_scale=0.125)
        dt = time.perf_counter() - t0
        got = full_out if (full and full_out is not None) else inferer(sub, net)
    what = (f"the WHOLE {size}^3 benchmark volume, all {nsub} windows" if full else f"{ext[0]}x{ext[1]}x{ext[2]} corner sub-volume, {nsub} of {nfull} windows")
    if more is not None:            # the same reference for other arithmetic families of the product (extra.fp32_exact)
        more["ref"], more["sub"], more["what"] = ref, sub, what
    parity = oracle.label_parity(got, ref, tol=1e-4)
    parity["compared"] = (f"complete inferer output (blended logits) of {what}; rule: max|dlogit| <= 1e-4 and "
                          "every argmax difference at a voxel whose oracle top-2 margin < 2 max|dlogit| (mismatch_outside_margin == 0); the raw counts "
                          "(argmax_mismatch_voxels, min_class_dice, min_top2_margin) are the literal north-star bars: bit-exact argmax, Dice == 1.0")
    per_win = dt / nsub
    return {
        "value": size ** 3 / (nfull * per_win),
        "unit": "voxels/s",
        "cores": procs * threads,
        "kind": "port",
        "kind_note": "the bit-pinned oracle (oracle/: the reference's own ATen CPU operators in its order, checked bit-for-bit against the real reference by tests/test_oracle_golden.py); "
                     "the reference package itself is not installed on the GPU box",
        "parity_vs_gpu": parity,
        "sample": f"complete CPU-oracle sliding-window inference (windows + BasicUNet + gaussian blend, sw_batch 4) of {what} ({roi}^3 windows) on {procs} worker "
                  f"processes x {threads} threads of {ncpu} host threads (the blend in window order in the parent): {dt:.2f} s = {per_win:.4f} s/window "
                  f"(one batch alone on one {threads}-thread group: {t_win:.3f} s/window; pool efficiency {t_win / max(per_win * procs, 1e-9):.2f}); "
                  + ("value = voxels of the volume / that time" if full else f"value = {size}^3 voxels / ({nfull} windows x that)"),
    }


_PMC_INRUN: dict = {}      # kernel key -> traffic record collected by pmc_inrun() in THIS run
_PMC_WIDE_READS = ("sw_blend_mosaic_kernel", "sw_blend_reg_kernel")      # 16 B / lane readers: FETCH_SIZE counts their 128-B requests as 64 B (MI355X_MICROARCH.md, HBM section)


def pmc_inrun(budget_s: float = 150.0) -> dict:
    """HBM bytes per launch of the blend and of the dominant convolution from the hardware counters, collected in THIS run: two `rocprofv3 --kernel-trace --pmc`
    child processes (FETCH_SIZE, then WRITE_SIZE -- separate passes, as MI355X_MICROARCH.md prescribes) over tools/pmc_probe.py, which launches the two kernels at the
    benchmark's configuration (the mosaic blend of 1000 windows into 5 x 512^3; 32 -> 32 channels @ 96^3 x 64 windows).  Counter unit KiB per dispatch; the guide's
    gfx950 correction (x 2 on FETCH_SIZE for 16-bytes-per-lane readers) applied to the blend only.  Every child runs under `timeout`; any failure leaves the
    committed builder pass (`from_file`) in place."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile

    here = os.path.dirname(os.path.abspath(__file__))
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rp is None:
        return {}
    tmp = tempfile.mkdtemp(prefix="monai_amd_pmc_", dir="/tmp")
    got: dict = {}
    t0 = time.perf_counter()
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = ["timeout", "-k", "5", str(int(budget_s)), rp, "--kernel-trace", "--pmc", counter, "-d", os.path.join(tmp, counter), "-o", "p", "--",
                   sys.executable, os.path.join(here, "tools", "pmc_probe.py"), "--only", "mosaic,conv", "--conv-cfgs", "h2"]
            subprocess.run(cmd, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=budget_s + 30, check=True)
            dbs = [os.path.join(r, f) for r, _, fs in os.walk(os.path.join(tmp, counter)) for f in fs if f.endswith(".db")]
            if not dbs:
                return {}
            db = sqlite3.connect(dbs[0])
            rows = db.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? group by kernel_name", (counter,)).fetchall()
            db.close()
            for key in ("sw_blend_mosaic_kernel", "conv3d_k3_h2_kernel"):
                hit = [(n, avg) for name, n, avg in rows if key in name and "pack" not in name and "scale" not in name]
                if len(hit) != 1:
                    return {}
                got.setdefault(key, {})[counter] = {"dispatches": hit[0][0], "bytes": hit[0][1] * 1024.0}
    except Exception as e:                                         # noqa: BLE001 -- a missing profiler, a timeout, a changed schema: keep the committed pass
        print(f"bench: in-run PMC pass failed ({type(e).__name__}: {e}); traffic stays from_file", file=sys.stderr)
        return {}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    alg = {"sw_blend_mosaic_kernel": 1000 * 5 * 96.0 ** 3 * 4 + 5 * 512.0 ** 3 * 4, "conv3d_k3_h2_kernel": 2 * 64 * 32 * 96.0 ** 3 * 4}
    on = {"sw_blend_mosaic_kernel": "1000 windows of 5 x 96^3 logits -> 5 x 512^3 (the benchmark's blend launch)",
          "conv3d_k3_h2_kernel": "32 -> 32 channels @ 96^3, 64 windows per launch (the benchmark's largest convolution shape)"}
    out = {}
    for key, c in got.items():
        fetch = c["FETCH_SIZE"]["bytes"] * (2.0 if key in _PMC_WIDE_READS else 1.0)
        total = fetch + c["WRITE_SIZE"]["bytes"]
        note = (f"in_run (this bench process ran `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... WRITE_SIZE` over tools/pmc_probe.py, {c['FETCH_SIZE']['dispatches']} dispatches each, "
                f"{time.perf_counter() - t0:.0f} s; FETCH_SIZE " + ("x 2: 16-bytes-per-lane reader" if key in _PMC_WIDE_READS else "as counted") + ")")
        if "conv3d" in key:
            note += "; FETCH_SIZE is uncalibrated for this kernel's 4-byte loads: a ratio below 1 is not evidence of under-fetch"
        out[key] = {"measured": note, "hbm_bytes_per_launch": total, "fetch_bytes": fetch, "write_bytes": c["WRITE_SIZE"]["bytes"], "algorithmic_bytes": alg[key],
                    "ratio": total / alg[key], "measured_on": on[key]}
    return out


def pmc_traffic(kernel_key: str):
    """HBM bytes per launch of `kernel_key`: the counters collected in this run (pmc_inrun) when there are any, else the committed builder pass (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 corrections applied); the newest profiles/r*_pmc_hbm_traffic.json wins."""
    if kernel_key in _PMC_INRUN:
        return _PMC_INRUN[kernel_key]
    pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    try:
        for name in sorted((f for f in os.listdir(pdir) if f.endswith("_pmc_hbm_traffic.json")), reverse=True):
            with open(os.path.join(pdir, name)) as f:
                k = json.load(f)["kernels"].get(kernel_key)
            if k is not None:
                note = ("from_file (a builder-run rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass, see `source`; the in-run pass was skipped or failed)")
                if "conv3d" in kernel_key:       # MI355X_MICROARCH.md calibrates FETCH_SIZE for 16 B/lane readers only
                    note += "; FETCH_SIZE is uncalibrated for this kernel's 4-byte loads: a ratio below 1 is not evidence of under-fetch"
                return {"measured": note,
                        "hbm_bytes_per_launch": k["hbm_bytes_per_launch"], "algorithmic_bytes": k["algorithmic_bytes"],
                        "ratio": k["hbm_bytes_per_launch"] / k["algorithmic_bytes"], "measured_on": k.get("measured_on", "the bench configuration"),
                        "source": "profiles/" + name.replace(".json", ".txt")}
    except (OSError, KeyError, ValueError):
        pass
    return None


def build_net(name: str, roi: int, dev, features=None):
    from monai_amd.networks.nets import UNETR, BasicUNet, DynUNet, SegResNet, SwinUNETR, UNet

    torch.manual_seed(1)       # weights exactly as SURVEY.md 8(d) config 1 / 3
    if name == "swinunetr":
        net = SwinUNETR(in_channels=1, out_channels=5, feature_size=48)
    elif name == "unetr":
        net = UNETR(in_channels=1, out_channels=5, img_size=(roi,) * 3)
    elif name == "unet":
        net = UNet(spatial_dims=3, in_channels=1, out_channels=5, channels=(16, 32, 64, 128, 256), strides=(2, 2, 2, 2), num_res_units=2)
    elif name == "dynunet":
        net = DynUNet(spatial_dims=3, in_channels=1, out_channels=5, kernel_size=[3] * 5, strides=[1, 2, 2, 2, 2], upsample_kernel_size=[2] * 4)
    elif name == "segresnet":
        net = SegResNet(spatial_dims=3, init_filters=16, in_channels=1, out_channels=5)
    else:
        net = BasicUNet(spatial_dims=3, in_channels=1, out_channels=5, **({"features": features} if features else {}))
    return net.eval().to(dev)


def timed_steps(inferer, vol, net, steps: int, warmup: int, sync):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + device synchronise on both sides"""
    from monai_amd import _prof

    out = None
    for _ in range(warmup):
        out = inferer(vol, net)
    sync()
    _prof.start()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = inferer(vol, net)
    sync()
    dt = time.perf_counter() - t0
    return dt, _prof.stop(), out


def conv_roofline(spans, steps: int, ms: float, roi: int):
    """the 3x3x3 convolution configuration with the largest share of the step, priced against the peak of the matrix instruction it issues"""
    from monai_amd import ops as _ops

    convs = {k: v for k, v in spans.items() if k.startswith("conv3d_k3/")}
    if not convs:
        return None
    key, conv = max(convs.items(), key=lambda kv: kv[1]["ms_total"])
    tf = conv["work"] / (conv["ms_total"] * 1e-3) / 1e12          # the convolution's own flops: 2 * 27 * Cin * Cout per voxel
    cfg_id = int(key.split("/cfg")[1])
    ncfg = _ops.conv3d_k3_num_configs()
    peak, extra, pmc_key = PEAK_FP32_TFLOPS, {}, "conv3d_k3_mfma_kernel"
    if cfg_id == _ops.conv3d_k3_h2_config():     # fp16 two-piece split precision: three fp16 MFMA products per fp32 multiply-add
        kname = (f"conv3d_k3_h2_kernel (z-streaming direct 3x3x3 convolution on v_mfma_f32_32x32x16_f16, every fp32 operand as hi + lo fp16 pieces scaled into "
                 f"fp16's range by the input records' magnitude bounds, products hi*hi + lo*hi + hi*lo, fp32 accumulate: fp32-equivalent; 32|64 -> 32 ch @ {roi}^3 and the levels below)")
        gain, peak, pmc_key = 1.0 / 3.0, PEAK_F16_TFLOPS, "conv3d_k3_h2_kernel"
        extra = {"fp32_equivalent_tflops": tf, "fp32_equivalent_vs_fp32_mfma_peak": tf / PEAK_FP32_TFLOPS, "piece_products_per_multiply": 3}
    elif cfg_id == _ops.conv3d_k3_h2c_config():     # the same kernel in output channel groups of 16: two z-taps per 32-column instruction, 6 instead of 9 per in-plane tap
        kname = (f"conv3d_k3_h2_kernel<C16> (the split-precision kernel in output channel groups of 16: columns [kz 0 | kz 1] and [kz 2 | 0], a completed plane = the sum of three "
                 f"partial planes; 16-couts layers @ {roi}^3)")
        gain, peak, pmc_key = 1.0 / 4.0, PEAK_F16_TFLOPS, "conv3d_k3_h2c_kernel"      # 6 x 32 columns issued per 3 x 16 useful ones, x 3 piece products
        extra = {"fp32_equivalent_tflops": tf, "fp32_equivalent_vs_fp32_mfma_peak": tf / PEAK_FP32_TFLOPS, "piece_products_per_multiply": 3, "columns_issued_per_useful": 4.0 / 3.0}
    elif cfg_id == ncfg:      # in-plane Winograd: 12 instead of 27 multiply-adds per (voxel, cin, cout)
        kname = f"conv3d_k3_wino2p_kernel (Winograd F(2x2,3x3) in-plane + 3 direct z taps on v_mfma_f32_16x16x4_f32, two waves per SIMD, 32|64 -> 32 ch @ {roi}^3 / {roi // 2}^3)"
        gain, pmc_key = 2.25, "conv3d_k3_wino2p_kernel"
    else:
        kname, gain = f"conv3d_k3_mfma_kernel (cfg{cfg_id}: direct 3x3x3 implicit GEMM on v_mfma_f32_32x32x2_f32 @ {roi}^3)", 1.0
    issued = tf / gain
    roof = {"bound": "mfma", "achieved": issued, "peak": peak, "unit": "TFLOP/s", "frac": issued / peak, "traffic": None, "kernel": kname,
            "note": "achieved = matrix-core flops ISSUED per launch / average launch time (HIP events in the timed region); frac = the fraction of the dense MFMA peak of the "
                    "kernel's matrix instruction (fp32: 157.3, fp16: 2500 TFLOP/s) the matrix pipe delivers.  algorithmic_tflops counts the 3x3x3 convolution's own flops "
                    "(2*27*Cin*Cout per voxel); Winograd needs winograd_algorithmic_gain x fewer multiply-adds for them",
            "algorithmic_tflops": tf, "algorithmic_frac_of_peak": tf / peak, "winograd_algorithmic_gain": gain if gain >= 1.0 else None, **extra,
            "launches": conv["launches"], "ms_avg": conv["ms_avg"], "flops_per_launch_issued": conv["work"] / conv["launches"] / gain,
            "flops_per_launch_algorithmic": conv["work"] / conv["launches"], "share_of_step": conv["ms_total"] / steps / ms}
    td = pmc_traffic(pmc_key)
    if td:
        roof["traffic"], roof["traffic_detail"] = td["hbm_bytes_per_launch"], td
    return roof


def device_copy_gbps(dev) -> float:
    """read + write rate of a plain 1 GiB stream on this box, measured in this run: the practical ceiling of any read-once / write-once kernel, printed next to
    every HBM-bound `frac` (`frac_of_copy_ceiling`) -- the 8 TB/s spec is not reachable by a copy on this chip.  The faster of (a) torch's device-to-device copy and
    (b) this library's own 16-bytes-per-lane point-wise stream kernel (mh_scale_intensity_range_f32: 4 B read + 4 B written per voxel; 6.3 TB/s in round 2, the float4
    copy of tools/ubench/hbm_stream.hip: 6.15)"""
    from monai_amd import _lib

    n = 1 << 28
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    stream = _lib.stream_ptr(a)

    def own():
        _lib.lib().call("mh_scale_intensity_range_f32", _lib.ptr(a), _lib.ptr(b), n, 0.0, 1.0, 0, 0.0, 0.0, 0, 0.0, 0, 0.0, stream)

    best = 0.0
    for fn in (lambda: b.copy_(a), own):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = max(best, 5 * 2.0 * 4 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    return best


def blend_roofline(spans, mosaic: bool):
    blend = spans.get("sw_blend")
    if not blend:
        return None
    gbs = blend["work"] / (blend["ms_total"] * 1e-3) / 1e9
    roof = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": None,
            "kernel": ("sw_blend_mosaic_kernel<5,2> (gather blend over the mosaic logits layout" if mosaic else "sw_blend_reg_kernel<5,4,2> (gather blend over window-major logits") + ": logits read once, output written once)", "launches": blend["launches"], "ms_avg": blend["ms_avg"],
            "bytes_per_launch": blend["work"] / blend["launches"],
            "streaming_ceilings": "float4 copy / read-only / write-only kernels of tools/ubench/hbm_stream.hip on MI355X: 6.15 / 6.55-7.0 / 6.07 TB/s (profiles/r02_ubench_hbm_stream_v1.txt)"}
    td = pmc_traffic("sw_blend_mosaic_kernel" if mosaic else "sw_blend_reg_kernel")
    if td:
        roof["traffic"], roof["traffic_detail"] = td["hbm_bytes_per_launch"], td
    return roof


def extra_fp32_exact(args, vol, net, inferer, sync, shared):
    """the same workload on the exact-fp32 convolution kernels (monai_amd.config.CONV_ALGO = "fp32"), and its parity on the cpu_baseline's windows"""
    import oracle
    from monai_amd import config

    saved = config.CONV_ALGO
    config.CONV_ALGO = "fp32"
    try:
        dt, spans, _ = timed_steps(inferer, vol, net, 2, 1, sync)
        ms = 1e3 * dt / 2
        res = {"conv_algo": "fp32", "steps": 2, "warmup": 1, "ms_per_step": ms, "value": float(args.size) ** 3 / (dt / 2), "unit": "voxels/s",
               "roofline": conv_roofline(spans, 2, ms, args.roi)}
        if shared.get("ref") is not None:
            par = oracle.label_parity(inferer(shared["sub"], net), shared["ref"], tol=1e-4)
            par["compared"] = shared["what"]
            res["parity_vs_cpu_oracle"] = par
    finally:
        config.CONV_ALGO = saved
    return res


def extra_config3(args, vol, sync, dev):
    """BASELINE.json configs[3]: UNETR (ViT-B/16 encoder) over the same volume -- 2 timed steps, the attention kernel's matrix-core rate, and one
    window against the CPU oracle (oracle/unetr.py, bit-pinned to the reference by tests/golden/unetr.npz)"""
    import oracle
    from monai_amd.inferers import SlidingWindowInferer
    from oracle import unetr as ounetr

    net = build_net("unetr", args.roi, dev)
    inferer = SlidingWindowInferer(roi_size=(args.roi,) * 3, sw_batch_size=4, overlap=0.5, mode="gaussian", sigma_scale=0.125)
    dt, spans, _ = timed_steps(inferer, vol, net, 2, 1, sync)
    ms = 1e3 * dt / 2
    res = {"workload": f"UNETR ViT-B/16 5-class (seed-1 init), the same {args.size}^3 volume, {args.roi}^3 windows overlap 0.5 gaussian", "steps": 2, "warmup": 1,
           "ms_per_step": ms, "value": float(args.size) ** 3 / (dt / 2), "unit": "voxels/s", "roofline": conv_roofline(spans, 2, ms, args.roi)}
    att = spans.get("attention")
    if att:
        tf = att["work"] / (att["ms_total"] * 1e-3) / 1e12
        res["attention"] = {"kernel": "attention_h2_kernel<64> (softmax(QK^T/sqrt(d))V per head on the fp16 matrix cores in split precision, streamed keys / values, online softmax; S = 216, 12 heads x 64; rate = fp32-equivalent flops vs the fp32-MFMA peak)", "ms_per_step": att["ms_total"] / 2,
                            "tflops": tf, "bound": "mfma", "peak": PEAK_FP32_TFLOPS, "frac": tf / PEAK_FP32_TFLOPS, "share_of_step": att["ms_total"] / 2 / ms}
    lin = spans.get("linear")
    if lin:
        res["linear"] = {"kernel": "linear_h2_big_kernel / linear_h2_kernel (nn.Linear + bias / GELU / residual on the fp16 matrix cores in split precision; 128 x 128 workgroup tiles at this token count)", "ms_per_step": lin["ms_total"] / 2,
                         "fp32_equivalent_tflops": lin["work"] / (lin["ms_total"] * 1e-3) / 1e12}
    # parity + CPU baseline: the complete inferer on a corner of the benchmark volume (27 windows at 96^3 / 512^3) against the CPU oracle of the same network
    # (oracle/unetr.py, pinned to the real reference by tests/golden/unetr.npz), its per-window network in worker processes (oracle/parallel_predict.py)
    from oracle import parallel_predict as pp
    from oracle.sliding_window import dense_patch_starts, get_scan_interval

    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    ext = sub_volume_extents(args.size, args.roi, 27)
    rr = tuple(min(args.roi, e) for e in ext)
    sub = vol[:, :, : ext[0], : ext[1], : ext[2]].contiguous()
    sub_cpu = sub.cpu()
    starts, _ = dense_patch_starts(ext, rr, get_scan_interval(ext, rr, (0.5,) * 3))
    nsub = len(starts[0]) * len(starts[1]) * len(starts[2])
    fstarts, _ = dense_patch_starts((args.size,) * 3, (args.roi,) * 3, get_scan_interval((args.size,) * 3, (args.roi,) * 3, (0.5,) * 3))
    nfull = len(fstarts[0]) * len(fstarts[1]) * len(fstarts[2])
    procs, threads, ncpu = _host_layout()
    torch.set_num_threads(threads)
    with torch.no_grad():
        t0 = time.perf_counter()
        if procs > 1:
            with pp.PoolPredictor(sub_cpu, rr, 4, (0.5,) * 3, pp.unetr_factory, (sd,), procs=procs, threads=threads) as pred:
                ref = oracle.sliding_window_inference(sub_cpu, rr, 4, pred, overlap=0.5, mode="gaussian", sigma_scale=0.125)
        else:
            ref = oracle.sliding_window_inference(sub_cpu, rr, 4, lambda w: ounetr.unetr_forward(sd, w), overlap=0.5, mode="gaussian", sigma_scale=0.125)
        dt_cpu = time.perf_counter() - t0
        got = inferer(sub, net)
    par = oracle.label_parity(got, ref, tol=1e-4)
    par["compared"] = f"complete inferer output (blended logits) of the {ext[0]}x{ext[1]}x{ext[2]} corner of the benchmark volume, {nsub} windows of {args.roi}^3, product vs CPU oracle"
    res["parity_vs_cpu_oracle"] = par
    res["cpu_baseline"] = {"value": float(args.size) ** 3 / (nfull * dt_cpu / nsub), "unit": "voxels/s", "cores": procs * threads, "kind": "port",
                           "sample": f"CPU-oracle UNETR sliding-window inference of those {nsub} of {nfull} windows on {procs} worker processes x {threads} threads (of {ncpu}, "
                                     f"worker start-up included): {dt_cpu:.2f} s = {dt_cpu / nsub:.3f} s/window; value = {args.size}^3 voxels / ({nfull} windows x that)"}
    return res


def extra_config4(dev):  # noqa: C901
    """BASELINE.json configs[4]: Spacing (affine diag(.8, .8, 1.6) -> pixdim 1, trilinear, border: 512^3 -> 410x410x819) and GaussianSmooth(sigma 1) on a
    batch of 4 x 512^3 volumes resident in HBM: transform and kernel-only times against the 8 TB/s spec (SURVEY.md 8d byte counts), and parity of both
    on a 128^3 volume against CPU restatements of the reference path (oracle/resample.py; F.pad + depthwise F.conv3d per axis)."""
    import numpy as np
    import torch.nn.functional as F

    from monai_amd import ops
    from monai_amd.data import MetaTensor
    from monai_amd.networks.layers import gaussian_1d
    from monai_amd.transforms import GaussianSmooth, Spacing
    from oracle import resample as ores

    def timeit(fn, iters=5, warm=2):
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

    n, e = 4, 512
    aff = np.diag([0.8, 0.8, 1.6, 1.0])
    vols = []
    for s in range(n):
        torch.manual_seed(s)
        vols.append(MetaTensor(torch.rand(1, e, e, e).to(dev), affine=aff))
    sp = Spacing(pixdim=(1.0, 1.0, 1.0), mode="bilinear", padding_mode="border")
    out = sp(vols[0])
    osz = tuple(int(v) for v in out.shape[1:])
    res = {"workload": f"{n} x {e}^3 fp32 volumes (seeds 0..{n - 1}) in HBM: Spacing(pixdim 1, bilinear, border; fp64 coordinates) -> {list(out.shape)}, GaussianSmooth(sigma=1)", "runs": []}
    copy_gbps = device_copy_gbps(dev)
    res["device_copy_GBps"] = copy_gbps
    res["device_copy_note"] = ("a 1 GiB read + write stream measured in this run (the faster of torch's device copy and this library's 16-bytes-per-lane point-wise kernel): "
                               "the practical streaming ceiling; frac = of the 8 TB/s spec, frac_of_copy_ceiling = of this")
    ms = timeit(lambda: [sp(v) for v in vols])
    nb = 4.0 * n * (e ** 3 + out.numel())
    res["runs"].append({"op": "Spacing transform (4 volumes, host-side affine algebra included)", "ms": ms, "GBps": nb / ms / 1e6, "frac": nb / ms / 1e6 / PEAK_HBM_GBS})
    gs = GaussianSmooth(sigma=1.0)
    plain = [v.as_tensor() for v in vols]
    ms = timeit(lambda: [gs(v) for v in plain])
    nb = 8.0 * n * e ** 3
    res["runs"].append({"op": "GaussianSmooth transform (4 volumes)", "ms": ms, "GBps": nb / ms / 1e6, "frac": nb / ms / 1e6 / PEAK_HBM_GBS})
    raw = plain[0]
    m = np.array([[1.25, 0, 0, 0], [0, 1.25, 0, 0], [0, 0, 0.625, 0]], dtype=np.float64)
    for f64 in (True, False):
        ms = timeit(lambda: ops.affine_resample(raw, m.reshape(-1), osz, "bilinear", "border", False, f64))
        nb = 4.0 * (raw.numel() + osz[0] * osz[1] * osz[2])
        res["runs"].append({"op": f"kernel: separable affine resample, {'fp64' if f64 else 'fp32'} interpolation (1 volume)", "bound": "hbm", "ms": ms, "GBps": nb / ms / 1e6,
                            "frac": nb / ms / 1e6 / PEAK_HBM_GBS, "bytes": nb})
    k = gaussian_1d(1.0).numpy()
    ms = timeit(lambda: ops.separable_filter3d(raw, [k, k, k]))
    res["runs"].append({"op": "kernel: fused 3-axis Gaussian, 9 taps (1 volume)", "bound": "hbm", "ms": ms, "GBps": 8.0 * raw.numel() / ms / 1e6,
                        "frac": 8.0 * raw.numel() / ms / 1e6 / PEAK_HBM_GBS, "bytes": 8.0 * raw.numel()})
    # parity on a 128^3 volume
    torch.manual_seed(11)
    small = torch.rand(1, 128, 128, 128)
    y = sp(MetaTensor(small.to(dev), affine=aff))
    xform = np.linalg.inv(aff) @ y.affine.cpu().numpy()
    ref = ores.spatial_resample_eager(small, torch.from_numpy(xform), tuple(y.shape[1:]), mode="bilinear", padding_mode="border")
    g = gs(small.to(dev)).cpu()
    kk = gaussian_1d(1.0)
    gref = small[None]
    for ax in range(3):        # separable_filtering (simplelayers.py:170-249): zero padding + depthwise conv per axis, fp32
        shape = [1, 1, 1, 1, 1]
        shape[2 + ax] = kk.numel()
        pad = [0, 0, 0, 0, 0, 0]
        pad[2 * (2 - ax)] = pad[2 * (2 - ax) + 1] = kk.numel() // 2
        gref = F.conv3d(F.pad(gref, pad), kk.reshape(shape))
    res["parity_vs_cpu_restatement"] = {"spacing_max_abs": float((y.cpu().as_tensor() - ref).abs().max()), "spacing_tol": 2e-6, "spacing_shape": list(y.shape),
                                        "gaussian_max_abs": float((g - gref[0]).abs().max()), "gaussian_tol": 1e-5,
                                        "compared": "128^3 volume, the same transforms: product vs oracle/resample.py (AffineTransform path of the reference) / zero-padded depthwise F.conv3d per axis"}
    for r_ in res["runs"]:
        r_["frac_of_copy_ceiling"] = r_["GBps"] / copy_gbps
    res["parity_vs_cpu_restatement"]["ok"] = bool(res["parity_vs_cpu_restatement"]["spacing_max_abs"] < 2e-6 and res["parity_vs_cpu_restatement"]["gaussian_max_abs"] < 1e-5)
    # CPU baseline: the reference path of both transforms -- restated with the same ATen operators (oracle/resample.py: img.to(dtype) -> normalised theta ->
    # F.affine_grid + F.grid_sample -> float32; separable_filtering: F.pad + depthwise F.conv3d per axis) -- on ONE 512^3 volume on the host cores, and the product's
    # result for that volume against it at full size
    procs_, threads_, ncpu = _host_layout()
    threads = min(32, max(threads_, procs_ * threads_))       # ONE process here: the whole share of the host (16 threads under the boxes' 16-CPU quota)
    torch.set_num_threads(threads)
    v0 = vols[0]
    x0 = v0.as_tensor().cpu()
    y0 = sp(v0)
    xf0 = torch.from_numpy(np.linalg.inv(aff) @ y0.affine.cpu().numpy())
    cpu = {"cores": threads, "of_host_threads": ncpu, "kind": "port", "volume": f"one {e}^3 fp32 volume"}
    with torch.no_grad():
        for name, dt_ in (("spacing_fp64_s", torch.float64), ("spacing_fp32_s", torch.float32)):
            t0 = time.perf_counter()
            r_ = ores.spatial_resample_eager(x0, xf0.to(dt_), tuple(y0.shape[1:]), mode="bilinear", padding_mode="border", dtype=dt_)
            cpu[name] = time.perf_counter() - t0
            if dt_ == torch.float64:
                cpu["spacing_512_max_abs_vs_product"] = float((y0.as_tensor().cpu() - r_).abs().max())
            del r_
        t0 = time.perf_counter()
        gref0 = x0[None]
        for ax in range(3):
            shape = [1, 1, 1, 1, 1]
            shape[2 + ax] = kk.numel()
            pad = [0, 0, 0, 0, 0, 0]
            pad[2 * (2 - ax)] = pad[2 * (2 - ax) + 1] = kk.numel() // 2
            gref0 = F.conv3d(F.pad(gref0, pad), kk.reshape(shape))
        cpu["gaussian_s"] = time.perf_counter() - t0
        cpu["gaussian_512_max_abs_vs_product"] = float((gs(plain[0]).cpu() - gref0[0]).abs().max())
    nvox_out = float(y0.numel())
    cpu["spacing_fp64_output_voxels_per_s"] = nvox_out / cpu["spacing_fp64_s"]
    cpu["spacing_fp32_output_voxels_per_s"] = nvox_out / cpu["spacing_fp32_s"]
    cpu["gaussian_voxels_per_s"] = float(e) ** 3 / cpu["gaussian_s"]
    res["cpu_baseline"] = cpu
    return res


def per_rank_breakdown(spans, steps: int, world: int, dev, dist):
    """predictor / gather-wait / blend milliseconds per step of every rank (what the first SCALE run needs to be read)"""
    keys = ("sw_predictor", "sw_gather_wait", "sw_blend")
    mine = torch.tensor([spans.get(k, {}).get("ms_total", 0.0) / steps for k in keys], dtype=torch.float64, device=dev)
    if world > 1:
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
    else:
        allr = [mine]
    return [{"rank": r, "predictor_ms": float(t[0]), "gather_wait_ms": float(t[1]), "blend_ms": float(t[2])} for r, t in enumerate(allr)]


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=512, help="volume edge (512 = the BASELINE.json workload)")
    ap.add_argument("--roi", type=int, default=96)
    ap.add_argument("--cpu-windows", type=int, default=1000, help="windows the CPU baseline / parity check runs: >= the volume's window count (1000) = the WHOLE volume, "
                                                                  "fewer = a corner sub-volume (125 = 288^3), 0 = skip")
    ap.add_argument("--cpu-budget-s", type=float, default=float(os.environ.get("MONAI_AMD_BENCH_CPU_BUDGET_S", "900")),
                    help="seconds the CPU leg may take; a probe batch prices it and the largest corner sub-volume that fits is taken when the whole volume would not")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run hardware-counter passes behind roofline*.traffic (two rocprofv3 child processes, ~1 min)")
    ap.add_argument("--no-extra", action="store_true", help="skip extra.fp32_exact / config3 / config4 (development runs)")
    ap.add_argument("--harness-features", default="", help="TEST HARNESS ONLY (emulator runs of tests/test_bench_harness.py): BasicUNet widths, e.g. 16,16,32,32,64,16; "
                                                           "refused on a GPU -- the benchmark network has the default widths")
    ap.add_argument("--net", default="basicunet", choices=sorted(NETS),
                    help="basicunet = the BASELINE.json metric (configs[1]); unetr = configs[3] (ViT-B/16 UNETR, MFMA attention path); unet = MONAI UNet 16..256, 2 res units (row a11); "
                         "dynunet = nnU-Net-shaped DynUNet (5 levels, 32..320 filters); segresnet = SegResNet(init_filters=16); swinunetr = SwinUNETR(feature_size=48) (SURVEY 8f-4)")
    args = ap.parse_args(argv)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    emulated = False
    if not torch.cuda.is_available():
        # TEST HARNESS ONLY (tests/test_bench_harness.py): the same code path on the SIMT-emulator build of the kernels with gloo,
        # so that the torch.distributed.run wiring, the rank-0 JSON line and its keys are exercised without a GPU.  Never a result.
        if os.environ.get("MONAI_AMD_BENCH_EMULATOR") != "1":
            raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from emu_backend import emu_backend

        emulated = True
        ctx = emu_backend()
        ctx.__enter__()
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")

    import torch.distributed as dist

    from monai_amd import config, parallel
    from monai_amd.inferers import SlidingWindowInferer

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if emulated:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        parallel.enable_window_sharding()

    feats = tuple(int(v) for v in args.harness_features.split(",")) if args.harness_features else None
    if feats and not emulated:
        raise SystemExit("--harness-features is for the emulator harness test only")
    net = build_net(args.net, args.roi, dev, feats)
    vol = benchmark_volume(args.size).to(dev)
    inferer = SlidingWindowInferer(roi_size=(args.roi,) * 3, sw_batch_size=4, overlap=0.5, mode="gaussian", sigma_scale=0.125)

    def sync():
        if world > 1:
            dist.barrier()
        if not emulated:
            torch.cuda.synchronize()

    dt, spans, out = timed_steps(inferer, vol, net, args.steps, args.warmup, sync)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ranks = per_rank_breakdown(spans, args.steps, world, dev, dist) if world > 1 else None

    if rank == 0:
        voxels = float(args.size) ** 3
        ms = 1e3 * dt / args.steps
        profiled = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")      # no profiler inside a profiler
        if world == 1 and not emulated and not args.no_pmc and not profiled and args.net == "basicunet" and (args.size, args.roi) == (512, 96):
            _PMC_INRUN.update(pmc_inrun())
        exact = config.conv_algo() in (config.CONV_ALGOS["fp32"], config.CONV_ALGOS["direct"], config.CONV_ALGOS["wino2d"])
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
            "dtype_note": ("every tensor, accumulator and elementwise op is fp32; 3x3x3 convolutions on the exact-fp32 kernels (monai_amd.config.CONV_ALGO / MONAI_AMD_CONV_ALGO set)"
                           if exact else
                           "every tensor, accumulator and elementwise op is fp32; the multiplications of the 3x3x3 convolutions are evaluated from two fp16 pieces per fp32 operand "
                           "(hi + lo, the input first scaled into fp16's range by a power of two from its records' magnitude bounds; three exact piece products on the fp16 matrix "
                           "cores, fp32 accumulation): fp32-equivalent for any finite input magnitude -- parity_vs_gpu below; extra.fp32_exact is the same run on the exact-fp32 kernels"),
            "data": "synthetic",
            "config": {
                "workload": f"{NETS[args.net]} 5-class (default features, seed-1 init), {args.size}^3 fp32 synthetic CT volume (the reference's create_test_image_3d phantom, "
                            f"SURVEY 8d config 1) resident in HBM, {args.roi}^3 windows overlap 0.5 gaussian blend, sw_batch_size 4 (engine batches up to 64 windows per launch)",
                "parallelism": "1 GPU" if world == 1 else f"windows sharded over {world} GPUs, RCCL all-gather of logits before the blend",
            },
            "roofline": conv_roofline(spans, args.steps, ms, args.roi),
            "roofline_hbm": blend_roofline(spans, mosaic=world == 1 and hasattr(net, "forward_into_windows") and os.environ.get("MONAI_AMD_LOGITS_LAYOUT") != "windows"),
            "conv_ms_per_step": conv_all,
            "checksum": float(out.double().sum().item()),
        }
        if line["roofline_hbm"] is not None and not emulated:
            cg = device_copy_gbps(dev)
            line["roofline_hbm"]["device_copy_GBps"] = cg
            line["roofline_hbm"]["frac_of_copy_ceiling"] = line["roofline_hbm"]["achieved"] / cg
        if ranks is not None:
            line["per_rank_ms_per_step"] = ranks
        if emulated:
            line["emulated"] = "SIMT emulator + gloo: harness test only, not a measurement"
        shared: dict = {}
        if world == 1 and args.cpu_windows > 0 and args.net == "basicunet":
            try:
                line["cpu_baseline"] = cpu_baseline(args.size, args.roi, args.cpu_windows, vol, net, inferer, out, args.cpu_budget_s, shared)
            except Exception as e:      # the checker's own failure (a dead pool worker, host memory) must not cost the headline line: fall back to one process on a 27-window corner
                err = f"{type(e).__name__}: {e}"
                try:
                    os.environ["MONAI_AMD_BENCH_CPU_PROCS"] = "1"
                    line["cpu_baseline"] = cpu_baseline(args.size, args.roi, 27, vol, net, inferer, None, args.cpu_budget_s, shared)
                    line["cpu_baseline"]["fallback_after"] = err
                except Exception as e2:
                    line["cpu_baseline"] = {"error": err, "fallback_error": f"{type(e2).__name__}: {e2}"}
        else:
            line["cpu_baseline"] = None
        if world == 1 and args.net == "basicunet" and not args.no_extra and not emulated:
            extra = {}
            for name, fn in (("fp32_exact", lambda: extra_fp32_exact(args, vol, net, inferer, sync, shared)), ("config3", lambda: extra_config3(args, vol, sync, dev)),
                             ("config4", lambda: extra_config4(dev))):
                try:
                    extra[name] = fn()
                except Exception as e:      # an extra must never cost the headline line
                    extra[name] = {"error": f"{type(e).__name__}: {e}"}
                torch.cuda.empty_cache()
            line["extra"] = extra
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
