"""Headline benchmark (BASELINE.json): output voxels/s of 3-D sliding-window segmentation --
512^3 fp32 synthetic CT volume, 96^3 windows, overlap 0.5 (1000 windows), 5-class BasicUNet, gaussian blend.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one complete ``SlidingWindowInferer(...)(volume, net)`` call: window gather, the BasicUNet forward of
all 1000 windows, (N > 1: RCCL all-gather of the per-window logits,) blend + normalise.  The volume -- the reference's own
``create_test_image_3d`` phantom of SURVEY.md 8(d) config 1, restated bit-identically in oracle/synthetic.py -- is resident
in HBM before the timed region.  N > 1 shards the windows of the SAME volume over the ranks (strong scaling, config 2 of
BASELINE.json).  Rank 0 prints ONE JSON line of numbers and short labels (< 6 KB: the driver keeps a bounded tail of stdout); what every
key means, how it is measured and which caveats apply is written down ONCE in DESIGN.md section 7 ("the bench line"), not in the line:

  parity              the headline family's whole-volume parity against the CPU oracle (134 217 728 voxels x 5 logits), at top level
  reference_self_spread   the oracle against ITSELF (1 thread vs the pool's thread layout; oneDNN off vs on): what "bit-exact argmax" means for the reference
  roofline / roofline_hbm the dominant convolution kernel (matrix flops issued / launch time vs the fp16 MFMA peak) and the blend (bytes / time vs 8 TB/s)
  cpu_baseline        the complete CPU-oracle inferer on the host cores (kind "port", bit-pinned to the reference by tests/test_oracle_golden.py)
  extra               fp32_exact (exact-fp32 kernels, same workload + parity), direct_split (the direct split-precision kernel everywhere: ms per step), config3 (UNETR), config4 (Spacing + GaussianSmooth on 4 x 512^3)
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


def _short_parity(rep: dict, family: str, full: bool = True) -> dict:
    """the numbers of oracle.label_parity under short keys (the rule itself: oracle/parity.py); full=False: the extras' brief form"""
    out = {"family": family, "voxels": rep["voxels"], "max_abs_logit_diff": rep["max_abs_logit_diff"], "tolerance": rep["tolerance"],
           "argmax_mismatch_voxels": rep["argmax_mismatch_voxels"], "mismatch_outside_margin": rep["mismatch_outside_margin"],
           "dice_deficit": rep["dice_deficit"], "ok": rep["ok"]}      # dice_deficit = 1 - min class Dice from the integer counts: 0.0 iff the label maps are identical
    if full:
        out.update({"max_top2_margin_at_mismatch": rep["max_top2_margin_at_mismatch"], "min_top2_margin": rep["min_top2_margin"],
                    "voxels_with_margin_below_1e-4": rep["voxels_with_margin_below_1e-4"]})
    return out


def _oracle_inferer(sd, sub_cpu, rr, procs: int, threads: int, factory=None):
    """the complete CPU-oracle sliding-window inference of `sub_cpu` with the per-window network on `procs` workers x `threads` ATen threads"""
    import oracle
    from oracle import parallel_predict as pp

    with torch.no_grad():
        if procs > 1 or factory is not None:
            with pp.PoolPredictor(sub_cpu, rr, 4, (0.5,) * 3, factory or pp.basic_unet_factory, (sd,), procs=procs, threads=threads) as pred:
                return oracle.sliding_window_inference(sub_cpu, rr, 4, pred, overlap=0.5, mode="gaussian", sigma_scale=0.125)
        torch.set_num_threads(threads)
        return oracle.sliding_window_inference(sub_cpu, rr, 4, lambda w: oracle.basic_unet_forward(sd, w), overlap=0.5, mode="gaussian", sigma_scale=0.125)


def reference_self_spread(sd, vol_cpu, size: int, roi: int, procs: int, threads: int, t_win: float, budget_s: float, product=None):
    """how far the REFERENCE arithmetic is from itself.  The CPU oracle (the reference's own ATen operators) on a corner of the benchmark
    volume in three configurations of the same host: (a) the pool's thread layout (what cpu_baseline runs), (b) ONE thread per worker (another oneDNN work split:
    other summation orders), (c) oneDNN disabled (ATen's native convolution).  max|dlogit| and the labels that flip between them are what "bit-exact argmax"
    can mean for the reference; `product` = (inferer, net, device volume): the product on the same voxels against (a), next to them."""
    import oracle
    from oracle import parallel_predict as pp

    total = max(1, procs * threads)
    out = {}
    # windows by budget: (a) and (b) cost about the same CPU-seconds per window, (c) several times more -> a smaller corner there
    per_win_pool = t_win / max(procs, 1) / 0.6
    for key, want, factory, p_, t_, cost in (("one_thread_vs_pool", 125, None, total, 1, 2.2), ("onednn_off_vs_on", 27, pp.basic_unet_factory_no_onednn, procs, threads, 8.0)):
        n = want
        while n > 8 and n * per_win_pool * cost > budget_s / 2:
            n = {125: 64, 64: 27, 27: 8}.get(n, 8)
        ext = sub_volume_extents(size, roi, n)
        rr = tuple(min(roi, e) for e in ext)
        sub_cpu = vol_cpu[:, :, : ext[0], : ext[1], : ext[2]].contiguous()
        t0 = time.perf_counter()
        base = _oracle_inferer(sd, sub_cpu, rr, procs, threads, pp.basic_unet_factory)
        other = _oracle_inferer(sd, sub_cpu, rr, p_, t_, factory or pp.basic_unet_factory)
        rep = oracle.label_parity(other, base, tol=1e-4)
        rec = {"voxels": rep["voxels"], "max_abs_logit_diff": rep["max_abs_logit_diff"], "label_flips": rep["argmax_mismatch_voxels"],
               "flips_outside_margin": rep["mismatch_outside_margin"], "layout": f"{p_}x{t_} vs {procs}x{threads} threads", "seconds": time.perf_counter() - t0}
        if product is not None:
            from monai_amd import config

            inferer, net, vol = product
            sub_dev = vol[:, :, : ext[0], : ext[1], : ext[2]].contiguous()
            pr = oracle.label_parity(inferer(sub_dev, net), base, tol=1e-4)
            rec["product_max_abs_logit_diff"], rec["product_label_flips"] = pr["max_abs_logit_diff"], pr["argmax_mismatch_voxels"]
            # the same voxels on the exact-fp32 kernels, and on the headline kernels with the InstanceNorm statistics from a separate two-pass kernel over the stored
            # fp32 tensor (mean, then centred squares: the order ATen's CPU batch-norm uses) instead of the convolution epilogue's per-tile records: is the distance
            # to the reference in the fp16 pieces, in the statistics, or in the summation order of the convolutions themselves?
            saved_algo, saved_stats = config.CONV_ALGO, getattr(net, "fused_stats", True)
            try:
                config.CONV_ALGO = "fp32"
                pe = oracle.label_parity(inferer(sub_dev, net), base, tol=1e-4)
                rec["product_fp32_exact"] = [pe["max_abs_logit_diff"], pe["argmax_mismatch_voxels"]]
                config.CONV_ALGO = saved_algo
                if hasattr(net, "fused_stats"):
                    net.fused_stats = False
                    ps = oracle.label_parity(inferer(sub_dev, net), base, tol=1e-4)
                    rec["product_two_pass_stats"] = [ps["max_abs_logit_diff"], ps["argmax_mismatch_voxels"]]
            finally:
                config.CONV_ALGO = saved_algo
                if hasattr(net, "fused_stats"):
                    net.fused_stats = saved_stats
        out[key] = rec
    return out


def cpu_baseline(size: int, roi: int, windows: int, vol: torch.Tensor, net, inferer, full_out, budget_s: float, more=None, spread_budget_s: float = 0.0):
    """CPU oracle = port of the reference path (kind "port": the same ATen CPU operators in the same order, pinned bit-for-bit to the real reference by
    tests/test_oracle_golden.py -- the GPU box has no MONAI): the COMPLETE sliding-window inference -- window loop, BasicUNet, importance-weighted blend -- on the host
    cores, and the product's output on the same voxels against it (the headline rule of oracle/parity.py on the BLENDED logits).
    Default: the WHOLE benchmark volume (1000 windows, every output voxel compared).  The per-window network runs in several worker processes
    (oracle/parallel_predict.py), the blend in this process in the reference's window order.  A probe batch prices the run first: when the whole volume would
    exceed `budget_s` seconds the largest corner sub-volume that fits is taken instead (`windows` < `windows_total` in the record).
    -> (cpu_baseline record, parity report, reference_self_spread record or None)"""
    import oracle
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
    eff = 0.6 if procs > 1 else 1.0                      # what several groups sharing the memory system keep of their stand-alone rate (the measured value is reported)
    fit = int(budget_s * procs * eff / max(t_win, 1e-6))
    full = windows >= nfull and fit >= nfull
    ext = (size,) * 3 if full else sub_volume_extents(size, roi, max(1, min(windows, fit, nfull)))
    sub = vol if full else vol[:, :, : ext[0], : ext[1], : ext[2]].contiguous()
    sub_cpu = vol_cpu if full else sub.cpu()
    rr = tuple(min(roi, e) for e in ext)
    starts, _ = dense_patch_starts(ext, rr, get_scan_interval(ext, rr, (0.5,) * 3))
    nsub = len(starts[0]) * len(starts[1]) * len(starts[2])
    t0 = time.perf_counter()
    ref = _oracle_inferer(sd, sub_cpu, rr, procs, threads)
    dt = time.perf_counter() - t0
    with torch.no_grad():
        got = full_out if (full and full_out is not None) else inferer(sub, net)
    what = f"whole {size}^3, {nsub} windows" if full else f"{ext[0]}x{ext[1]}x{ext[2]} corner, {nsub} of {nfull} windows"
    if more is not None:            # the same reference for other arithmetic families of the product (extra.fp32_exact)
        more["ref"], more["sub"], more["what"], more["s_per_window"] = ref, sub, what, dt / nsub
    parity = oracle.label_parity(got, ref, tol=1e-4)
    per_win = dt / nsub
    rec = {"value": size ** 3 / (nfull * per_win), "unit": "voxels/s", "cores": procs * threads, "kind": "port",
           "sample": f"{what}, {procs}x{threads} of {ncpu} threads",
           "windows": nsub, "windows_total": nfull, "seconds": dt, "s_per_window": per_win}
    spread = None
    if spread_budget_s > 0:
        try:
            spread = reference_self_spread(sd, vol_cpu, size, roi, procs, threads, t_win, spread_budget_s, product=(inferer, net, vol))
        except Exception as e:      # noqa: BLE001 -- the spread is evidence next to the parity, never a reason to lose it
            spread = {"error": f"{type(e).__name__}: {e}"[:160]}
    return rec, parity, spread


_PMC_INRUN: dict = {}      # kernel key -> traffic record collected by pmc_inrun() in THIS run
_PMC_WIDE_READS = ("sw_blend_mosaic_kernel", "sw_blend_reg_kernel")      # 16 B / lane readers: FETCH_SIZE counts their 128-B requests as 64 B (MI355X_MICROARCH.md, HBM section)


_PMC_STATUS = {"status": "not_run"}      # why the in-run pass did or did not deliver (printed as line["pmc"])


def _pmc_conv_key() -> str:
    """kernel name of the configuration the selector gives the headline's 32 -> 32 layers at 96^3 (what tools/pmc_probe.py --conv-cfgs auto launches)"""
    from monai_amd import ops as _ops

    return "conv3d_k3_h2w_kernel" if _ops.conv3d_k3_select(32, 32, 96, 96, 96, bounded=True) == _ops.conv3d_k3_h2w_config() else "conv3d_k3_h2_kernel"


def pmc_inrun(budget_s: float = 150.0) -> dict:
    """HBM bytes per launch of the blend and of the dominant convolution from the hardware counters, collected in THIS run: two `rocprofv3 --kernel-trace --pmc`
    child processes (FETCH_SIZE, then WRITE_SIZE -- separate passes, as MI355X_MICROARCH.md prescribes) over tools/pmc_probe.py, which launches the two kernels at the
    benchmark's configuration (the mosaic blend of 1000 windows into 5 x 512^3; 32 -> 32 channels @ 96^3 x 64 windows).  Counter unit KiB per dispatch; the guide's
    gfx950 correction (x 2 on FETCH_SIZE for 16-bytes-per-lane readers) applied to the blend only (FETCH_SIZE is uncalibrated for the convolution's 4-byte loads: its
    read side says nothing, its write side does).  Every child runs under `timeout`; a failure is retried once, then the committed builder pass (`from_file`) stays in
    place and `_PMC_STATUS` says what failed (the child's last stderr line included)."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile

    here = os.path.dirname(os.path.abspath(__file__))
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rp is None:
        _PMC_STATUS["status"] = "failed: rocprofv3 not found"
        return {}
    tmp = tempfile.mkdtemp(prefix="monai_amd_pmc_", dir="/tmp")
    got: dict = {}
    t0 = time.perf_counter()
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            rows, why = None, ""
            for attempt in range(2):
                out_dir = os.path.join(tmp, f"{counter}_{attempt}")
                cmd = ["timeout", "-k", "5", str(int(budget_s)), rp, "--kernel-trace", "--pmc", counter, "-d", out_dir, "-o", "p", "--",
                       sys.executable, os.path.join(here, "tools", "pmc_probe.py"), "--only", "mosaic,conv", "--conv-cfgs", "auto"]
                try:
                    r = subprocess.run(cmd, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=budget_s + 30, text=True)
                except subprocess.TimeoutExpired:
                    why = f"{counter}: timeout after {budget_s:.0f} s"
                    continue
                if r.returncode != 0:
                    tail = (r.stderr or "").strip().splitlines()[-1:] or [""]
                    why = f"{counter}: rocprofv3 rc {r.returncode}: {tail[0][:100]}"
                    continue
                dbs = [os.path.join(d_, f) for d_, _, fs in os.walk(out_dir) for f in fs if f.endswith(".db")]
                if not dbs:
                    why = f"{counter}: no counter database written"
                    continue
                db = sqlite3.connect(dbs[0])
                rows = db.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? group by kernel_name", (counter,)).fetchall()
                db.close()
                break
            if rows is None:
                _PMC_STATUS["status"] = "failed: " + why
                return {}
            for key in ("sw_blend_mosaic_kernel", _pmc_conv_key()):
                hit = [(n, avg) for name, n, avg in rows if key in name and "pack" not in name and "scale" not in name]
                if len(hit) != 1:
                    _PMC_STATUS["status"] = f"failed: {counter}: {len(hit)} kernels match {key}"
                    return {}
                got.setdefault(key, {})[counter] = {"dispatches": hit[0][0], "bytes": hit[0][1] * 1024.0}
    except Exception as e:                                         # noqa: BLE001 -- a changed schema, a full disk: keep the committed pass, say why
        _PMC_STATUS["status"] = f"failed: {type(e).__name__}: {e}"[:160]
        return {}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    alg = {"sw_blend_mosaic_kernel": 1000 * 5 * 96.0 ** 3 * 4 + 5 * 512.0 ** 3 * 4, _pmc_conv_key(): 2 * 64 * 32 * 96.0 ** 3 * 4}
    out = {}
    for key, c in got.items():
        fetch = c["FETCH_SIZE"]["bytes"] * (2.0 if key in _PMC_WIDE_READS else 1.0)
        total = fetch + c["WRITE_SIZE"]["bytes"]
        out[key] = {"measured": "in_run", "hbm_bytes_per_launch": total, "fetch_bytes": fetch, "write_bytes": c["WRITE_SIZE"]["bytes"], "algorithmic_bytes": alg[key],
                    "ratio": total / alg[key], "dispatches": c["FETCH_SIZE"]["dispatches"], "fetch_calibrated": key in _PMC_WIDE_READS}
    _PMC_STATUS["status"] = "in_run"
    _PMC_STATUS["seconds"] = round(time.perf_counter() - t0, 1)
    return out


def pmc_traffic(kernel_key: str):
    """HBM bytes per launch of `kernel_key`: the counters collected in this run (pmc_inrun) when there are any, else the committed builder pass (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 corrections applied); the newest profiles/r*_pmc_hbm_traffic.json wins.  `measured`: "in_run" |
    "from_file:<name>" -- a kernel the in-run probe does not launch (the exact-fp32 family's) is always from_file."""
    if kernel_key in _PMC_INRUN:
        return _PMC_INRUN[kernel_key]
    pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    try:
        for name in sorted((f for f in os.listdir(pdir) if f.endswith("_pmc_hbm_traffic.json")), reverse=True):
            with open(os.path.join(pdir, name)) as f:
                k = json.load(f)["kernels"].get(kernel_key)
            if k is not None:
                return {"measured": "from_file:" + name, "hbm_bytes_per_launch": k["hbm_bytes_per_launch"], "algorithmic_bytes": k["algorithmic_bytes"],
                        "ratio": k["hbm_bytes_per_launch"] / k["algorithmic_bytes"], "fetch_calibrated": "conv3d" not in kernel_key}
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


def _traffic(roof: dict, key: str) -> None:
    td = pmc_traffic(key)
    if td:
        roof["traffic"] = td["hbm_bytes_per_launch"]
        roof["traffic_src"], roof["traffic_ratio"], roof["fetch_calibrated"] = td["measured"], td["ratio"], td["fetch_calibrated"]


def conv_roofline(spans, steps: int, ms: float, roi: int):
    """the 3x3x3 convolution configuration with the largest share of the step, priced against the peak of the matrix instruction it issues:
    achieved = matrix-core flops ISSUED per launch / average launch time (HIP events in the timed region); the convolution's own flops
    (2 * 27 * Cin * Cout per voxel) are `algorithmic_tflops`"""
    from monai_amd import ops as _ops

    convs = {k: v for k, v in spans.items() if k.startswith("conv3d_k3/")}
    if not convs:
        return None
    key, conv = max(convs.items(), key=lambda kv: kv[1]["ms_total"])
    tf = conv["work"] / (conv["ms_total"] * 1e-3) / 1e12
    cfg_id = int(key.split("/cfg")[1])
    ncfg = _ops.conv3d_k3_num_configs()
    peak, pmc_key = PEAK_FP32_TFLOPS, "conv3d_k3_mfma_kernel"
    if cfg_id == _ops.conv3d_k3_h2_config():        # fp16 two-piece split precision: three fp16 MFMA products per fp32 multiply-add
        kname, gain, peak, pmc_key = "conv3d_k3_h2_kernel (v_mfma_f32_32x32x16_f16, hi+lo split)", 1.0 / 3.0, PEAK_F16_TFLOPS, "conv3d_k3_h2_kernel"
    elif cfg_id == _ops.conv3d_k3_h2w_config():     # in-plane Winograd F(2x2,3x3) in front of the split product: 12 x 3 = 36 instead of 27 x 3 = 81 issued multiply-adds per (voxel, cin, cout)
        kname, gain, peak, pmc_key = "conv3d_k3_h2w_kernel (F(2x2,3x3) + hi/lo split, v_mfma_f32_16x16x32_f16)", 27.0 / 36.0, PEAK_F16_TFLOPS, "conv3d_k3_h2w_kernel"
    elif cfg_id == _ops.conv3d_k3_h2c_config():     # the same kernel in output channel groups of 16: 6 x 32 columns issued per 3 x 16 useful ones, x 3 piece products
        kname, gain, peak, pmc_key = "conv3d_k3_h2_kernel<C16> (two z-taps per instruction)", 1.0 / 4.0, PEAK_F16_TFLOPS, "conv3d_k3_h2c_kernel"
    elif cfg_id == ncfg:                            # in-plane Winograd: 12 instead of 27 multiply-adds per (voxel, cin, cout)
        kname, gain, pmc_key = "conv3d_k3_wino2p_kernel (F(2x2,3x3), v_mfma_f32_16x16x4_f32)", 2.25, "conv3d_k3_wino2p_kernel"
    else:
        kname, gain = f"conv3d_k3_mfma_kernel cfg{cfg_id} (v_mfma_f32_32x32x2_f32)", 1.0
    issued = tf / gain
    roof = {"bound": "mfma", "achieved": issued, "peak": peak, "unit": "TFLOP/s", "frac": issued / peak, "traffic": None, "kernel": kname,
            "algorithmic_tflops": tf, "issued_per_algorithmic_flop": 1.0 / gain,
            "launches": conv["launches"], "ms_avg": conv["ms_avg"], "share_of_step": conv["ms_total"] / steps / ms}
    _traffic(roof, pmc_key)
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
            "kernel": "sw_blend_mosaic_kernel<5,2>" if mosaic else "sw_blend_reg_kernel<5,4,2>", "launches": blend["launches"], "ms_avg": blend["ms_avg"],
            "bytes_per_launch": blend["work"] / blend["launches"]}
    _traffic(roof, "sw_blend_mosaic_kernel" if mosaic else "sw_blend_reg_kernel")
    return roof


def _roof_brief(roof):
    """a roofline object reduced to its numbers (the extras: the headline's objects carry the rest)"""
    if not roof:
        return None
    keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_src", "kernel", "ms_avg", "launches", "share_of_step", "algorithmic_tflops")
    return {k: roof[k] for k in keep if k in roof}


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
               "roofline": _roof_brief(conv_roofline(spans, 2, ms, args.roi))}
        if shared.get("ref") is not None:
            res["parity"] = _short_parity(oracle.label_parity(inferer(shared["sub"], net), shared["ref"], tol=1e-4), "fp32-exact", full=False)
            res["parity"]["compared"] = shared["what"]
    finally:
        config.CONV_ALGO = saved
    return res


def extra_direct_split(args, vol, net, inferer, sync):
    """the same workload with the DIRECT split-precision kernel on every layer (CONV_ALGO = "h2": what ran before the in-plane Winograd form of round 6): the A/B of
    `roofline.kernel` on the box of this very run"""
    from monai_amd import config

    saved = config.CONV_ALGO
    config.CONV_ALGO = "h2"
    try:
        dt, _, _ = timed_steps(inferer, vol, net, 3, 1, sync)
    finally:
        config.CONV_ALGO = saved
    return {"conv_algo": "h2", "ms_per_step": 1e3 * dt / 3}


def extra_config3(args, vol, sync, dev, shared=None):
    """BASELINE.json configs[3]: UNETR (ViT-B/16 encoder) over the same volume -- 2 timed steps, the attention kernel's matrix-core rate, and one
    window against the CPU oracle (oracle/unetr.py, bit-pinned to the reference by tests/golden/unetr.npz)"""
    import oracle
    from monai_amd.inferers import SlidingWindowInferer
    from oracle import unetr as ounetr

    net = build_net("unetr", args.roi, dev)
    inferer = SlidingWindowInferer(roi_size=(args.roi,) * 3, sw_batch_size=4, overlap=0.5, mode="gaussian", sigma_scale=0.125)
    dt, spans, _ = timed_steps(inferer, vol, net, 2, 1, sync)
    ms = 1e3 * dt / 2
    res = {"workload": f"UNETR ViT-B/16 5-class, same {args.size}^3 volume, {args.roi}^3 win ov 0.5", "steps": 2, "warmup": 1,
           "ms_per_step": ms, "value": float(args.size) ** 3 / (dt / 2), "unit": "voxels/s", "roofline": _roof_brief(conv_roofline(spans, 2, ms, args.roi))}
    att = spans.get("attention")
    if att:
        tf = att["work"] / (att["ms_total"] * 1e-3) / 1e12
        res["attention"] = {"kernel": "attention_h2_kernel<64> (S=216, 12 heads; fp32-eq flops)", "ms_per_step": att["ms_total"] / 2,
                            "tflops": tf, "bound": "mfma", "peak": PEAK_FP32_TFLOPS, "frac": tf / PEAK_FP32_TFLOPS, "share_of_step": att["ms_total"] / 2 / ms}
    lin = spans.get("linear")
    if lin:
        ltf = lin["work"] / (lin["ms_total"] * 1e-3) / 1e12
        res["linear"] = {"kernel": "linear_h2_big_kernel (128x128 tiles, hi+lo split)", "ms_per_step": lin["ms_total"] / 2, "fp32_equivalent_tflops": ltf,
                         "issued_frac_of_fp16_peak": 3.0 * ltf / PEAK_F16_TFLOPS, "share_of_step": lin["ms_total"] / 2 / ms}
    # parity + CPU baseline: the complete inferer on a corner of the benchmark volume (27 windows at 96^3 / 512^3) against the CPU oracle of the same network
    # (oracle/unetr.py, pinned to the real reference by tests/golden/unetr.npz), its per-window network in worker processes (oracle/parallel_predict.py)
    from oracle import parallel_predict as pp
    from oracle.sliding_window import dense_patch_starts, get_scan_interval

    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    # 125 windows (288^3) when the BasicUNet leg's measured CPU rate says the UNETR oracle (0.6x the flops per window) finishes them in about a minute, else 27
    spw = (shared or {}).get("s_per_window")
    ext = sub_volume_extents(args.size, args.roi, 125 if (spw is not None and 0.6 * spw * 125 < 75.0) else 27)
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
    res["parity"] = _short_parity(oracle.label_parity(got, ref, tol=1e-4), "split-fp16", full=False)
    res["parity"]["compared"] = f"{ext[0]}x{ext[1]}x{ext[2]} corner, {nsub} windows"
    res["cpu_baseline"] = {"value": float(args.size) ** 3 / (nfull * dt_cpu / nsub), "unit": "voxels/s", "cores": procs * threads, "kind": "port",
                           "sample": f"{nsub} of {nfull} windows, {procs}x{threads} of {ncpu} threads", "seconds": dt_cpu, "s_per_window": dt_cpu / nsub}
    return res


def extra_unet(args, vol, sync, dev):
    """MONAI UNet (16..256, strides 2, two residual units) over the same volume -- 2 timed steps, and the complete inferer on a 27-window corner
    against the CPU oracle of the same network (oracle/unet.py, pinned to the real reference by tests/golden/unet*.npz)"""
    import oracle
    from monai_amd.inferers import SlidingWindowInferer
    from oracle import unet as ounet

    net = build_net("unet", args.roi, dev)
    inferer = SlidingWindowInferer(roi_size=(args.roi,) * 3, sw_batch_size=4, overlap=0.5, mode="gaussian", sigma_scale=0.125)
    dt, _, _ = timed_steps(inferer, vol, net, 2, 1, sync)
    res = {"workload": f"UNet 16..256 res2 5-class, same {args.size}^3 volume", "ms_per_step": 1e3 * dt / 2, "value": float(args.size) ** 3 / (dt / 2), "unit": "voxels/s"}
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    ext = sub_volume_extents(args.size, args.roi, 27)
    rr = tuple(min(args.roi, e) for e in ext)
    sub = vol[:, :, : ext[0], : ext[1], : ext[2]].contiguous()
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    with torch.no_grad():
        t0 = time.perf_counter()
        ref = oracle.sliding_window_inference(sub.cpu(), rr, 4, lambda w: ounet.unet_forward(sd, w, (16, 32, 64, 128, 256), (2, 2, 2, 2), 2), overlap=0.5, mode="gaussian", sigma_scale=0.125)
        dt_cpu = time.perf_counter() - t0
        got = inferer(sub, net)
    res["parity"] = _short_parity(oracle.label_parity(got, ref, tol=1e-4), "split-fp16", full=False)
    res["parity"]["compared"] = f"{ext[0]}x{ext[1]}x{ext[2]} corner, 27 windows"
    res["cpu_s_per_window"] = dt_cpu / 27
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
    res = {"workload": f"{n} x {e}^3 fp32 in HBM: Spacing(pixdim 1) -> {list(out.shape)[1:]}, GaussianSmooth(1)", "runs": []}
    copy_gbps = device_copy_gbps(dev)
    res["device_copy_GBps"] = copy_gbps
    ms = timeit(lambda: [sp(v) for v in vols])
    nb = 4.0 * n * (e ** 3 + out.numel())
    res["runs"].append({"op": "Spacing x4 (with host algebra)", "ms": ms, "GBps": nb / ms / 1e6, "frac": nb / ms / 1e6 / PEAK_HBM_GBS})
    gs = GaussianSmooth(sigma=1.0)
    plain = [v.as_tensor() for v in vols]
    ms = timeit(lambda: [gs(v) for v in plain])
    nb = 8.0 * n * e ** 3
    res["runs"].append({"op": "GaussianSmooth x4", "ms": ms, "GBps": nb / ms / 1e6, "frac": nb / ms / 1e6 / PEAK_HBM_GBS})
    raw = plain[0]
    m = np.array([[1.25, 0, 0, 0], [0, 1.25, 0, 0], [0, 0, 0.625, 0]], dtype=np.float64)
    for f64 in (True, False):
        ms = timeit(lambda: ops.affine_resample(raw, m.reshape(-1), osz, "bilinear", "border", False, f64))
        nb = 4.0 * (raw.numel() + osz[0] * osz[1] * osz[2])
        res["runs"].append({"op": f"kernel: resample {'fp64' if f64 else 'fp32'}", "ms": ms, "GBps": nb / ms / 1e6,
                            "frac": nb / ms / 1e6 / PEAK_HBM_GBS})
    k = gaussian_1d(1.0).numpy()
    ms = timeit(lambda: ops.separable_filter3d(raw, [k, k, k]))
    res["runs"].append({"op": "kernel: gaussian 9 taps", "ms": ms, "GBps": 8.0 * raw.numel() / ms / 1e6,
                        "frac": 8.0 * raw.numel() / ms / 1e6 / PEAK_HBM_GBS})
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
                                        "gaussian_max_abs": float((g - gref[0]).abs().max()), "gaussian_tol": 1e-5, "compared": "128^3 volume"}
    for r_ in res["runs"]:
        r_["frac_of_copy_ceiling"] = r_.pop("GBps") / copy_gbps          # GB/s = frac x 8000
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
    cpu = {"cores": threads, "kind": "port", "volume": f"one {e}^3"}
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
    res["cpu_baseline"] = cpu          # output voxels / s = 410 * 410 * 819 / spacing_*_s, 512^3 / gaussian_s
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


def _rounded(obj, digits: int = 5):
    """floats to `digits` significant digits: the line is a record of measurements, not of double-precision noise (and stays short)"""
    if isinstance(obj, float):
        return float(f"{obj:.{digits}g}") if obj == obj and abs(obj) != float("inf") else obj
    if isinstance(obj, dict):
        return {k: (v if k in ("value", "checksum") else _rounded(v, digits)) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_rounded(v, digits) for v in obj]
    return obj


_EXTRA_DROP = ("steps", "warmup", "launches", "algorithmic_tflops", "cores", "kind", "tolerance", "traffic_src")
LINE_BYTES_MAX = 5800      # the driver's parsed copy keeps a line of this size whole (the 16 KB line of round 4 was cut)
_TRIM_ORDER = (("extra", "config4", "cpu_baseline"), ("extra", "unet", "workload"), ("extra", "config3", "workload"), ("extra", "config4", "workload"), ("extra", "config3", "attention"),
               ("extra", "fp32_exact", "roofline"), ("conv_ms_per_step",), ("upconv",), ("extra", "config4", "parity_vs_cpu_restatement"), ("extra", "config3", "cpu_baseline"))


def _slim_extra(obj):
    """the extras without what the headline's objects already say (steps, units, core counts, labels of the CPU leg): numbers and short labels only"""
    if isinstance(obj, dict):
        out = {}
        for k, v in obj.items():
            if k in _EXTRA_DROP or v is None or (k == "unit" and v == "voxels/s"):
                continue
            out[k] = int(round(v)) if k == "value" and isinstance(v, float) else _slim_extra(v)
        return out
    if isinstance(obj, (list, tuple)):
        return [_slim_extra(v) for v in obj]
    return obj


def _fit_line(line, limit: int = LINE_BYTES_MAX):
    """drop the least important sub-objects (in _TRIM_ORDER) until the JSON line fits `limit` bytes; what was dropped is named in `trimmed`"""
    dropped = []
    for path in _TRIM_ORDER:
        if len(json.dumps(line)) <= limit:
            break
        node = line
        for k in path[:-1]:
            node = node.get(k) if isinstance(node, dict) else None
            if node is None:
                break
        if isinstance(node, dict) and path[-1] in node:
            del node[path[-1]]
            dropped.append(".".join(path))
    if dropped:
        line["trimmed"] = dropped
    return line


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
    ap.add_argument("--no-spread", action="store_true", help="skip reference_self_spread (the oracle against itself under other thread layouts / without oneDNN)")
    ap.add_argument("--spread-budget-s", type=float, default=90.0, help="seconds the reference_self_spread leg may take (it sizes its corner sub-volumes by it)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run hardware-counter passes behind roofline*.traffic (two rocprofv3 child processes, ~1 min)")
    ap.add_argument("--no-extra", action="store_true", help="skip extra.fp32_exact / config3 / config4 (development runs)")
    ap.add_argument("--harness-features", default="", help="TEST HARNESS ONLY (emulator runs of tests/test_bench_harness.py): BasicUNet widths, e.g. 16,16,32,32,64,16; "
                                                           "refused on a GPU -- the benchmark network has the default widths")
    ap.add_argument("--net", default="basicunet", choices=sorted(NETS),
                    help="basicunet = the BASELINE.json metric (configs[1]); unetr = configs[3] (ViT-B/16 UNETR, MFMA attention path); unet = MONAI UNet 16..256, 2 res units (row a11); "
                         "dynunet = nnU-Net-shaped DynUNet (5 levels, 32..320 filters); segresnet = SegResNet(init_filters=16); swinunetr = SwinUNETR(feature_size=48) (SURVEY 8f-4)")
    ap.add_argument("--force-shard", action="store_true", help="--gpus 1 only: initialise RCCL with ONE rank and run the window-sharded code (round schedule with its tail, padded rows, the in-place "
                                                               "probe, one async all_gather_into_tensor per round) -- the N > 1 path exercised on a one-GPU box; the line is a development record "
                                                               "(`forced_shard`), not the headline")
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

    forced = bool(args.force_shard) and world == 1
    if world > 1 or forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if forced:
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if emulated:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        parallel.enable_window_sharding(force=forced)

    feats = tuple(int(v) for v in args.harness_features.split(",")) if args.harness_features else None
    if feats and not emulated:
        raise SystemExit("--harness-features is for the emulator harness test only")
    net = build_net(args.net, args.roi, dev, feats)
    vol = benchmark_volume(args.size).to(dev)
    inferer = SlidingWindowInferer(roi_size=(args.roi,) * 3, sw_batch_size=4, overlap=0.5, mode="gaussian", sigma_scale=0.125)

    def sync():
        if world > 1 or forced:
            dist.barrier()
        if not emulated:
            torch.cuda.synchronize()

    dt, spans, out = timed_steps(inferer, vol, net, args.steps, args.warmup, sync)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ranks = per_rank_breakdown(spans, args.steps, world, dev, dist) if (world > 1 or forced) else None

    if rank == 0:
        voxels = float(args.size) ** 3
        ms = 1e3 * dt / args.steps
        profiled = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")      # no profiler inside a profiler
        if world == 1 and not forced and not emulated and not args.no_pmc and not profiled and args.net == "basicunet" and (args.size, args.roi) == (512, 96):
            _PMC_INRUN.update(pmc_inrun())
        exact = config.conv_algo() in (config.CONV_ALGOS["fp32"], config.CONV_ALGOS["direct"], config.CONV_ALGOS["wino2d"])
        conv_all = {k.split("/")[1]: {"ms": round(v["ms_total"] / args.steps, 3), "tflops": round(v["work"] / (v["ms_total"] * 1e-3) / 1e12, 2)}
                    for k, v in spans.items() if k.startswith("conv3d_k3/")}
        family = "fp32-exact" if exact else "split-fp16"
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
            "dtype_note": "fp32 tensors + accumulators; conv products " + ("exact fp32" if exact else "from hi+lo fp16 pieces (fp32-equivalent); extra.fp32_exact = exact kernels"),
            "data": "synthetic",
            "config": {
                "workload": f"{NETS[args.net]} 5-class seed 1, {args.size}^3 CT phantom in HBM, {args.roi}^3 win ov 0.5 gaussian, sw_batch 4 (engine: <= 64 windows per launch)",
                "parallelism": ("1 GPU, sharded code forced on a one-rank RCCL group" if forced else "1 GPU") if world == 1 else f"windows sharded over {world} GPUs, RCCL all-gather of logits before the blend",
            },
            "parity": None,
            "roofline": conv_roofline(spans, args.steps, ms, args.roi),
            "roofline_hbm": blend_roofline(spans, mosaic=world == 1 and not forced and hasattr(net, "forward_into_windows") and os.environ.get("MONAI_AMD_LOGITS_LAYOUT") != "windows"),
            "conv_ms_per_step": conv_all,
            "upconv": ({"kernel": "upconv_k4s2_h2_kernel (UpCat: convT k4 s2 of the low-res tensor)", "ms_per_step": spans["upconv_k4s2"]["ms_total"] / args.steps,
                        "ms_avg": spans["upconv_k4s2"]["ms_avg"], "fp32_equivalent_tflops": spans["upconv_k4s2"]["work"] / (spans["upconv_k4s2"]["ms_total"] * 1e-3) / 1e12}
                       if "upconv_k4s2" in spans else None),
            "checksum": float(out.double().sum().item()),
            "pmc": dict(_PMC_STATUS),
        }
        if line["roofline_hbm"] is not None and not emulated:
            cg = device_copy_gbps(dev)
            line["roofline_hbm"]["device_copy_GBps"] = cg
            line["roofline_hbm"]["frac_of_copy_ceiling"] = line["roofline_hbm"]["achieved"] / cg
        if ranks is not None:
            line["per_rank_ms_per_step"] = ranks
            pred = max(r["predictor_ms"] for r in ranks)
            line["exposed_comm_share"] = max(r["gather_wait_ms"] for r in ranks) / ms      # what the compute stream waited for gathers, of the step
            line["predictor_ms_max"] = pred
        if forced:
            line["forced_shard"] = {"backend": dist.get_backend(), "inplace_gather_probe": [bool(v) for v in parallel.inplace_gather_verdicts().values()],
                                    "rounds": [n for _, n in parallel.partition(1000 if (args.size, args.roi) == (512, 96) else 1, 1, 0, force=True).schedule(64)][:12]}
        if emulated:
            line["emulated"] = "SIMT emulator + gloo: harness test only, not a measurement"
        shared: dict = {}
        if world == 1 and args.cpu_windows > 0 and args.net == "basicunet" and not forced:
            spread_s = 0.0 if (emulated or args.no_spread) else args.spread_budget_s
            try:
                rec, par, spread = cpu_baseline(args.size, args.roi, args.cpu_windows, vol, net, inferer, out, args.cpu_budget_s, shared, spread_s)
            except Exception as e:      # the checker's own failure (a dead pool worker, host memory) must not cost the headline line: fall back to one process on a 27-window corner
                err = f"{type(e).__name__}: {e}"[:160]
                try:
                    os.environ["MONAI_AMD_BENCH_CPU_PROCS"] = "1"
                    rec, par, spread = cpu_baseline(args.size, args.roi, 27, vol, net, inferer, None, args.cpu_budget_s, shared, 0.0)
                    rec["fallback_after"] = err
                except Exception as e2:
                    rec, par, spread = {"error": err, "fallback_error": f"{type(e2).__name__}: {e2}"[:160]}, None, None
            if par is not None:
                line["parity"] = _short_parity(par, family)
                line["parity"]["compared"] = shared.get("what", "")
                rec["parity_vs_gpu"] = {k: par[k] for k in ("ok", "mismatch_outside_margin", "max_abs_logit_diff", "voxels")}
            line["cpu_baseline"] = rec
            if spread is not None:
                line["reference_self_spread"] = spread
        else:
            line["cpu_baseline"] = None
        if world == 1 and args.net == "basicunet" and not args.no_extra and not emulated and not forced:
            extra = {}
            for name, fn in (("fp32_exact", lambda: extra_fp32_exact(args, vol, net, inferer, sync, shared)), ("direct_split", lambda: extra_direct_split(args, vol, net, inferer, sync)), ("config3", lambda: extra_config3(args, vol, sync, dev, shared)), ("unet", lambda: extra_unet(args, vol, sync, dev)),
                             ("config4", lambda: extra_config4(dev))):
                try:
                    extra[name] = fn()
                except Exception as e:      # an extra must never cost the headline line
                    extra[name] = {"error": f"{type(e).__name__}: {e}"[:160]}
                torch.cuda.empty_cache()
            line["extra"] = _slim_extra(extra)
        line = _fit_line(_rounded(line))
        print(json.dumps(line))
    if world > 1 or forced:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
