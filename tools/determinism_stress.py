"""Run-to-run determinism stress of the headline path, with a per-kernel bisect (VERDICT r03 "Next round" 1b).

    python tools/determinism_stress.py --procs 4 --runs 20 --size 512            # parent: spawns the workers, prints one JSON report
    python tools/determinism_stress.py --worker --runs 20 --size 512             # one worker (also usable alone)

Each worker runs the complete 512^3 inference `runs` times on the SAME GPU as its siblings (they time-slice the CUs: the condition under
which round 3 recorded one run-to-run mismatch) and compares, BITWISE, every launch's outputs against its own first run:

  * every `ops.*` launch of the engine is wrapped; after the launch an int64 digest of each output tensor (sum of its 32-bit words as
    integers, wrapping -- order-independent, exact) is queued on the device; one run = a list of (op, digest) pairs, read back at the end;
  * the first pair that differs from the first run's names the kernel whose output changed first (its inputs' digests were equal);
  * `--poison` additionally re-fills the allocator's free blocks with a different bit pattern before every run (uninitialised reads);
  * `--fresh-plans` drops the networks' cached activation buffers between runs so the allocator hands out different addresses.

Test infrastructure (tools/): nothing in monai_amd/ imports it."""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# outputs of each wrapped launch: positional indices / keyword names of the tensors the kernel WRITES (monai_amd/ops.py signatures)
OUTPUTS = {
    "window_extract": [5],
    "conv3d_k3": [5],            # (the statistics buffer, argument 6, is a shared scratch whose tail is stale by design: its consumer instnorm_finalize is digested)
    "instnorm_stats": [1],
    "instnorm_finalize": [8],
    "groupnorm_finalize": [9],
    "maxpool2": [2, 3],
    "deconv_k2s2": [4, 5],
    "conv1x1": [4],
    "nrm_identity": [0],
    "pad_replicate": [1, 3],
    "add_act": [5, 6],
    "sw_blend_mosaic": [2],
    "sw_blend": [2],
}


def digest(t):
    import torch

    if t is None or not isinstance(t, torch.Tensor) or t.numel() == 0:
        return None
    # wrapping int32 sum of the words: integer addition is associative, so the digest is exact and independent of the reduction order; no int64 copy of
    # a 7 GB activation tensor and no .contiguous() of a channel-sliced view (the reduction walks the strides)
    w = t.view(torch.int32) if t.element_size() == 4 else t.contiguous().view(torch.uint8).to(torch.int32)
    return w.sum(dtype=torch.int32)


def install_hooks(log):
    import torch

    from monai_amd import ops

    def wrap(name, fn, outs):
        def inner(*a, **k):
            r = fn(*a, **k)
            for i in outs:
                t = a[i] if i < len(a) else None
                if isinstance(t, torch.Tensor):
                    log.append((name, i, digest(t)))
            return r

        return inner

    for name, outs in OUTPUTS.items():
        if hasattr(ops, name):
            setattr(ops, name, wrap(name, getattr(ops, name), outs))
    # the network's last kernel writes into the mosaic (an object holding the class arrays): digest its flat storage
    if hasattr(ops, "conv1x1_windows"):
        real = ops.conv1x1_windows

        def c1w(src, src_nrm, w, b, mosaic, w0):
            r = real(src, src_nrm, w, b, mosaic, w0)
            # the windows this launch wrote (the flat allocation also holds alignment gaps and the other windows: never digested as a whole)
            d = None
            for w in range(int(w0), int(w0) + int(src.shape[0])):
                dw = digest(mosaic.window_view(w))
                d = dw if d is None else d + dw
            log.append(("conv1x1_windows", int(w0), d))
            return r

        ops.conv1x1_windows = c1w


def worker(a):
    import torch

    from bench import benchmark_volume, build_net
    from monai_amd.inferers import SlidingWindowInferer

    emu = a.device == "cpu"          # plumbing check of this tool on the SIMT emulator (tiny sizes)
    if emu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from emu_backend import emu_backend

        ctx = emu_backend()
        ctx.__enter__()
        dev = torch.device("cpu")
        torch.cuda.synchronize = lambda *x, **k: None
    else:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
    net = build_net(a.net, a.roi, dev, tuple(int(v) for v in a.features.split(",")) if a.features else None)
    vol = benchmark_volume(a.size).to(dev)
    inferer = SlidingWindowInferer(roi_size=(a.roi,) * 3, sw_batch_size=4, overlap=0.5, mode="gaussian", sigma_scale=0.125)
    log: list = []
    if not a.no_hooks:
        install_hooks(log)
    patterns = [0x7FC00000, 0x7F7FFFFF, 0x00000000, 0x3F800000, 0xFF7FFFFE]
    first, first_out, report = None, None, {"pid": os.getpid(), "runs": a.runs, "mismatching_runs": [], "launches_per_run": None, "schedules": []}
    # --vary: the schedules the product picks under memory pressure (windows per launch from the free HBM, the slab-wise path when the all-window
    # logits do not fit), forced here one after the other: the OUTPUT must not depend on them
    vary = [("MONAI_AMD_SW_BATCH", "64"), ("MONAI_AMD_SW_BATCH", "32"), ("MONAI_AMD_SW_BATCH", "50"), ("MONAI_AMD_SW_BATCH", "13"),
            ("MONAI_AMD_MAX_LOGITS_BYTES", str(int(6e9))), ("MONAI_AMD_MAX_LOGITS_BYTES", str(int(9e9))), ("MONAI_AMD_LOGITS_LAYOUT", "windows")]
    if not a.vary and not a.no_hooks:
        os.environ.setdefault("MONAI_AMD_SW_BATCH", "32")       # per-launch digests are compared as a sequence: the schedule must not follow the free memory
    t0 = time.time()
    for r in range(a.runs):
        if a.fresh_plans and hasattr(net, "_plans"):
            net._plans.clear()
        if a.poison and not emu:
            torch.cuda.empty_cache()
            free = torch.cuda.mem_get_info(dev)[0]
            n = int(min(free * 0.5, 60e9)) // 4
            junk = torch.empty(n, dtype=torch.int32, device=dev)
            p = patterns[r % len(patterns)]
            junk.fill_(p if p < 2**31 else p - 2**32)
            del junk
        log.clear()
        if a.vary:
            for k_, _ in vary:
                os.environ.pop(k_, None)
            if r > 0:
                k_, v_ = vary[(r - 1) % len(vary)]
                os.environ[k_] = v_
                report["schedules"].append(f"{k_}={v_}")
        out = inferer(vol, net)
        od = digest(out)
        torch.cuda.synchronize()
        run = [(n_, i_, None if d_ is None else int(d_.item())) for n_, i_, d_ in log] + [("output", 0, int(od.item()))]
        if first is None:
            first, first_out = run, out.clone() if a.keep_output else None
            report["launches_per_run"] = len(run)
            report["first_run_output_digest"] = run[-1][2]
            continue
        if a.vary or a.no_hooks:          # outputs only
            if run[-1] != first[-1]:
                report["mismatching_runs"].append({"run": r, "op": "output", "schedule": report["schedules"][-1] if report["schedules"] else None})
        elif run != first:
            where = next((j for j, (x, y) in enumerate(zip(run, first)) if x != y), min(len(run), len(first)))
            entry = {"run": r, "output_equal": run[-1] == first[-1], "first_differing_launch": where, "op": run[where][0] if where < len(run) else "length",
                     "launch_arg": run[where][1] if where < len(run) else None, "differing_launches": sum(1 for x, y in zip(run, first) if x != y),
                     "launches": len(run), "launches_first_run": len(first)}
            if first_out is not None:
                entry["output_words_differing"] = int((out.view(torch.int32) != first_out.view(torch.int32)).sum())
            report["mismatching_runs"].append(entry)
        del out
    report["seconds"] = time.time() - t0
    report["all_equal"] = not report["mismatching_runs"]
    print("STRESS_WORKER " + json.dumps(report), flush=True)
    return 0 if report["all_equal"] else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("--procs", type=int, default=4)
    ap.add_argument("--runs", type=int, default=20)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--roi", type=int, default=96)
    ap.add_argument("--net", default="basicunet")
    ap.add_argument("--poison", action="store_true")
    ap.add_argument("--fresh-plans", action="store_true")
    ap.add_argument("--no-hooks", action="store_true", help="outputs only (no per-launch digests: the unperturbed timing of the product)")
    ap.add_argument("--keep-output", action="store_true")
    ap.add_argument("--vary", action="store_true", help="force a different schedule per run (windows per launch, slab-wise path, window-major logits); compare outputs")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--features", default="", help="BasicUNet widths for emulator plumbing checks")
    a = ap.parse_args()
    if a.worker:
        return worker(a)
    cmd = [sys.executable, os.path.abspath(__file__), "--worker"] + [x for x in sys.argv[1:] if x != "--worker"]
    t0 = time.time()
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for _ in range(a.procs)]
    reports, bad = [], 0
    for p in procs:
        out, _ = p.communicate()
        line = next((ln for ln in out.splitlines() if ln.startswith("STRESS_WORKER ")), None)
        if line is None:
            bad += 1
            reports.append({"error": out[-2000:]})
        else:
            rep = json.loads(line[len("STRESS_WORKER "):])
            reports.append(rep)
            bad += 0 if rep["all_equal"] else 1
    # the workers' first runs must agree with each other too (same binary, same inputs)
    firsts = {r.get("first_run_output_digest") for r in reports if "error" not in r}
    if len(firsts) > 1:
        bad += 1
    summary = {"procs": a.procs, "runs_per_proc": a.runs, "size": a.size, "net": a.net, "poison": a.poison, "fresh_plans": a.fresh_plans, "hooks": not a.no_hooks, "vary": a.vary,
               "inferences_total": a.procs * a.runs, "workers_with_mismatch": bad, "first_run_output_digests": sorted(str(f) for f in firsts), "seconds": time.time() - t0, "workers": reports}
    print("STRESS " + json.dumps(summary))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
