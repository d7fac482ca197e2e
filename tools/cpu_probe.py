"""What the GPU box's host gives the CPU-oracle leg of bench.py: visible threads (cpuset vs machine), and how N concurrent 32-thread oracle workers scale
(one BasicUNet window each, started together) -- decides whether oracle/parallel_predict.py's pool is worth its processes.  Test infrastructure."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--worker":
    import torch

    import oracle

    torch.set_num_threads(int(sys.argv[2]))
    torch.manual_seed(1)
    sd = oracle.make_basic_unet_state(1, 5)
    x = torch.rand(2, 1, 96, 96, 96)
    with torch.no_grad():
        oracle.basic_unet_forward(sd, x[:1])
        t0 = time.perf_counter()
        for _ in range(int(sys.argv[3])):
            oracle.basic_unet_forward(sd, x)
        print("WORKER", (time.perf_counter() - t0) / (2 * int(sys.argv[3])), flush=True)
    sys.exit(0)

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
print(subprocess.run("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA node' | head -8", shell=True, capture_output=True, text=True).stdout)
cfgs = [(int(a), int(b)) for a, b in (c.split("x") for c in os.environ.get("PROBE", "16x1,8x2,4x4,2x8,16x2,8x4,4x8,8x8").split(","))]      # threads x procs
for threads, procs in cfgs:
    t0 = time.perf_counter()
    ps = [subprocess.Popen([sys.executable, __file__, "--worker", str(threads), "2"], stdout=subprocess.PIPE, text=True) for _ in range(procs)]
    per = [float(next(ln for ln in p.communicate()[0].splitlines() if ln.startswith("WORKER")).split()[1]) for p in ps]
    print(f"{procs} x {threads} threads: {sum(per) / len(per):.3f} s/window per worker -> {procs / (sum(per) / len(per)):.2f} windows/s aggregate (wall {time.perf_counter() - t0:.1f} s)", flush=True)
