"""bench.py's own code path without a GPU: the SIMT-emulator build of the kernels + gloo (MONAI_AMD_BENCH_EMULATOR=1), tiny sizes.
Checks the wiring the driver relies on -- `python bench.py` and `python -m torch.distributed.run ... bench.py --gpus N` print exactly
one JSON line on rank 0 with the contract's keys, the `roofline` / `roofline_hbm` / `cpu_baseline` objects, the enforced parity rule --
and that sharding the windows over two ranks reproduces the single-process result bit for bit (same checksum).  Not a measurement."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--size", "40", "--roi", "32", "--steps", "1", "--warmup", "0", "--harness-features", "16,16,32,32,64,16"]      # quarter widths: the SIMT emulator pays for every flop
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
        "roofline", "roofline_hbm", "cpu_baseline"}


def _run(cmd):
    # exact-fp32 convolutions: the fp16 split-precision default is an order of magnitude slower to emulate (its parity is the GPU suite's job)
    env = dict(os.environ, MONAI_AMD_BENCH_EMULATOR="1", MONAI_AMD_CONV_ALGO="fp32", OMP_NUM_THREADS="4", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_single_process_line():
    line = _run([sys.executable, "bench.py", "--cpu-windows", "2"] + ARGS)
    assert KEYS <= set(line) and line["n_gpus"] == 1 and line["unit"] == "voxels/s" and line["dtype"] == "f32" and line["emulated"]
    assert line["roofline"]["bound"] == "mfma" and 0.0 < line["roofline"]["frac"] and line["roofline"]["unit"] == "TFLOP/s"
    assert line["roofline_hbm"]["bound"] == "hbm" and line["roofline_hbm"]["peak"] == 8000.0
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    par = cb["parity_vs_gpu"]
    assert par["ok"] and par["mismatch_outside_margin"] == 0 and par["max_abs_logit_diff"] <= 1e-4
    test_bench_single_process_line.checksum = line["checksum"]


def test_bench_two_ranks_line_and_bitwise_sharding():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    line = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                 "bench.py", "--gpus", "2"] + ARGS)
    assert KEYS <= set(line) and line["n_gpus"] == 2 and line["scaling"] == "strong" and line["cpu_baseline"] is None
    assert "sharded over 2" in line["config"]["parallelism"]
    single = getattr(test_bench_single_process_line, "checksum", None)
    if single is None:
        single = _run([sys.executable, "bench.py", "--cpu-windows", "0"] + ARGS)["checksum"]
    assert line["checksum"] == single          # the replicated deterministic blend: sharded == unsharded, bit for bit


def test_bench_four_ranks_line_and_bitwise_sharding():
    """the driver's `--gpus 4` launch of SCALE_rNN: four ranks, one JSON line from rank 0, per-rank breakdown of all four, checksum == single process"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    line = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1", "--master-port", str(port),
                 "bench.py", "--gpus", "4"] + ARGS)
    assert KEYS <= set(line) and line["n_gpus"] == 4 and line["scaling"] == "strong" and line["cpu_baseline"] is None
    assert "sharded over 4" in line["config"]["parallelism"] and [r["rank"] for r in line["per_rank_ms_per_step"]] == [0, 1, 2, 3]
    single = getattr(test_bench_single_process_line, "checksum", None)
    if single is None:
        single = _run([sys.executable, "bench.py", "--cpu-windows", "0"] + ARGS)["checksum"]
    assert line["checksum"] == single
