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
ARGS = ["--size", "40", "--roi", "32", "--steps", "1", "--warmup", "0", "--harness-features", "8,8,16,16,32,8"]      # an eighth of the widths: the SIMT emulator pays for every flop
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
    top = line["parity"]            # the same report at top level (the driver's parsed copy keeps it), the line short enough for the driver's stdout tail
    assert top["ok"] and top["family"] == "fp32-exact" and top["voxels"] == par["voxels"] and top["max_abs_logit_diff"] == par["max_abs_logit_diff"]
    assert len(json.dumps(line)) < 6000
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


def test_pmc_inrun_reads_the_counter_databases(tmp_path, monkeypatch):
    """bench.pmc_inrun(): two profiler child processes (FETCH_SIZE, WRITE_SIZE), per-kernel averages out of the rocpd `counters_collection` view, KiB -> bytes, the
    guide's x 2 on FETCH_SIZE for the 16-bytes-per-lane blend only, pack / scale helper kernels not mistaken for the convolution -- against a stand-in `rocprofv3`
    that writes the database a real pass would (no GPU here); a failing profiler leaves the committed pass in place."""
    fake = tmp_path / "rocprofv3"
    fake.write_text(f"""#!{sys.executable}
import os, sqlite3, sys
a = sys.argv[1:]
d, counter = a[a.index("-d") + 1], a[a.index("--pmc") + 1]
assert "--kernel-trace" in a and a[a.index("--") + 2].endswith("pmc_probe.py")
if os.environ.get("FAKE_PMC_FAIL"):
    sys.exit(3)
os.makedirs(os.path.join(d, "host"), exist_ok=True)
db = sqlite3.connect(os.path.join(d, "host", "p_results.db"))
db.execute("create table counters_collection (kernel_name text, counter_name text, value real)")
kib = {{"FETCH_SIZE": (9000000.0, 7000000.0), "WRITE_SIZE": (2621440.0, 7077888.0)}}[counter]
for i in range(3):
    db.execute("insert into counters_collection values (?, ?, ?)", ("void mh::sw_blend_mosaic_kernel<5, 2>(mh::Mosaic)", counter, kib[0] + i - 1))
    db.execute("insert into counters_collection values (?, ?, ?)", ("void mh::conv3d_k3_h2w_kernel<true, false, false>(mh::Tensor)", counter, kib[1]))
db.execute("insert into counters_collection values (?, ?, ?)", ("mh::conv3d_k3_h2w_pack_kernel(float const*)", counter, 1.0))
db.execute("insert into counters_collection values (?, ?, ?)", ("mh::conv3d_k3_h2w_scale_fix_kernel(float*)", counter, 1.0))
db.execute("insert into counters_collection values (?, ?, ?)", ("mh::conv3d_k3_h2_scale_kernel(float const*)", counter, 1.0))
db.commit()
""")
    fake.chmod(0o755)
    sys.path.insert(0, ROOT)
    import bench

    monkeypatch.setattr("shutil.which", lambda name: str(fake) if name == "rocprofv3" else None)
    got = bench.pmc_inrun(budget_s=60)
    blend, conv = got["sw_blend_mosaic_kernel"], got["conv3d_k3_h2w_kernel"]      # the configuration the selector gives 32 -> 32 channels at 96^3 (bench._pmc_conv_key)
    assert blend["fetch_bytes"] == 2 * 9000000.0 * 1024 and blend["write_bytes"] == 2621440.0 * 1024
    assert conv["fetch_bytes"] == 7000000.0 * 1024 and conv["hbm_bytes_per_launch"] == (7000000.0 + 7077888.0) * 1024
    assert abs(blend["ratio"] - (2 * 9000000.0 + 2621440.0) * 1024 / (1000 * 5 * 96 ** 3 * 4 + 5 * 512 ** 3 * 4)) < 1e-12
    assert blend["measured"] == "in_run" and blend["fetch_calibrated"] and not conv["fetch_calibrated"] and bench._PMC_STATUS["status"] == "in_run"
    monkeypatch.setenv("FAKE_PMC_FAIL", "1")
    assert bench.pmc_inrun(budget_s=60) == {} and bench._PMC_STATUS["status"].startswith("failed: FETCH_SIZE: rocprofv3 rc 3")
    assert bench.pmc_traffic("sw_blend_mosaic_kernel")["measured"].startswith("from_file")
