# round 4, final validation: the whole -m gpu suite + smoke(), the driver's bench command (timed), the headline under rocprofv3 --kernel-trace --stats
export TMPDIR=/tmp
O=gpurun_out/r4final; mkdir -p $O
python -m monai_amd.build > /dev/null
( time timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/gpu_tests.txt ) 2> $O/gpu_tests.time; tail -4 $O/gpu_tests.txt; tail -3 $O/gpu_tests.time
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 > $O/smoke.txt; cat $O/smoke.txt
( time timeout 1700 python bench.py > $O/bench_line.json 2> $O/bench_line.err ) 2> $O/bench_line.time; tail -3 $O/bench_line.time
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4final/bench_line.json").read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], "value", d["value"], "roofline.frac", d["roofline"]["frac"], "hbm", d["roofline_hbm"]["frac"], d["roofline_hbm"].get("frac_of_copy_ceiling"))
    c = d["cpu_baseline"]; print("cpu", c["value"], c["cores"], {k: v for k, v in c["parity_vs_gpu"].items() if k != "compared"}); print(c["sample"])
    for k, v in d.get("extra", {}).items():
        print(k, v.get("ms_per_step"), v.get("parity_vs_cpu_oracle", v.get("parity_vs_cpu_restatement")), json.dumps(v.get("cpu_baseline"))[:400], v.get("error"))
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/r4final/bench_line.err").read()[-3000:])
PY
timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 4 --warmup 1 --cpu-windows 0 --no-extra > $O/prof_bench.json 2> $O/prof_bench.err
find $O/prof -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/bench_kernel_trace_stats.txt 2>&1; rm -rf $O/prof
head -22 $O/bench_kernel_trace_stats.txt | cut -c1-175
