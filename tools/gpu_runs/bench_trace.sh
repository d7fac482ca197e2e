# the driver's bench command (default flags: headline + cpu_baseline on 125 windows + extra.fp32_exact / config3 / config4), then the same headline under
# rocprofv3 --kernel-trace --stats (per-kernel summary by tools/rocpd_stats.py); writes gpurun_out/bench/*   [STEPS=5 by default]
O=gpurun_out/bench; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py --gpus 1 --steps ${STEPS:-5} --warmup 2 > $O/bench_line.json 2> $O/bench_line.err; cut -c1-400 $O/bench_line.json; tail -3 $O/bench_line.err
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 3 --warmup 1 --cpu-windows 0 --no-extra > $O/bench_line_traced.json 2> $O/trace.err
find $O/prof -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/bench_kernel_trace_stats.txt 2>&1; rm -rf $O/prof
head -16 $O/bench_kernel_trace_stats.txt | cut -c1-175
