# end-of-round verification on a fresh box: what the driver runs (pytest -m gpu, smoke, bench) + the kernel trace of the bench command
O=gpurun_out/r2verify; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/gpu_tests.txt; tail -3 $O/gpu_tests.txt
python __graft_entry__.py smoke 2>&1 | tail -2 > $O/smoke.txt; cat $O/smoke.txt
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py > $O/bench_line.json 2> $O/bench_line.err
find $O/prof -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/bench_kernel_trace_stats.txt 2>&1
rm -rf $O/prof
head -12 $O/bench_kernel_trace_stats.txt | cut -c1-150; cut -c1-330 $O/bench_line.json
