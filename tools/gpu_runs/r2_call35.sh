O=gpurun_out/r2c35; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/gpu_tests.txt
tail -6 $O/gpu_tests.txt
for net in unetr swinunetr; do
  timeout 600 python bench.py --net $net --steps 3 --warmup 1 --cpu-windows 0 > $O/bench_line_$net.json 2> $O/bench_line_$net.err
  cut -c1-260 $O/bench_line_$net.json; tail -2 $O/bench_line_$net.err
done
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o b -- python bench.py --net unetr --steps 2 --warmup 1 --cpu-windows 0 > /dev/null 2> $O/prof_unetr.err
find $O/prof -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/unetr_kernel_trace_stats.txt 2>&1
rm -rf $O/prof
head -16 $O/unetr_kernel_trace_stats.txt | cut -c1-160
