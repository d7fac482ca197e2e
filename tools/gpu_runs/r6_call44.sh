# round 6, call 44 (final tree): the Winograd split-precision convolution (staging and finishing in the matrix phase, new Z layout, accumulating form with its loads ahead of the stores,
# pooling form by v_permlane32_swap): kernel A/B, the whole -m gpu suite, the driver's bench command with every extra, its kernel trace
export TMPDIR=/tmp
O=${O:-gpurun_out/r6c44}; mkdir -p $O
timeout 600 python tools/h2w_bench.py > $O/h2w_bench.json 2> $O/h2w_bench.err; cut -c1-1500 $O/h2w_bench.json
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/gpu_tests.txt
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; wc -c $O/bench_line.json; cut -c1-700 $O/bench_line.json; tail -3 $O/bench_line.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -o bench -- python $OLDPWD/bench.py --steps 5 --warmup 2 --cpu-windows 0 --no-extra --no-pmc ) > $O/bench_line_traced.json 2> $O/trace.err
find $O/prof -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/bench_kernel_trace_stats.txt 2>&1; rm -rf $O/prof
head -14 $O/bench_kernel_trace_stats.txt | cut -c1-175
