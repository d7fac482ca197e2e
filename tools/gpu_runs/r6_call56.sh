# round 6, call 56: the tree after the composite-kernel pipeline, the UpCat order and the two-launch 64-channel convolution: whole -m gpu suite, smoke, the driver's bench command, its kernel trace
export TMPDIR=/tmp
O=${O:-gpurun_out/r6c56}; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/gpu_tests.txt
python __graft_entry__.py smoke 2>&1 | tail -2 | tee $O/smoke.txt
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; wc -c $O/bench_line.json; cut -c1-700 $O/bench_line.json; tail -3 $O/bench_line.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -o bench -- python $OLDPWD/bench.py --steps 5 --warmup 2 --cpu-windows 0 --no-extra --no-pmc ) > $O/bench_line_traced.json 2> $O/trace.err
find $O/prof -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/bench_kernel_trace_stats.txt 2>&1; rm -rf $O/prof
head -16 $O/bench_kernel_trace_stats.txt | cut -c1-175
