# round 6, call 26: the Winograd split-precision convolution selected by default: e2e + kernel GPU tests, then a short headline bench with a 125-window parity check and its kernel trace
export TMPDIR=/tmp
O=${O:-gpurun_out/r6c26}; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -x -q -m gpu 2>&1 | tail -8 | tee $O/gpu_tests.txt
timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 --cpu-windows 125 --no-spread --no-extra > $O/bench_line.json 2> $O/bench_line.err; cut -c1-1800 $O/bench_line.json; tail -3 $O/bench_line.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -o bench -- python $OLDPWD/bench.py --steps 5 --warmup 2 --cpu-windows 0 --no-extra --no-pmc ) > $O/bench_line_traced.json 2> $O/trace.err
find $O/prof -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/bench_kernel_trace_stats.txt 2>&1; rm -rf $O/prof
head -14 $O/bench_kernel_trace_stats.txt | cut -c1-175
