# round 6, call 37: per-form dealing in the Winograd split kernel: kernel A/B, its GPU cases, a short headline bench with 125-window parity and the kernel trace
export TMPDIR=/tmp
O=${O:-gpurun_out/r6c37}; mkdir -p $O
timeout 600 python tools/h2w_bench.py > $O/h2w_bench.json 2> $O/h2w_bench.err; cut -c1-1500 $O/h2w_bench.json
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "h2w or accumulating or pooling" 2>&1 | tail -4 | tee $O/gpu_tests.txt
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --cpu-windows 125 --no-spread --no-extra > $O/bench_line.json 2> $O/bench_line.err; cut -c1-400 $O/bench_line.json; tail -3 $O/bench_line.err
MONAI_AMD_CONV_ALGO=h2 timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --cpu-windows 0 --no-spread --no-extra --no-pmc > $O/bench_line_direct.json 2> $O/bench_line_direct.err; cut -c1-400 $O/bench_line_direct.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -o bench -- python $OLDPWD/bench.py --steps 5 --warmup 2 --cpu-windows 0 --no-extra --no-pmc ) > $O/bench_line_traced.json 2> $O/trace.err
find $O/prof -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/bench_kernel_trace_stats.txt 2>&1; rm -rf $O/prof
head -12 $O/bench_kernel_trace_stats.txt | cut -c1-175
