# one-input-channel convolution + packed transposed convolution on the MI355X: their kernel cases, the bench line with its kernel trace,
# the A/B line with both switched off, then the whole -m gpu suite and smoke() (what the driver runs at round end)
O=gpurun_out/r2c1; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -n 0 -k "one_input_channel or pool_deconv" 2>&1 | tail -15 > $O/new_kernel_tests.txt; tail -3 $O/new_kernel_tests.txt
rm -rf $O/prof
timeout 420 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py > $O/bench_line.json 2> $O/bench_line.err
find $O/prof -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/bench_kernel_trace_stats.txt 2>&1
rm -rf $O/prof
head -14 $O/bench_kernel_trace_stats.txt | cut -c1-150; cut -c1-330 $O/bench_line.json
MONAI_AMD_C1=0 MONAI_AMD_DECONV_IMPL=scalar timeout 200 python bench.py --steps 3 --warmup 1 --cpu-windows 0 > $O/bench_line_c1off_deconvscalar.json 2> $O/bench_line_ab.err
cut -c1-260 $O/bench_line_c1off_deconvscalar.json
timeout 200 python bench.py --steps 3 --warmup 1 --cpu-windows 0 > $O/bench_line_default_nocpu.json 2>> $O/bench_line_ab.err
cut -c1-260 $O/bench_line_default_nocpu.json
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/gpu_tests.txt; tail -3 $O/gpu_tests.txt
python __graft_entry__.py smoke 2>&1 | tail -2 > $O/smoke.txt; cat $O/smoke.txt
