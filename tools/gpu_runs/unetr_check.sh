# attention / UNETR on the MI355X: kernel cases, the UNETR e2e cases, the UNETR bench line with its kernel trace; writes gpurun_out/unetr/*
O=gpurun_out/unetr; mkdir -p $O; export TMPDIR=/tmp
python -m monai_amd.build > /dev/null
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -q -n 2 -k "attention or unetr" 2>&1 | tail -6 > $O/tests.txt; cat $O/tests.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o b -- python bench.py --net unetr --steps 2 --warmup 1 --cpu-windows 0 > $O/bench_line_unetr.json 2> $O/err.txt
find $O/prof -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/unetr_kernel_trace_stats.txt 2>&1; rm -rf $O/prof
head -14 $O/unetr_kernel_trace_stats.txt | cut -c1-180; tail -1 $O/bench_line_unetr.json | cut -c1-260
