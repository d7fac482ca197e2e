# round 4: kernel trace of one UNETR step (after the statistics moved into the shortcut convolution)
export TMPDIR=/tmp
O=gpurun_out/r4x1; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o b -- python bench.py --net unetr --steps 1 --warmup 1 --cpu-windows 0 --no-extra > $O/bench_prof.json 2> $O/err.txt
find $O/prof -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/unetr_kernel_trace_stats.txt 2>&1; rm -rf $O/prof
head -16 $O/unetr_kernel_trace_stats.txt | cut -c1-170
