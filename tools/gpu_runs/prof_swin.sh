O=gpurun_out/swin; mkdir -p $O; export TMPDIR=/tmp; rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o b -- python bench.py --net swinunetr --steps 1 --warmup 1 --cpu-windows 0 > $O/line.json 2> $O/err.txt
find $O/prof -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/swin_kernel_trace_stats.txt 2>&1
rm -rf $O/prof
head -24 $O/swin_kernel_trace_stats.txt | cut -c1-165; tail -1 $O/swin_kernel_trace_stats.txt
