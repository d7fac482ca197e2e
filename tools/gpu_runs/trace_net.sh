# kernel trace of bench.py --net $NET (default dynunet): per-kernel summary into gpurun_out/trace_$NET/
export TMPDIR=/tmp
NET=${NET:-dynunet}; O=$PWD/gpurun_out/trace_$NET; mkdir -p $O
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $OLDPWD/bench.py --net $NET --steps 3 --warmup 1 --cpu-windows 0 --no-extra --no-pmc --no-spread ) > $O/bench_line.json 2> $O/trace.err
find $O/prof -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/kernel_trace_stats.txt 2>&1; rm -rf $O/prof
grep -o '"ms_per_step": [0-9.]*' $O/bench_line.json | head -1; head -24 $O/kernel_trace_stats.txt | cut -c1-165
