mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_unetr
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_unetr -o u -- python bench.py --steps 1 --warmup 1 --net unetr --cpu-windows 0 > gpurun_out/prof_unetr.log 2>&1
find gpurun_out/prof_unetr -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > gpurun_out/prof_unetr_stats.txt 2>&1
find gpurun_out/prof_unetr -name "*.db" -delete
head -32 gpurun_out/prof_unetr_stats.txt | cut -c1-150
