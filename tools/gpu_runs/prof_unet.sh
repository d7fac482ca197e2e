mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_unet
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_unet -o u -- python bench.py --steps 1 --warmup 1 --net unet --cpu-windows 0 > gpurun_out/prof_unet.log 2>&1
find gpurun_out/prof_unet -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > gpurun_out/prof_unet_stats.txt 2>&1
find gpurun_out/prof_unet -name "*.db" -delete
head -18 gpurun_out/prof_unet_stats.txt | cut -c1-165
