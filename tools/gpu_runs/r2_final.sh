# round-2 measurement run: bench + rocprofv3 kernel trace of the same command, PMC HBM traffic (separate passes), the other networks, transforms
O=gpurun_out/r2final; mkdir -p $O; export TMPDIR=/tmp
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py > $O/bench_line.json 2> $O/bench_line.err
find $O/prof -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/bench_kernel_trace_stats.txt 2>&1
rm -rf $O/prof
head -25 $O/bench_kernel_trace_stats.txt | cut -c1-170
cut -c1-300 $O/bench_line.json
for net in unet unetr dynunet segresnet swinunetr; do
  timeout 600 python bench.py --net $net --steps 3 --warmup 1 --cpu-windows 0 > $O/bench_line_$net.json 2> $O/bench_line_$net.err
  cut -c1-260 $O/bench_line_$net.json
done
MONAI_AMD_CONV_ALGO=fp32 python bench.py --steps 3 --warmup 1 --cpu-windows 0 > $O/bench_line_fp32.json 2> $O/bench_line_fp32.err
cut -c1-260 $O/bench_line_fp32.json
python tools/transform_bench.py > $O/transform_bench.json 2> $O/transform_bench.err
python tools/preproc_bench.py > $O/preproc_bench.json 2> $O/preproc_bench.err
bash tools/gpu_runs/pmc.sh > $O/pmc.log 2>&1
cp gpurun_out/pmc_fetch_stats.txt gpurun_out/pmc_write_stats.txt $O/
tail -8 $O/pmc.log | cut -c1-200
