O=gpurun_out/r2c12; mkdir -p $O; export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f16.hip -o /tmp/mfma_f16 2>/dev/null && /tmp/mfma_f16 > $O/ubench_mfma_f16.txt 2>&1
cat $O/ubench_mfma_f16.txt
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/gpu_tests.txt
tail -5 $O/gpu_tests.txt
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py > $O/bench_line.json 2> $O/bench_line.err
find $O/prof -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/bench_kernel_trace_stats.txt 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
rm -rf $O/prof
head -30 $O/bench_kernel_trace_stats.txt | cut -c1-200
cut -c1-300 $O/bench_line.json
bash tools/gpu_runs/pmc.sh > $O/pmc.log 2>&1
cp gpurun_out/pmc_fetch_stats.txt gpurun_out/pmc_write_stats.txt $O/
tail -12 $O/pmc.log | cut -c1-200
