mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f -- python tools/pmc_probe.py > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w -- python tools/pmc_probe.py > gpurun_out/pmc_write.log 2>&1
find gpurun_out/pmc_fetch -name "*.db" | head -1 | xargs -I{} python tools/pmc_stats.py {} "%mh::%" > gpurun_out/pmc_fetch_stats.txt 2>&1
find gpurun_out/pmc_write -name "*.db" | head -1 | xargs -I{} python tools/pmc_stats.py {} "%mh::%" > gpurun_out/pmc_write_stats.txt 2>&1
find gpurun_out -name "*.db" -delete
grep -A1 "h2_kernel\|sw_blend_reg\|sw_blend_mosaic\|rowvec" gpurun_out/pmc_fetch_stats.txt gpurun_out/pmc_write_stats.txt | cut -c1-160
