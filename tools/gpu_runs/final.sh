# Round-end measurement run: tests, smoke, headline bench (+ rocprofv3 kernel trace), the other configs, kernel sweeps,
# HBM-traffic PMC passes, micro-benchmarks.  Everything lands in gpurun_out/; the summaries are copied to profiles/ by hand.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu.log 2>&1
tail -3 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
python bench.py > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-600
rm -rf gpurun_out/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o b -- python bench.py --steps 2 --warmup 1 --cpu-windows 0 > gpurun_out/prof.log 2>&1
find gpurun_out/prof -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > gpurun_out/bench_kernel_trace_stats.txt 2>&1
python bench.py --steps 2 --warmup 1 --net unetr --cpu-windows 0 > gpurun_out/bench_unetr.log 2>&1; tail -1 gpurun_out/bench_unetr.log | cut -c1-300
python bench.py --steps 2 --warmup 1 --net unet --cpu-windows 0 > gpurun_out/bench_unet.log 2>&1; tail -1 gpurun_out/bench_unet.log | cut -c1-300
python tools/transform_bench.py > gpurun_out/transform_bench.json 2> gpurun_out/transform_bench.err
KB_BATCH=64 python tools/kernel_bench.py > gpurun_out/kernel_bench.json 2> gpurun_out/kernel_bench.err
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f -- python tools/pmc_probe.py > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w -- python tools/pmc_probe.py > gpurun_out/pmc_write.log 2>&1
find gpurun_out/pmc_fetch -name "*.db" | head -1 | xargs -I{} python tools/pmc_stats.py {} "%mh::%" > gpurun_out/pmc_fetch_stats.txt 2>&1
find gpurun_out/pmc_write -name "*.db" | head -1 | xargs -I{} python tools/pmc_stats.py {} "%mh::%" > gpurun_out/pmc_write_stats.txt 2>&1
find gpurun_out -name "*.db" -delete
(cd tools/ubench && hipcc --offload-arch=gfx950 -O3 -Wno-unused-value issue.hip -o /tmp/issue && /tmp/issue && hipcc --offload-arch=gfx950 -O3 -Wno-unused-value coexec.hip -o /tmp/coexec && /tmp/coexec) > gpurun_out/ubench.txt 2>&1
head -12 gpurun_out/bench_kernel_trace_stats.txt | cut -c1-180
python -c "
import json;r=json.load(open('gpurun_out/transform_bench.json'))
for x in r['runs']: print(x['op'], round(x['ms'],3),'ms', round(x['GBps'],1),'GB/s')"
