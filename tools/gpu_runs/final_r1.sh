# Round-end measurement run (trimmed): tests, smoke, headline bench (+ rocprofv3 kernel trace), the other configs, transforms.
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu.log 2>&1
grep -v "^Extension modules" gpurun_out/pytest_gpu.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
python bench.py > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-260
rm -rf gpurun_out/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o b -- python bench.py --steps 2 --warmup 1 --cpu-windows 0 > gpurun_out/prof.log 2>&1
find gpurun_out/prof -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > gpurun_out/bench_kernel_trace_stats.txt 2>&1
find gpurun_out -name "*.db" -delete
python bench.py --steps 2 --warmup 1 --net unetr --cpu-windows 0 > gpurun_out/bench_unetr.log 2>&1; tail -1 gpurun_out/bench_unetr.log | cut -c100-230
python bench.py --steps 2 --warmup 1 --net unet --cpu-windows 0 > gpurun_out/bench_unet.log 2>&1; tail -1 gpurun_out/bench_unet.log | cut -c100-230
python tools/transform_bench.py > gpurun_out/transform_bench.json 2> gpurun_out/transform_bench.err
python tools/pushpull_bench.py 2>/dev/null | tail -1 > gpurun_out/pushpull_bench.json
head -8 gpurun_out/bench_kernel_trace_stats.txt | cut -c1-170
python -c "
import json;r=json.load(open('gpurun_out/transform_bench.json'))
print(r['device_copy'])
for x in r['runs']: print(x['op'], round(x['ms'],3),'ms', round(x['GBps'],1),'GB/s', round(x['frac_of_device_copy'],2))"
