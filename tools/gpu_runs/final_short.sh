mkdir -p gpurun_out
python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu.log 2>&1; tail -2 gpurun_out/pytest_gpu.log
python bench.py > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-200
python bench.py --steps 2 --warmup 1 --net unetr --cpu-windows 0 > gpurun_out/bench_unetr.log 2>&1; tail -1 gpurun_out/bench_unetr.log | cut -c100-230
python bench.py --steps 2 --warmup 1 --net unet --cpu-windows 0 > gpurun_out/bench_unet.log 2>&1; tail -1 gpurun_out/bench_unet.log | cut -c100-230
python tools/transform_bench.py > gpurun_out/transform_bench.json 2> gpurun_out/transform_bench.err
python -c "
import json;r=json.load(open('gpurun_out/transform_bench.json'))
print(r['device_copy'])
for x in r['runs']: print(x['op'], round(x['ms'],3),'ms', round(x['GBps'],1),'GB/s', round(x['frac_of_device_copy'],2))"
