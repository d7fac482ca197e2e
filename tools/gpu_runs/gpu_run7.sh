set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu.log 2>&1
python tools/transform_bench.py > gpurun_out/transform_bench.json 2> gpurun_out/transform_bench.err
python bench.py --steps 2 --warmup 1 --net unet --cpu-windows 0 > gpurun_out/bench_unet.log 2>&1
tail -12 gpurun_out/pytest_gpu.log; tail -1 gpurun_out/bench_unet.log | cut -c1-2500; python -c "
import json;r=json.load(open('gpurun_out/transform_bench.json'))
for x in r['runs']: print(x['op'], round(x['ms'],3),'ms', round(x['GBps'],1),'GB/s')"
