set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu.log 2>&1
python tools/transform_bench.py > gpurun_out/transform_bench.json 2> gpurun_out/transform_bench.err
python bench.py --steps 2 --warmup 1 > gpurun_out/bench.log 2>&1
tail -6 gpurun_out/pytest_gpu.log; tail -1 gpurun_out/bench.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline'])"; python -c "
import json;r=json.load(open('gpurun_out/transform_bench.json'))
for x in r['runs']: print(x['op'], round(x['ms'],3),'ms', round(x['GBps'],1),'GB/s')"
