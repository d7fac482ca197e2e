set -x
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke.log 2>&1
python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu.log 2>&1
python tools/transform_bench.py > gpurun_out/transform_bench.json 2> gpurun_out/transform_bench.err
python bench.py --steps 2 --warmup 1 > gpurun_out/bench.log 2>&1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f -- python tools/pmc_probe.py > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w -- python tools/pmc_probe.py > gpurun_out/pmc_write.log 2>&1
tail -3 gpurun_out/smoke.log; tail -8 gpurun_out/pytest_gpu.log; tail -1 gpurun_out/bench.log | cut -c1-1200; python -c "
import json;r=json.load(open('gpurun_out/transform_bench.json'))
for x in r['runs']: print(x['op'], round(x['ms'],3),'ms', round(x['GBps'],1),'GB/s')"
