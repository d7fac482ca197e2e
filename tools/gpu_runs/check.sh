mkdir -p gpurun_out
python -m pytest tests -m gpu -q --tb=short -x > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
python bench.py > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log > gpurun_out/bench_line.json; python -c "
import json;l=json.load(open('gpurun_out/bench_line.json'));print(l['value'],l['ms_per_step']);print(l['cpu_baseline'])"
