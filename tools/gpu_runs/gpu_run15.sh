set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --tb=short -x > gpurun_out/pytest_gpu.log 2>&1
tail -4 gpurun_out/pytest_gpu.log
python bench.py --steps 2 --warmup 1 > gpurun_out/bench.log 2>&1
tail -1 gpurun_out/bench.log | cut -c1-1800
MONAI_AMD_CONV_ALGO=direct python bench.py --steps 2 --warmup 1 --cpu-windows 0 > gpurun_out/bench_direct.log 2>&1
tail -1 gpurun_out/bench_direct.log | cut -c1-400
