set -x
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke.log 2>&1
python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu.log 2>&1
python tools/kernel_bench.py > gpurun_out/kernel_bench.json 2> gpurun_out/kernel_bench.err
python tools/transform_bench.py > gpurun_out/transform_bench.json 2> gpurun_out/transform_bench.err
python bench.py --steps 2 --warmup 1 > gpurun_out/bench.log 2>&1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r1b -- python bench.py --steps 1 --warmup 1 --cpu-windows 0 > gpurun_out/prof.log 2>&1
tail -3 gpurun_out/smoke.log; tail -8 gpurun_out/pytest_gpu.log; tail -1 gpurun_out/bench.log | cut -c1-900; cat gpurun_out/transform_bench.json
