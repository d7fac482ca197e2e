mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -k "wino2d" 2>&1 | tail -3
python tools/wino_bench.py 2>&1 | tail -1 | tee gpurun_out/wino_bench.log
