mkdir -p gpurun_out
python -m pytest tests/test_transforms_gpu.py -m gpu -q --tb=short 2>&1 | tail -3
python tools/pushpull_bench.py 2>&1 | tail -1 | tee gpurun_out/pushpull_bench.json
