mkdir -p gpurun_out
python -m pytest tests/test_transforms_gpu.py -m gpu -q --tb=short -x > gpurun_out/pytest_transforms.log 2>&1; grep -v "^Extension modules" gpurun_out/pytest_transforms.log | tail -25
python tools/pushpull_bench.py 2>&1 | tail -1 | tee gpurun_out/pushpull_bench.json | cut -c1-200
