python -m pytest tests/test_transforms_gpu.py -m gpu -q --tb=short -k "post" 2>&1 | grep -v "^Extension" | tail -4
