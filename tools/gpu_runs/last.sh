python -m pytest tests/test_e2e_gpu.py tests/test_abi.py -m "gpu or not gpu" -q --tb=short -k "patch or abi or slabwise" 2>&1 | grep -v "^Extension" | tail -4
