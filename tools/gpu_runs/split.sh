mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -x -k "split" 2>&1 | grep -v "^Extension" | tail -4
WB_SKIP_DIRECT=1 KB_BATCH=64 python tools/wino_bench.py 2>&1 | tail -1 | tee gpurun_out/wb_split.json | cut -c1-900
