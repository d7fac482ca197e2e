mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "wino2d" --tb=short -x 2>&1 | tail -1
export WB_SKIP_DIRECT=1 KB_BATCH=64
python tools/wino_bench.py 2>&1 | tail -1 | cut -c1-420
for v in variants/*.so; do MH_LIB=$v python tools/wino_bench.py 2>&1 | tail -1 | cut -c1-420; done
