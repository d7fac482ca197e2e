mkdir -p gpurun_out
python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu.log 2>&1
grep -v "^Extension modules" gpurun_out/pytest_gpu.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
WB_SKIP_DIRECT=1 KB_BATCH=64 WB_FIRST=1 python tools/wino_bench.py 2>&1 | tail -1 | cut -c1-300
