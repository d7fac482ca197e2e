O=gpurun_out/r2c16; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "fp16_split or wino2d" -n 0 2>&1 | tail -4 > $O/gpu_tests_h2.txt
cat $O/gpu_tests_h2.txt
KB_BATCH=64 WB_SKIP_SPLIT=1 WB_SKIP_DIRECT=1 python tools/wino_bench.py > $O/wino_bench.json 2> $O/wino_bench.err
cat $O/wino_bench.json; tail -3 $O/wino_bench.err
