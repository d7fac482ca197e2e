# Round 2, GPU call 5: de-phased two-wave Winograd kernel vs round-1 kernel; class-stride probe of the blend shape
O=gpurun_out/r2c5; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "wino2d" 2>&1 | tail -5 > $O/gpu_tests.txt
KB_BATCH=64 WB_SKIP_SPLIT=1 WB_SKIP_DIRECT=1 python tools/wino_bench.py > $O/wino_bench_p.json 2> $O/wino_bench_p.err
MONAI_AMD_W2_IMPL=d KB_BATCH=64 WB_SKIP_SPLIT=1 WB_SKIP_DIRECT=1 python tools/wino_bench.py > $O/wino_bench_d.json 2> $O/wino_bench_d.err
( cd tools/ubench && hipcc --offload-arch=gfx950 -O3 -Wno-unused-value hbm_stream.hip -o /tmp/hbm_stream && /tmp/hbm_stream ) > $O/hbm_stream.txt 2>&1
cat $O/gpu_tests.txt; cat $O/wino_bench_p.json $O/wino_bench_d.json; tail -9 $O/hbm_stream.txt
