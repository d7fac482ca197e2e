# Round 2, GPU call 6: specialised (matrix wave + staging wave per SIMD) Winograd kernel vs the two earlier implementations
O=gpurun_out/r2c6; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "wino2d" 2>&1 | tail -5 > $O/gpu_tests.txt
for impl in s p d; do MONAI_AMD_W2_IMPL=$impl KB_BATCH=64 WB_SKIP_SPLIT=1 WB_SKIP_DIRECT=1 python tools/wino_bench.py > $O/wino_bench_$impl.json 2> $O/wino_bench_$impl.err; done
python bench.py --steps 3 --warmup 1 > $O/bench_s.json 2> $O/bench_s.err
cat $O/gpu_tests.txt; cat $O/wino_bench_s.json $O/wino_bench_p.json $O/wino_bench_d.json; cut -c1-1500 $O/bench_s.json; tail -3 $O/bench_s.err
