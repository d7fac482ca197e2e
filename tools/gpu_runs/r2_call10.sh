O=gpurun_out/r2c10; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/gpu_tests.txt
for impl in p d; do MONAI_AMD_W2_IMPL=$impl KB_BATCH=64 WB_SKIP_SPLIT=1 WB_SKIP_DIRECT=1 python tools/wino_bench.py > $O/wino_bench_$impl.json 2> $O/wino_bench_$impl.err; done
python bench.py --steps 5 --warmup 2 --cpu-windows 0 > $O/bench_line.json 2> $O/bench_line.err
cat $O/gpu_tests.txt; cat $O/wino_bench_p.json $O/wino_bench_d.json; cut -c1-400 $O/bench_line.json
