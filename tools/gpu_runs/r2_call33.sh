O=gpurun_out/r2c33; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/gpu_tests.txt
tail -4 $O/gpu_tests.txt
python bench.py --steps 5 --warmup 2 --cpu-windows 0 > $O/bench_line.json 2> $O/bench_line.err
cut -c1-400 $O/bench_line.json
KB_BATCH=64 WB_SKIP_SPLIT=1 WB_SKIP_DIRECT=1 python tools/wino_bench.py > $O/wino_bench.json 2> $O/wino_bench.err
cat $O/wino_bench.json
