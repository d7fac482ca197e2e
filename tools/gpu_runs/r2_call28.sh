O=gpurun_out/r2c28; mkdir -p $O; export TMPDIR=/tmp
python bench.py --steps 5 --warmup 2 > $O/bench_line.json 2> $O/bench_line.err
cut -c1-1500 $O/bench_line.json; tail -2 $O/bench_line.err
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/gpu_tests.txt
tail -8 $O/gpu_tests.txt
KB_BATCH=64 WB_SKIP_SPLIT=1 python tools/wino_bench.py > $O/wino_bench.json 2> $O/wino_bench.err
cat $O/wino_bench.json
