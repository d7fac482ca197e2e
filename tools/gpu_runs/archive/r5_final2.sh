# round 5, second session's final tree: the whole -m gpu suite + smoke(), then the driver's bench command (whole-volume CPU-oracle leg, reference self-spread, in-run counters, extras)
export TMPDIR=/tmp
O=gpurun_out/r5final2; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -30 > $O/gpu_tests.txt; grep "^E  .*assert\|^E  .*Error" $O/gpu_tests.txt | head -5; tail -3 $O/gpu_tests.txt
python __graft_entry__.py smoke 2>&1 | tail -2 | tee $O/smoke.txt
timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 > $O/bench_line.json 2> $O/bench_line.err; cut -c1-600 $O/bench_line.json; tail -3 $O/bench_line.err
