# round 5, call 1: (1) the whole -m gpu suite on the round's first tree (the last commits of round 4 were not fully GPU-verified; h2z removed, bench line rewritten),
# (2) the stream-lanes A/B on the headline workload (tools/streams_ab.py)
export TMPDIR=/tmp
O=gpurun_out/r5c1; mkdir -p $O
( time timeout 420 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > $O/gpu_tests.txt ) 2> $O/gpu_tests.time; cat $O/gpu_tests.txt; tail -3 $O/gpu_tests.time
timeout 240 python tools/streams_ab.py --out $O/streams_ab.json 2>&1 | grep -v "^BasicUNet" | tee $O/streams_ab.txt
