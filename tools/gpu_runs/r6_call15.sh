# round 6, call 15: the 8-wave two-phase form of the in-plane Winograd split-precision convolution (conv3d_wino_h2.h) on the MI355X: its kernel cases, then the A/B against the direct kernel
export TMPDIR=/tmp
O=gpurun_out/r6c29; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "h2w" 2>&1 | tail -5 | tee $O/gpu_tests.txt
timeout 900 python tools/h2w_bench.py > $O/h2w_bench.json 2> $O/h2w_bench.err; cat $O/h2w_bench.json; tail -3 $O/h2w_bench.err
