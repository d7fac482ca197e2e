# round 4, last calls (the GPU budget no longer holds the whole suite + the bench in one call): the -m gpu tests of everything that changed after the last full-suite run
# (profiles/r04_gpu_tests_final.txt): every network's end-to-end goldens (16-couts layers now on the split-precision kernel), transforms (Gaussian), kernels
export TMPDIR=/tmp
O=gpurun_out/r4final2; mkdir -p $O
( time timeout 200 python -m pytest tests/test_widen_gpu.py tests/test_transforms_gpu.py tests/test_kernels_gpu.py -q -m gpu -x -n 6 2>&1 | tail -6 > $O/gpu_tests_subset.txt ) 2> $O/gpu_tests_subset.time; cat $O/gpu_tests_subset.txt; tail -3 $O/gpu_tests_subset.time
