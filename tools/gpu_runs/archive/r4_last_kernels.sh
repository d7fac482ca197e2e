# round 4, the remaining GPU seconds: the kernel test file on the final tree (its tail did not fit into r4_final_tests.sh's limit)
export TMPDIR=/tmp
O=gpurun_out/r4final2; mkdir -p $O
timeout 75 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -n 8 2>&1 | tail -4 > $O/gpu_tests_kernels.txt; cat $O/gpu_tests_kernels.txt
