# first GPU call of round 3: what round 2 prepared on the CPU and could not measure any more
#  1. the transposed convolution as one GEMM on the fp32 matrix cores (kernels/nn_simple.h: deconv_k2s2_mfma_kernel, opt-in): its GPU cases and the
#     bench line with / without it (the one-voxel kernel is 6.5 % of the step: 0.89 / 2.3 ms per launch at 32 -> 32 ch; not arithmetic-bound -- it writes at 3.0 TB/s,
#     profiles/r02_deconv_bench_v1.json: the new form writes complete 256-byte runs)
#     + the 2-D BasicUNet / DynUNet cases (one plane of the 3-D engine, SliceInferer over them) that were written after the budget was spent
#  2. the whole -m gpu suite + smoke() on the state the round starts from, the bench line with its kernel trace
O=gpurun_out/r3first; mkdir -p $O; export TMPDIR=/tmp
MONAI_AMD_TEST_UNVERIFIED_ON_GPU=1 timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py tests/test_widen_gpu.py -q -n 0 -k "deconv_on_the_matrix or pool_deconv or 2d_and_slice_inferer" 2>&1 | tail -5 > $O/deconv_mfma_tests.txt; cat $O/deconv_mfma_tests.txt
python tools/deconv_bench.py > $O/deconv_bench.json 2> $O/deconv_bench.err; tr -d "\n " < $O/deconv_bench.json | cut -c1-900; echo
for impl in scalar mfma; do
  MONAI_AMD_DECONV_IMPL=$impl timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$impl -o bench -- python bench.py --steps 3 --warmup 1 --cpu-windows 0 > $O/bench_line_deconv_$impl.json 2> $O/bench_$impl.err
  find $O/prof_$impl -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/kernel_trace_deconv_$impl.txt 2>&1
  rm -rf $O/prof_$impl
  grep -i "deconv" $O/kernel_trace_deconv_$impl.txt | cut -c1-170; cut -c1-260 $O/bench_line_deconv_$impl.json
done
MONAI_AMD_C1_COT=8 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c1 -o bench -- python bench.py --steps 3 --warmup 1 --cpu-windows 0 > $O/bench_line_c1_cot8.json 2> $O/bench_c1.err
find $O/prof_c1 -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/kernel_trace_c1_cot8.txt 2>&1; rm -rf $O/prof_c1
grep -i "c1_kernel" $O/kernel_trace_c1_cot8.txt | cut -c1-170; cut -c1-200 $O/bench_line_c1_cot8.json
bash tools/gpu_runs/pmc_deconv.sh > $O/pmc_deconv.log 2>&1; tail -12 $O/pmc_deconv.log | cut -c1-150
MONAI_AMD_TEST_UNVERIFIED_ON_GPU=1 timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 > $O/gpu_tests.txt; tail -3 $O/gpu_tests.txt
python __graft_entry__.py smoke 2>&1 | tail -2 > $O/smoke.txt; cat $O/smoke.txt
