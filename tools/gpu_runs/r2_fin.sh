# last call of round 2: the row-mapped general resample (cases + transform bench with its A/B against the linear-index kernel), the bench line of the
# final state (one-input-channel kernel on, scalar transposed convolution) with its kernel trace, then the kernel / transform / end-to-end GPU cases
O=gpurun_out/r2fin; mkdir -p $O; export TMPDIR=/tmp
timeout 150 python -m pytest tests/test_transforms_gpu.py -q -n 0 -k "general_rows or separable_fast" 2>&1 | tail -3 > $O/rows_tests.txt; cat $O/rows_tests.txt
timeout 150 python tools/transform_bench.py > $O/transform_bench.json 2> $O/transform_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2fin/transform_bench.json"))
for r in d["runs"]:
    print(f'{r["op"][:80]:80s} {r["ms"]:.3f} ms  {r["frac_of_8TBps"]:.3f}')
PY
rm -rf $O/prof
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py > $O/bench_line.json 2> $O/bench_line.err
find $O/prof -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/bench_kernel_trace_stats.txt 2>&1
rm -rf $O/prof
head -12 $O/bench_kernel_trace_stats.txt | cut -c1-150; cut -c1-330 $O/bench_line.json
timeout 200 python -m pytest tests/test_kernels_gpu.py tests/test_transforms_gpu.py tests/test_e2e_gpu.py -q 2>&1 | tail -5 > $O/gpu_tests_subset.txt; tail -3 $O/gpu_tests_subset.txt
