# round 5, call 11: MaxPool3d(2) out of the producing convolution's epilogue (conv3d_h2.h, POOL): cases, headline A/B on one box, kernel trace
export TMPDIR=/tmp
O=${O:-gpurun_out/r5c11}; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -q -m gpu -x -k "pooling_epilogue or upcat or accumulating or net_single or headline_many" 2>&1 | tail -4 | tee $O/gpu_tests_subset.txt
for f in 0 1 0 1; do
  MONAI_AMD_POOL_FUSED=$f timeout 200 python bench.py --steps 3 --warmup 1 --cpu-windows 0 --no-extra --no-pmc 2>/dev/null | grep "^{" > $O/bench_pool$f.json
  python - <<PY
import json
d = json.load(open("$O/bench_pool$f.json"))
print("POOL_FUSED=$f", round(d["ms_per_step"], 1), "ms", round(d["value"] / 1e6, 1), "Mvox/s", d["conv_ms_per_step"], "checksum", d["checksum"])
PY
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-windows 0 --no-extra --no-pmc > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
find $O/trace -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/kernel_trace_stats.txt 2>&1; head -16 $O/kernel_trace_stats.txt | cut -c1-150
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
