# round 5, call 19: UNETR / SwinUNETR with their transposed convolutions on the matrix cores (bound records through the decoder chains) -- reference goldens, A/B on one box,
# UNETR's kernel trace
export TMPDIR=/tmp
O=${O:-gpurun_out/r5c19}; mkdir -p $O
timeout 500 python -m pytest tests/test_widen_gpu.py tests/test_e2e_gpu.py -q -m gpu -x -k "unetr or swin" 2>&1 | tail -4 | tee $O/gpu_tests_subset.txt
line() { python - "$1" "$2" <<PY
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], round(d["ms_per_step"], 1), "ms", round(d["value"] / 1e6, 1), "Mvox/s", "checksum", d["checksum"])
PY
}
for f in 0 1 0 1; do
  MONAI_AMD_DECONV_H2=$f timeout 200 python bench.py --net unetr --steps 2 --warmup 1 --cpu-windows 0 --no-extra --no-pmc 2>/dev/null | grep "^{" > $O/bench_unetr_dc_$f.json
  line $O/bench_unetr_dc_$f.json "unetr DECONV_H2=$f"
done
for f in 0 1; do
  MONAI_AMD_DECONV_H2=$f timeout 300 python bench.py --net swinunetr --steps 1 --warmup 1 --cpu-windows 0 --no-extra --no-pmc 2>/dev/null | grep "^{" > $O/bench_swinunetr_dc_$f.json
  line $O/bench_swinunetr_dc_$f.json "swinunetr DECONV_H2=$f"
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --net unetr --steps 2 --warmup 1 --cpu-windows 0 --no-extra --no-pmc > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
find $O/trace -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/unetr_kernel_trace_stats.txt 2>&1; head -24 $O/unetr_kernel_trace_stats.txt | cut -c1-150
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
