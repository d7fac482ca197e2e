# round 5, call 14: SwinUNETR -- window attention with bias / mask evaluated in the kernel, the vector LayerNorm, the block's copies folded into the gathering LayerNorm
# and the projection's scattering epilogue: cases, A/B on one box (each switch alone and both), kernel trace of the final form
export TMPDIR=/tmp
O=${O:-gpurun_out/r5c14}; mkdir -p $O
timeout 420 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py tests/test_widen_gpu.py -q -m gpu -x -k "window_attention or layernorm or linear or swin or unetr" 2>&1 | tail -4 | tee $O/gpu_tests_subset.txt
for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  MONAI_AMD_SWIN_REL_ATTENTION=$1 MONAI_AMD_SWIN_FUSED_MOVES=$2 timeout 240 python bench.py --net swinunetr --steps 2 --warmup 1 --cpu-windows 0 --no-extra --no-pmc 2>/dev/null | grep "^{" > $O/bench_swin_$1$2.json
  python - <<PY
import json
d = json.load(open("$O/bench_swin_$1$2.json"))
print("REL_ATTENTION=$1 FUSED_MOVES=$2", round(d["ms_per_step"], 1), "ms", round(d["value"] / 1e6, 1), "Mvox/s", "checksum", d["checksum"])
PY
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --net swinunetr --steps 2 --warmup 1 --cpu-windows 0 --no-extra --no-pmc > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
find $O/trace -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/swinunetr_kernel_trace_stats.txt 2>&1; head -24 $O/swinunetr_kernel_trace_stats.txt | cut -c1-150
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
