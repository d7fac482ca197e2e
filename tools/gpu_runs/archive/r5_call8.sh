# round 5, call 8: the tuned upconv kernel (operand fetch a group ahead, conversion inside the matrix phase, bias table in registers) -- its cases, the headline with and
# without the fused UpCat path on one box, a kernel trace of the fused step -- and kernel traces of DynUNet and SwinUNETR
export TMPDIR=/tmp
O=${O:-gpurun_out/r5c8}; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -q -m gpu -x -k "upcat or accumulating or buffered" 2>&1 | tail -4 | tee $O/gpu_tests_subset.txt
for f in 0 1 1; do
  MONAI_AMD_UPCAT_FUSED=$f timeout 200 python bench.py --steps 3 --warmup 1 --cpu-windows 0 --no-extra --no-pmc 2>/dev/null | grep "^{" > $O/bench_fused$f.json
  python - <<PY
import json
d = json.load(open("$O/bench_fused$f.json"))
print("UPCAT_FUSED=$f", round(d["ms_per_step"], 1), "ms", round(d["value"] / 1e6, 1), "Mvox/s", d["conv_ms_per_step"], "checksum", d["checksum"])
PY
done
trace() { name=$1; shift; cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace_$name -o t -- python $GRAFT_REPO_ROOT/bench.py "$@" --cpu-windows 0 --no-extra --no-pmc > $GRAFT_REPO_ROOT/$O/line_$name.json 2>/dev/null; cd $GRAFT_REPO_ROOT
  find $O/trace_$name -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/kernel_trace_stats_$name.txt 2>&1; echo "== $name"; grep -o '"ms_per_step": [0-9.]*' $O/line_$name.json | head -1; head -14 $O/kernel_trace_stats_$name.txt | cut -c1-140; }
trace basicunet --steps 3 --warmup 1
trace dynunet --net dynunet --steps 2 --warmup 1
trace swinunetr --net swinunetr --steps 1 --warmup 1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
