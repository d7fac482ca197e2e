# round 5, call 16: the one-read 1x1x1 convolution for more than 512 input channels (SwinUNETR's decoder5 shortcut, 768 -> 384 @ 6^3: 24 passes of the VALU kernel before)
export TMPDIR=/tmp
O=${O:-gpurun_out/r5c16}; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_widen_gpu.py -q -m gpu -x -k "conv1x1 or swin" 2>&1 | tail -4 | tee $O/gpu_tests_subset.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --net swinunetr --steps 2 --warmup 1 --cpu-windows 0 --no-extra --no-pmc 2>/dev/null | grep "^{" > $GRAFT_REPO_ROOT/$O/bench_line_swinunetr.json; cd $GRAFT_REPO_ROOT
python - <<PY
import json
d = json.load(open("$O/bench_line_swinunetr.json"))
print("swinunetr (under the profiler)", round(d["ms_per_step"], 1), "ms", round(d["value"] / 1e6, 1), "Mvox/s", "checksum", d["checksum"])
PY
find $O/trace -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/swinunetr_kernel_trace_stats.txt 2>&1; head -30 $O/swinunetr_kernel_trace_stats.txt | cut -c1-150
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
for net in swinunetr dynunet; do
timeout 300 python bench.py --net $net --steps 2 --warmup 1 --cpu-windows 27 --no-extra --no-pmc 2>/dev/null | grep "^{" > $O/bench_line_${net}_parity.json
python - <<PY
import json
d = json.load(open("$O/bench_line_${net}_parity.json"))
print("$net", round(d["ms_per_step"], 1), "ms", round(d["value"] / 1e6, 1), "Mvox/s", "parity", d.get("parity"))
PY
done
