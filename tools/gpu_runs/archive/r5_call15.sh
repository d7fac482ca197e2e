# round 5, call 15: the 1x1x1 shortcut convolution with all output channels from one read (conv1x1_h2.h), the residual join inside the output convolution, bounds for
# SwinUNETR's normalised hidden states: cases + the networks' goldens, SwinUNETR / UNETR / DynUNet step times, SwinUNETR trace
export TMPDIR=/tmp
O=${O:-gpurun_out/r5c15}; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py tests/test_widen_gpu.py -q -m gpu -x -k "conv1x1 or residual_join or swin or unetr or dynunet or simple_layers or nets_with_spread or nonfinite" 2>&1 | tail -4 | tee $O/gpu_tests_subset.txt
for net in swinunetr unetr; do
  timeout 300 python bench.py --net $net --steps 2 --warmup 1 --cpu-windows 0 --no-extra --no-pmc 2>/dev/null | grep "^{" > $O/bench_$net.json
  python - <<PY
import json
d = json.load(open("$O/bench_$net.json"))
print("$net", round(d["ms_per_step"], 1), "ms", round(d["value"] / 1e6, 1), "Mvox/s", "checksum", d["checksum"])
PY
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --net swinunetr --steps 2 --warmup 1 --cpu-windows 0 --no-extra --no-pmc > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
find $O/trace -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/swinunetr_kernel_trace_stats.txt 2>&1; head -24 $O/swinunetr_kernel_trace_stats.txt | cut -c1-150
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
