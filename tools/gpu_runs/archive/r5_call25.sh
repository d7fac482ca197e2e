# round 5, call 25: the small-volume kernel with up to 768 input channels (SwinUNETR's 6^3 level): cases, SwinUNETR's reference golden, SwinUNETR with and without it
export TMPDIR=/tmp
O=${O:-gpurun_out/r5c25}; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_widen_gpu.py -q -m gpu -x -k "small_volume or swin_unetr_vs" 2>&1 | tail -3 | tee $O/gpu_tests_subset.txt
for f in 0 1; do
  MONAI_AMD_SMALL_VOLUME_H2=$f timeout 200 python bench.py --net swinunetr --steps 1 --warmup 1 --cpu-windows 0 --no-extra --no-pmc 2>/dev/null | grep "^{" > $O/bench_swinunetr_sv_$f.json
  python - <<PY
import json
d = json.load(open("$O/bench_swinunetr_sv_$f.json"))
print("swinunetr SMALL_VOLUME_H2=$f", round(d["ms_per_step"], 1), "ms", round(d["value"] / 1e6, 1), "Mvox/s", "checksum", d["checksum"], d.get("conv_ms_per_step"))
PY
done
