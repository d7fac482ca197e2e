# round 5, call 10: the composite kernel with its epilogue parked one plane (emitted between the next plane's matrix instructions): cases, headline A/B on one box, trace
export TMPDIR=/tmp
O=${O:-gpurun_out/r5c10}; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -q -m gpu -x -k "upcat or accumulating" 2>&1 | tail -4 | tee $O/gpu_tests_subset.txt
for f in 0 1 1; do
  MONAI_AMD_UPCAT_FUSED=$f timeout 200 python bench.py --steps 3 --warmup 1 --cpu-windows 0 --no-extra --no-pmc 2>/dev/null | grep "^{" > $O/bench_fused$f.json
  python - <<PY
import json
d = json.load(open("$O/bench_fused$f.json"))
print("UPCAT_FUSED=$f", round(d["ms_per_step"], 1), "ms", round(d["value"] / 1e6, 1), "Mvox/s", d["conv_ms_per_step"], d.get("upconv"), "checksum", d["checksum"])
PY
done
