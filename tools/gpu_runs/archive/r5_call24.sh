# round 5, call 24: the stride-2 GEMM with two operand register sets in flight (split form) -- cases, per-layer timing, DynUNet
export TMPDIR=/tmp
O=${O:-gpurun_out/r5c24}; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_widen_gpu.py -q -m gpu -x -k "strided or dynunet_vs or segresnet_vs" 2>&1 | tail -3 | tee $O/gpu_tests_subset.txt
timeout 200 python tools/s2_bench.py --layers "32,64,96;64,128,48;128,256,24;16,32,96" 2>&1 | grep "^{" | tee $O/s2_layers_deep.txt
timeout 200 python bench.py --net dynunet --steps 2 --warmup 1 --cpu-windows 0 --no-extra --no-pmc 2>/dev/null | grep "^{" > $O/bench_dynunet.json
python - <<PY
import json
d = json.load(open("$O/bench_dynunet.json"))
print("dynunet", round(d["ms_per_step"], 1), "ms", round(d["value"] / 1e6, 1), "Mvox/s", "checksum", d["checksum"])
PY
