# round 5, call 20: the stride-2 kernel's fused form (conversion inside the GEMM's staging) against the split form: cases (bit-identity of the two forms), per-layer timing,
# DynUNet / SegResNet with MONAI_AMD_STRIDED_H2_FUSED = 0 / 1 / auto on one box
export TMPDIR=/tmp
O=${O:-gpurun_out/r5c20}; mkdir -p $O
timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_widen_gpu.py -q -m gpu -x -k "strided or dynunet_vs or segresnet_vs" 2>&1 | tail -4 | tee $O/gpu_tests_subset.txt
timeout 200 python tools/s2_bench.py --layers "32,64,96;64,128,48;128,256,24;16,32,96;32,64,48;64,128,24" 2>&1 | grep "^{" | tee $O/s2_layers_fused.txt
line() { python - "$1" "$2" <<PY
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], round(d["ms_per_step"], 1), "ms", round(d["value"] / 1e6, 1), "Mvox/s", "checksum", d["checksum"])
PY
}
for net in dynunet segresnet; do
for f in 0 1 auto; do
  MONAI_AMD_STRIDED_H2_FUSED=$f timeout 200 python bench.py --net $net --steps 2 --warmup 1 --cpu-windows 0 --no-extra --no-pmc 2>/dev/null | grep "^{" > $O/bench_${net}_fused_$f.json
  line $O/bench_${net}_fused_$f.json "$net STRIDED_H2_FUSED=$f"
done
done
