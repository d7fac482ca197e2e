# round 5, call 22 (the round's last tree): the whole -m gpu suite + smoke(); the small-volume split-precision convolution (kernels/conv3d_vol_h2.h) against the fp32 tiles
# it replaces: headline and DynUNet with MONAI_AMD_SMALL_VOLUME_H2 = 0 / 1 on one box
export TMPDIR=/tmp
O=${O:-gpurun_out/r5c22}; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -30 > $O/gpu_tests.txt; grep "^E  .*assert\|^E  .*Error" $O/gpu_tests.txt | head -5; tail -3 $O/gpu_tests.txt
python __graft_entry__.py smoke 2>&1 | tail -1 | tee $O/smoke.txt
line() { python - "$1" "$2" <<PY
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], round(d["ms_per_step"], 1), "ms", round(d["value"] / 1e6, 1), "Mvox/s", "checksum", d["checksum"], d.get("conv_ms_per_step"))
PY
}
for f in 0 1 0 1; do
  MONAI_AMD_SMALL_VOLUME_H2=$f timeout 200 python bench.py --steps 3 --warmup 1 --cpu-windows 0 --no-extra --no-pmc 2>/dev/null | grep "^{" > $O/bench_basicunet_sv_$f.json
  line $O/bench_basicunet_sv_$f.json "basicunet SMALL_VOLUME_H2=$f"
done
for f in 0 1; do
  MONAI_AMD_SMALL_VOLUME_H2=$f timeout 200 python bench.py --net dynunet --steps 2 --warmup 1 --cpu-windows 0 --no-extra --no-pmc 2>/dev/null | grep "^{" > $O/bench_dynunet_sv_$f.json
  line $O/bench_dynunet_sv_$f.json "dynunet SMALL_VOLUME_H2=$f"
done
