# round 5, call 2: same-box A/B of (a) the barrier-free separable resample (separable_resample_wave_kernel) and (b) the packed-math transposed convolution against the
# development library built from the previous commit (libmonai_amd_dev.so = tree 7a72e7e), then the GPU cases of the kernels touched and a short headline run
export TMPDIR=/tmp
O=gpurun_out/r5c2; mkdir -p $O
for i in 1 2; do
  MONAI_AMD_LIB=$PWD/monai_amd/csrc/libmonai_amd_dev.so timeout 200 python tools/transform_bench.py > $O/transform_old_$i.json 2> $O/err.txt
  timeout 200 python tools/transform_bench.py > $O/transform_new_$i.json 2>> $O/err.txt
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5c2/transform_*_?.json")):
    try:
        r = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f)
    for x in r.get("runs", []):
        print("   ", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in x.items() if k in ("op", "ms", "GBps")})
PY
tail -2 $O/err.txt
MONAI_AMD_LIB=$PWD/monai_amd/csrc/libmonai_amd_dev.so timeout 120 python tools/deconv_bench.py > $O/deconv_old.json 2>> $O/err.txt
timeout 120 python tools/deconv_bench.py > $O/deconv_new.json 2>> $O/err.txt
python - <<'PY'
import json
for tag in ("old", "new"):
    try:
        r = json.load(open(f"gpurun_out/r5c2/deconv_{tag}.json"))
    except Exception as e:
        print(tag, "unreadable", e); continue
    for row in r["runs"]:
        print(tag, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in row.items() if not k.endswith("GBps")})
PY
timeout 300 python -m pytest tests/test_transforms_gpu.py tests/test_kernels_gpu.py -q -m gpu -x -k "deconv or separable or spacing or resample or lazy or affine" 2>&1 | tail -4 | tee $O/gpu_tests_subset.txt
timeout 200 python bench.py --steps 3 --warmup 1 --cpu-windows 0 --no-extra --no-pmc 2>/dev/null | grep "^{" | tee $O/bench_line_short.json | cut -c1-400
