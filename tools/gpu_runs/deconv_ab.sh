# transposed-convolution variants (dev library) and the Gaussian with non-temporal result stores (dev library built with MONAI_AMD_DEV_FLAGS=-DMH_DEV_NT_STORES);
# writes gpurun_out/deconv/*
O=gpurun_out/deconv; mkdir -p $O
DEV=$PWD/monai_amd/csrc/libmonai_amd_dev.so
MONAI_AMD_LIB=$DEV DB_VARS=0,1,2,3 timeout 400 python tools/deconv_bench.py > $O/deconv_bench.json 2> $O/err.txt
python - <<'PY'
import json
r = json.load(open("gpurun_out/deconv/deconv_bench.json"))
for row in r["runs"]:
    print(row["cin"], row["cout"], row["edge"], {k: round(v, 3) for k, v in row.items() if k.endswith("_ms") or k.endswith("diff")})
PY
for lib in "" "$DEV"; do
  echo "== transform bench, library: ${lib:-product}"
  MONAI_AMD_LIB=$lib timeout 300 python tools/transform_bench.py 2>> $O/err.txt | python -c "
import json,sys
r=json.load(sys.stdin)
for x in r['runs']:
    if 'kernel' in x['op'] and ('separable' in x['op'] or 'filter3d' in x['op']): print(round(x['ms'],4), round(x['GBps']/8000,3), x['op'])
"
done
tail -3 $O/err.txt
