# config-4 transform kernels with the development library (= whatever tree it was built from: build it BEFORE a change to get a same-box A/B) and the product library;
# writes gpurun_out/transform/{old,new}_{1,2}.json
O=gpurun_out/transform; mkdir -p $O
for i in 1 2; do
  MONAI_AMD_LIB=$PWD/monai_amd/csrc/libmonai_amd_dev.so timeout 200 python tools/transform_bench.py > $O/old_$i.json 2> $O/err.txt
  timeout 200 python tools/transform_bench.py > $O/new_$i.json 2>> $O/err.txt
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/transform/*_?.json")):
    try:
        r = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    rows = r.get("runs") or r.get("rows") or r.get("results") or []
    print(f)
    for x in rows:
        print("   ", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in x.items() if k in ("op", "name", "ms", "frac", "GBps", "ms_per_volume", "frac_of_8TBps", "kernel")})
PY
tail -2 $O/err.txt
