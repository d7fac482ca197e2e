O=gpurun_out/rs; mkdir -p $O
for t in 512 256; do MONAI_AMD_RS_THREADS=$t python tools/transform_bench.py > $O/tb_$t.json 2> $O/tb_$t.err; python - <<PY
import json
d=json.load(open("$O/tb_$t.json"))
print("threads $t")
for r in d["runs"]:
    if "resample" in r["op"] or "Spacing" in r["op"]: print("  %.3f ms  %.3f of 8 TB/s  %s" % (r["ms"], r.get("frac_of_8TBps",0), r["op"]))
PY
done
timeout 600 python -m pytest tests/test_transforms_gpu.py -q -m gpu -k "spacing or resample or affine" 2>&1 | tail -2
