# round 4, call 7: the LDS-column transposed convolution against the round-3 kernel (same box, development library: MONAI_AMD_DECONV_VAR=5 = old), its GPU cases,
# and what it does to the headline and to config 3
export TMPDIR=/tmp
O=gpurun_out/r4c7; mkdir -p $O
MONAI_AMD_LIB=monai_amd/csrc/libmonai_amd_dev.so DB_VARS=0,5 timeout 300 python tools/deconv_bench.py > $O/deconv_ab.json 2> $O/deconv_ab.err; python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r4c7/deconv_ab.json"))
    for r in d["runs"]:
        print(r["cin"], r["cout"], r["edge"], {k: round(v, 3) for k, v in r.items() if k.endswith("_ms") or "diff" in k})
except Exception as e:
    print("deconv bench failed", e, open("gpurun_out/r4c7/deconv_ab.err").read()[-1500:])
PY
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -n 0 -k "deconv or bound" 2>&1 | tail -3
for net in basicunet unetr; do
  timeout 600 python bench.py --net $net --steps 3 --warmup 1 --cpu-windows 0 --no-extra > $O/bench_$net.json 2> $O/bench_$net.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r4c7/bench_$net.json").read().strip().splitlines()[-1])
    print("$net", round(d["ms_per_step"], 1), "ms", "checksum", d["checksum"])
except Exception as e:
    print("$net failed", e, open("gpurun_out/r4c7/bench_$net.err").read()[-800:])
PY
done
