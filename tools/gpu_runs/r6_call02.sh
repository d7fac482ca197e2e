# round 6, call 2: rounds of windows on several HIP streams (config.SW_STREAMS): bit-identity test, then the headline interleaved 1 / 2 / 1 / 2 / 3 / 1 streams on ONE box
export TMPDIR=/tmp
O=${O:-gpurun_out/r6c02}; mkdir -p $O
timeout 600 python -m pytest tests/test_e2e_gpu.py -q -m gpu -x -n 0 -s -k "several_streams" 2>&1 | tail -5 | tee $O/streams_test.txt
for tag in s1a s2a s1b s2b s3a s1c s2c; do
  k=${tag:1:1}
  MONAI_AMD_SW_STREAMS=$k timeout 600 python bench.py --steps 5 --warmup 2 --cpu-windows 0 --no-extra --no-pmc 2>$O/bench_$tag.err | grep "^{" > $O/bench_$tag.json
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_$tag.json"))
    print("$tag", "streams", d.get("streams", 1), round(d["ms_per_step"], 2), "ms", round(d["value"] / 1e6, 1), "Mvox/s", "conv ms_avg", d["roofline"]["ms_avg"], "frac", d["roofline"]["frac"], "checksum", d["checksum"])
except Exception as e:
    print("$tag failed", e)
PY
done 2>&1 | tee $O/summary.txt
