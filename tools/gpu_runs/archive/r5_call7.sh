# round 5, call 7: the whole -m gpu suite on the tree with the fused UpCat path, then the driver's bench command (whole-volume CPU leg, reference_self_spread, extras, in-run PMC)
export TMPDIR=/tmp
O=${O:-gpurun_out/r5c7}; mkdir -p $O
( time timeout 480 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > $O/gpu_tests.txt ) 2> $O/gpu_tests.time; cat $O/gpu_tests.txt; tail -3 $O/gpu_tests.time
( time timeout 900 python bench.py 2> $O/bench.err | grep "^{" > $O/bench_line.json ) 2> $O/bench.time; tail -3 $O/bench.time; tail -5 $O/bench.err
python - <<PY
import json
s = open("$O/bench_line.json").read()
print("line bytes", len(s))
d = json.loads(s)
print({k: d[k] for k in ("value", "ms_per_step", "parity", "reference_self_spread", "pmc")})
print("roofline", d["roofline"]); print("roofline_hbm", d["roofline_hbm"]); print("cpu_baseline", d["cpu_baseline"])
for k, v in d.get("extra", {}).items(): print(k, json.dumps(v)[:900])
PY
