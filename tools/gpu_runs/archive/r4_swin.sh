# round 4: the direct split kernel with a half-filled last cout group (48 / 80 couts): its GPU cases, the networks that reach it, SwinUNETR / UNETR / DynUNet step times
export TMPDIR=/tmp
O=gpurun_out/r4swin; mkdir -p $O
timeout 900 python -m pytest tests -q -x -m gpu -n 2 -k "fp16_split or swin or unetr or dynunet or segresnet" 2>&1 | tail -4 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
for net in swinunetr unetr dynunet; do
  timeout 400 python bench.py --net $net --steps 2 --warmup 1 --cpu-windows 0 --no-extra > $O/bench_$net.json 2> $O/bench_$net.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r4swin/bench_$net.json").read().strip().splitlines()[-1])
    print("$net", round(d["ms_per_step"], 1), "ms", round(d["value"] / 1e6, 1), "Mvoxel/s", {k: round(v["ms_total"], 1) for k, v in d["conv_ms_per_step"].items()})
except Exception as e:
    print("$net failed", e, open("gpurun_out/r4swin/bench_$net.err").read()[-600:])
PY
done
