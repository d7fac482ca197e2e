# round 4: the other networks after the 16-couts kernel and the statistics-fused shortcut (SegResNet(16), UNet 16..256 and SwinUNETR have 16-couts / shortcut layers)
export TMPDIR=/tmp
O=gpurun_out/r4nets2; mkdir -p $O
for net in segresnet unet swinunetr; do
  timeout 100 python bench.py --net $net --steps 2 --warmup 1 --cpu-windows 0 --no-extra > $O/bench_$net.json 2> $O/err_$net.txt
  python - $O/bench_$net.json $net <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d["ms_per_step"], 1), "ms", round(d["value"] / 1e6, 1), "Mvoxel/s", {k: round(v["ms_total"], 1) for k, v in d["conv_ms_per_step"].items()})
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
done
