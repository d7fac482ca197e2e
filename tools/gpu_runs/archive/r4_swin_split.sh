# round 4, last GPU call: SwinUNETR(48) with its 48-couts convolutions split over the 32-couts and the 16-couts form of the split-precision kernel
export TMPDIR=/tmp
O=gpurun_out/r4swin4; mkdir -p $O
timeout 75 python -m pytest tests/test_e2e_gpu.py tests/test_widen_gpu.py -q -x -n 2 -k "splits_couts or swin" 2>&1 | tail -3 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
timeout 70 python bench.py --net swinunetr --steps 2 --warmup 1 --cpu-windows 0 --no-extra > $O/bench_swinunetr.json 2> $O/err.txt
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4swin4/bench_swinunetr.json").read().strip().splitlines()[-1]); print("swinunetr", round(d["ms_per_step"], 1), "ms", round(d["value"] / 1e6, 1), "Mvoxel/s", {k: round(v["ms_total"], 1) for k, v in d["conv_ms_per_step"].items()})
except Exception as e:
    print("bench failed", e)
PY
