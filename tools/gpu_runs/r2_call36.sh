O=gpurun_out/r2c36; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "fp16_split" -n 0 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 --cpu-windows 0 > $O/bench_line.json 2> $O/bench_line.err
cut -c1-330 $O/bench_line.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2c36/bench_line.json").read().strip().split("\n")[-1])
print(d["conv_ms_per_step"])
PY
