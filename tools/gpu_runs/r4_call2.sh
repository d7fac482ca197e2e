# round 4, call 2: z-Winograd split-precision kernel -- A/B against the direct kernel, its GPU parity cases; the rest of the determinism checks
export TMPDIR=/tmp
bash tools/gpu_runs/h2z_variants.sh
O=gpurun_out/h2zv
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "z_winograd or fp16_split" -p no:xdist 2>&1 | tail -5 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
# whole network with the z-Winograd kernel wherever it fits: bench line (no CPU leg, no extras) for h2 and h2z
for algo in h2 h2z; do
  MONAI_AMD_CONV_ALGO=$algo timeout 600 python bench.py --steps 3 --warmup 1 --cpu-windows 0 --no-extra > $O/bench_$algo.json 2> $O/bench_$algo.err; python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$algo.json").read().strip().splitlines()[-1])
    print("$algo", round(d["ms_per_step"], 1), "ms", {k: round(v["ms_total"], 1) for k, v in d["conv_ms_per_step"].items()}, "checksum", d["checksum"])
except Exception as e:
    print("$algo failed", e, open("$O/bench_$algo.err").read()[-600:])
PY
done
O=gpurun_out/determinism; mkdir -p $O
timeout 900 python tools/determinism_stress.py --procs 4 --runs 8 --size 512 --vary 2>&1 | grep "^STRESS " > $O/stress_vary.json; cut -c1-700 $O/stress_vary.json
timeout 900 python tools/determinism_stress.py --procs 4 --runs 6 --size 512 2>&1 | grep "^STRESS " > $O/stress_hooks.json; cut -c1-500 $O/stress_hooks.json
