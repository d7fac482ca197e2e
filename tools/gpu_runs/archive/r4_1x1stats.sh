# round 4: InstanceNorm statistics out of the 1x1x1 shortcut convolution: kernel cases, UNETR / DynUNet goldens, UNETR and SwinUNETR step times
export TMPDIR=/tmp
O=gpurun_out/r4x1; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_widen_gpu.py tests/test_e2e_gpu.py -q -x -n 4 -k "pool_deconv_1x1_stats or unetr or dynunet or swin or mosaic or blend" 2>&1 | tail -3 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
for net in unetr swinunetr; do
  timeout 400 python bench.py --net $net --steps 2 --warmup 1 --cpu-windows 0 --no-extra > $O/bench_$net.json 2> $O/err_$net.txt
  python - $O/bench_$net.json $net <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d["ms_per_step"], 1), "ms", round(d["value"] / 1e6, 1), "Mvoxel/s")
PY
done
