# round 4: the 16-couts form of the split-precision convolution: kernel cases, UNETR goldens, UNETR step time with and without it (dev library, alternating), kernel trace
export TMPDIR=/tmp
O=gpurun_out/r4h2c; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_widen_gpu.py tests/test_e2e_gpu.py -q -x -n 4 -k "16_couts or unetr or fp16_split" 2>&1 | tail -3 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
DEVLIB=$PWD/monai_amd/csrc/libmonai_amd_dev.so
for i in 1 2; do
  for on in 0 1; do
    MONAI_AMD_LIB=$DEVLIB MONAI_AMD_H2C=$on timeout 400 python bench.py --net unetr --steps 2 --warmup 1 --cpu-windows 0 --no-extra > $O/bench_unetr_h2c${on}_$i.json 2> $O/err.txt
    python - $O/bench_unetr_h2c${on}_$i.json $on <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("unetr h2c=" + sys.argv[2], round(d["ms_per_step"], 1), "ms", round(d["value"] / 1e6, 1), "Mvoxel/s", {k: round(v["ms_total"], 1) for k, v in d["conv_ms_per_step"].items()})
PY
  done
done
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o b -- python bench.py --net unetr --steps 1 --warmup 1 --cpu-windows 0 --no-extra > $O/bench_prof.json 2>> $O/err.txt
find $O/prof -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/unetr_kernel_trace_stats.txt 2>&1; rm -rf $O/prof
head -8 $O/unetr_kernel_trace_stats.txt | cut -c1-170
