# round 4: SwinUNETR with the matrix-core window attention: kernel cases + golden test + step time + kernel trace
export TMPDIR=/tmp
O=gpurun_out/r4swin3; mkdir -p $O
timeout 600 python -m pytest tests/test_widen_gpu.py tests/test_kernels_gpu.py -q -x -n 0 -k "swin or window_attention" 2>&1 | tail -3 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
timeout 400 python bench.py --net swinunetr --steps 2 --warmup 1 --cpu-windows 0 --no-extra > $O/bench_swinunetr.json 2> $O/err0.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4swin3/bench_swinunetr.json").read().strip().splitlines()[-1]); print("swinunetr", round(d["ms_per_step"], 1), "ms", round(d["value"] / 1e6, 1), "Mvoxel/s")
PY
timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof -o b -- python bench.py --net swinunetr --steps 1 --warmup 1 --cpu-windows 0 --no-extra > $O/bench_prof.json 2> $O/err.txt
find $O/prof -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/swinunetr_kernel_trace_stats.txt 2>&1; rm -rf $O/prof
head -14 $O/swinunetr_kernel_trace_stats.txt | cut -c1-170
