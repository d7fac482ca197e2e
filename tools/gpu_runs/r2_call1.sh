# Round 2, GPU call 1: HBM ceilings (tools/ubench/hbm_stream.hip), two-waves-per-SIMD probe (coexec2.hip), blend variants,
# the whole -m gpu suite with the regular-grid blend, and everything round 1 left unmeasured (preproc kernels, DynUNet, SegResNet).
#   gpurun --timeout 1200 -- 'bash tools/gpu_runs/r2_call1.sh'
O=gpurun_out/r2c1; mkdir -p $O
( cd tools/ubench && hipcc --offload-arch=gfx950 -O3 -Wno-unused-value hbm_stream.hip -o /tmp/hbm_stream && /tmp/hbm_stream ) > $O/hbm_stream.txt 2>&1
( cd tools/ubench && hipcc --offload-arch=gfx950 -O3 -Wno-unused-value coexec2.hip -o /tmp/coexec2 && /tmp/coexec2 ) > $O/coexec2.txt 2>&1
python tools/blend_bench.py > $O/blend_bench.json 2> $O/blend_bench.err
timeout 600 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 > $O/gpu_tests.txt
python tools/preproc_bench.py > $O/preproc_bench.json 2> $O/preproc_bench.err
python bench.py --net dynunet --steps 2 --warmup 1 --cpu-windows 0 > $O/bench_dynunet.json 2> $O/bench_dynunet.err
python bench.py --net segresnet --steps 2 --warmup 1 --cpu-windows 0 > $O/bench_segresnet.json 2> $O/bench_segresnet.err
python bench.py --steps 3 --warmup 1 > $O/bench_line.json 2> $O/bench_line.err
cat $O/hbm_stream.txt $O/coexec2.txt; tail -4 $O/gpu_tests.txt; python - <<'P'
import json
d=json.load(open('gpurun_out/r2c1/blend_bench.json'))
for r in d['runs']: print(r)
P
cut -c1-300 $O/bench_line.json
