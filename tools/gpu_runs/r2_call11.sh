O=gpurun_out/r2c11; mkdir -p $O
timeout 600 python -m pytest tests/test_transforms_gpu.py tests/test_widen_gpu.py -q -k "gaussian or swin" -p no:xdist 2>&1 | tail -5 > $O/gpu_tests.txt
python tools/transform_bench.py > $O/transform_bench_rowvec.json 2> $O/tb1.err
MONAI_AMD_GS_IMPL=tile python tools/transform_bench.py > $O/transform_bench_tile.json 2> $O/tb2.err
cat $O/gpu_tests.txt; python - <<'P'
import json
for f in ('rowvec','tile'):
    d=json.load(open(f'gpurun_out/r2c11/transform_bench_{f}.json'))
    for r in d['runs']:
        if 'Gauss' in r['op'] or 'filter3d' in r['op']: print(f, round(r['ms'],3), round(r['frac_of_8TBps'],3), r['op'])
P
