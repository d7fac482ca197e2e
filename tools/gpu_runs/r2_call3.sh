# Round 2, GPU call 3: the two-waves-per-SIMD Winograd kernel (conv3d_wino2p.h) against the round-1 kernel, padded logits stride.
O=gpurun_out/r2c3; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -q -x -k "wino2d or blend or extract or headline or slice_inferer or bench_shaped or config0" 2>&1 | tail -15 > $O/gpu_tests.txt
KB_BATCH=64 WB_SKIP_SPLIT=1 python tools/wino_bench.py > $O/wino_bench_p.json 2> $O/wino_bench_p.err
MONAI_AMD_W2_IMPL=d KB_BATCH=64 WB_SKIP_SPLIT=1 WB_SKIP_DIRECT=1 python tools/wino_bench.py > $O/wino_bench_d.json 2> $O/wino_bench_d.err
python tools/blend_bench.py > $O/blend_bench.json 2> $O/blend_bench.err
python bench.py --steps 3 --warmup 1 --cpu-windows 0 > $O/bench_p.json 2> $O/bench_p.err
MONAI_AMD_W2_IMPL=d MONAI_AMD_LOGITS_PAD=0 python bench.py --steps 3 --warmup 1 --cpu-windows 0 > $O/bench_d.json 2> $O/bench_d.err
cat $O/gpu_tests.txt; cat $O/wino_bench_p.json $O/wino_bench_d.json; python - <<'P'
import json
d=json.load(open('gpurun_out/r2c3/blend_bench.json'))
for r in d['runs']: print(round(r['ms'],3), round(r['frac_of_8TBps'],3), r['variant'], r.get('bitwise_equal_to_round1',''))
for f in ('bench_p','bench_d'):
    try:
        l=json.loads(open(f'gpurun_out/r2c3/{f}.json').read().strip().split('\n')[-1])
        print(f, l['ms_per_step'], l['value'], l['roofline']['ms_avg'], l['roofline']['mfma_pipe_frac'], l['roofline_hbm']['ms_avg'], l['roofline_hbm']['frac'])
    except Exception as e: print(f, 'ERR', e)
P
tail -3 $O/*.err
