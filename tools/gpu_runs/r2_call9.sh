# Round 2, GPU call 9: whole -m gpu suite (new: headline-192, slice inferer, fused argmax, lazy resampling, SwinUNETR), transform bench with
# the lazy-fusion row, SwinUNETR bench, headline bench
O=gpurun_out/r2c9; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/gpu_tests.txt
python tools/transform_bench.py > $O/transform_bench.json 2> $O/transform_bench.err
python bench.py --net swinunetr --steps 2 --warmup 1 --cpu-windows 0 > $O/bench_swin.json 2> $O/bench_swin.err
python bench.py --steps 5 --warmup 2 > $O/bench_line.json 2> $O/bench_line.err
cat $O/gpu_tests.txt; tail -3 $O/transform_bench.err; python - <<'P'
import json
d=json.load(open('gpurun_out/r2c9/transform_bench.json'))
for r in d['runs']: print(round(r['ms'],3), round(r['frac_of_8TBps'],3), r['op'])
print(d.get('lazy_fusion')); print(d['device_copy'])
for f in ('bench_swin','bench_line'):
    try:
        l=json.loads(open(f'gpurun_out/r2c9/{f}.json').read().strip().split('\n')[-1])
        print(f, l['ms_per_step'], l['value'], (l.get('roofline') or {}).get('frac'), (l.get('roofline_hbm') or {}).get('frac'), (l.get('cpu_baseline') or {}).get('parity_vs_gpu'))
    except Exception as e: print(f,'ERR',e)
P
tail -3 $O/bench_swin.err $O/bench_line.err
