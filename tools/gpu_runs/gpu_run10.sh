set -x
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -k "winograd" > gpurun_out/pytest_gpu.log 2>&1
tail -5 gpurun_out/pytest_gpu.log
KB_BATCH=25 python tools/kernel_bench.py > gpurun_out/kernel_bench.json 2> gpurun_out/kernel_bench.err
python -c "
import json;r=json.load(open('gpurun_out/kernel_bench.json'))
for row in r['conv']:
    c=row['cfgs']; best=min((k for k in c if k!='15'), key=lambda k:c[k]['ms'])
    w=c.get('15')
    print(row['layer'],row['cin'],row['cout'],row['edge'],'sel',row['selected'],'best-direct',best,round(c[best]['ms'],3),'ms',round(c[best]['tflops'],1),'TF', '| wino', (round(w['ms'],3), round(w['tflops'],1)) if w else None)
"
