set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --tb=short -k "unet or gauss or strided or separable or spacing" > gpurun_out/pytest_gpu.log 2>&1
python tools/transform_bench.py > gpurun_out/transform_bench.json 2> gpurun_out/transform_bench.err
rm -rf gpurun_out/prof_unet
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_unet -o u -- python bench.py --steps 1 --warmup 1 --net unet --cpu-windows 0 > gpurun_out/prof_unet.log 2>&1
find gpurun_out/prof_unet -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > gpurun_out/prof_unet_stats.txt 2>&1
find gpurun_out/prof_unet -name "*.db" -delete
tail -5 gpurun_out/pytest_gpu.log
tail -1 gpurun_out/prof_unet.log | cut -c1-300
head -16 gpurun_out/prof_unet_stats.txt | cut -c1-200
python -c "
import json;r=json.load(open('gpurun_out/transform_bench.json'))
for x in r['runs']: print(x['op'], round(x['ms'],3),'ms', round(x['GBps'],1),'GB/s')"
