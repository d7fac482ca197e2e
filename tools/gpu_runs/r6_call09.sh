# round 6, call 9: (a) the product library's resample after the register cap + chunk rule (tools/transform_bench.py), (b) Gaussian z-chunk sweep with the development library
export TMPDIR=/tmp
O=gpurun_out/r6c09; mkdir -p $O
show() { python -c "
import json, sys
r = json.load(sys.stdin)
print('$1', [(x['op'][:44], round(x['ms'], 4)) for x in r['runs'] if 'separable' in x['op'] or 'Spacing' in x['op'] or 'aussian' in x['op']])"; }
for rep in 1 2; do timeout 200 python tools/transform_bench.py 2>/dev/null | show "product rep $rep" | tee -a $O/transform.txt; done
for c in 0 2 3 4 5 6 8 11 16; do
  if [ $c = 0 ]; then unset MONAI_AMD_GS_CHUNKS; else export MONAI_AMD_GS_CHUNKS=$c; fi
  MONAI_AMD_LIB=$PWD/monai_amd/csrc/libmonai_amd_dev.so timeout 200 python tools/transform_bench.py 2>/dev/null | show "GS_CHUNKS=$c" | tee -a $O/gauss_chunks.txt
done
unset MONAI_AMD_GS_CHUNKS
timeout 900 python -m pytest tests/test_transforms_gpu.py -q -m gpu -x 2>&1 | tail -3 | tee $O/gpu_tests_transforms.txt
