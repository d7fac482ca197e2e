# round 6, call 11: the product library with the ring form for fp64 resampling: transform tests, the development library's MONAI_AMD_RS_RING=0/1 A/B, config-4 numbers
export TMPDIR=/tmp
O=gpurun_out/r6c11; mkdir -p $O
timeout 900 python -m pytest tests/test_transforms_gpu.py tests/test_widen_gpu.py -q -m gpu -x 2>&1 | tail -4 | tee $O/gpu_tests.txt
show() { python -c "
import json, sys
r = json.load(sys.stdin)
print('$1', [(x['op'][:44], round(x['ms'], 4)) for x in r['runs'] if 'separable' in x['op'] or 'Spacing' in x['op']])"; }
for rep in 1 2; do
  for v in 1 0; do MONAI_AMD_LIB=$PWD/monai_amd/csrc/libmonai_amd_dev.so MONAI_AMD_RS_RING=$v timeout 200 python tools/transform_bench.py 2>/dev/null | show "RS_RING=$v rep $rep" | tee -a $O/ring_ab.txt; done
done
timeout 200 python tools/transform_bench.py 2>/dev/null | show "product" | tee -a $O/ring_ab.txt
