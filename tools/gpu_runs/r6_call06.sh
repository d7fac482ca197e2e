# round 6, call 6: micro-variants of conv3d_k3_h2_kernel on the standalone harness (32 -> 32, 96^3 x 64, resident slabs = the headline's dominant instantiation), interleaved twice
export TMPDIR=/tmp
O=gpurun_out/r6c06; mkdir -p $O; : > $O/variants.txt
build() { hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -Imonai_amd/csrc $1 tools/ubench/h2_variants.hip -o $2 2>/dev/null; }
i=0
IFS=';' read -ra VS <<< "${VARIANTS:--DH2V_RES=true;-DH2V_RES=true -DH2X_SETPRIO=1;-DH2V_RES=true -DH2X_SETPRIO=2;-DH2V_RES=true -DH2X_NV=4;-DH2V_RES=true -DH2X_NV=6;-DH2V_RES=true -DH2X_NV=7}"
for v in "${VS[@]}"; do build "$v" /tmp/h2v_$i; i=$((i+1)); done
for rep in 1 2 3; do
  i=0
  for v in "${VS[@]}"; do /tmp/h2v_$i 32 "$v" >> $O/variants.txt 2>&1; i=$((i+1)); done
done
cat $O/variants.txt
# (b) the separable resample with 16-byte stores (VEC, the product default) against the scalar-store form: the development library reads MONAI_AMD_RS_VEC; interleaved
for rep in 1 2; do
  for v in 1 0; do
    MONAI_AMD_LIB=$PWD/monai_amd/csrc/libmonai_amd_dev.so MONAI_AMD_RS_VEC=$v timeout 200 python tools/transform_bench.py 2>/dev/null | python -c "
import json, sys
r = json.load(sys.stdin)
print('RS_VEC=$v rep $rep', [(x['op'][:40], round(x['ms'], 4)) for x in r['runs']])" | tee -a $O/resample_vec_ab.txt
  done
done
timeout 900 python -m pytest tests/test_transforms_gpu.py tests/test_widen_gpu.py -q -m gpu -x -k "separable or spacing or Spacing or resampl or unet_activations or lazy" 2>&1 | tail -4 | tee $O/gpu_tests_subset.txt
