# round 4, call 9: the mosaic blend with one class per thread (5x the threads, a fifth of the read streams per wave) against the product form, same box (development library)
export TMPDIR=/tmp
O=gpurun_out/r4c9; mkdir -p $O
show() { python - "$1" <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
for x in r["runs"]:
    print(f"{x['ms']:8.3f} ms {x['frac_of_8TBps']:.3f}  {x['variant']}  {x.get('bitwise_equal_to_window_major', '')}")
PY
}
for ks in 0 1 2 0 1; do
  echo "== MONAI_AMD_BLEND_KSPLIT=$ks"
  MONAI_AMD_LIB=$PWD/monai_amd/csrc/libmonai_amd_dev.so MONAI_AMD_BLEND_KSPLIT=$ks BB_QUICK=1 timeout 300 python tools/blend_bench.py > $O/blend_ks${ks}.json 2>> $O/err.txt && show $O/blend_ks${ks}.json | grep mosaic
done
tail -3 $O/err.txt
