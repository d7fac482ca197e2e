# A/B of the blend over the two logits layouts + the two logits writers (tools/blend_bench.py), then -- with the -DMH_DEV_KNOBS library
# (python -m monai_amd.build --dev, built in the build container) -- the window-batch size / non-temporal variants of the mosaic kernel; writes gpurun_out/blend/*
O=gpurun_out/blend; mkdir -p $O
show() { python - "$1" <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
for x in r["runs"]:
    print(f"{x['ms']:8.3f} ms {x['frac_of_8TBps']:.3f}  {x['variant']}  {x.get('bitwise_equal_to_window_major', '')}")
PY
}
timeout 600 python tools/blend_bench.py > $O/blend_bench.json 2> $O/blend_bench.err; show $O/blend_bench.json; tail -3 $O/blend_bench.err
if [ -f monai_amd/csrc/libmonai_amd_dev.so ]; then
  for g in 1 2 4; do for nt in 0 1; do
    echo "== mosaic kernel G=$g NT=$nt"
    MONAI_AMD_LIB=$PWD/monai_amd/csrc/libmonai_amd_dev.so MONAI_AMD_BLEND_G=$g MONAI_AMD_BLEND_NT=$nt BB_QUICK=1 timeout 300 python tools/blend_bench.py > $O/blend_g${g}_nt${nt}.json 2>> $O/blend_bench.err && show $O/blend_g${g}_nt${nt}.json | grep mosaic
  done; done
fi
