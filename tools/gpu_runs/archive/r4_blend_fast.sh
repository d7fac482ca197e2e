# round 4: the mosaic blend with 24-bit / 32-bit index arithmetic and no per-accumulate selects against the previous build, alternating processes on one box
# (tools/blend_bench.py BB_QUICK: both layouts, every mosaic result compared bitwise with the unchanged window-major kernel)
export TMPDIR=/tmp
O=gpurun_out/r4blend; mkdir -p $O
show() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for r in d["runs"]:
    print(f'  {r["variant"][:80]:80s} {r["ms"]:.3f} ms  {r["frac_of_8TBps"]:.3f}  {r.get("bitwise_equal_to_window_major", "")}')
PY
}
for i in 1 2; do
  echo "== previous build"; MONAI_AMD_LIB=$PWD/tools/ubench/_old/libmonai_amd_prev.so BB_QUICK=1 timeout 300 python tools/blend_bench.py > $O/prev_$i.json 2>> $O/err.txt && show $O/prev_$i.json
  echo "== this tree"; BB_QUICK=1 timeout 300 python tools/blend_bench.py > $O/new_$i.json 2>> $O/err.txt && show $O/new_$i.json
done
